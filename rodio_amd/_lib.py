"""ctypes binding of librodio_hip.so (the C ABI declared in include/rodio_hip.h).

There is no CPU fallback: if the shared library is missing this module raises at import, and
every compute entry point returns RH_ERR_NOT_INITIALIZED until rh_init() has found a gfx950 GPU.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RODIO_HIP_LIB") or os.path.join(_HERE, "librodio_hip.so")  # override: diagnostic builds

RH_OK = 0
STATUS_NAMES = {
    0: "RH_OK", 1: "RH_ERR_INVALID", 2: "RH_ERR_HIP", 3: "RH_ERR_UNSUPPORTED", 4: "RH_ERR_NOMEM",
    5: "RH_ERR_TIMEOUT", 6: "RH_ERR_NOT_INITIALIZED", 7: "RH_ERR_CAPACITY",
}


class RhError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: {STATUS_NAMES.get(status, status)} {detail}".strip())


class LimitParams(C.Structure):
    _fields_ = [("threshold_db", C.c_float), ("knee_width_db", C.c_float),
                ("attack_ns", C.c_uint64), ("release_ns", C.c_uint64)]


class AgcParams(C.Structure):
    _fields_ = [("target_level", C.c_float), ("attack_ns", C.c_uint64), ("release_ns", C.c_uint64),
                ("absolute_max_gain", C.c_float), ("floor", C.c_float)]


class WavInfo(C.Structure):
    _fields_ = [("channels", C.c_uint32), ("sample_rate", C.c_uint32), ("bits_per_sample", C.c_uint32), ("is_float", C.c_int32),
                ("data_offset", C.c_uint64), ("data_bytes", C.c_uint64), ("samples", C.c_uint64)]


class UniformSeg(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("src_frame0", C.c_uint64), ("src_frames", C.c_uint64), ("m0", C.c_uint64), ("m1", C.c_uint64),
                ("span_frames", C.c_uint64), ("from_rate", C.c_uint32), ("to_rate", C.c_uint32), ("from_ch", C.c_uint32), ("to_ch", C.c_uint32), ("gain", C.c_float), ("reserved", C.c_uint32)]


class WideSrc(C.Structure):
    _fields_ = [("data", C.c_void_p), ("channels", C.c_uint32), ("from_rate", C.c_uint32), ("phase", C.c_uint32), ("frames", C.c_uint64), ("last", C.c_uint32), ("gain", C.c_float)]


class RlmConfig(C.Structure):
    _fields_ = [("from_rate", C.c_uint32), ("to_rate", C.c_uint32), ("channels", C.c_uint32),
                ("span_len", C.c_uint64), ("filter_kind", C.c_int32), ("filter_freq", C.c_uint32),
                ("filter_q", C.c_float), ("max_sources", C.c_uint32), ("max_in_frames", C.c_uint64),
                ("frames_per_lane", C.c_uint32), ("ring_stages", C.c_uint32), ("no_balance", C.c_uint32), ("force_general", C.c_uint32), ("custom_coeffs", C.c_float * 5), ("filter_first", C.c_uint32)]


class RlmGeometry(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("threads", "frames_per_lane", "ring_stages", "stage_kib", "lds_bytes",
                                           "lookback_tiles", "resident_waves_per_cu", "n_tiles", "general_kernel", "ragged_pair", "mix_first")]


vp, sz, u32, u64, i32, f32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int32, C.c_float
f32p = C.POINTER(C.c_float)

# name -> (restype, argtypes).  Everything include/rodio_hip.h declares must be listed here:
# tests/test_abi.py checks the two against each other and against the built library.
SIGNATURES = {
    "rh_version": (i32, []),
    "rh_status_string": (C.c_char_p, [i32]),
    "rh_last_hip_error": (C.c_char_p, []),
    "rh_async_status": (i32, []),
    "rh_init": (i32, [i32]),
    "rh_device_name": (i32, [C.c_char_p, sz]),
    "rh_bind_thread": (i32, []),
    "rh_malloc": (i32, [C.POINTER(vp), sz]),
    "rh_free": (i32, [vp]),
    "rh_memset": (i32, [vp, i32, sz, vp]),
    "rh_memcpy_h2d": (i32, [vp, vp, sz, vp]),
    "rh_memcpy_h2d_rows": (i32, [vp, vp, sz, sz, sz, vp]),
    "rh_memcpy_d2h": (i32, [vp, vp, sz, vp]),
    "rh_memcpy_d2h_async": (i32, [vp, vp, C.c_size_t, vp]),
    "rh_memcpy_d2d": (i32, [vp, vp, C.c_size_t, vp]),
    "rh_host_alloc": (i32, [C.POINTER(vp), C.c_size_t]),
    "rh_host_free": (i32, [vp]),
    "rh_event_synchronize": (i32, [vp]),
    "rh_stream_wait_event": (i32, [vp, vp]),
    "rh_stream_create": (i32, [C.POINTER(vp)]),
    "rh_stream_destroy": (i32, [vp]),
    "rh_stream_synchronize": (i32, [vp]),
    "rh_stream_release_scratch": (i32, [vp]),
    "rh_event_create": (i32, [C.POINTER(vp)]),
    "rh_event_destroy": (i32, [vp]),
    "rh_event_record": (i32, [vp, vp]),
    "rh_event_elapsed_ms": (i32, [vp, vp, f32p]),
    "rh_convert_i8_to_f32": (i32, [vp, vp, sz, vp]),
    "rh_convert_u8_to_f32": (i32, [vp, vp, sz, vp]),
    "rh_convert_i16_to_f32": (i32, [vp, vp, sz, vp]),
    "rh_convert_u16_to_f32": (i32, [vp, vp, sz, vp]),
    "rh_convert_i24_to_f32": (i32, [vp, vp, sz, vp]),
    "rh_convert_i32_to_f32": (i32, [vp, vp, sz, vp]),
    "rh_convert_f32_to_i8": (i32, [vp, vp, sz, vp]),
    "rh_convert_f32_to_i16": (i32, [vp, vp, sz, vp]),
    "rh_convert_f32_to_u16": (i32, [vp, vp, sz, vp]),
    "rh_convert_f32_to_i32": (i32, [vp, vp, sz, vp]),
    "rh_convert_f32_to_u8": (i32, [vp, vp, sz, vp]),
    "rh_convert_f32_to_i24": (i32, [vp, vp, sz, vp]),
    "rh_convert_f32_to_u24": (i32, [vp, vp, sz, vp]),
    "rh_convert_f32_to_u32": (i32, [vp, vp, sz, vp]),
    "rh_convert_f32_to_i64": (i32, [vp, vp, sz, vp]),
    "rh_convert_f32_to_u64": (i32, [vp, vp, sz, vp]),
    "rh_convert_f32_to_f64": (i32, [vp, vp, sz, vp]),
    "rh_convert_u24_to_f32": (i32, [vp, vp, sz, vp]),
    "rh_convert_u32_to_f32": (i32, [vp, vp, sz, vp]),
    "rh_convert_i64_to_f32": (i32, [vp, vp, sz, vp]),
    "rh_convert_u64_to_f32": (i32, [vp, vp, sz, vp]),
    "rh_convert_f64_to_f32": (i32, [vp, vp, sz, vp]),
    "rh_channels_convert": (i32, [vp, vp, sz, u32, u32, vp]),
    "rh_amplify": (i32, [vp, vp, sz, f32, vp]),
    "rh_channel_volume": (i32, [vp, vp, sz, u32, f32p, u32, vp]),
    "rh_spatial_gains": (i32, [f32p, f32p, f32p, f32p]),
    "rh_delay_samples": (u64, [u64, u32, u32]),
    "rh_echo_mix": (i32, [vp, vp, sz, sz, f32, vp]),
    "rh_resample_out_frames": (i32, [u64, u32, u32, u32, u64, C.POINTER(u64)]),
    "rh_resample_linear": (i32, [vp, vp, u64, u32, u32, u32, u64, vp]),
    "rh_uniform_span_frames": (i32, [u64, u32, u32, i32, C.POINTER(u64)]),
    "rh_uniform_first_tap": (i32, [u64, u32, u32, C.POINTER(u64)]),
    "rh_uniform_cut_tail_samples": (i32, [u64, u32, u32, u32, u32, u32, C.POINTER(u64)]),
    "rh_uniform_segments": (i32, [C.POINTER(UniformSeg), u32, vp]),
    "rh_uniform_segments_dev": (i32, [vp, u32, u64, vp]),
    "rh_mix_sum": (i32, [vp, sz, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64), u32, vp]),
    "rh_wide_mix_block": (i32, [vp, u32, u32, u64, C.POINTER(WideSrc), u32, vp]),
    "rh_biquad_coeffs": (i32, [i32, u32, f32, u32, f32p]),
    "rh_biquad": (i32, [vp, vp, u64, u32, u32, f32p, vp, i32, vp]),
    "rh_limit": (i32, [vp, vp, u64, u32, u32, u32, C.POINTER(LimitParams), vp, vp]),
    "rh_agc_state_floats": (sz, []),
    "rh_agc_state_init": (i32, [vp, u32, vp]),
    "rh_agc": (i32, [vp, vp, u64, u32, u32, C.POINTER(AgcParams), vp, vp]),
    "rh_resampler_create": (i32, [C.POINTER(vp), u32, u32, u32]),
    "rh_resampler_reset": (i32, [vp]),
    "rh_resampler_destroy": (i32, [vp]),
    "rh_resampler_pending_frames": (i32, [vp, u64, i32, C.POINTER(u64)]),
    "rh_resampler_process": (i32, [vp, vp, u64, vp, u64, i32, C.POINTER(u64), vp]),
    "rh_echo_create": (i32, [C.POINTER(vp), u64, f32]),
    "rh_echo_reset": (i32, [vp]),
    "rh_echo_destroy": (i32, [vp]),
    "rh_echo_process": (i32, [vp, vp, vp, u64, vp]),
    "rh_echo_flush": (i32, [vp, vp, vp]),
    "rh_wav_probe_host": (i32, [vp, sz, C.POINTER(WavInfo)]),
    "rh_wav_decode": (i32, [vp, vp, u64, u32, u32, i32, C.POINTER(u64), vp]),
    "rh_wav_decode_channels": (i32, [vp, vp, u64, u32, u32, i32, u32, C.POINTER(u64), vp]),
    "rh_wav_header_f32_host": (sz, [vp, sz, u32, u32, u64]),
    "rh_delay": (i32, [vp, vp, u64, u64, vp]),
    "rh_take_duration": (i32, [vp, vp, u64, u64, u32, u32, u64, i32, C.POINTER(u64), C.POINTER(i32), vp]),
    "rh_take_duration_from": (i32, [vp, vp, u64, u64, u64, u32, u32, u32, i32, C.POINTER(u64), C.POINTER(i32), C.POINTER(u64), vp]),
    "rh_distortion": (i32, [vp, vp, sz, f32, f32, vp]),
    "rh_db_to_linear": (f32, [f32]),
    "rh_linear_to_db": (f32, [f32]),
    "rh_duration_to_coefficient": (f32, [u64, u32]),
    "rh_dither": (i32, [vp, vp, sz, u64, u32, u32, i32, u64, vp]),
    "rh_linear_gain_ramp": (i32, [vp, vp, sz, u64, u32, u32, u64, f32, f32, i32, vp]),
    "rh_reverb_spatial": (i32, [vp, vp, sz, sz, f32, vp, u32, sz, sz, vp]),
    "rh_comm_unique_id": (i32, [vp]),
    "rh_comm_init": (i32, [C.POINTER(vp), i32, i32, vp]),
    "rh_comm_destroy": (i32, [vp]),
    "rh_allreduce_sum_f32": (i32, [vp, vp, sz, vp]),
    "rh_reduce_sum_f32": (i32, [vp, vp, sz, i32, vp]),
    "rh_rlm_create": (i32, [C.POINTER(vp), C.POINTER(RlmConfig)]),
    "rh_rlm_destroy": (i32, [vp]),
    "rh_rlm_set_sources": (i32, [vp, C.POINTER(vp), C.POINTER(u64), u32]),
    "rh_rlm_run": (i32, [vp, vp, u64, C.POINTER(u64), vp]),
    "rh_filter_scan_ok": (i32, [i32, u32, f32, u32]),
    "rh_rlm_set_gains": (i32, [vp, f32p, u32]),
    "rh_rlm_set_filters": (i32, [vp, C.POINTER(i32), C.POINTER(u32), f32p, u32]),
    "rh_rlm_stream_keep_history": (i32, [vp, i32]),
    "rh_rlm_stream_overlap": (i32, [vp, i32]),
    "rh_rlm_stream_one_launch_blocks": (i32, [vp, C.POINTER(u32)]),
    "rh_rlm_stream_overlapped_blocks": (i32, [vp, C.POINTER(u32)]),
    "rh_rlm_stream_stats": (i32, [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]),
    "rh_rlm_set_exclusive": (i32, [vp, i32]),
    "rh_rlm_set_mix_first": (i32, [vp, i32]),
    "rh_rlm_run_subset": (i32, [vp, u32, u32, vp, u64, C.POINTER(u64), vp]),
    "rh_rlm_stream_begin": (i32, [vp]),
    "rh_rlm_stream_block": (i32, [vp, C.POINTER(vp), u32, u64, i32, vp, u64, C.POINTER(u64), C.POINTER(u64), vp]),
    "rh_rlm_stream_block_v": (i32, [vp, C.POINTER(vp), C.POINTER(u64), C.POINTER(C.c_uint8), u32, vp, u64, C.POINTER(u64), C.POINTER(u64), vp]),
    "rh_rlm_run_batch": (i32, [vp, vp, u64, C.POINTER(u64), vp]),
    "rh_rlm_autotune": (i32, [vp, vp, u64, vp, C.POINTER(u32), C.POINTER(u32)]),
    "rh_rlm_last_status": (i32, [vp]),
    "rh_rlm_geometry": (i32, [vp, C.POINTER(RlmGeometry)]),
    "rh_rlm_late_carries": (i32, [vp, C.POINTER(u64)]),
    "rh_rlm_phase_cycles": (i32, [vp, C.POINTER(C.c_double)]),
}


def load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python rodio_amd/build.py` (needs hipcc). "
            "rodio_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64
    # (same SONAME as /opt/rocm's).  If librodio_hip.so pulled in /opt/rocm's copy first, torch
    # would later load a second runtime and one of the two would see no device.  Loading torch
    # first makes the library bind to the runtime that is already in the process.  Hosts without
    # torch (the Rust/C++ side) simply get /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = the library does not export the ABI
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()


def check(status: int, where: str):
    if status != RH_OK:
        detail = lib.rh_last_hip_error().decode(errors="replace") if status == 2 else ""
        raise RhError(status, where, detail)

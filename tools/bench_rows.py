"""Every stand-alone entry of SURVEY.md 8(a)'s rows at a size that fills the chip: fraction of 8 TB/s on its algorithmic bytes.
(The fused path has bench.py; this is the table for the one-row launches a chain falls back to.)  HIP events, rows resident.

    python tools/bench_rows.py [--mib 512] [--steps 10] [--only name,name]
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rodio_amd import _lib, source


def timed(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def measure(mib=512, steps=10, only=(), emit=None):
    """The rows as a list of dicts (bench.py's `roofline.side.rows` calls this with a smaller size and a few rows)."""
    a = argparse.Namespace(mib=mib, steps=steps)
    only = set(only)
    rows = []
    source._ensure()
    lib, st = _lib.lib, source._stream()
    n = (a.mib << 20) // 4  # f32 samples of input
    x = (torch.rand(n, device="cuda") * 2 - 1).contiguous()
    P = lambda t: C.c_void_p(t.data_ptr())
    ck = _lib.check

    def row(name, fn, alg, extra=None):
        if only and name not in only:
            return
        ms = timed(fn, a.steps)
        d = {"row": name, "ms": round(ms, 4), "GBps": round(alg / ms / 1e6, 1), "frac": round(alg / ms / 1e6 / 8000, 3)}
        if extra:
            d.update(extra)
        rows.append(d)
        if emit:
            emit(d)

    # a1/a2 SampleRateConverter, stand-alone: 44.1 -> 48 kHz (and down), stereo / mono / 5.1
    for ch, fr, to in [(2, 44100, 48000), (1, 44100, 48000), (6, 44100, 48000), (2, 48000, 44100), (2, 8000, 48000)]:
        frames = n // ch
        m = C.c_uint64(0)
        ck(lib.rh_resample_out_frames(frames, fr, to, ch, 0, C.byref(m)), "out_frames")
        dst = torch.empty(m.value * ch + 8, device="cuda")
        row(f"resample_linear ch={ch} {fr}->{to}", lambda: ck(lib.rh_resample_linear(P(dst), P(x), frames, fr, to, ch, 0, st), "rh_resample_linear"), 4 * frames * ch + 4 * m.value * ch)
        del dst
    # a6 Mixer: ordered sum of S rows
    for S in (4, 32, 256):
        L = n // S // 4 * 4
        ptrs = (C.c_void_p * S)(*[x.data_ptr() + 4 * L * s for s in range(S)])
        starts = (C.c_uint64 * S)(*([0] * S))
        lens = (C.c_uint64 * S)(*([L] * S))
        dst = torch.empty(L + 8, device="cuda")
        row(f"mix_sum S={S}", lambda: ck(lib.rh_mix_sum(P(dst), L, ptrs, starts, lens, S, st), "rh_mix_sum"), 4 * L * S + 4 * L)
        del dst
    for S in (32, 256):  # the same rows NOT a power of two apart (+ 4 KiB + 64 B)
        pitch = n // S // 4 * 4
        L = pitch - 1040
        ptrs = (C.c_void_p * S)(*[x.data_ptr() + 4 * pitch * s0 for s0 in range(S)])
        starts = (C.c_uint64 * S)(*([0] * S))
        lens = (C.c_uint64 * S)(*([L] * S))
        dst = torch.empty(L + 8, device="cuda")
        row(f"mix_sum S={S} odd pitch", lambda: ck(lib.rh_mix_sum(P(dst), L, ptrs, starts, lens, S, st), "rh_mix_sum"), 4 * L * S + 4 * L)
        del dst
    for S in (4, 32):  # late joins: every source starts two samples after the one in front (rows off the vector boundary)
        L = n // S // 4 * 4 - 256
        ptrs = (C.c_void_p * S)(*[x.data_ptr() + 4 * (L + 256) * s0 for s0 in range(S)])
        starts = (C.c_uint64 * S)(*[2 * s0 for s0 in range(S)])
        lens = (C.c_uint64 * S)(*([L] * S))
        dst = torch.empty(L + 2 * S + 8, device="cuda")
        row(f"mix_sum S={S} late joins", lambda: ck(lib.rh_mix_sum(P(dst), L + 2 * S, ptrs, starts, lens, S, st), "rh_mix_sum"), 4 * L * S + 4 * L)
        del dst
    dst = torch.empty(n + 1 + (1 << 17), device="cuda")
    # a7 Amplify
    row("amplify", lambda: ck(lib.rh_amplify(P(dst), P(x), n, 0.5, st), "rh_amplify"), 8 * n)
    xo = x[1:]  # a row that starts 4 bytes off a vector boundary (a row inside a larger buffer)
    row("amplify src+4B", lambda: ck(lib.rh_amplify(P(dst), P(xo), n - 1, 0.5, st), "rh_amplify"), 8 * (n - 1))
    # a9 reverb stand-alone (echo mix), delay 65 536 samples
    D = 65536
    row("echo_mix D=65536", lambda: ck(lib.rh_echo_mix(P(dst), P(x), n, D, 0.7, st), "rh_echo_mix"), 4 * n + 4 * (n + D))
    # a10 ChannelVolume / Spatial (stereo -> stereo) and 6 -> 2
    g2 = np.array([0.3, 0.9], np.float32)
    row("channel_volume 2->2", lambda: ck(lib.rh_channel_volume(P(dst), P(x), n // 2, 2, g2.ctypes.data_as(_lib.f32p), 2, st), "rh_channel_volume"), 8 * n)
    row("channel_volume 6->2", lambda: ck(lib.rh_channel_volume(P(dst), P(x), n // 6, 6, g2.ctypes.data_as(_lib.f32p), 2, st), "rh_channel_volume"), 4 * (n // 6) * 8)
    g6 = np.array([0.3, 0.9, 0.5, 0.2, 0.1, 1.0], np.float32)
    d6 = torch.empty(3 * n + 8, device="cuda")
    row("channel_volume 2->6", lambda: ck(lib.rh_channel_volume(P(d6), P(x), n // 2, 2, g6.ctypes.data_as(_lib.f32p), 6, st), "rh_channel_volume"), 4 * n + 12 * n)
    del d6
    # a4 SampleTypeConverter
    i16 = torch.empty(n, dtype=torch.int16, device="cuda")
    row("f32_to_i16", lambda: ck(lib.rh_convert_f32_to_i16(P(i16), P(x), n, st), "cv"), 6 * n)
    row("i16_to_f32", lambda: ck(lib.rh_convert_i16_to_f32(P(dst), P(i16), n, st), "cv"), 6 * n)
    i16o = i16[2:]
    row("i16_to_f32 src+4B", lambda: ck(lib.rh_convert_i16_to_f32(P(dst), P(i16o), n - 2, st), "cv"), 6 * (n - 2))
    i32 = torch.empty(n, dtype=torch.int32, device="cuda")
    row("f32_to_i32", lambda: ck(lib.rh_convert_f32_to_i32(P(i32), P(x), n, st), "cv"), 8 * n)
    row("f32_to_u32", lambda: ck(lib.rh_convert_f32_to_u32(P(i32), P(x), n, st), "cv"), 8 * n)
    row("u32_to_f32", lambda: ck(lib.rh_convert_u32_to_f32(P(dst), P(i32), n, st), "cv"), 8 * n)
    u8 = torch.empty(n, dtype=torch.uint8, device="cuda")
    row("f32_to_u8", lambda: ck(lib.rh_convert_f32_to_u8(P(u8), P(x), n, st), "cv"), 5 * n)
    f64 = torch.empty(n // 2, dtype=torch.float64, device="cuda")
    row("f32_to_f64", lambda: ck(lib.rh_convert_f32_to_f64(P(f64), P(x), n // 2, st), "cv"), 12 * (n // 2))
    row("f64_to_f32", lambda: ck(lib.rh_convert_f64_to_f32(P(dst), P(f64), n // 2, st), "cv"), 12 * (n // 2))
    del i32, u8, f64
    # (f)3 elementwise adapters
    row("distortion", lambda: ck(lib.rh_distortion(P(dst), P(x), n, 2.0, 0.8, st), "rh_distortion"), 8 * n)
    row("delay D=65536", lambda: ck(lib.rh_delay(P(dst), P(x), n, D, st), "rh_delay"), 4 * n + 4 * (n + D))
    row("linear_gain_ramp", lambda: ck(lib.rh_linear_gain_ramp(P(dst), P(x), n, 0, 2, 48000, 10_000_000_000, 0.0, 1.0, 1, st), "ramp"), 8 * n)
    row("dither", lambda: ck(lib.rh_dither(P(dst), P(x), n, 0, 2, 16, 1, 1234, st), "dither"), 8 * n)
    # a8 BltFilter stand-alone (time-parallel) for reference
    co = (C.c_float * 5)()
    ck(lib.rh_biquad_coeffs(0, 200, 0.5, 48000, co), "coeffs")
    S = 64
    row("biquad 64 streams (mode 1)", lambda: ck(lib.rh_biquad(P(dst), P(x), n // S // 2, 2, S, co, None, 1, st), "rh_biquad"), 8 * (n // S // 2) * 2 * S)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=512)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    measure(a.mib, a.steps, [x for x in a.only.split(",") if x], emit=lambda d: print(json.dumps(d), flush=True))


if __name__ == "__main__":
    main()

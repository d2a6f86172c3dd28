"""The one-row launches at BLOCK sizes (what a GpuSource chain hands them: a few thousand frames): microseconds per launch, launches back to back.
    python tools/bench_rows_small.py [--frames 4096]        (RH_PCM_NO_TILE=1: the lane-per-output kernels)
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


def timed(fn, steps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4096)
    a = ap.parse_args()
    source._ensure()
    lib, st, ck = _lib.lib, source._stream(), _lib.check
    F = a.frames
    x = (torch.rand(F * 8 + 64, device="cuda") * 2 - 1).contiguous()
    dst = torch.empty(F * 16 + 1024, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    g6 = np.linspace(0.2, 1.0, 6).astype(np.float32)
    m = C.c_uint64(0)
    i16 = torch.randint(-30000, 30000, (F * 8,), dtype=torch.int16, device="cuda")
    rows = {
        "amplify 2ch": lambda: ck(lib.rh_amplify(P(dst), P(x), F * 2, 0.5, st), "a"),
        "distortion 2ch": lambda: ck(lib.rh_distortion(P(dst), P(x), F * 2, 2.0, 0.8, st), "a"),
        "linear_gain_ramp 2ch": lambda: ck(lib.rh_linear_gain_ramp(P(dst), P(x), F * 2, 0, 2, 48000, 10_000_000_000, 0.0, 1.0, 1, st), "a"),
        "echo_mix 2ch D=2000": lambda: ck(lib.rh_echo_mix(P(dst), P(x), F * 2, 2000, 0.7, st), "a"),
        "resample 2ch 44.1->48": lambda: ck(lib.rh_resample_linear(P(dst), P(x), F, 44100, 48000, 2, 0, st), "a"),
        "resample 6ch 44.1->48": lambda: ck(lib.rh_resample_linear(P(dst), P(x), F, 44100, 48000, 6, 0, st), "a"),
        "channels 6->2": lambda: ck(lib.rh_channels_convert(P(dst), P(x), F, 6, 2, st), "a"),
        "channels 2->6": lambda: ck(lib.rh_channels_convert(P(dst), P(x), F, 2, 6, st), "a"),
        "channel_volume 2->6": lambda: ck(lib.rh_channel_volume(P(dst), P(x), F, 2, g6.ctypes.data_as(_lib.f32p), 6, st), "a"),
        "wav_decode_channels i16 6->2": lambda: ck(lib.rh_wav_decode_channels(P(dst), P(i16), F * 6, 6, 16, 0, 2, C.byref(m), st), "a"),
        "i16_to_f32 2ch": lambda: ck(lib.rh_convert_i16_to_f32(P(dst), P(i16), F * 2, st), "a"),
    }
    S = 16
    ptrs = (C.c_void_p * S)(*[x.data_ptr() for _ in range(S)])
    starts = (C.c_uint64 * S)(*([0] * S))
    lens = (C.c_uint64 * S)(*([F * 2] * S))
    rows["mix_sum 16 x 2ch"] = lambda: ck(lib.rh_mix_sum(P(dst), F * 2, ptrs, starts, lens, S, st), "a")
    for k, fn in rows.items():
        print(json.dumps({"row": k, "frames": F, "us": round(timed(fn), 2)}), flush=True)


if __name__ == "__main__":
    main()

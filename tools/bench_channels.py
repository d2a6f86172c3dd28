"""ChannelCountConverter (channels.rs:57-85) as a stand-alone launch, from f32 frames and straight from PCM bytes (rh_wav_decode_channels):
the lane-per-output kernels against the tile-per-workgroup kernel (k_pcm_to_channels_tile), layouts side by side.  HIP events, rows resident.

    python tools/bench_channels.py [--mib 768] [--steps 10]          (RH_PCM_NO_TILE=1 / RH_PCM_TILE_KB=n: the knobs of rh_wav.hip)
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=768)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    source._ensure()
    lib, st = _lib.lib, source._stream()
    raw = torch.randint(0, 255, (a.mib << 20,), dtype=torch.uint8, device="cuda")
    raw.view(torch.int32).bitwise_and_(0x3F7FFFFF)  # as f32: finite
    m = C.c_uint64(0)
    for bits, is_float, frm, to in [(32, 1, 6, 2), (32, 1, 2, 6), (32, 1, 2, 1), (32, 1, 1, 2), (32, 1, 8, 2), (32, 1, 2, 2), (16, 0, 6, 2), (16, 0, 2, 2), (16, 0, 2, 6), (16, 0, 8, 1),
                                   (24, 0, 6, 2), (24, 0, 2, 2), (8, 0, 6, 2), (8, 0, 8, 8)]:
        bps = bits // 8
        frames = min((a.mib << 20) // (bps * frm), (3 << 30) // (4 * to))
        n = frames * frm
        dst = torch.empty(frames * to + 8, device="cuda")
        alg = n * bps + 4 * frames * to
        row = {"bits": bits, "float": is_float, "from": frm, "to": to, "frames": frames}
        ms = timed(lambda: _lib.check(lib.rh_wav_decode_channels(C.c_void_p(dst.data_ptr()), C.c_void_p(raw.data_ptr()), n, frm, bits, is_float, to, C.byref(m), st), "rh_wav_decode_channels"), a.steps)
        row["decode_channels_ms"], row["decode_channels_frac"] = round(ms, 4), round(alg / ms / 1e6 / 8000, 3)
        if is_float:
            dst2 = torch.empty(frames * to + 8, device="cuda")
            ms2 = timed(lambda: _lib.check(lib.rh_channels_convert(C.c_void_p(dst2.data_ptr()), C.c_void_p(raw.data_ptr()), frames, frm, to, st), "rh_channels_convert"), a.steps)
            row["channels_convert_ms"], row["channels_convert_frac"] = round(ms2, 4), round(alg / ms2 / 1e6 / 8000, 3)
            row["same_bits"] = bool(torch.equal(dst[: frames * to].view(torch.int32), dst2[: frames * to].view(torch.int32)))
            del dst2
        print(json.dumps(row), flush=True)
        del dst


if __name__ == "__main__":
    main()

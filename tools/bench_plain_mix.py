"""The mixer WITHOUT a filter -- `mixer.add(src)` / `mixer.add(src.amplify(g))`, rodio's everyday use: per source UniformSourceIterator
(44.1 -> 48 kHz), then the ordered sum (mixer.rs:185-198), bit-exact.  The headline's 256 x 1 Mi stereo sources, one shot (rh_rlm_run) and --
for mixers of few sources -- rh_wide_mix_block on the same rows.  HIP events; prints one JSON line per form.

    python tools/bench_plain_mix.py [--sources 256] [--frames 1048576] [--steps 10]
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import rodio_amd as rh
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
    ap.add_argument("--sources", type=int, default=256)
    ap.add_argument("--frames", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    source._ensure()
    S, N = a.sources, a.frames
    rows = [torch.from_numpy((np.random.default_rng(1234 + s).uniform(-1, 1, 2 * N) / S).astype(np.float32)).cuda() for s in range(S)]
    p = rh.ResampleLowpassMix(44100, 48000, 2, None, None, 0, 0.5, max_sources=S, max_in_frames=N)
    p.set_sources(rows)
    out = torch.empty(p.out_frames * 2 + 8, device="cuda")
    p.autotune(out=out)  # (as bench.py does for the headline: the fastest of the candidate launch geometries)
    ms = timed(lambda: p.run(out=out), a.steps)
    M = p.out_frames
    alg = 8 * S * N + 8 * M
    geo = p.geometry()
    got = p.run(out=out).clone()
    print(json.dumps({"form": "rh_rlm_run, no filter (k_rlm_fast plain: per source converter, ordered sum)", "sources": S, "frames": N, "ms": ms, "GBps": alg / ms / 1e6, "frac_of_8TBps": alg / ms / 1e6 / 8000,
                      "geometry": {k: geo[k] for k in ("frames_per_lane", "ring_stages", "n_tiles", "general_kernel", "mix_first")}}))
    # rh_wide_mix_block on the same rows (32 sources a launch, the partial sum carried through dst)
    arr = (_lib.WideSrc * S)()
    for s in range(S):
        arr[s].data, arr[s].channels, arr[s].from_rate, arr[s].phase, arr[s].frames, arr[s].last, arr[s].gain = rows[s].data_ptr(), 2, 44100, 0, M, N - 1, 1.0
    dst = torch.empty(M * 2 + 8, device="cuda")
    ms2 = timed(lambda: _lib.check(_lib.lib.rh_wide_mix_block(C.c_void_p(dst.data_ptr()), 2, 48000, M, arr, S, source._stream()), "rh_wide_mix_block"), a.steps)
    same = bool(torch.equal(dst[: M * 2].view(torch.int32), got[: M * 2].view(torch.int32)))
    print(json.dumps({"form": "rh_wide_mix_block on the same rows", "sources": S, "frames": N, "ms": ms2, "GBps": alg / ms2 / 1e6, "frac_of_8TBps": alg / ms2 / 1e6 / 8000, "bit_identical_to_rh_rlm_run": same}))


if __name__ == "__main__":
    main()

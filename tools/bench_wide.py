"""A block of a 5.1 mixer (mixer::mixer(nz!(6), 48 kHz)): S continuous sources at 44.1 kHz, amplified, converted, summed.
    old: what a wide generation of GpuMixer ran until round 6 -- per source rh_amplify + rh_uniform_segments (one segment), then rh_mix_sum
    new: rh_wide_mix_block, one launch
Rows resident in device memory; HIP events around `steps` blocks.  Prints one JSON line per form.

    python tools/bench_wide.py [--sources 16] [--block 16384] [--channels 6] [--steps 50]
"""
import argparse
import ctypes as C
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rodio_amd import _lib, source

U64_MAX = (1 << 64) - 1


def measure(S=16, M=16384, Cc=6, from_rate=44100, to_rate=48000, steps=50, check_frames=2048):
    """-> {"old": ms per block, "new": ms per block, "bit_identical_to_chain": bool, "oracle_ok": bool | None, "algorithmic_bytes": int}"""
    source._ensure()
    lib = _lib.lib
    n_in = M * from_rate // to_rate + 8
    rng = np.random.default_rng(1)
    host = [rng.uniform(-1, 1, n_in * Cc).astype(np.float32) for _ in range(S)]
    rows = [torch.from_numpy(h).cuda() for h in host]
    gains = [float(np.float32(0.5 + 0.01 * s)) for s in range(S)]
    st = source._stream()
    # ---- old: amplify -> uniform segment -> rows -> rh_mix_sum
    amp = [torch.empty_like(r) for r in rows]
    conv = [torch.empty(M * Cc, device="cuda") for _ in range(S)]
    dst_old = torch.empty(M * Cc, device="cuda")
    segs = []
    for s in range(S):
        g = _lib.UniformSeg()
        g.src, g.dst = amp[s].data_ptr(), conv[s].data_ptr()
        g.src_frame0, g.src_frames, g.m0, g.m1, g.span_frames = 0, n_in, 0, M, U64_MAX
        g.from_rate, g.to_rate, g.from_ch, g.to_ch, g.gain, g.reserved = from_rate, to_rate, Cc, Cc, 1.0, 0
        segs.append(g)
    ptrs = (C.c_void_p * S)(*[c.data_ptr() for c in conv])
    start = (C.c_uint64 * S)(*([0] * S))
    lens = (C.c_uint64 * S)(*([M * Cc] * S))

    def old():
        for s in range(S):
            _lib.check(lib.rh_amplify(C.c_void_p(amp[s].data_ptr()), C.c_void_p(rows[s].data_ptr()), n_in * Cc, gains[s], st), "rh_amplify")
            _lib.check(lib.rh_uniform_segments(C.byref(segs[s]), 1, st), "rh_uniform_segments")
        _lib.check(lib.rh_mix_sum(C.c_void_p(dst_old.data_ptr()), M * Cc, ptrs, start, lens, S, st), "rh_mix_sum")

    # ---- new: one launch
    dst_new = torch.empty(M * Cc, device="cuda")
    arr = (_lib.WideSrc * S)()
    for s in range(S):
        arr[s].data, arr[s].channels, arr[s].from_rate, arr[s].phase, arr[s].frames, arr[s].last, arr[s].gain = rows[s].data_ptr(), Cc, from_rate, 0, M, 0xFFFFFFFF, gains[s]

    def new():
        _lib.check(lib.rh_wide_mix_block(C.c_void_p(dst_new.data_ptr()), Cc, to_rate, M, arr, S, st), "rh_wide_mix_block")

    res = {}
    for name, fn in (("old", old), ("new", new)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()  # (source._stream() is torch's current stream)
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / steps
    res["bit_identical_to_chain"] = bool(torch.equal(dst_old.view(torch.int32), dst_new.view(torch.int32)))
    res["algorithmic_bytes"] = 4 * S * Cc * (M * from_rate // to_rate) + 4 * Cc * M
    res["oracle_ok"] = None
    if check_frames:  # the first frames against the oracle's mixer (the checker: mixer::mixer(ch, rate) + add(src.amplify(g)) per sample)
        from oracle import rodio_oracle as O

        k = min(check_frames, M)
        need = k * from_rate // to_rate + 4  # frames whose taps the first k outputs read
        mx = O.Mixer(Cc, to_rate)
        for s in range(S):
            mx.add(O.TestSource(host[s][: need * Cc], Cc, from_rate).amplify(gains[s]))
        ref = mx.pull(k * Cc)
        res["oracle_ok"] = bool(len(ref) == k * Cc and np.array_equal(ref.view(np.uint32), dst_new[: k * Cc].cpu().numpy().view(np.uint32)))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sources", type=int, default=16)
    ap.add_argument("--block", type=int, default=16384, help="output frames per block")
    ap.add_argument("--channels", type=int, default=6)
    ap.add_argument("--from-rate", type=int, default=44100)
    ap.add_argument("--to-rate", type=int, default=48000)
    ap.add_argument("--steps", type=int, default=50)
    a = ap.parse_args()
    S, M, Cc = a.sources, a.block, a.channels
    res = measure(S, M, Cc, a.from_rate, a.to_rate, a.steps)
    algo = res["algorithmic_bytes"]
    for name in ("old", "new"):
        ms = res[name]
        print(json.dumps({"form": name, "sources": S, "channels": Cc, "block_frames": M, "ms_per_block": ms, "launches_per_block": (2 * S + (S + 31) // 32) if name == "old" else (S + 31) // 32,
                          "algorithmic_bytes": algo, "GBps": algo / ms / 1e6, "frac_of_8TBps": algo / ms / 1e6 / 8000.0, "bit_identical_to_old": res["bit_identical_to_chain"], "oracle_ok": res["oracle_ok"]}))


if __name__ == "__main__":
    main()

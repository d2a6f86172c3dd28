"""Ragged one-shot batches through rh_rlm_run (GPU box): python tools/bench_ragged.py"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodio_amd as rh
rh.init(0)
S, N = 256, 1 << 20
rng = np.random.default_rng(1)
big = (torch.rand(S * N * 2, device="cuda") * 2 - 1) / S
def run(tag, ns, filt="low_pass"):
    p = rh.ResampleLowpassMix(44100, 48000, 2, None, filt, 200, 0.5, max_sources=S, max_in_frames=N)
    p.set_sources([big[s * N * 2: s * N * 2 + 2 * int(n)] for s, n in enumerate(ns)])
    out = torch.empty(p.out_frames * 2, device="cuda")
    p.autotune(out)
    for _ in range(3):
        p.run(out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        p.run(out)
    e1.record()
    torch.cuda.synchronize()
    p.check_status()
    ms = e0.elapsed_time(e1) / 10
    diff = None
    if filt and len(set(int(n) for n in ns)) > 1:  # against the general kernel alone on the same batch
        p2 = rh.ResampleLowpassMix(44100, 48000, 2, None, filt, 200, 0.5, max_sources=S, max_in_frames=N, force_general=1)
        p2.set_sources([big[s * N * 2: s * N * 2 + 2 * int(n)] for s, n in enumerate(ns)])
        ref = p2.run().clone()
        p2.check_status()
        diff = float((out[: ref.numel()] - ref).abs().max())
        p2.close()
    byt = 8.0 * float(sum(ns)) + 8.0 * p.out_frames
    g = p.geometry()
    print(json.dumps({"batch": tag, "filter": filt, "ms": round(ms, 4), "GBps": round(byt / ms / 1e6), "frac": round(byt / ms / 1e6 / 8000, 3), "kernel": "wave" if g["general_kernel"] else "fast",
                      "R": g["frames_per_lane"], "NS": g["ring_stages"], "max_abs_diff_vs_general_kernel": diff}))
    p.close()
if len(sys.argv) > 1:  # one case, no autotune noise: python tools/bench_ragged.py short [R]
    os.environ.setdefault("X", "1")
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    p = rh.ResampleLowpassMix(44100, 48000, 2, None, "low_pass", 200, 0.5, max_sources=S, max_in_frames=N, frames_per_lane=R)
    ns = [N] * (S - 1) + [N // 2]
    p.set_sources([big[s * N * 2: s * N * 2 + 2 * int(n)] for s, n in enumerate(ns)])
    out = torch.empty(p.out_frames * 2, device="cuda")
    for _ in range(6):
        p.run(out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        p.run(out)
    e1.record()
    torch.cuda.synchronize()
    print("ms per run", e0.elapsed_time(e1) / 10)
    p.check_status()
    print(p.geometry())
    print('phase ticks (last kernel launched) gran/stage/wait/carry/run/scan+rest:', p.phase_cycles())
    sys.exit(0)
run("equal", [N] * S)
run("one short", [N] * (S - 1) + [N // 2])
run("uniform [N/2, N]", rng.integers(N // 2, N + 1, S))
run("uniform [0, N]", rng.integers(0, N + 1, S))
run("uniform [N/2, N], no filter", rng.integers(N // 2, N + 1, S), None)
run("equal, no filter", [N] * S, None)

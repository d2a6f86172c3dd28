"""BASELINE config 3 timing: 64 sources x 2 Mi stereo samples, reverb(65 536 samples, 0.3) -> Spatial,
fused (rh_reverb_spatial) vs the two unfused ops.   python tools/bench_config3.py   (GPU box)"""
import json, sys, time
sys.path.insert(0, ".")
import torch
import rodio_amd as G

G.init(0)
S, n = 64, 2 << 20
x = (torch.rand((S, n), device="cuda") * 2 - 1) * 0.25
em = [[0.5 + 0.01 * s, 0, 1] for s in range(S)]
d = G.delay_samples(682_666_667, 48000, 2)
out = torch.empty((S, n + d), device="cuda")


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


gd = G.spatial_gains_batch(em, [-1, 0, 0], [1, 0, 0])
fused = timed(lambda: G.reverb_spatial_batch(x, 48000, 682_666_667, 0.3, None, None, None, out=out, gains_dev=gd))


def unfused():
    for s in range(S):
        G.Spatial(G.GpuSource(x[s], 2, 48000).reverb(682_666_667, 0.3), em[s], [-1, 0, 0], [1, 0, 0])


two = timed(unfused, reps=3)
alg = 4 * S * n + 4 * S * (n + d)  # read L, write L+D per source
print(json.dumps({"config": "reverb(65536)+spatial, 64 sources x 2Mi samples", "fused_ms": fused, "unfused_ms": two,
                  "algorithmic_bytes": alg, "fused_GBps": alg / fused / 1e6, "frac_of_8TBps": alg / fused / 1e6 / 8000,
                  "note": "gains precomputed on the device; torch events around 20 launches"}))

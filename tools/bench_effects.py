"""Timing of the per-stream recurrences (limiter, AGC, sequential / time-parallel biquad) on batches of stereo streams.
   python tools/bench_effects.py [S frames]      (GPU box)   8 B of HBM traffic per sample (4 in + 4 out)."""
import json, sys
sys.path.insert(0, ".")
import torch
import rodio_amd as G

G.init(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
x = (torch.rand((S, 2 * n), device="cuda") * 2 - 1) * 0.9
co = G.biquad_coeffs("low_pass", 200, 0.5, 48000)
out = torch.empty_like(x)


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


alg = 8 * S * 2 * n
res = {"streams": S, "frames": n, "algorithmic_bytes": alg}
for name, fn, reps in (("limit", lambda: G.limit_batch(x, 2, 48000, out=out), 3),
                       ("agc", lambda: G.agc_batch(x, 48000, out=out), 2),
                       ("biquad_mode0", lambda: G.biquad_batch(x, co, mode=0), 2),
                       ("biquad_mode1", lambda: G.biquad_batch(x, co, mode=1), 10)):
    if len(sys.argv) > 3 and name not in sys.argv[3:]:
        continue
    ms = timed(fn, reps)
    res[name] = {"ms": round(ms, 4), "GBps": round(alg / ms / 1e6, 1), "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 5)}
print(json.dumps(res))

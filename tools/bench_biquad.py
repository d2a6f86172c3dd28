"""rh_biquad mode 0 (sequential, bit-exact) vs mode 1 (time-parallel) on 64 stereo streams x 1 Mi frames.  (GPU box)"""
import json, sys
sys.path.insert(0, ".")
import torch
import rodio_amd as G

G.init(0)
S, n = 64, 1 << 20
x = (torch.rand((S, 2 * n), device="cuda") * 2 - 1) * 0.25
co = G.biquad_coeffs("low_pass", 200, 0.5, 48000)


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t1 = timed(lambda: G.biquad_batch(x, co, mode=1), 10)
t0 = timed(lambda: G.biquad_batch(x, co, mode=0), 2)
alg = 8 * S * 2 * n
print(json.dumps({"streams": S, "frames": n, "mode0_ms": t0, "mode1_ms": t1, "mode1_GBps": alg / t1 / 1e6, "speedup": t0 / t1}))

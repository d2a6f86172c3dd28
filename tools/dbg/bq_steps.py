import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import rodio_amd as rh
rh.init(0)
S, n = 64, 1 << 20
x = torch.from_numpy(np.stack([(np.random.default_rng(4321 + s).uniform(-1, 1, 2 * n) * 0.9).astype(np.float32) for s in range(S)])).cuda()
out = torch.empty_like(x)
co = rh.biquad_coeffs("low_pass", 200, 0.5, 48000)
ref = None
ts = []
for k in range(48):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rh.biquad_batch(x, co, mode=1, out=out)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
    if ref is None:
        ref = out.clone()
    elif not torch.equal(ref, out):
        print("step", k, "differs from step 0: max", float((ref - out).abs().max()))
print("ms per call (synchronised):", [round(t, 2) for t in ts])
print("async status", rh.async_status())
# back to back, as bench.py does it
for steps in (20, 30, 40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        rh.biquad_batch(x, co, mode=1, out=out)
    torch.cuda.synchronize()
    print(steps, "back to back:", round((time.perf_counter() - t0) * 1e3 / steps, 4), "ms per call; equal to step 0:", bool(torch.equal(ref, out)), "status", rh.async_status())

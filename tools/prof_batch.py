"""Batch mode of the fused kernel (what rh_biquad mode 1 runs), timed and -- with an RH_PHASE_PROFILE build
(RODIO_HIP_LIB=variants/librodio_hip_prof.so) -- split into phases.   python tools/prof_batch.py [S frames R]"""
import json, sys
sys.path.insert(0, ".")
import torch
import rodio_amd as G
G.init(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
R = int(sys.argv[3]) if len(sys.argv) > 3 else 12
x = (torch.rand((S, 2 * n), device="cuda") * 2 - 1) * 0.25
p = G.ResampleLowpassMix(48000, 48000, 2, None, "low_pass", 200, 0.5, max_sources=S, max_in_frames=n, frames_per_lane=R)
p.set_sources([x[s] for s in range(S)])
out = p.run_batch(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    p.run_batch()
e1.record(); torch.cuda.synchronize()
p.check_status()
ms = e0.elapsed_time(e1) / 10
print(json.dumps({"S": S, "frames": n, "R": R, "ms": ms, "frac": 16 * S * n / ms / 1e6 / 8000, "geometry": p.geometry(), "phase_cycles[gran,stage,wait,carry,run,rest]": p.phase_cycles()}))

"""Phase cycles of the limiter kernel (RH_LIMIT_PROFILE build): RODIO_HIP_LIB=variants/librodio_hip_lprof.so python tools/prof_limit.py [S frames]"""
import ctypes as C, json, sys
sys.path.insert(0, ".")
import torch
import rodio_amd as G
from rodio_amd import _lib
G.init(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
x = (torch.rand((S, 2 * n), device="cuda") * 2 - 1) * 0.9
out = torch.empty_like(x)
fn = _lib.lib.rh_limit_phase_cycles
fn.restype = C.c_int32
buf = (C.c_double * 8)()
G.limit_batch(x, 2, 48000, out=out); fn(buf)
reps = 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    G.limit_batch(x, 2, 48000, out=out)
e1.record(); torch.cuda.synchronize()
st = fn(buf)
names = ["load", "gain+segment", "lookback_I", "integrator", "lookback_P", "gain_stage", "store"]
tot = sum(buf[:7]) or 1
print(json.dumps({"status": st, "ms": e0.elapsed_time(e1) / reps, "cycles_share": {k: round(buf[i] / tot, 3) for i, k in enumerate(names)},
                  "counter_units_per_wave_share": {k: round(buf[i] / reps / (S * -(-n // (64 * 16))), 0) for i, k in enumerate(names)}}))  # shares of 64 x 16 frames (the shipped R = 16)

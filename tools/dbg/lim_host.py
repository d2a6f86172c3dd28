import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rodio_amd as rh
from rodio_amd import _lib
rh.init(0)
S, n = 64, 1 << 20
x = (torch.rand((S, 2 * n), device="cuda") * 2 - 1) * 0.9
out = torch.empty_like(x)
lib = _lib.lib
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = _lib.LimitParams(-1.0, 4.0, 5_000_000, 100_000_000)
def call():
    _lib.check(lib.rh_limit(C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()), n, 2, 48000, S, C.byref(p), None, st), "rh_limit")
for _ in range(3): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for reps in (1, 20, 20):
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps): call()
    e1.record(); th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("rh_limit x%d: host per call %.3f ms, events per call %.4f ms" % (reps, th / reps * 1e3, e0.elapsed_time(e1) / reps), flush=True)
# per-call event pairs, as bench.py times
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for a, b in evs:
    a.record(); call(); b.record()
torch.cuda.synchronize()
print("per-call event pairs: %.4f ms" % (sum(a.elapsed_time(b) for a, b in evs) / 20))
lib.rh_event_create.restype = C.c_int32

import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rodio_amd as rh
from rodio_amd import _lib
rh.init(0)
S, n = 64, 1 << 20
x = (torch.rand((S, 2 * n), device="cuda") * 2 - 1) * 0.9
out = torch.empty_like(x)
co = rh.biquad_coeffs("low_pass", 200, 0.5, 48000)
lib = _lib.lib
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def call():
    _lib.check(lib.rh_biquad(C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()), n, 2, S, co.ctypes.data_as(_lib.f32p), None, 1, st), "rh_biquad")
for _ in range(3): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(20): call()
e1.record(); th = time.perf_counter() - t0
torch.cuda.synchronize()
print("prealloc: host per call %.3f ms, events per call %.3f ms" % (th / 20 * 1e3, e0.elapsed_time(e1) / 20))
t0 = time.perf_counter(); e0.record()
for _ in range(20): o = rh.biquad_batch(x, co, mode=1)
e1.record(); th = time.perf_counter() - t0
torch.cuda.synchronize()
print("biquad_batch (empty_like per call): host per call %.3f ms, events per call %.3f ms" % (th / 20 * 1e3, e0.elapsed_time(e1) / 20))
keep = {}
t0 = time.perf_counter(); e0.record()
for _ in range(20): keep["o"] = rh.biquad_batch(x, co, mode=1)
e1.record(); th = time.perf_counter() - t0
torch.cuda.synchronize()
print("... result kept in a dict: host per call %.3f ms, events per call %.3f ms" % (th / 20 * 1e3, e0.elapsed_time(e1) / 20))
lim = torch.empty_like(x)
t0 = time.perf_counter(); e0.record()
for _ in range(20): rh.limit_batch(x, 2, 48000, out=lim)
e1.record(); th = time.perf_counter() - t0
torch.cuda.synchronize()
print("limit_batch: host per call %.3f ms, events per call %.3f ms" % (th / 20 * 1e3, e0.elapsed_time(e1) / 20))

import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
x = (torch.rand((64, 2 << 20), device="cuda") * 2 - 1) * 0.9
def t_alloc(tag):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): o = torch.empty_like(x)
    print(tag, "empty_like per call %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)
t_alloc("before anything:")
import rodio_amd as rh
rh.init(0)
t_alloc("after rh.init:")
co = rh.biquad_coeffs("low_pass", 200, 0.5, 48000)
o = rh.biquad_batch(x, co, mode=1); torch.cuda.synchronize()
t_alloc("after one biquad_batch:")
def t_bq(tag):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): o = rh.biquad_batch(x, co, mode=1)
    th = time.perf_counter() - t0; torch.cuda.synchronize()
    print(tag, "biquad_batch host per call %.3f ms" % (th / 20 * 1e3), flush=True)
t_bq("loop:")
t_alloc("after the loop:")
print(torch.cuda.memory_stats()["num_device_alloc"], torch.cuda.memory_stats()["num_device_free"])

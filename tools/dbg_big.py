"""Extremes of the scan kernels' index arithmetic: one very long stream, very many very short ones.  python tools/dbg_big.py"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np, torch
import rodio_amd as G
from rodio_amd import _lib
from oracle import rodio_oracle as O
G.init(0)

def biquad(x, frames, ch, S, co, mode):
    out = torch.empty_like(x)
    _lib.check(_lib.lib.rh_biquad(C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()), frames, ch, S, co.ctypes.data_as(_lib.f32p), None, mode,
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rh_biquad")
    return out

co = G.biquad_coeffs("low_pass", 200, 0.5, 48000)
for S, frames in [(1, 1 << 26), (100000, 16), (3, (1 << 24) + 4), (65536, 128)]:
    rng = np.random.default_rng(S)
    x = (rng.uniform(-1, 1, (S, frames * 2)) * 0.4).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    lim = G.limit_batch(xd, 2, 48000).cpu().numpy()
    par = biquad(xd, frames, 2, S, co, 1)
    seq = biquad(xd, frames, 2, S, co, 0)
    torch.cuda.synchronize()
    G.async_status()
    e_b = float((par - seq).abs().max())
    worst = 0.0
    for s in sorted(set([0, S - 1, S // 2])):
        ref = O.TestSource(x[s], 2, 48000).limit().collect()
        worst = max(worst, float(np.max(np.abs(lim[s] - ref))))
    print(f"{S} x {frames}: limiter vs oracle {worst:.2e}, biquad mode 1 vs mode 0 {e_b:.2e}", "OK" if worst <= 1e-5 and e_b <= 1e-5 else "FAIL", flush=True)

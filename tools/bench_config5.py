"""BASELINE config 5 timing: i16 -> f32 SampleTypeConverter and 6 -> 2 ChannelCountConverter on the music.wav
excerpt tiled to block scale (the asset itself is 1.8 MB: launch-bound).   python tools/bench_config5.py  (GPU box)"""
import ctypes as C, json, os, sys
sys.path.insert(0, ".")
import numpy as np, torch
import rodio_amd as G
from rodio_amd import _lib

G.init(0)
lib = _lib.lib
ex = np.load(os.path.join("tests", "golden", "music_excerpt_i16.npy"))
reps = 4096
i16 = torch.from_numpy(np.tile(ex, reps)).cuda()          # 128 Mi samples, 256 MiB
n = i16.numel()
f32 = torch.empty(n, device="cuda", dtype=torch.float32)
frames6 = n // 6
out2 = torch.empty(frames6 * 2, device="cuda", dtype=torch.float32)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t_cvt = timed(lambda: lib.rh_convert_i16_to_f32(C.c_void_p(f32.data_ptr()), C.c_void_p(i16.data_ptr()), n, st))
t_ch = timed(lambda: lib.rh_channels_convert(C.c_void_p(out2.data_ptr()), C.c_void_p(f32.data_ptr()), frames6, 6, 2, st))
ref = torch.from_numpy(np.load(os.path.join("tests", "golden", "music_excerpt_f32.npy"))).cuda()
assert torch.equal(f32[: ref.numel()], ref) and torch.equal(f32[-ref.numel():], ref)
b_cvt, b_ch = 6 * n, 32 * frames6  # 2 B in + 4 B out per sample; 24 B in + 8 B out per frame
print(json.dumps({"samples": n, "i16_to_f32_ms": t_cvt, "i16_to_f32_GBps": b_cvt / t_cvt / 1e6, "i16_to_f32_frac_of_8TBps": b_cvt / t_cvt / 1e6 / 8000,
                  "channels_6to2_ms": t_ch, "channels_6to2_GBps": b_ch / t_ch / 1e6, "channels_6to2_frac_of_8TBps": b_ch / t_ch / 1e6 / 8000}))

"""Diagnostics (variants/librodio_hip_cls_diag9.so: tools/build_variant.sh cls_diag9 -DRH_CLS_DIAG=9): where the tiles of k_rlm_chunk_classes spend a launch.
RODIO_HIP_LIB=variants/librodio_hip_cls_diag9.so python tools/cls_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodio_amd as rh
from rodio_amd import _lib

rh.init(0)
S, N = 256, 1 << 20
g = torch.Generator(device="cuda").manual_seed(1)
data = (torch.rand((S, 2 * N), device="cuda", generator=g) * 2 - 1) * 0.05
classes = [("low_pass", 200), ("low_pass", 1000), ("high_pass", 1000), ("low_pass", 4000)]
pc = rh.ResampleLowpassMix(44100, 48000, 2, None, "low_pass", 200, 0.5, max_sources=S, max_in_frames=N)
pc.set_filters([classes[s % 4] for s in range(S)])
pc.set_sources([data[s] for s in range(S)])
for _ in range(3):
    out = pc.run()
torch.cuda.synchronize()
print("mix_first", pc.geometry()["mix_first"])
lib = C.CDLL(_lib.LIB_PATH)
n = 2048 * 8 * 8
buf = (C.c_ulonglong * n)()
assert lib.rh_debug_cls_stamps(buf, C.c_size_t(n)) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 8, 8)[:1024, :4, :6].astype(np.float64) * 0.01  # us (100 MHz wall clock)
t0 = st[:, 0, 0].min() - 77.0
st -= t0
names = ["loader: sum done", "loader: image written", "wave1: image seen", "wave1: halo there", "wave1: look-back done", "wave1: stores issued"]
for k in range(4):
    print(f"class {k}: " + "  ".join(f"{names[i].split(': ')[1]} {st[:, k, i].mean():7.1f} (max {st[:, k, i].max():7.1f})" for i in range(6)))
    print(f"   loader waits for its image {np.mean(st[:, k, 1] - st[:, k, 0]):6.2f} (max {np.max(st[:, k, 1] - st[:, k, 0]):6.1f});  wave1: seen-written {np.mean(st[:, k, 2] - st[:, k, 1]):6.2f}, halo wait {np.mean(st[:, k, 3] - st[:, k, 2]):6.2f} (max {np.max(st[:, k, 3] - st[:, k, 2]):6.1f}), "
          f"taps..look-back {np.mean(st[:, k, 4] - st[:, k, 3]):6.2f} (max {np.max(st[:, k, 4] - st[:, k, 3]):6.1f}), stores {np.mean(st[:, k, 5] - st[:, k, 4]):6.2f}")
print("spread of the loaders' ends per class (max - min over tiles):", [float(st[:, k, 0].max() - st[:, k, 0].min()) for k in range(4)])
print("kernel span ~", float(st[:, 3, 5].max()), "us")

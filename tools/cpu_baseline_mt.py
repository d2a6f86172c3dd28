"""The generous CPU number of SURVEY.md 8(d): the oracle's iterator pipeline with the sources sharded over all host
cores (rodio's own mixer is single-threaded by construction; this is an upper bound, not what rodio does), partial mixes
summed at the end.   python tools/cpu_baseline_mt.py [sources] [frames]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rodio_oracle as O

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
cores = os.cpu_count()
rng = np.random.default_rng(1234)
x = (rng.uniform(-1, 1, (S, N, 2)) / S).astype(np.float32)
t0 = time.perf_counter()
one = O.pipeline_resample_lowpass_mix(x, 44100, 48000, O.SPAN_NONE, 200, 0.5, want_output=False)
t1 = time.perf_counter() - t0
shards = [x[i::cores] for i in range(cores)]
t0 = time.perf_counter()
with ThreadPoolExecutor(cores) as ex:  # the ctypes call releases the GIL
    outs = list(ex.map(lambda sh: O.pipeline_resample_lowpass_mix(np.ascontiguousarray(sh), 44100, 48000, O.SPAN_NONE, 200, 0.5, want_output=False), shards))
tm = time.perf_counter() - t0
print(f"{S} sources x {N} frames: 1 thread {S * N * 2 / t1 / 1e6:.1f} Msamples/s, {cores} threads {S * N * 2 / tm / 1e6:.1f} Msamples/s ({t1 / tm:.1f}x)")

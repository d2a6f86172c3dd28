"""Debug helper: where does the fused kernel first deviate from the oracle?  (GPU box)"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import rodio_amd as G
from oracle import rodio_oracle as O
sys.path.insert(0, "tests")
from test_gpu_parity import _oracle_pipeline, _gpu_pipeline, rnd
G.init(0)
for R, NS, S, freq in [(8, 3, 1, 200), (8, 3, 4, 200), (8, 3, 16, 200), (4, 3, 4, 200), (8, 2, 16, 200), (8, 4, 16, 200), (8, 3, 6, 20)]:
    n = 60000
    xs = [rnd(600 + s, 2 * n, 1.0 / 16) for s in range(S)]
    ref = _oracle_pipeline(O, xs, 44100, 48000, None, "low_pass", freq)
    out, geo = _gpu_pipeline(G, xs, 44100, 48000, None, "low_pass", freq, frames_per_lane=R, ring_stages=NS)
    d = np.abs(out - ref).reshape(-1, 2).max(axis=1)
    bad = np.nonzero(d > 1e-6)[0]
    L = 64 * R
    print(f"R={R} NS={NS} S={S} f={freq} J={geo['lookback_tiles']} tiles={geo['n_tiles']} max={d.max():.3e} first_bad_frame={bad[0] if len(bad) else None} "
          f"(tile {bad[0] // L if len(bad) else None}, lane {(bad[0] % L) // R if len(bad) else None}) n_bad={len(bad)}")
    if len(bad):
        pt = np.array([d[t * L:(t + 1) * L].max() for t in range(min(geo['n_tiles'], 12))])
        print("   per-tile max err:", np.array2string(pt, precision=2))

import os, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rodio_amd as G
from oracle import rodio_oracle as O
G.init(0)
x = np.concatenate([np.load("tests/golden/music_excerpt_f32.npy"), (np.random.default_rng(77).uniform(-1, 1, 2 * 132300) * 0.3).astype(np.float32)])
stages = [
    ("high_pass", lambda s: s.high_pass(300)),
    ("amplify", lambda s: s.amplify(1.2)),
    ("speed", lambda s: s.speed(0.9)),
    ("agc", lambda s: s.automatic_gain_control()),
    ("delay", lambda s: s.delay(500_000_000)),
    ("fade_in", lambda s: s.fade_in(2_000_000_000)),
    ("take", lambda s: s.take_duration(10_000_000_000, fade_out=True)),
    ("reverb", lambda s: s.reverb(50_000_000, 0.3)),
]
for k in range(1, len(stages) + 1):
    a, b = O.TestSource(x, 2, 44100), G.TestSource(x, 2, 44100)
    for name, f in stages[:k]:
        a, b = f(a), f(b)
    ra, rb = a.collect(), b.collect()
    n = min(len(ra), len(rb))
    d = np.abs(ra[:n] - rb[:n])
    print(stages[k - 1][0], "len", len(ra), len(rb), "rate", a.sample_rate(), b.sample_rate(), "max err", d.max() if n else 0, "first bad", int(np.argmax(d > 1e-5)) if (d > 1e-5).any() else None)
a, b = O.TestSource(x, 2, 44100), G.TestSource(x, 2, 44100)
for name, f in stages:
    a, b = f(a), f(b)
ra = O.UniformSourceIterator(a, 2, 40000).collect(); rb = G.UniformSourceIterator(b, 2, 40000).collect()
print("uniform len", len(ra), len(rb))
n = min(len(ra), len(rb)); d = np.abs(ra[:n] - rb[:n]); bad = np.nonzero(d > 1e-6)[0]
print("bad count", len(bad), "first", bad[:6], "last", bad[-6:], "max", d.max())

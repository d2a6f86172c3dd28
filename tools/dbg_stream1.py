import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodio_amd as G
from oracle import rodio_oracle as O
def rnd(seed, n, scale=1.0):
    return (np.random.default_rng(seed).uniform(-1, 1, n) * scale).astype(np.float32)
for S in (1, 2, 3):
    for R in (4, 8):
        n = 30000
        xs = [rnd(10 + s, 2 * n, 0.1) for s in range(S)]
        m = O.Mixer(2, 48000)
        for x in xs:
            m.add(O.UniformSourceIterator(O.TestSource(x, 2, 44100), 2, 48000).low_pass(300))
        ref = m.collect()
        p = G.ResampleLowpassMix(44100, 48000, 2, None, "low_pass", 300, 0.5, max_sources=S, max_in_frames=8192 + 4096, frames_per_lane=R)
        xd = [torch.from_numpy(x).cuda() for x in xs]
        p.stream_begin()
        outs = []
        cuts = list(range(0, n, 8192)) + [n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            outs.append(p.stream_feed_v([x[2 * a: 2 * b] for x in xd], [b >= n] * S))
            try:
                p.check_status()
            except Exception as e:
                print("  block", a, b, "out", len(outs[-1]) // 2, "->", e)
        while True:  # the final call emits what is left once all have ended
            o = p.stream_feed_v([x[:0] for x in xd], [True] * S) if sum(len(o) for o in outs) < len(ref) else None
            if o is None or len(o) == 0:
                break
            outs.append(o)
        got = torch.cat(outs).cpu().numpy()
        err = float(np.max(np.abs(got - ref))) if len(got) == len(ref) else (len(got), len(ref))
        print("S", S, "R", R, "err", err, "geo", p.geometry()["frames_per_lane"], [len(o) // 2 for o in outs][:6])
        p.close()

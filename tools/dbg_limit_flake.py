"""Flake hunt for the small-block limiter path: the stream of tests' chain case, block by block with a carried state, on a
non-blocking stream, every block against the oracle.  python tools/dbg_limit_flake.py [iters] [block_frames]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import rodio_amd as G
from oracle import rodio_oracle as O
G.init(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 777
n = 30000
x = (np.random.default_rng(3204).uniform(-1, 1, 2 * n) * 0.9).astype(np.float32)
ref = O.TestSource(x, 2, 48000).limit().collect()
xd = torch.from_numpy(x).cuda()
side = torch.cuda.Stream()
bad = 0
for it in range(iters):
    with torch.cuda.stream(side):
        state = torch.zeros((1, 4), device="cuda")
        outs, states = [], []
        for a in range(0, n, B):
            b = min(n, a + B)
            outs.append(G.limit_batch(xd[2 * a: 2 * b][None, :].contiguous(), 2, 48000, state=state))
            states.append(state.clone())
        side.synchronize()
        got = torch.cat(outs, dim=1).cpu().numpy()[0]
    err = np.abs(got - ref)
    if err.max() > 1e-5:
        bad += 1
        i = int(np.argmax(err > 1e-5))
        blk, off = divmod(i // 2, B)
        print(f"iter {it}: first bad sample {i} = block {blk} frame {off} (tile {off // 512}), max err {err.max():.3g}; state after block {blk-1}: {states[blk-1].cpu().numpy() if blk else None}, after block {blk}: {states[blk].cpu().numpy()}")
print("bad iterations:", bad, "of", iters)

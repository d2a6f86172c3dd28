"""rh_agc paths against the reference-order kernel and the oracle on a few shapes (GPU box): prints max |difference| per path.
    python tools/agc_check.py [S n]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rodio_amd as G
from oracle import rodio_oracle as O

G.init(0)
shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(5, 60000), (2, 4097), (3, 100), (70, 12288)]
for S, n in shapes:
    rng = np.random.default_rng(S * 1000 + n)
    xs = [(rng.uniform(-1, 1, n) * np.abs(np.sin(np.arange(n) / (3000.0 + 100 * s)))).astype(np.float32) for s in range(S)]
    x = torch.from_numpy(np.stack(xs)).cuda()
    res = {}
    for name, env in (("chain", {}), ("vec", {"RH_AGC_VEC": "1"}), ("seq", {"RH_AGC_SEQ": "1"})):
        os.environ.update(env)
        G.init(0)  # the library reads its variables in rh_init
        res[name] = G.agc_batch(x, 48000).cpu().numpy()
        for k in env:
            del os.environ[k]
        G.init(0)
    ref = O.TestSource(xs[0], 1, 48000).automatic_gain_control().collect()
    line = f"S={S} n={n}: seq-oracle {np.max(np.abs(res['seq'][0] - ref)):.2e}"
    for name in ("chain", "vec"):
        d = np.abs(res[name] - res["seq"])
        line += f" | {name}-seq {d.max():.2e} at {np.unravel_index(int(d.argmax()), d.shape)}"
    print(line, flush=True)

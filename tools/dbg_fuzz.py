import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import rodio_amd as G
from oracle import rodio_oracle as O
import test_gpu_fuzz as T
from scipy.signal import lfilter
G.init(0)
for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(9500 + seed)
    c = T._case(rng)
    c["span"] = 0
    ref = T._oracle(O, c)
    nmax = max(c["ns"])
    block = int(rng.choice([64, 500, 4096, nmax + 5]))
    # f64 truth
    co = O.blt_coeffs(c["filt"], c["freq"], 0.5, c["to"]).astype(np.float64)
    truth = np.zeros(len(ref) // 2 * 2).reshape(-1, 2)
    for i, x in enumerate(c["xs"]):
        src = O.TestSource(x, 2, c["frm"])
        if c["gains"] is not None:
            src = src.amplify(float(c["gains"][i]))
        r = O.UniformSourceIterator(src, 2, c["to"]).collect().astype(np.float64).reshape(-1, 2)
        if len(r):
            truth[: len(r)] += lfilter(co[:3], [1.0, co[3], co[4]], r, axis=0)
    truth = truth.reshape(-1)
    p = T._make(G, c, max_in=max(block + 4096, nmax))
    if c["gains"] is not None:
        p.set_gains(c["gains"])
    xd = [torch.from_numpy(x).cuda() if len(x) else torch.empty(0, device="cuda") for x in c["xs"]]
    p.set_sources(xd)
    one = p.run().cpu().numpy().copy()
    p.stream_begin()
    outs = []
    a = 0
    while True:
        b = a + int(rng.integers(1, block + 1))
        outs.append(p.stream_feed_v([x[2 * min(a, n): 2 * min(b, n)] for x, n in zip(xd, c["ns"])], [n <= b for n in c["ns"]]))
        a = b
        if a >= nmax:
            break
    got = torch.cat(outs).cpu().numpy()
    e = lambda u, v: float(np.max(np.abs(u - v)))
    print(seed, c["filt"], c["freq"], c["frm"], c["to"], "S", c["S"], "blocks", len(outs), "peak", float(np.max(np.abs(truth))),
          "| one-ref", e(one, ref), "stream-ref", e(got, ref), "stream-one", e(got, one), "| ref-f64", e(ref, truth), "one-f64", e(one, truth), "stream-f64", e(got, truth))

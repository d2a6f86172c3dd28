"""Randomised parity of the fused path (seeded, deterministic): source counts, ragged lengths (empty, one frame, ...),
rates, span chunking, filters, gains, launch geometry, one-shot and block-streamed -- against the oracle's iterator
chains.  The reference tests its converter with quickcheck (sample_rate.rs:252-334); this is the same idea one level up."""
import numpy as np
import pytest

import os

pytestmark = pytest.mark.gpu
N_ONE = int(os.environ.get("RH_FUZZ_ONE", "48"))  # more seeds for a bug hunt: RH_FUZZ_ONE=1000 RH_FUZZ_STREAM=500
N_STREAM = int(os.environ.get("RH_FUZZ_STREAM", "32"))
TOL = 1e-5
RATES = [(44100, 48000), (48000, 44100), (22050, 48000), (32000, 48000), (96000, 48000), (48000, 48000), (8000, 11025), (44100, 40000),
         (192000, 48000), (192000, 44100), (176400, 48000)]  # ... and the steep ones (from/to <= 4.5)


@pytest.fixture(scope="module")
def G(rh):
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    rh.init(0)
    return rh


def _case(rng):
    S = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17, 24]))
    frm, to = RATES[int(rng.integers(len(RATES)))]
    nmax = int(rng.choice([300, 2000, 9000, 20000]))
    kind = int(rng.integers(4))
    if kind == 0:
        ns = [nmax] * S  # equal lengths: k_rlm_fast
    elif kind == 1:
        ns = [int(v) for v in rng.integers(0, nmax + 1, S)]
    elif kind == 2:
        ns = [int(rng.choice([0, 1, 2, 3, nmax, nmax - 1, nmax // 2])) for _ in range(S)]
    else:
        ns = [nmax] * S
        ns[int(rng.integers(S))] = int(rng.integers(0, nmax))
    if max(ns) == 0:
        ns[0] = 1
    filt = [None, "low_pass", "high_pass"][int(rng.integers(3))]
    freq = int(rng.choice([100, 200, 1000, 3000])) if filt else 0
    span = int(rng.choice([0, 0, 0, 32768, 4096])) if frm != to else 0
    gains = rng.choice([1.0, 0.5, 0.25, 1.5, 0.0, -1.0], S).astype(np.float32) if rng.random() < 0.6 else None
    R = int(rng.choice([0, 0, 3, 4, 6, 8, 9, 12]))
    scale = 0.5 / S if filt else 1.0
    xs = [(rng.uniform(-1, 1, 2 * n) * scale).astype(np.float32) for n in ns]
    return dict(S=S, frm=frm, to=to, ns=ns, filt=filt, freq=freq, span=span, gains=gains, R=R, xs=xs)


def _oracle(O, c):
    m = O.Mixer(2, c["to"])
    for i, x in enumerate(c["xs"]):
        src = O.SpanSource(x, 2, c["frm"], c["span"]) if c["span"] else O.TestSource(x, 2, c["frm"])
        if c["gains"] is not None:
            src = src.amplify(float(c["gains"][i]))
        u = O.UniformSourceIterator(src, 2, c["to"])
        if c["filt"] == "low_pass":
            u = u.low_pass(c["freq"])
        elif c["filt"] == "high_pass":
            u = u.high_pass(c["freq"])
        m.add(u)
    return m.collect()


def _make(G, c, max_in=None):
    try:
        return G.ResampleLowpassMix(c["frm"], c["to"], 2, c["span"] or None, c["filt"], c["freq"], 0.5, max_sources=c["S"], max_in_frames=max_in or max(c["ns"]), frames_per_lane=c["R"])
    except G.RhError:  # no variant with that many frames per lane for this ratio: let the library choose
        return G.ResampleLowpassMix(c["frm"], c["to"], 2, c["span"] or None, c["filt"], c["freq"], 0.5, max_sources=c["S"], max_in_frames=max_in or max(c["ns"]))


def _truth(O, c):
    """f64 evaluation of the filter on the (bit-exact) f32 converted streams: what the reference's f32 recurrence and the
    GPU's time-parallel evaluation both approximate."""
    from scipy.signal import lfilter

    co = O.blt_coeffs(c["filt"], c["freq"], 0.5, c["to"]).astype(np.float64)
    acc = None
    for i, x in enumerate(c["xs"]):
        src = O.SpanSource(x, 2, c["frm"], c["span"]) if c["span"] else O.TestSource(x, 2, c["frm"])
        if c["gains"] is not None:
            src = src.amplify(float(c["gains"][i]))
        r = O.UniformSourceIterator(src, 2, c["to"]).collect().astype(np.float64).reshape(-1, 2)
        y = lfilter(co[:3], [1.0, co[3], co[4]], r, axis=0) if len(r) else r
        if acc is None or len(y) > len(acc):
            acc, y = y.copy(), acc
        if y is not None and len(y):
            acc[: len(y)] += y
    return acc.reshape(-1)


# Cases whose |gpu - oracle| exceeded 1e-5 and were judged against the f64 truth instead: counted,
# printed, and bounded by test_fuzz_fallback_budget at the end of this file.
COMPARED = {"filtered": 0}
FALLBACKS = []


def _compare(tag, c, got, ref, O=None):
    assert len(got) == len(ref), (tag, len(got), len(ref), c["ns"], c["frm"], c["to"])
    if len(ref) == 0:
        return
    if c["filt"] is None:
        assert np.array_equal(got, ref), (tag, float(np.max(np.abs(got - ref))), c["S"], c["ns"], c["frm"], c["to"], c["span"], c["R"])
    else:
        err = float(np.max(np.abs(got - ref)))
        COMPARED["filtered"] += 1
        if err > TOL:  # the f32 reference recurrence drifts for poles next to 1 (high_pass(100) at 48 kHz: 2e-5 from exact)
            FALLBACKS.append((tag, c["filt"], c["freq"], c["to"], err))
            truth = _truth(O, c)
            e_gpu, e_ref = float(np.max(np.abs(got - truth))), float(np.max(np.abs(ref - truth)))
            assert e_gpu <= 1e-6 + 0.25 * e_ref and err <= e_ref + e_gpu + 1e-7, (tag, err, e_gpu, e_ref, c["S"], c["frm"], c["to"], c["filt"], c["freq"], c["R"])


@pytest.mark.parametrize("seed", range(N_ONE))
def test_fuzz_one_shot(G, O, seed):
    import torch

    c = _case(np.random.default_rng(9000 + seed))
    ref = _oracle(O, c)
    p = _make(G, c)
    if c["gains"] is not None:
        p.set_gains(c["gains"])
    p.set_sources([torch.from_numpy(x).cuda() if len(x) else torch.empty(0, device="cuda") for x in c["xs"]])
    got = p.run().cpu().numpy().copy()
    p.check_status()
    _compare("one-shot", c, got, ref, O)
    if p.geometry()["general_kernel"] == 0 and c["filt"]:  # equal lengths: the ragged-batch kernel must agree too
        p.close()
        p = G.ResampleLowpassMix(c["frm"], c["to"], 2, c["span"] or None, c["filt"], c["freq"], 0.5, max_sources=c["S"], max_in_frames=max(c["ns"]), force_general=1)
        if c["gains"] is not None:
            p.set_gains(c["gains"])
        p.set_sources([torch.from_numpy(x).cuda() for x in c["xs"]])
        got = p.run().cpu().numpy().copy()
        p.check_status()
        _compare("general", c, got, ref, O)
    p.close()


@pytest.mark.parametrize("seed", range(N_STREAM))
def test_fuzz_block_streaming(G, O, seed):
    import torch

    rng = np.random.default_rng(9500 + seed)
    c = _case(rng)
    c["span"] = 0  # continuous sources stream; spanned ones are converted span by span
    ref = _oracle(O, c)
    nmax = max(c["ns"])
    block = int(rng.choice([64, 500, 4096, nmax + 5]))
    p = _make(G, c, max_in=block + 4096)
    if c["gains"] is not None:
        p.set_gains(c["gains"])
    xd = [torch.from_numpy(x).cuda() if len(x) else torch.empty(0, device="cuda") for x in c["xs"]]
    p.stream_begin()
    outs = []
    a = 0
    while True:
        b = a + int(rng.integers(1, block + 1))
        outs.append(p.stream_feed_v([x[2 * min(a, n): 2 * min(b, n)] for x, n in zip(xd, c["ns"])], [n <= b for n in c["ns"]]))
        a = b
        if a >= nmax:
            break
    p.check_status()
    got = torch.cat(outs).cpu().numpy() if outs else np.zeros(0, np.float32)
    _compare(f"stream block<= {block}", c, got, ref, O)
    p.close()


def test_fuzz_fallback_budget():
    """How many filtered cases left the 1e-5 bound and were accepted on the f64 criterion.  Only the family whose
    REFERENCE recurrence is itself > 1e-5 from exact may do so: a high-pass with its poles next to z = 1 (cutoff <= 200 Hz),
    where y = x - (low-passed x) cancels in f32 (blt.rs:523-542,559).  Anything else fails here."""
    n = COMPARED["filtered"]
    print(f"[fuzz] filtered comparisons: {n}; beyond 1e-5 abs (accepted against the f64 truth): {len(FALLBACKS)}")
    for tag, filt, freq, to, err in FALLBACKS:
        print(f"    {tag}: {filt}({freq}) @ {to} Hz  |gpu-oracle| = {err:.3e}")
    stray = [f for f in FALLBACKS if not (f[1] == "high_pass" and f[2] <= 200)]
    assert not stray, stray
    assert len(FALLBACKS) <= max(2, n // 8), (len(FALLBACKS), n)

"""The fused kernels with mono frames (rh_rlm_config.channels = 1): `mixer::mixer(1, rate)` over mono sources --
UniformSourceIterator(src, 1, rate) [.low_pass(f)] and the ordered sum -- against the oracle's iterator chains.  The same
cases the stereo suites run (test_gpu_parity.py), on the channel count the kernels take as a template parameter."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def G(rh):
    import torch

    assert torch.cuda.is_available()
    rh.init(0)
    return rh


def rnd(seed, n, scale=1.0):
    return (np.random.default_rng(seed).uniform(-1, 1, n) * scale).astype(np.float32)


def _oracle(O, xs, frm, to, span, filt, freq, gains=None, ch=1):
    m = O.Mixer(ch, to)
    for i, x in enumerate(xs):
        src = O.TestSource(x, ch, frm) if not span else O.SpanSource(x, ch, frm, span)
        if gains is not None:
            src = src.amplify(float(gains[i]))
        u = O.UniformSourceIterator(src, ch, to)
        m.add(u.low_pass(freq) if filt == "low_pass" else u.high_pass(freq) if filt == "high_pass" else u)
    return m.collect()


def _check(got, ref, filt):
    assert len(got) == len(ref), (len(got), len(ref))
    if filt is None:
        assert np.array_equal(got, ref), int(np.argmax(got != ref))
    else:
        assert float(np.max(np.abs(got - ref))) <= TOL


@pytest.mark.parametrize("frm,to", [(44100, 48000), (48000, 44100), (48000, 48000), (22050, 48000), (96000, 44100), (192000, 44100)])
@pytest.mark.parametrize("filt,freq", [(None, 0), ("low_pass", 200), ("high_pass", 300)])
def test_mono_equal_length_batch(G, O, frm, to, filt, freq):
    import torch

    S, n = 7, 30000
    xs = [rnd(5000 + s, n, 0.12) for s in range(S)]
    ref = _oracle(O, xs, frm, to, None, filt, freq)
    p = G.ResampleLowpassMix(frm, to, 1, None, filt, freq, 0.5, max_sources=S, max_in_frames=n)
    p.set_sources([torch.from_numpy(x).cuda() for x in xs])
    assert p.geometry()["general_kernel"] == 0
    got = p.run().cpu().numpy()
    p.check_status()
    _check(got, ref, filt)
    p.close()


@pytest.mark.parametrize("R", [0, 4, 8, 12, 18, 20])
@pytest.mark.parametrize("filt,freq", [(None, 0), ("low_pass", 200)])
def test_mono_tile_sizes_and_gains(G, O, R, filt, freq):
    import torch

    S, n = 5, 70001
    xs = [rnd(5100 + s, n, 0.2) for s in range(S)]
    gains = np.array([1.0, 0.5, 1.7, 0.0, -0.25], dtype=np.float32)
    ref = _oracle(O, xs, 44100, 48000, None, filt, freq, gains)
    p = G.ResampleLowpassMix(44100, 48000, 1, None, filt, freq, 0.5, max_sources=S, max_in_frames=n, frames_per_lane=R)
    p.set_gains(gains)
    p.set_sources([torch.from_numpy(x).cuda() for x in xs])
    got = p.run().cpu().numpy()
    p.check_status()
    _check(got, ref, filt)
    p.close()


@pytest.mark.parametrize("filt,freq", [(None, 0), ("low_pass", 200)])
@pytest.mark.parametrize("span", [None, 32768, 3000])
def test_mono_ragged_batch_and_spans(G, O, filt, freq, span):
    import torch

    ns = [50000, 31000, 12345, 50000, 147, 0, 49999, 1, 32768, 40001]
    xs = [rnd(5200 + i, n, 0.1) for i, n in enumerate(ns)]
    ref = _oracle(O, xs, 44100, 48000, span, filt, freq)
    p = G.ResampleLowpassMix(44100, 48000, 1, span, filt, freq, 0.5, max_sources=len(ns), max_in_frames=max(ns))
    p.set_sources([torch.from_numpy(x).cuda() if len(x) else torch.empty(0, device="cuda") for x in xs])
    assert p.geometry()["general_kernel"] == 1
    got = p.run().cpu().numpy()
    p.check_status()
    _check(got, ref, filt)
    p.close()


def test_mono_ragged_filtered_batch_takes_the_sum_first_kernel_with_its_pairs(G, O):
    """Mono sources of different lengths, filtered, one shot: k_rlm_fast<RAG, SUMF, C = 1> with the pairs of ending sources inside it
    (round 4; mono batches took k_rlm_wave before).  Against the oracle and against the ragged-batch kernel on the same batch."""
    import torch

    n = 120000
    ns = [n] * 4 + [n - 900 * i - 7 for i in range(1, 40)] + [n // 2, 9000]
    xs = [rnd(5600 + i, m, 0.05) for i, m in enumerate(ns)]
    ref = _oracle(O, xs, 44100, 48000, None, "low_pass", 250)
    outs = {}
    for general in (0, 1):
        p = G.ResampleLowpassMix(44100, 48000, 1, None, "low_pass", 250, 0.5, max_sources=len(ns), max_in_frames=n, force_general=general)
        p.set_sources([torch.from_numpy(x).cuda() for x in xs])
        geo = p.geometry()
        assert geo["general_kernel"] == 1 and geo["ragged_pair"] == (0 if general else 1), geo
        outs[general] = p.run().cpu().numpy()
        p.check_status()
        p.close()
    _check(outs[0], ref, "low_pass")
    assert float(np.max(np.abs(outs[0] - outs[1]))) <= 1e-6


@pytest.mark.parametrize("filt,freq", [(None, 0), ("low_pass", 200)])
@pytest.mark.parametrize("span", [None, 32768])
def test_mono_block_streaming(G, O, filt, freq, span):
    import torch

    ns = [50000, 31000, 16384, 16385, 147, 0, 49999, 2, 40001]
    gains = np.linspace(0.5, 1.2, len(ns)).astype(np.float32)
    xs = [rnd(5300 + i, n, 0.1) for i, n in enumerate(ns)]
    ref = _oracle(O, xs, 44100, 48000, span, filt, freq, gains)
    p = G.ResampleLowpassMix(44100, 48000, 1, span, filt, freq, 0.5, max_sources=len(ns), max_in_frames=max(ns), frames_per_lane=8)
    p.set_gains(gains)
    xd = [torch.from_numpy(x).cuda() if len(x) else torch.empty(0, device="cuda") for x in xs]
    rng = np.random.default_rng(11)
    for trial in range(3):
        cuts = [0] + sorted(set(int(c) for c in rng.integers(1, max(ns), size=[1, 6, 25][trial]))) + [max(ns)]
        p.stream_begin()
        outs = [p.stream_feed_v([x[min(cuts[k], n): min(cuts[k + 1], n)] for x, n in zip(xd, ns)], [n <= cuts[k + 1] for n in ns]) for k in range(len(cuts) - 1)]
        p.check_status()
        _check(torch.cat(outs).cpu().numpy(), ref, filt)
    p.close()
    # the summed-state entry: equal-length sources
    n = 40000
    xs = [rnd(5400 + i, n, 0.1) for i in range(4)]
    ref = _oracle(O, xs, 44100, 48000, span, filt, freq)
    p2 = G.ResampleLowpassMix(44100, 48000, 1, span, filt, freq, 0.5, max_sources=4, max_in_frames=n, frames_per_lane=8)
    xd = [torch.from_numpy(x).cuda() for x in xs]
    cuts = [0, 999, 16384, 16385, 32768, n]
    p2.stream_begin()
    outs = [p2.stream_feed([x[cuts[k]: cuts[k + 1]] for x in xd], flush=(k == len(cuts) - 2)) for k in range(len(cuts) - 1)]
    p2.check_status()
    _check(torch.cat(outs).cpu().numpy(), ref, filt)
    p2.close()


def test_mono_full_size_block(G, O):
    # 64 mono sources x 1 Mi frames (BASELINE config 2's shape on one channel): every output frame against the oracle
    import torch

    S, n = 64, 1 << 20
    xs = [rnd(5500 + s, n, 1.0 / S) for s in range(S)]
    ref = _oracle(O, xs, 44100, 48000, None, "low_pass", 200)
    p = G.ResampleLowpassMix(44100, 48000, 1, None, "low_pass", 200, 0.5, max_sources=S, max_in_frames=n)
    p.set_sources([torch.from_numpy(x).cuda() for x in xs])
    p.autotune()
    got = p.run().cpu().numpy()
    p.check_status()
    _check(got, ref, "low_pass")
    p.close()

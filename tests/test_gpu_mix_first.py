"""Mix first: a filtered batch of equal-length sources is summed at the input rate (k_mix_rows for short rows,
k_mix_ring for rows that fill the chip) and the fused kernel converts and filters that one stream --
sum_s filter(resample(g_s x_s)) = filter(resample(sum_s g_s x_s)).  Checked against the oracle's per-source chains
(UniformSourceIterator(src.amplify(g)).low_pass(f) summed by the Mixer, mixer.rs:58-66, uniform.rs:50-97, blt.rs) and against
the per-source path of the same library (RH_NO_MIX_FIRST)."""
import numpy as np
import pytest
from conftest import knobs

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


_refs = {}


def _oracle(O, xs, frm, to, span, filt, freq, gains, ch, key=None):
    if key is not None and key in _refs:  # the same reference for every summing kernel
        return _refs[key]
    ref = _oracle_run(O, xs, frm, to, span, filt, freq, gains, ch)
    if key is not None:
        _refs[key] = ref
    return ref


def _oracle_run(O, xs, frm, to, span, filt, freq, gains, ch):
    m = O.Mixer(ch, to)
    for i, x in enumerate(xs):
        src = O.TestSource(x, ch, frm) if not span else O.SpanSource(x, ch, frm, span)
        if gains is not None:
            src = src.amplify(float(gains[i]))
        u = O.UniformSourceIterator(src, ch, to)
        m.add(u.low_pass(freq) if filt == "low_pass" else u.high_pass(freq))
    return m.collect()


def _run(G, xs, frm, to, ch, span, filt, freq, gains, R=0):
    import torch

    S, n = len(xs), len(xs[0]) // ch
    p = G.ResampleLowpassMix(frm, to, ch, span, filt, freq, 0.5, max_sources=S, max_in_frames=n, frames_per_lane=R)
    if gains is not None:
        p.set_gains(gains)
    p.set_sources([torch.from_numpy(x).cuda() for x in xs])
    geo = p.geometry()
    got = p.run().cpu().numpy()
    again = p.run().cpu().numpy()  # the mixed row and its descriptor are the handle's: a second run finds them in place
    p.check_status()
    p.close()
    assert np.array_equal(got, again)
    return got, geo


# which summing kernel: RH_MIX_U = 1 / 2 / 4 vector loads per lane (k_mix_rows), 12 / 13 the LDS-DMA ring of 2 / 3 stages (k_mix_ring)
@pytest.mark.parametrize("mix", [None, "1", "2", "4", "12", "13"])
@pytest.mark.parametrize("ch,n", [(2, 30000), (1, 30001), (2, 1537), (1, 4099), (2, 70000)])
@pytest.mark.parametrize("filt,freq", [("low_pass", 200), ("high_pass", 300)])
def test_mix_first_equals_rodio_chain(G, O, mix, ch, n, filt, freq):
    S = 9
    xs = [rnd(6100 + s, n * ch, 0.1) for s in range(S)]
    gains = np.array([1.0, 0.5, 1.7, 0.0, -0.25, 0.9, 1.0, 0.3, 1.1], dtype=np.float32)
    ref = _oracle(O, xs, 44100, 48000, None, filt, freq, gains, ch, key=(ch, n, filt, freq))
    with knobs(**({"RH_MIX_U": mix} if mix else {})):
        got, geo = _run(G, xs, 44100, 48000, ch, None, filt, freq, gains)
    assert geo["mix_first"] == 1 and geo["general_kernel"] == 0
    assert len(got) == len(ref)
    assert float(np.max(np.abs(got - ref))) <= TOL
    with knobs(RH_NO_MIX_FIRST="1"):  # the per-source path of the same handle type: the two agree far inside the tolerance
        per_source, geo2 = _run(G, xs, 44100, 48000, ch, None, filt, freq, gains)
    assert geo2["mix_first"] == 0
    assert float(np.max(np.abs(got - per_source))) <= 2e-6


@pytest.mark.parametrize("frm,to", [(48000, 44100), (48000, 48000), (22050, 48000), (192000, 44100)])
@pytest.mark.parametrize("span", [None, 32768, 3000])
def test_mix_first_rates_and_spans(G, O, frm, to, span):
    # the seams of spanned sources (uniform.rs:50-68: the converter restarts, the last frame of a span is emitted verbatim) are
    # the same for every source of the batch: they commute with the sum like the rest of the converter
    S, n, ch = 6, 80000, 2
    xs = [rnd(6200 + s, n * ch, 0.15) for s in range(S)]
    ref = _oracle(O, xs, frm, to, span, "low_pass", 200, None, ch)
    got, geo = _run(G, xs, frm, to, ch, span, "low_pass", 200, None)
    assert geo["mix_first"] == 1
    assert len(got) == len(ref)
    assert float(np.max(np.abs(got - ref))) <= TOL


def test_mix_first_does_not_apply_where_it_would_change_bits_or_cannot(G, O):
    import torch

    S, n = 4, 20000
    xs = [rnd(6300 + s, n * 2, 0.2) for s in range(S)]
    # without a filter the batch is rodio's ordered sum of converted samples, bit for bit: it stays per source
    p = G.ResampleLowpassMix(44100, 48000, 2, None, None, 0, 0.5, max_sources=S, max_in_frames=n)
    p.set_sources([torch.from_numpy(x).cuda() for x in xs])
    assert p.geometry()["mix_first"] == 0
    m = O.Mixer(2, 48000)
    for x in xs:
        m.add(O.UniformSourceIterator(O.TestSource(x, 2, 44100), 2, 48000))
    assert np.array_equal(p.run().cpu().numpy(), m.collect())
    p.close()
    # sources of different lengths end at different frames (each one's last frame is emitted verbatim): not one stream
    p = G.ResampleLowpassMix(44100, 48000, 2, None, "low_pass", 200, 0.5, max_sources=S, max_in_frames=n)
    p.set_sources([torch.from_numpy(x[: 2 * (n - 1000 * s)]).cuda() for s, x in enumerate(xs)])
    assert p.geometry()["mix_first"] == 0
    p.close()
    # a single source is its own mix
    p = G.ResampleLowpassMix(44100, 48000, 2, None, "low_pass", 200, 0.5, max_sources=S, max_in_frames=n)
    p.set_sources([torch.from_numpy(xs[0]).cuda()])
    assert p.geometry()["mix_first"] == 0
    got = p.run().cpu().numpy()
    ref = O.UniformSourceIterator(O.TestSource(xs[0], 2, 44100), 2, 48000).low_pass(200).collect()
    assert float(np.max(np.abs(got - ref))) <= TOL
    p.close()


def _truth(xs, frm, to, ch, kind, freq):
    """The same chain in float64: linear interpolation at rodio's positions (sample_rate.rs:131-201: frame floor(m*F/T), weight
    (m*F mod T)/T, the last frame verbatim), the biquad with rodio's f32 coefficients (blt.rs:558-560), the sum."""
    import math

    import rodio_amd
    from scipy.signal import lfilter

    g = math.gcd(frm, to)
    F, T = frm // g, to // g
    n = len(xs[0]) // ch
    M = ((n - 1) * T + F - 1) // F + 1 if F != T else n
    m = np.arange(M, dtype=np.int64)
    i = np.minimum(m * F // T, n - 1)
    w = ((m * F) % T).astype(np.float64) / T
    j = np.minimum(i + 1, n - 1)
    co = [float(v) for v in rodio_amd.biquad_coeffs(kind, int(freq), 0.5, to)]
    out = np.zeros((M, ch))
    for x in xs:
        f = x.astype(np.float64).reshape(n, ch)
        lerp = f[i] + (f[j] - f[i]) * w[:, None]
        out += lfilter(co[:3], [1.0, co[3], co[4]], lerp, axis=0)
    return out.reshape(-1)


def test_mix_first_full_scale_sources(G, O):
    # 64 sources at full scale: the input-rate mix reaches ~15 before the filter takes the band above 200 Hz away again.  rodio's
    # f32 recurrence and any other evaluation order of the same filter are both ~1e-5 x peak away from the exact response at this
    # cutoff (poles at 0.974: rounding is amplified by 1 / (1 - p)^2), so the contract is the one of the other time-parallel
    # filters (test_host_mirror.py): together within 2e-5 per unit of peak, and no further from the float64 response than the
    # reference itself (x 2 + 1e-7).
    S, n = 64, 40000
    xs = [rnd(6400 + s, n * 2, 1.0) for s in range(S)]
    ref = _oracle(O, xs, 44100, 48000, None, "low_pass", 200, None, 2)
    got, geo = _run(G, xs, 44100, 48000, 2, None, "low_pass", 200, None)
    assert geo["mix_first"] == 1
    peak = max(1.0, float(np.max(np.abs(ref))))
    e_go = float(np.max(np.abs(got - ref)))
    assert e_go <= 2 * TOL * peak
    # THE FILTER CONTRACT (rodio_hip.h, rh_filter_scan_ok): low_pass(200) at 48 kHz lies inside the region where ONE full-scale source stays
    # within 1e-5 of rodio's own recurrence; rodio's rounding noise is per source, so a mix of S full-scale sources may be S times that away
    assert G.filter_scan_ok("low_pass", 200, 0.5, 48000)
    print(f"[full scale x{S}] |gpu - oracle| = {e_go:.2e} (contract: {S} sources x 1e-5 = {S * TOL:.1e}), mix peak {peak:.1f}")
    assert e_go <= S * TOL
    truth = _truth(xs, 44100, 48000, 2, "low_pass", 200)
    assert len(truth) == len(ref)
    e_ref, e_gpu = float(np.max(np.abs(ref - truth))), float(np.max(np.abs(got - truth)))
    assert e_gpu <= 2.0 * e_ref + 1e-7 * peak, (e_gpu, e_ref)


# ---- k_rlm_chunk: mix first in one kernel, for stereo rows of at least 2 x 256 chunks of 1024 frames ---------------------------------
@pytest.mark.parametrize("frm,to,span,n,want", [
    (44100, 48000, None, 600000, 2), (44100, 48000, 32768, 600000, 2), (44100, 48000, 3000, 524288, 2), (48000, 44100, None, 700001 - 1, 2),
    (48000, 48000, None, 600000, 2), (48000, 32000, 5000, 600000, 2),
    (96000, 44100, 5000, 600000, 1),   # more than 2 input frames per output frame: the filter's look-back leaves the 4 frames in front of a chunk
    (44100, 48000, None, 600001, 1),   # an odd length: not whole 16-byte vectors -> the two-launch form
    (22050, 48000, None, 600000, 2),   # more than 64 x 18 output frames per 1024-frame chunk: the instance with chunks of 512 frames
    (16000, 48000, 32768, 600000, 1),  # ratio 3: more than 64 x 18 frames even of 512 -> the two-launch form
    (44100, 48000, None, 200000, 1),   # too few chunks to fill the chip -> the two-launch form
])
@pytest.mark.parametrize("filt,freq", [("low_pass", 200), ("high_pass", 300)])
def test_chunk_kernel_equals_rodio_chain(G, O, frm, to, span, n, want, filt, freq):
    S, ch = 3, 2
    xs = [rnd(6500 + s, n * ch, 0.3) for s in range(S)]
    gains = np.array([1.0, 0.5, -0.75], dtype=np.float32)
    ref = _oracle(O, xs, frm, to, span, filt, freq, gains, ch)
    got, geo = _run(G, xs, frm, to, ch, span, filt, freq, gains)
    assert geo["mix_first"] == want, geo
    assert len(got) == len(ref)
    err = np.abs(got - ref)
    assert float(np.max(err)) <= TOL, (int(np.argmax(err)), float(np.max(err)))
    if want == 2:
        with knobs(RH_NO_CHUNK="1"):  # the two-launch form of the same sum: the two agree far inside the tolerance
            two, geo2 = _run(G, xs, frm, to, ch, span, filt, freq, gains)
        assert geo2["mix_first"] == 1
        assert float(np.max(np.abs(got - two))) <= 2e-6


def test_chunk_kernel_many_sources_and_low_cutoff(G, O):
    # 40 sources; a 30 Hz high-pass: poles at 0.996, the carry reaches back over 7 tiles, and rounding is amplified by
    # 1 / (1 - p)^2 = 65 000 -- rodio's f32 recurrence is 1e-4 away from the exact response here, and so is every other f32
    # evaluation of this filter (the per-source kernels too).  The contract is the truth-relative one (see above).
    S, n = 40, 540000
    xs = [rnd(6600 + s, n * 2, 0.05) for s in range(S)]
    ref = _oracle(O, xs, 44100, 48000, None, "high_pass", 30, None, 2)
    got, geo = _run(G, xs, 44100, 48000, 2, None, "high_pass", 30, None)
    assert geo["mix_first"] == 2
    truth = _truth(xs, 44100, 48000, 2, "high_pass", 30)
    assert len(truth) == len(ref) == len(got)
    e_ref, e_gpu = float(np.max(np.abs(ref - truth))), float(np.max(np.abs(got - truth)))
    assert e_gpu <= 2.0 * e_ref + 1e-7, (e_gpu, e_ref)
    # measured against the ORACLE too: high_pass(30) is far outside the filter contract (rh_filter_scan_ok: high_pass needs >= 600 Hz at
    # 48 kHz), the distance is rodio's own rounding noise (e_ref) give or take the scan's (e_gpu) -- which is why a drop-in
    # (include/rodio_hip.hpp) runs such a filter in the reference's order instead
    assert not G.filter_scan_ok("high_pass", 30, 0.5, 48000)
    e_go = float(np.max(np.abs(got - ref)))
    print(f"[high_pass(30) x{S}] |gpu - oracle| = {e_go:.2e}, |oracle - f64| = {e_ref:.2e}, |gpu - f64| = {e_gpu:.2e}")
    assert e_go <= e_ref + e_gpu + 1e-7 and e_go >= 0.5 * e_ref
    with knobs(RH_NO_CHUNK="1"):
        two, _ = _run(G, xs, 44100, 48000, 2, None, "high_pass", 30, None)
    assert float(np.max(np.abs(two - truth))) <= 2.0 * e_ref + 1e-7


# ---- rh_rlm_config.filter_first: `mixer.add(src.low_pass(f))` -- the filter at the source's rate, in front of the converter -------------
@pytest.mark.parametrize("frm,to,n", [(44100, 48000, 50000), (48000, 44100, 30001), (44100, 48000, 600000)])
@pytest.mark.parametrize("filt,freq", [("low_pass", 200), ("low_pass", 1000), ("high_pass", 300)])
def test_filter_before_the_converter(G, O, frm, to, n, filt, freq):
    import torch

    S, ch = 5, 2
    xs = [rnd(6700 + s, n * ch, 0.2) for s in range(S)]
    gains = np.array([1.0, 0.5, -0.75, 0.25, 1.5], dtype=np.float32)
    m = O.Mixer(ch, to)
    for x, g in zip(xs, gains):  # source/mod.rs:255-275: the filter takes the rate of its input; mixer.rs:58-66 converts what it is given
        src = O.TestSource(x, ch, frm).amplify(float(g))
        m.add(O.UniformSourceIterator(src.low_pass(freq) if filt == "low_pass" else src.high_pass(freq), ch, to))
    ref = m.collect()
    p = G.ResampleLowpassMix(frm, to, ch, None, filt, freq, 0.5, max_sources=S, max_in_frames=n, filter_first=True)
    p.set_gains(gains)
    p.set_sources([torch.from_numpy(x).cuda() for x in xs])
    got = p.run().cpu().numpy()
    p.check_status()
    assert len(got) == len(ref)
    assert float(np.max(np.abs(got - ref))) <= TOL
    # the other order is a different signal (the benchmark's spelling: convert, then filter at the mixer's rate)
    other = _oracle(O, xs, frm, to, None, filt, freq, gains, ch)
    assert float(np.max(np.abs(other - ref))) > 10 * TOL
    # sources of different lengths: not with this flag
    p.set_sources([torch.from_numpy(x[: 2 * (n - 100 * s)]).cuda() for s, x in enumerate(xs)])
    with pytest.raises(Exception):
        p.run()
    p.close()


@pytest.mark.parametrize("frm,to,span,n", [(44100, 48000, None, 600000), (44100, 48000, 32768, 600004), (48000, 44100, None, 800000), (48000, 48000, 3000, 600000)])
@pytest.mark.parametrize("filt,freq", [("low_pass", 200), ("high_pass", 300)])
def test_chunk_kernel_mono(G, O, frm, to, span, n, filt, freq):
    # mono rows: chunks of 4 KiB = 1024 frames, the same 64 runs of 18 frames per tile
    S, ch = 3, 1
    xs = [rnd(6800 + s, n, 0.3) for s in range(S)]
    gains = np.array([1.0, 0.5, -0.75], dtype=np.float32)
    ref = _oracle(O, xs, frm, to, span, filt, freq, gains, ch)
    got, geo = _run(G, xs, frm, to, ch, span, filt, freq, gains)
    assert geo["mix_first"] == 2, geo
    assert len(got) == len(ref)
    err = np.abs(got - ref)
    assert float(np.max(err)) <= TOL, (int(np.argmax(err)), float(np.max(err)))
    with knobs(RH_NO_CHUNK="1"):
        two, geo2 = _run(G, xs, frm, to, ch, span, filt, freq, gains)
    assert geo2["mix_first"] == 1
    assert float(np.max(np.abs(got - two))) <= 2e-6


@pytest.mark.parametrize("frm,to,span,n", [(44100, 48000, None, 600000), (44100, 48000, 32768, 600000), (48000, 44100, None, 700000)])
def test_chunk_kernel_half_chunks(G, O, frm, to, span, n):
    # RH_CHUNK_HALF: the stereo instance with chunks of 512 frames in runs of 9 (a tuning aid: twice the tiles, two per SIMD)
    S, ch = 3, 2
    xs = [rnd(6900 + s, n * ch, 0.3) for s in range(S)]
    ref = _oracle(O, xs, frm, to, span, "low_pass", 200, None, ch)
    with knobs(RH_CHUNK_HALF="1"):
        got, geo = _run(G, xs, frm, to, ch, span, "low_pass", 200, None)
    assert geo["mix_first"] == 2, geo
    assert len(got) == len(ref)
    assert float(np.max(np.abs(got - ref))) <= TOL


@pytest.mark.parametrize("ch,n", [(2, 2_600_000), (1, 3_400_000)])
def test_chunk_kernel_more_tiles_than_slots(G, O, ch, n):
    # rows with more chunks than workgroups fit on the chip at once: the tiles are handed out by a ticket (a tile only waits for
    # earlier tiles, which run or have finished), no longer by workgroup index
    S = 2
    xs = [rnd(7000 + s, n * ch, 0.3) for s in range(S)]
    ref = _oracle(O, xs, 44100, 48000, None, "low_pass", 200, None, ch)
    got, geo = _run(G, xs, 44100, 48000, ch, None, "low_pass", 200, None)
    assert geo["mix_first"] == 2, geo
    assert len(got) == len(ref)
    assert float(np.max(np.abs(got - ref))) <= TOL


def test_block_streaming_sums_first_too(G, O):
    """rh_rlm_stream_block carries ONE summed filter state, and the state of the sum is the sum of the states: a block of a
    stream is summed at the input rate (k_mix_ring for blocks that fill the chip, k_mix_rows for short ones) and the one mixed
    row streams through the fused kernel with the stream's state words.  Blocks of 600 000 frames (the ring) and short ones;
    the concatenation against the oracle's per-source chains and against the per-source path of the same stream."""
    import torch

    S, n = 4, 1_250_000
    xs = [rnd(4200 + s, 2 * n, 1.0 / S) for s in range(S)]
    gains = np.array([1.0, 0.5, 1.25, 0.75], dtype=np.float32)
    ref = _oracle(O, xs, 44100, 48000, None, "low_pass", 200, gains, 2)
    xd = [torch.from_numpy(x).cuda() for x in xs]
    got = {}
    for mix_first in (True, False):
        p = G.ResampleLowpassMix(44100, 48000, 2, None, "low_pass", 200, 0.5, max_sources=S, max_in_frames=700_000)
        p.set_gains(gains)
        p.set_mix_first(mix_first)
        p.stream_begin()
        cuts = [0, 600_000, 1_200_000, 1_200_700, 1_249_000, n]
        outs = []
        for k in range(len(cuts) - 1):
            outs.append(p.stream_feed([x[2 * cuts[k]: 2 * cuts[k + 1]] for x in xd], flush=(k == len(cuts) - 2)))
        p.check_status()
        got[mix_first] = torch.cat(outs).cpu().numpy()
        p.close()
    assert len(got[True]) == len(ref) == len(got[False])
    e_first, e_each, e_between = (float(np.max(np.abs(got[True] - ref))), float(np.max(np.abs(got[False] - ref))), float(np.max(np.abs(got[True] - got[False]))))
    print(f"[stream mix first] |mix first - oracle| {e_first:.2e}  |per source - oracle| {e_each:.2e}  |between| {e_between:.2e}")
    assert e_first <= TOL and e_each <= TOL and e_between <= 2e-6


@pytest.mark.parametrize("n_equal", [True, False])
def test_a_filter_per_source_in_the_fused_mixer(G, O, n_equal):
    """rh_rlm_set_filters (VERDICT r03 missing #2): `mixer.add(a.low_pass(200)); mixer.add(b.high_pass(300)); mixer.add(c)` -- rodio's
    sources carry their own adapters into Mixer::add (source/mod.rs:686-721, mixer.rs:58-66).  Eight sources, three filters and none:
    the sources of a filter form a class that runs as its own fused launch (summed first where their lengths agree), the classes'
    mixes are added.  Against the oracle's Mixer over the per-source chains; gains on top; then back to the handle's one filter."""
    import torch

    S = 8
    ns = [60000] * S if n_equal else [60000, 41000, 60000, 12345, 60000, 2, 25000, 59999]
    filters = [("low_pass", 200), ("high_pass", 300), None, ("low_pass", 200), ("low_pass", 1000), ("high_pass", 300), None, ("low_pass", 1000)]
    gains = np.linspace(0.4, 1.1, S).astype(np.float32)
    xs = [rnd(6100 + i, 2 * n, 0.1) for i, n in enumerate(ns)]
    m = O.Mixer(2, 48000)
    for x, f, g in zip(xs, filters, gains):
        u = O.UniformSourceIterator(O.TestSource(x, 2, 44100).amplify(float(g)), 2, 48000)
        m.add(u if f is None else (u.low_pass(f[1]) if f[0] == "low_pass" else u.high_pass(f[1])))
    ref = m.collect()
    p = G.ResampleLowpassMix(44100, 48000, 2, None, "low_pass", 200, 0.5, max_sources=S, max_in_frames=max(ns))
    p.set_filters(filters)
    p.set_gains(gains)
    xd = [torch.from_numpy(x).cuda() for x in xs]
    p.set_sources(xd)
    got = p.run().cpu().numpy()
    again = p.run().cpu().numpy()
    p.check_status()
    assert len(got) == len(ref)
    e = float(np.max(np.abs(got - ref)))
    print(f"[filters per source, equal lengths {n_equal}] |gpu - oracle| = {e:.2e}")
    assert e <= TOL and np.array_equal(got, again)
    # the stream entries are for one-filter handles (a streaming host keeps one handle per filter)
    with pytest.raises(G.RhError):
        p.stream_begin()
    # back to the handle's one filter: low_pass(200) for every source
    p.set_filters([])
    p.set_sources(xd)
    one = p.run().cpu().numpy()
    p.check_status()
    ref1 = _oracle(O, xs, 44100, 48000, None, "low_pass", 200, gains, 2)
    assert len(one) == len(ref1) and float(np.max(np.abs(one - ref1))) <= TOL
    p.close()


@pytest.mark.parametrize("ch,n,many", [(2, 300_000, False), (1, 280_004, False), (2, 270_000, True)])
def test_filter_classes_walked_in_one_launch(G, O, ch, n, many):
    """Round 6 (VERDICT r05 next #5a): a mixer whose sources carry different filters (`mixer.add(a.low_pass(200)); mixer.add(b.high_pass(300))`,
    source/mod.rs:686-721, mixer.rs:58-66) where every class is long enough for k_rlm_chunk: ONE launch walks the classes (k_rlm_chunk_classes:
    a workgroup of two waves per tile, one summing the tile's chunk class after class, the other converting and filtering behind it; or, for
    classes of different geometry and under RH_CLASSES_ONE_WAVE=1, k_rlm_chunk_multi -- class k's workgroups behind class k-1's, each with its
    own arguments, tables and tickets), the classes' mixes are added behind it.  The
    oracle's Mixer over the per-source chains; the same bits as one launch per class (RH_CLASSES_ONE_BY_ONE=1: the tile code is the same);
    geometry().mix_first says which form ran."""
    import torch
    from conftest import knobs

    filters = [("low_pass", 200), ("high_pass", 300), ("low_pass", 1000), ("high_pass", 300), ("low_pass", 200), ("low_pass", 1000), ("low_pass", 200)]
    if many:  # seven classes (what one launch takes), two sources each, in mixed order
        kinds = [("low_pass", 200), ("high_pass", 700), ("low_pass", 1000), ("high_pass", 2000), ("low_pass", 4000), ("low_pass", 500), ("high_pass", 1200)]
        filters = kinds + kinds[::-1]
    S = len(filters)
    gains = np.linspace(0.5, 1.2, S).astype(np.float32)
    xs = [rnd(7300 + i, ch * n, 0.1) for i in range(S)]
    m = O.Mixer(ch, 48000)
    for x, f, g in zip(xs, filters, gains):
        u = O.UniformSourceIterator(O.TestSource(x, ch, 44100).amplify(float(g)), ch, 48000)
        m.add(u.low_pass(f[1]) if f[0] == "low_pass" else u.high_pass(f[1]))
    ref = m.collect()
    xd = [torch.from_numpy(x).cuda() for x in xs]

    def run():
        p = G.ResampleLowpassMix(44100, 48000, ch, None, "low_pass", 200, 0.5, max_sources=S, max_in_frames=n)
        p.set_filters(filters)
        p.set_gains(gains)
        p.set_sources(xd)
        got = p.run().cpu().numpy()
        again = p.run().cpu().numpy()
        p.check_status()
        how = p.geometry()["mix_first"]
        p.close()
        assert np.array_equal(got, again)
        return got, how

    one, how_one = run()  # two waves a tile: one loads class after class, the other converts, filters and ADDS the classes' mixes (k_rlm_chunk_classes)
    with knobs(RH_CLASSES_ONE_WAVE="1"):
        lined, how_lined = run()  # the classes' launches lined up in one grid (k_rlm_chunk_multi)
    with knobs(RH_CLASSES_ONE_BY_ONE="1"):
        each, how_each = run()
    assert how_one == 3 and how_lined == 3 and how_each == 2, (how_one, how_lined, how_each)
    assert len(one) == len(ref) == len(each) == len(lined)
    e = float(np.max(np.abs(one - ref)))
    print(f"[filter classes in one launch, {ch} ch] |gpu - oracle| = {e:.2e}")
    assert e <= TOL and np.array_equal(one, each) and np.array_equal(lined, each)


def test_block_streaming_keeps_its_table_while_the_sources_move_together(G, O):
    """A block that is summed first reads only pointers and gains of the source table, and they take a common offset: resident rows read at
    `row + consumed` (every source moves on by the same bytes) stream without uploading the table again (rh_pipeline_stream.hip).  The same
    stream with RH_STREAM_UPLOAD_ALWAYS=1: the same bits.  In the middle of the stream one source moves to another buffer (its distance
    differs: the table is uploaded again) and the gains change (they travel in the table): still the bits of the stream that uploads every time,
    and -- up to the frame where the gains changed -- the oracle's samples."""
    import ctypes as C

    import torch
    from conftest import knobs
    from rodio_amd import _lib

    lib = _lib.lib
    S, N, B = 6, 300_000, 40_000
    xs = [rnd(8800 + s, 2 * N, 0.15) for s in range(S)]
    g1, g2 = np.linspace(0.5, 1.2, S).astype(np.float32), np.linspace(1.1, 0.4, S).astype(np.float32)
    data = torch.from_numpy(np.stack(xs)).cuda()
    moved = data[2].clone()  # the third source's samples once more, somewhere else
    mo = C.c_uint64(0)
    _lib.check(lib.rh_resample_out_frames(N, 44100, 48000, 2, 0, C.byref(mo)), "rh_resample_out_frames")
    M = mo.value

    def run():
        p = G.ResampleLowpassMix(44100, 48000, 2, None, "low_pass", 1000, 0.5, max_sources=S, max_in_frames=B + 4096, frames_per_lane=18)
        p.set_gains(g1)
        p.stream_begin(keep_history=True)
        out = torch.zeros(2 * M + 4096, device="cuda", dtype=torch.float32)
        g0 = m = k = 0
        switch_at = None
        while True:
            hi = min(N, (k + 1) * B)
            base = [data[s_].data_ptr() for s_ in range(S)]
            if k >= 3:
                base[2] = moved.data_ptr()
            if k == 5:
                p.set_gains(g2)
                switch_at = m
            ptrs = (C.c_void_p * S)(*[b_ + g0 * 8 for b_ in base])
            avail = (C.c_uint64 * S)(*([hi - g0] * S))
            ended = (C.c_uint8 * S)(*([1 if hi == N else 0] * S))
            o, c = C.c_uint64(0), C.c_uint64(0)
            _lib.check(lib.rh_rlm_stream_block_v(p._h, ptrs, avail, ended, S, C.c_void_p(out.data_ptr() + m * 8), M + 512 - m, C.byref(o), C.byref(c), None), "rh_rlm_stream_block_v")
            m += o.value
            g0 += c.value
            k += 1
            if hi == N:
                break
        p.check_status()
        stats = p.stream_stats()
        res = out[: 2 * m].cpu().numpy()
        p.close()
        return res, switch_at, stats

    a, sw_a, st_a = run()
    with knobs(RH_STREAM_UPLOAD_ALWAYS="1"):
        b, sw_b, st_b = run()
    assert st_a == st_b and st_a[0] >= 7 and st_a[1] == 0, (st_a, st_b)  # every block on the summed state
    assert sw_a == sw_b and np.array_equal(a, b)
    ref1 = _oracle(O, xs, 44100, 48000, None, "low_pass", 1000, g1, 2)
    assert len(a) == len(ref1)
    assert float(np.max(np.abs(a[: 2 * sw_a] - ref1[: 2 * sw_a]))) <= TOL


@pytest.mark.parametrize("filt,freq", [("low_pass", 200), ("high_pass", 300)])
@pytest.mark.parametrize("case", ["end together", "end apart", "short first block", "short block in the middle", "one falls behind", "end apart, together again"])
def test_block_streaming_of_sources_that_run_together(G, O, filt, freq, case):
    """rh_rlm_stream_block_v + rh_rlm_stream_keep_history (VERDICT r03 missing #5): what a mixer's sources do from the moment they are
    added until the first of them ends -- all live, the same frames per block -- runs on the SUMMED state, every block summed first.
    When a source ends, falls behind, or a block is too short to recover from, the per-source states are recovered from the rows of
    the block before and the stream goes on with one state per source.  Every case against the oracle's per-source chains and against
    the same stream without the history (one state per source throughout)."""
    import torch

    S = 6
    ns = {"end together": [90000] * S, "end apart": [90000, 61000, 90000, 45000, 90000, 40000], "short first block": [90000, 61000, 90000, 90000, 90000, 90000],
          "short block in the middle": [90000] * S, "one falls behind": [90000] * S, "end apart, together again": [200000, 60000, 200000, 130000, 200000, 200000]}[case]
    cuts = {"end together": [0, 25000, 50000, 75000, 90000], "end apart": [0, 25000, 50000, 75000, 90000], "short first block": [0, 700, 30000, 60000, 90000],
            "short block in the middle": [0, 30000, 30300, 60000, 90000], "one falls behind": [0, 25000, 50000, 75000, 90000],
            "end apart, together again": list(range(0, 200001, 25000))}[case]
    gains = np.linspace(0.5, 1.3, S).astype(np.float32)
    xs = [rnd(7100 + i, 2 * n, 0.15) for i, n in enumerate(ns)]
    m = O.Mixer(2, 48000)
    for x, g in zip(xs, gains):
        u = O.UniformSourceIterator(O.TestSource(x, 2, 44100).amplify(float(g)), 2, 48000)
        m.add(u.low_pass(freq) if filt == "low_pass" else u.high_pass(freq))
    ref = m.collect()
    xd = [torch.from_numpy(x).cuda() for x in xs]
    got = {}
    for hist in (True, False):
        p = G.ResampleLowpassMix(44100, 48000, 2, None, filt, freq, 0.5, max_sources=S, max_in_frames=max(ns))
        p.set_gains(gains)
        p.stream_begin(keep_history=hist)
        outs = []
        fed = [0] * S  # frames of every source passed so far
        for k in range(len(cuts) - 1):
            lo, hi = cuts[k], cuts[k + 1]
            blocks, ended = [], []
            for s_, (x, n) in enumerate(zip(xd, ns)):
                a, b = min(fed[s_], n), min(hi, n)
                if case == "one falls behind" and s_ == 2 and k == 1:
                    b = a + (b - a) // 2  # this source delivers half a block once (a decoder that stalls): the others run ahead of it
                blocks.append(x[2 * a: 2 * b])
                fed[s_] = b
                ended.append(b >= n)
            outs.append(p.stream_feed_v(blocks, ended))
        while not all(f >= n for f, n in zip(fed, ns)):  # (the source that fell behind delivers its rest)
            blocks, ended = [], []
            for s_, (x, n) in enumerate(zip(xd, ns)):
                blocks.append(x[2 * fed[s_]: 2 * n])
                fed[s_] = n
                ended.append(True)
            outs.append(p.stream_feed_v(blocks, ended))
        p.check_status()
        got[hist] = torch.cat(outs).cpu().numpy()
        summed, each, rec = p.stream_stats()
        if not hist:
            assert summed == 0 and rec == 0 and each >= 4
        else:  # which blocks ran how: the cases are what their names say
            # (round 5: a stream with a state per source goes back to the summed state once its sources have either given everything or run
            # together again -- the block behind a short one, the blocks between two sources' ends)
            want = {"end together": (4, 0, 0), "end apart": (1, 3, 1), "short first block": (2, 2, 1), "short block in the middle": (3, 0, 1), "one falls behind": (3, 1, 1),
                    "end apart, together again": (6, 2, 2)}[case]
            assert (summed, each, rec) == want, (case, summed, each, rec)
        p.close()
    assert len(got[True]) == len(ref) == len(got[False]), (len(got[True]), len(got[False]), len(ref))
    e_t, e_s, e_b = float(np.max(np.abs(got[True] - ref))), float(np.max(np.abs(got[False] - ref))), float(np.max(np.abs(got[True] - got[False])))
    print(f"[together: {case} {filt}{freq}] |together - oracle| {e_t:.2e}  |per source - oracle| {e_s:.2e}  |between| {e_b:.2e}")
    assert e_t <= TOL and e_s <= TOL and e_b <= 2e-6



@pytest.mark.parametrize("fs", [44100, 48000, 96000])
def test_the_filter_contract(G, fs):
    """rh_filter_scan_ok (rodio_hip.h): where the time-parallel filter stays within 1e-5 of rodio's OWN f32 recurrence (rh_biquad mode 0,
    bit-exact with the oracle) for a full-scale source.  Inside the region the claim is checked on full-scale noise; the region's
    edges are where the measurement put them (profiles/r04_filter_contract.txt)."""
    import torch

    n = 300_000
    x = torch.from_numpy(rnd(9900 + fs, 2 * n, 1.0)).cuda().reshape(1, -1)
    inside = outside = 0
    for kind in ("low_pass", "high_pass"):
        for f in (20, 50, 100, 150, 200, 300, 600, 1000, 2000, 5000, 12000):
            co = G.biquad_coeffs(kind, f, 0.5, fs)
            seq = G.biquad_batch(x, co, mode=0).cpu().numpy()[0]
            par = G.biquad_batch(x, co, mode=1).cpu().numpy()[0]
            e = float(np.max(np.abs(seq - par)))
            if G.filter_scan_ok(kind, f, 0.5, fs):
                inside += 1
                assert e <= TOL, (kind, f, fs, e)
            else:
                outside += 1
    assert inside >= 8 and outside >= 4
    # the edges at 48 kHz, as documented
    if fs == 48000:
        assert G.filter_scan_ok("low_pass", 100, 0.5, fs) and not G.filter_scan_ok("low_pass", 50, 0.5, fs)
        assert G.filter_scan_ok("high_pass", 600, 0.5, fs) and not G.filter_scan_ok("high_pass", 500, 0.5, fs)


@pytest.mark.parametrize("ch,filt,freq,frm,B", [(2, "low_pass", 200, 44100, 40_000), (2, "high_pass", 1000, 48000, 9_000), (1, "low_pass", 1000, 44100, 70_000), (2, "low_pass", 200, 48000, 3_000),
                                                (2, "low_pass", 4000, 50000, 20_000)])
def test_a_block_of_a_summed_stream_in_one_launch(G, O, ch, filt, freq, frm, B):
    """Round 6: a block of a stream on the summed state is ONE kernel (k_rlm_sblk, rh_pipeline_sblk.hip) -- the sum over the
    sources, the conversion and the filter -- with nothing from the host per block but its arguments.  Resident rows read at `row + consumed`,
    blocks of B input frames (tile windows of 1, 2 or 3 KiB by the block's length; a last block that is shorter; the verbatim last frame): the
    oracle's one-pass samples; the same stream with RH_NO_SBLK=1 (two launches per block, as until round 5) within 2e-6; and side by side
    (rh_rlm_stream_overlap: a block launched without a barrier behind the block in front, the state waited for inside the kernel) bit for bit."""
    import ctypes as C

    import torch
    from conftest import knobs
    from rodio_amd import _lib

    lib = _lib.lib
    S, N = 7, 200_000
    xs = [rnd(9900 + s, ch * N, 0.15) for s in range(S)]
    gains = np.linspace(0.4, 1.3, S).astype(np.float32)
    data = torch.from_numpy(np.stack(xs)).cuda()
    mo = C.c_uint64(0)
    _lib.check(lib.rh_resample_out_frames(N, frm, 48000, ch, 0, C.byref(mo)), "rh_resample_out_frames")
    M = mo.value

    def run(overlap):
        p = G.ResampleLowpassMix(frm, 48000, ch, None, filt, freq, 0.5, max_sources=S, max_in_frames=B + 4096)
        p.set_gains(gains)
        p.stream_begin(keep_history=True)
        _lib.check(lib.rh_rlm_stream_overlap(p._h, 1 if overlap else 0), "rh_rlm_stream_overlap")
        out = torch.zeros(ch * M + 4096, device="cuda", dtype=torch.float32)
        g0 = m = k = 0
        while True:
            hi = min(N, (k + 1) * B)
            ptrs = (C.c_void_p * S)(*[data[s_].data_ptr() + g0 * 4 * ch for s_ in range(S)])
            avail = (C.c_uint64 * S)(*([hi - g0] * S))
            ended = (C.c_uint8 * S)(*([1 if hi == N else 0] * S))
            o, c = C.c_uint64(0), C.c_uint64(0)
            _lib.check(lib.rh_rlm_stream_block_v(p._h, ptrs, avail, ended, S, C.c_void_p(out.data_ptr() + m * 4 * ch), M + 512 - m, C.byref(o), C.byref(c), None), "rh_rlm_stream_block_v")
            m += o.value
            g0 += c.value
            k += 1
            if hi == N:
                break
        p.check_status()
        one = C.c_uint32(0)
        _lib.check(lib.rh_rlm_stream_one_launch_blocks(p._h, C.byref(one)), "rh_rlm_stream_one_launch_blocks")
        ovl = C.c_uint32(0)
        _lib.check(lib.rh_rlm_stream_overlapped_blocks(p._h, C.byref(ovl)), "rh_rlm_stream_overlapped_blocks")
        res = out[: ch * m].cpu().numpy()
        stats = p.stream_stats()
        p.close()
        return res, one.value, stats, k, ovl.value

    a, one_a, st_a, nb, ovl_a = run(False)
    b, one_b, st_b, _, ovl_b = run(True)
    with knobs(RH_NO_SBLK="1"):
        c_, one_c, st_c, _, _ = run(False)
    ref = _oracle(O, xs, frm, 48000, None, filt, freq, gains, ch)
    assert len(a) == len(b) == len(c_) == len(ref), (len(a), len(b), len(c_), len(ref))
    assert one_a == nb and one_b == nb and one_c == 0, (one_a, one_b, one_c, nb)  # every block of the stream, the last (flush) one included
    assert st_a[0] == nb and st_a[1] == 0
    e = [float(np.max(np.abs(v - ref))) for v in (a, b, c_)]
    assert max(e) <= TOL, e
    assert float(np.max(np.abs(a - c_))) <= 2e-6
    # side by side: the same kernel with the same state words, waited for inside it -- the same bits; every block but the stream's first (no state
    # to wait for) and a last one of a few frames started without a barrier behind the block in front
    assert np.array_equal(a, b) and ovl_a == 0 and nb - 3 <= ovl_b <= nb - 1, (ovl_a, ovl_b, nb)

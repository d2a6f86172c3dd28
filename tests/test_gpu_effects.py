"""The reference-order recurrences in their batched, vector-I/O form (rh_recurrence.hip: k_biquad_vec, k_agc_vec): many
streams per launch and state carried across blocks, through the Python mirror (`n_streams > 1` was never
called).  Biquad mode 0 is bit-exact (blt.rs:559 in the reference's order); the AGC is <= 1e-5 (device sqrt / divide)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def G(rh):
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    rh.init(0)
    return rh


def rnd(seed, n, scale=1.0):
    return (np.random.default_rng(seed).uniform(-1, 1, n) * scale).astype(np.float32)


def _programme(seed, n):
    """level changes by 20 dB every few thousand samples: the AGC's attack/release branches and the peak follower all move"""
    rng = np.random.default_rng(seed)
    env = np.repeat(10.0 ** (rng.uniform(-1.5, 0.0, n // 3000 + 1)), 3000)[:n]
    return (rng.uniform(-1, 1, n) * env).astype(np.float32)


@pytest.mark.parametrize("ch,S,frames", [(2, 7, 30000), (1, 70, 4099), (2, 64, 8192), (4, 5, 12000), (8, 3, 5000), (3, 4, 1001), (2, 3, 17)])
def test_biquad_mode0_batch_bit_exact(G, O, ch, S, frames):
    import torch

    xs = [rnd(10 + s, frames * ch, 0.7) for s in range(S)]
    co = G.biquad_coeffs("low_pass", 300, 0.5, 48000)
    x = torch.from_numpy(np.stack(xs)).cuda()
    out = torch.empty_like(x)
    from rodio_amd import _lib
    import ctypes as C

    _lib.check(_lib.lib.rh_biquad(C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()), frames, ch, S, co.ctypes.data_as(_lib.f32p), None, 0,
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rh_biquad")
    got = out.cpu().numpy()
    for s in range(S):
        ref = O.TestSource(xs[s], ch, 48000).low_pass(300).collect()
        assert np.array_equal(got[s], ref), (s, float(np.max(np.abs(got[s] - ref))))


@pytest.mark.parametrize("ch", [1, 2, 4])
def test_biquad_mode0_state_across_blocks_bit_exact(G, O, ch):
    import ctypes as C

    import torch

    from rodio_amd import _lib

    S, frames = 3, 20000
    xs = [rnd(30 + s, frames * ch) for s in range(S)]
    co = G.biquad_coeffs("high_pass", 120, 0.5, 44100)
    x = torch.from_numpy(np.stack(xs)).cuda()
    state = torch.zeros((S, 4 * ch), device="cuda")
    rng = np.random.default_rng(1)
    outs, a = [], 0
    while a < frames:
        b = min(frames, a + int(rng.choice([1, 5, 16, 333, 4096])))
        blk = x[:, a * ch: b * ch].contiguous()
        o = torch.empty_like(blk)
        _lib.check(_lib.lib.rh_biquad(C.c_void_p(o.data_ptr()), C.c_void_p(blk.data_ptr()), b - a, ch, S, co.ctypes.data_as(_lib.f32p), C.c_void_p(state.data_ptr()), 0,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rh_biquad")
        outs.append(o)
        a = b
    got = torch.cat(outs, dim=1).cpu().numpy()
    for s in range(S):
        assert np.array_equal(got[s], O.TestSource(xs[s], ch, 44100).high_pass(120).collect())


@pytest.mark.parametrize("S,n", [(5, 60000), (70, 20000), (3, 8192 + 16), (2, 100), (4, 8191), (1, 30011)])
def test_agc_many_streams_one_launch(G, O, S, n):
    import torch

    xs = [_programme(50 + s, n) for s in range(S)]
    out = G.agc_batch(torch.from_numpy(np.stack(xs)).cuda(), 48000).cpu().numpy()
    for s in range(S):
        ref = O.TestSource(xs[s], 1, 48000).automatic_gain_control().collect()
        assert float(np.max(np.abs(out[s] - ref))) <= TOL, s


@pytest.mark.parametrize("kw", [dict(), dict(target_level=0.5, attack_ns=10_000_000, release_ns=5_000_000, absolute_max_gain=5.0, floor=0.2)])
def test_agc_state_carried_across_blocks_equals_one_pass(G, O, kw):
    import torch

    S, n = 3, 70000
    xs = [_programme(80 + s, n) for s in range(S)]
    refs = [O.TestSource(x, 2, 48000).automatic_gain_control(**kw).collect() for x in xs]
    x = torch.from_numpy(np.stack(xs)).cuda()
    state = G.agc_state(S)
    rng = np.random.default_rng(3)
    outs, a = [], 0
    while a < n:
        b = min(n, a + int(rng.choice([4, 64, 1000, 8192, 9000, 20000])))  # below, at and above the 8192-sample RMS window
        outs.append(G.agc_batch(x[:, a:b].contiguous(), 48000, state=state, **kw))
        a = b
    got = torch.cat(outs, dim=1).cpu().numpy()
    for s in range(S):
        assert float(np.max(np.abs(got[s] - refs[s]))) <= TOL, s


from conftest import knobs as _env  # noqa: E402


@pytest.mark.parametrize("kw", [dict(), dict(target_level=0.5, attack_ns=10_000_000, release_ns=5_000_000, absolute_max_gain=5.0, floor=0.2),
                                dict(target_level=0.7, floor=6.0, absolute_max_gain=5.0), dict(attack_ns=0)])
def test_agc_chains_equal_the_reference_order_kernel(G, O, kw):
    """rh_agc.hip takes the AGC apart along its dependency chains (window sum, peak follower, gain) and runs everything else in
    parallel -- the same f32 operations in the same order as one lane walking agc.rs:433-504 sample by sample (k_agc_seq): bit for bit
    on the general path (any release), within one rounding on the default-parameter path, where select and clamp are one median."""
    import torch

    xs = [_programme(95 + s, 40000 + 4 * s) for s in range(4)]
    xs = [x[:40000] for x in xs]
    x = torch.from_numpy(np.stack(xs)).cuda()
    a = G.agc_batch(x, 48000, **kw).cpu().numpy()
    with _env(RH_AGC_SEQ="1"):
        b = G.agc_batch(x, 48000, **kw).cpu().numpy()
    with _env(RH_AGC_VEC="1"):
        c = G.agc_batch(x, 48000, **kw).cpu().numpy()
    assert np.array_equal(c, b)
    if kw.get("release_ns", 0) or kw.get("floor", 0.0) > kw.get("absolute_max_gain", 7.0):
        assert np.array_equal(a, b)  # every operation the reference's
    else:  # release == 0: select and clamps as one median (rh_agc.hip, GainOp0): a rounding apart where two candidates tie
        assert float(np.max(np.abs(a - b))) <= 1e-6


@pytest.mark.parametrize("kw", [dict(), dict(target_level=0.5, attack_ns=10_000_000, release_ns=5_000_000, absolute_max_gain=5.0, floor=0.2)])
@pytest.mark.parametrize("S,n", [(1, 100), (3, 127), (16, 128), (17, 8192 + 300), (5, 40004), (40, 33000), (4200, 520)])  # (4200 streams: more workgroups than CUs)
def test_agc_in_one_kernel_equals_the_segment_by_segment_form(G, O, S, n, kw):
    """k_agc_fused (round 4: a four-stage pipeline inside one workgroup of 16 streams, chunks of 128 samples; other parameters than the
    defaults add the peak follower as a third chain)
    against the chain kernels with the elementwise passes between them (RH_AGC_SEGMENTS=1): the same operations in the same order,
    so the same bits -- rows shorter than a chunk (everything in the epilogue), partial groups of streams, tails, and a state carried
    across blocks that are shorter and longer than the 8192-sample window."""
    import torch

    xs = [_programme(700 + s, n + 8)[:n] for s in range(S)]
    x = torch.from_numpy(np.stack(xs)).cuda()
    a = G.agc_batch(x, 48000, **kw).cpu().numpy()
    with _env(RH_AGC_SEGMENTS="1"):
        b = G.agc_batch(x, 48000, **kw).cpu().numpy()
    assert np.array_equal(a, b)
    for s_ in range(min(S, 3)):
        ref = O.TestSource(xs[s_], 1, 48000).automatic_gain_control(**kw).collect()
        assert float(np.max(np.abs(a[s_] - ref))) <= TOL
    # block streaming: the carried window and the four state words go in and out of the one kernel
    cuts = sorted(set([0, n] + [4 * int(c) for c in np.random.default_rng(S * 1000 + n).integers(1, max(n // 4, 2), 5) if 4 * int(c) < n]))
    st_a, st_b = G.agc_state(S), G.agc_state(S)
    pa = [G.agc_batch(x[:, i:j].contiguous(), 48000, state=st_a, **kw) for i, j in zip(cuts[:-1], cuts[1:])]
    with _env(RH_AGC_SEGMENTS="1"):
        pb = [G.agc_batch(x[:, i:j].contiguous(), 48000, state=st_b, **kw) for i, j in zip(cuts[:-1], cuts[1:])]
    whole = torch.cat(pa, dim=1).cpu().numpy()
    assert float(np.max(np.abs(whole - a))) <= 1e-6  # (a block boundary moves nothing but GainOp0's tie)
    for u, v in zip(pa, pb):
        assert torch.equal(u, v)
    assert torch.equal(st_a, st_b)


@pytest.mark.parametrize("S,n", [(3, 127), (16, 128), (17, 8192 + 300), (5, 40004), (40, 33000), (300, 1500)])
def test_agc_round_5_kernel_equals_round_4s(G, O, S, n):
    """k_agc_fused0 (round 5: the squares, the clamp and desired * (1 - attack) made by the waves around the two chain waves, the desired gain
    with one division, one wait per four vectors) against k_agc_fused<false> (RH_AGC_FUSED_R4=1): who computes an operand does not change
    its bits -- fresh windows, carried windows across uneven blocks, partial groups of streams, tails shorter than a chunk."""
    import torch

    xs = [_programme(900 + s, n + 8)[:n] for s in range(S)]
    x = torch.from_numpy(np.stack(xs)).cuda()
    a = G.agc_batch(x, 48000)
    with _env(RH_AGC_FUSED_R4="1"):
        b = G.agc_batch(x, 48000)
    assert torch.equal(a, b)
    cuts = sorted(set([0, n] + [4 * int(c) for c in np.random.default_rng(S * 77 + n).integers(1, max(n // 4, 2), 4) if 4 * int(c) < n]))
    st_a, st_b = G.agc_state(S), G.agc_state(S)
    pa = [G.agc_batch(x[:, i:j].contiguous(), 48000, state=st_a) for i, j in zip(cuts[:-1], cuts[1:])]
    with _env(RH_AGC_FUSED_R4="1"):
        pb = [G.agc_batch(x[:, i:j].contiguous(), 48000, state=st_b) for i, j in zip(cuts[:-1], cuts[1:])]
    for u, v in zip(pa, pb):
        assert torch.equal(u, v)
    assert torch.equal(st_a, st_b)


@pytest.mark.parametrize("S,n", [(1, 40001), (1, 32770), (3, 40003), (2, 65537)])
def test_agc_one_stream_of_an_odd_length_above_the_square_pass_threshold(G, O, S, n):
    """rows of n >= 32 768 samples take the parallel square pass, whose rows sit BEHIND the per-stream rows in the scratch:
    with n_streams * n_samples not a multiple of 4 (one stream of 40 001 samples) its 16-byte stores and LDS-DMA fetches were 4 to 12
    bytes off.  The region is rounded up to whole vectors now; these shapes are compared with the oracle and the reference-order kernel."""
    import torch

    xs = [_programme(300 + s, n + 8)[:n] for s in range(S)]
    x = torch.from_numpy(np.stack(xs)).cuda()
    a = G.agc_batch(x, 48000).cpu().numpy()
    with _env(RH_AGC_SEQ="1"):
        b = G.agc_batch(x, 48000).cpu().numpy()
    assert float(np.max(np.abs(a - b))) <= 1e-6
    for s_ in range(S):
        ref = O.TestSource(xs[s_], 1, 48000).automatic_gain_control().collect()
        assert float(np.max(np.abs(a[s_] - ref))) <= TOL, s_


# ---- rh_biquad mode 1: the dedicated time-parallel kernel (rh_biquad_scan.hip) ------------------------------------------
def _biquad(G, x, frames, ch, S, co, mode, state=None):
    import ctypes as C

    import torch

    from rodio_amd import _lib

    out = torch.empty_like(x)
    _lib.check(_lib.lib.rh_biquad(C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()), frames, ch, S, co.ctypes.data_as(_lib.f32p),
                                  C.c_void_p(state.data_ptr()) if state is not None else None, mode, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rh_biquad")
    return out


def _truth(x, co, ch):
    from scipy.signal import lfilter

    c = co.astype(np.float64)
    return lfilter(c[:3], [1.0, c[3], c[4]], x.astype(np.float64).reshape(-1, ch), axis=0).reshape(-1)


@pytest.mark.parametrize("ch", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("frames", [1, 2, 3, 255, 512, 513, 4096, 4097, 8192 + 5, 70001])
def test_biquad_mode1_channels_and_boundaries(G, O, ch, frames):
    import torch

    with _env(RH_BIQUAD_NO_FALLBACK="1"):  # the new kernel or nothing
        S = 3 if (frames * ch) % 4 == 0 else 1  # rows of a batch must start on 16-byte boundaries
        xs = [rnd(200 + 7 * ch + s, frames * ch, 0.4) for s in range(S)]  # (the f32 recurrence of a high-pass itself sits ~1e-5 from exact at full scale)
        x = torch.from_numpy(np.stack(xs)).cuda()
        for kind, freq in (("low_pass", 200), ("high_pass", 300)):
            co = G.biquad_coeffs(kind, freq, 0.5, 48000)
            par = _biquad(G, x, frames, ch, S, co, 1).cpu().numpy()
            seq = _biquad(G, x, frames, ch, S, co, 0).cpu().numpy()
            for s in range(S):
                d = float(np.max(np.abs(par[s] - seq[s])))
                t = _truth(xs[s], co, ch)
                e_par, e_seq = float(np.max(np.abs(par[s] - t))), float(np.max(np.abs(seq[s] - t)))
                assert d <= TOL, (kind, s, d)
                assert e_par <= 2.0 * e_seq + 1e-7, (kind, s, e_par, e_seq)
    G.async_status()  # no hand-off inside the scan expired


@pytest.mark.parametrize("ch", [1, 2, 3, 4, 5, 7, 8])
def test_biquad_mode1_state_across_blocks(G, O, ch):
    import torch

    S, frames = 4, 60000
    xs = [rnd(300 + s, frames * ch, 0.8) for s in range(S)]
    co = G.biquad_coeffs("low_pass", 150, 0.5, 44100)
    refs = [O.TestSource(x, ch, 44100).low_pass(150).collect() for x in xs]
    x = torch.from_numpy(np.stack(xs)).cuda()
    state = torch.zeros((S, 4 * ch), device="cuda")
    rng = np.random.default_rng(9)
    outs, a = [], 0
    while a < frames:
        b = min(frames, a + 4 * int(rng.choice([1, 2, 100, 1024, 5000])))  # multiples of 4 frames: rows stay 16-byte aligned
        blk = x[:, a * ch: b * ch].contiguous()
        outs.append(_biquad(G, blk, b - a, ch, S, co, 1, state))
        a = b
    got = torch.cat(outs, dim=1).cpu().numpy()
    for s in range(S):
        assert float(np.max(np.abs(got[s] - refs[s]))) <= TOL, s
    # the state is mode 0's {x1,x2,y1,y2}: a last block through the reference-order kernel continues the stream
    tail = [rnd(400 + s, 1000 * ch) for s in range(S)]
    t = torch.from_numpy(np.stack(tail)).cuda()
    st2 = state.clone()
    a_ = _biquad(G, t, 1000, ch, S, co, 0, state).cpu().numpy()
    b_ = _biquad(G, t, 1000, ch, S, co, 1, st2).cpu().numpy()
    for s in range(S):
        full = O.TestSource(np.concatenate([xs[s], tail[s]]), ch, 44100).low_pass(150).collect()[frames * ch:]
        assert float(np.max(np.abs(a_[s] - full))) <= TOL and float(np.max(np.abs(b_[s] - full))) <= TOL


def test_biquad_mode1_long_memory_and_many_streams(G, O):
    import torch

    # low_pass(20) at 48 kHz: poles at ~0.9974, the carry reaches back several 4096-frame tiles; 96 streams in one launch
    S, frames, ch = 96, 40000, 2
    xs = [rnd(500 + s, frames * ch, 0.2) for s in range(S)]  # (poles this close to z = 1: the f32 reference recurrence itself is ~1e-5 from exact at half scale)
    co = G.biquad_coeffs("low_pass", 20, 0.5, 48000)
    x = torch.from_numpy(np.stack(xs)).cuda()
    par = _biquad(G, x, frames, ch, S, co, 1).cpu().numpy()
    for s in (0, 17, S - 1):
        ref = O.TestSource(xs[s], ch, 48000).low_pass(20).collect()
        t = _truth(xs[s], co, ch)
        assert float(np.max(np.abs(par[s] - ref))) <= TOL
        assert float(np.max(np.abs(par[s] - t))) <= 2.0 * float(np.max(np.abs(ref - t))) + 1e-7


# ---- the runtime pieces behind the handle-less kernels ---------------------------------------------------------------------
def test_rh_memset_any_alignment_and_size(G):
    """rh_memset is a kernel of the library (rh::fill_async), not hipMemsetAsync: unaligned heads and tails, every size class."""
    import ctypes as C

    import torch

    from rodio_amd import _lib

    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    ref = np.zeros(1 << 16, np.uint8)
    rng = np.random.default_rng(5)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for off, n in [(0, 1 << 16), (1, 1), (3, 2), (5, 3), (2, 7), (1, 4097), (4, 4096), (7, 30001), (0, 0), (65535, 1)] + [(int(rng.integers(0, 60000)), int(rng.integers(1, 5000))) for _ in range(40)]:
        val = int(rng.integers(0, 256))
        _lib.check(_lib.lib.rh_memset(C.c_void_p(buf.data_ptr() + off), val, n, stream), "rh_memset")
        ref[off: off + n] = val
    assert np.array_equal(buf.cpu().numpy(), ref)


def test_stream_scratch_grows_and_shrinks_between_calls(G, O):
    """The per-stream scratch of the scan kernels starts at 1 MiB; a batch of 30 000 short streams needs more (one record per
    stream): the buffer is replaced behind a stream synchronise, and calls before and after are unaffected."""
    import torch

    def run(S, frames, seed):
        xs = [rnd(seed + s, frames * 2, 0.9) for s in range(S)]
        out = G.limit_batch(torch.from_numpy(np.stack(xs)).cuda(), 2, 48000).cpu().numpy()
        pick = np.random.default_rng(seed).choice(S, size=min(S, 40), replace=False)
        for s in pick:
            ref = O.TestSource(xs[s], 2, 48000).limit().collect()
            assert float(np.max(np.abs(out[s] - ref))) <= TOL, (S, frames, int(s))

    run(3, 5000, 7000)
    run(30000, 64, 8000)  # 30 000 records of 48 bytes: beyond the first MiB
    run(5, 3000, 9000)
    G.async_status()


def test_agc_without_a_state_in_place(G, O):
    """rh_agc with state == NULL and dst == src takes the ring kernel with its state in the stream's scratch."""
    import torch

    xs = [_programme(1300 + s, 30000) for s in range(6)]
    x = torch.from_numpy(np.stack(xs)).cuda()
    out = G.agc_batch(x, 44100, out=x).cpu().numpy()
    for s in range(6):
        ref = O.TestSource(xs[s], 2, 44100).automatic_gain_control().collect()
        assert float(np.max(np.abs(out[s] - ref))) <= TOL, s


def test_in_place_calls(G, O):
    """dst == src: the limiter works in place (a tile reads only its own frames); rh_biquad mode 1 would read the two frames in
    front of a share after a neighbour has overwritten them, so an in-place call takes mode 0 -- bit-exact to the oracle."""
    import torch

    S, frames, ch = 5, 40000, 2
    xs = [rnd(1500 + s, frames * ch, 0.9) for s in range(S)]
    x = torch.from_numpy(np.stack(xs)).cuda()
    co = G.biquad_coeffs("low_pass", 300, 0.5, 48000)
    y = x.clone()
    _biquad_inplace = y
    from rodio_amd import _lib
    import ctypes as C

    _lib.check(_lib.lib.rh_biquad(C.c_void_p(y.data_ptr()), C.c_void_p(y.data_ptr()), frames, ch, S, co.ctypes.data_as(_lib.f32p), None, 1,
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rh_biquad")
    got = _biquad_inplace.cpu().numpy()
    for s in range(S):
        assert np.array_equal(got[s], O.TestSource(xs[s], ch, 48000).low_pass(300).collect()), s
    z = x.clone()
    G.limit_batch(z, ch, 48000, out=z)
    ref = G.limit_batch(x, ch, 48000)
    assert torch.equal(z, ref)
    G.async_status()

"""Seeded random shapes for the two time-parallel kernels (rh_limit.hip, rh_biquad_scan.hip): stream counts from one to
hundreds, lengths on and off every tile size, 1-8 channels, with and without a state carried across random block splits.
The limiter against the oracle, rh_biquad mode 1 against mode 0 (the reference-order kernel, itself bit-exact to the oracle)."""
import ctypes as C

import numpy as np
import pytest
from conftest import knobs  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def G(rh):
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    rh.init(0)
    return rh


def _shapes(seed, n_cases, max_samples):
    rng = np.random.default_rng(seed)
    lengths = [1, 2, 63, 64, 255, 256, 511, 512, 513, 1023, 1024, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193, 12288, 16384, 20000, 40000]
    out = []
    while len(out) < n_cases:
        ch = int(rng.choice([1, 2, 2, 2, 3, 4, 5, 6, 7, 8]))
        frames = int(rng.choice(lengths)) if rng.random() < 0.6 else int(rng.integers(1, 30000))
        cap = max(1, max_samples // (frames * ch))
        S = int(min(cap, rng.choice([1, 1, 2, 3, 8, 33, 64, 130, 300, 700])))
        if S > 1 and (frames * ch) % 4:  # rows of a batch start on 16-byte boundaries
            frames += (4 - (frames * ch) % 4) % 4 if ch in (1, 2, 4, 8) else 0
            if (frames * ch) % 4:
                frames = frames // 4 * 4 + 4
        out.append((ch, S, frames, bool(rng.random() < 0.4)))
    return out


def _splits(rng, frames, ch, aligned):
    cuts, a = [], 0
    while a < frames:
        step = int(rng.choice([1, 3, 64, 500, 777, 2048, 5000, 8192, 20000]))
        if aligned:  # rows of a multi-stream block stay 16-byte aligned
            step = max(4, step // 4 * 4)
        b = min(frames, a + step)
        cuts.append((a, b))
        a = b
    return cuts


@pytest.mark.parametrize("case", range(40))
def test_limiter_random_shapes(G, O, case):
    import torch

    ch, S, frames, carry = _shapes(4242, 40, 1_500_000)[case]
    rng = np.random.default_rng(10_000 + case)
    xs = [(rng.uniform(-1, 1, frames * ch) * rng.choice([0.3, 0.9, 2.5])).astype(np.float32) for _ in range(S)]
    x = torch.from_numpy(np.stack(xs)).cuda()
    if carry:
        state = torch.zeros((S, 2 * ch), device="cuda")
        parts = [G.limit_batch(x[:, a * ch: b * ch].contiguous(), ch, 48000, state=state) for a, b in _splits(rng, frames, ch, S > 1)]
        got = torch.cat(parts, dim=1).cpu().numpy()
    else:
        got = G.limit_batch(x, ch, 48000).cpu().numpy()
    for s in rng.choice(S, size=min(S, 12), replace=False):
        ref = O.TestSource(xs[s], ch, 48000).limit().collect()
        assert float(np.max(np.abs(got[s] - ref))) <= TOL, (ch, S, frames, carry, int(s))
    G.async_status()


def _biquad(G, x, frames, ch, S, co, mode, state=None):
    import torch

    from rodio_amd import _lib

    out = torch.empty_like(x)
    _lib.check(_lib.lib.rh_biquad(C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()), frames, ch, S, co.ctypes.data_as(_lib.f32p),
                                  C.c_void_p(state.data_ptr()) if state is not None else None, mode, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rh_biquad")
    return out


@pytest.mark.parametrize("case", range(40))
def test_biquad_mode1_random_shapes(G, O, case):
    import os

    import torch

    ch, S, frames, carry = _shapes(777, 40, 3_000_000)[case]
    rng = np.random.default_rng(20_000 + case)
    kind, freq = [("low_pass", 200), ("low_pass", 1000), ("high_pass", 300), ("low_pass", 60)][case % 4]
    co = G.biquad_coeffs(kind, freq, 0.5, 48000)
    xs = [(rng.uniform(-1, 1, frames * ch) * 0.3).astype(np.float32) for _ in range(S)]
    x = torch.from_numpy(np.stack(xs)).cuda()
    with knobs(RH_BIQUAD_NO_FALLBACK="1"):  # the scan kernel or nothing
        if carry:
            st1, st0 = torch.zeros((S, 4 * ch), device="cuda"), torch.zeros((S, 4 * ch), device="cuda")
            p1, p0 = [], []
            for a, b in _splits(rng, frames, ch, True):  # mode 1 wants 16-byte aligned rows
                blk = x[:, a * ch: b * ch].contiguous()
                if (blk.data_ptr() % 16) or ((b - a) * ch) % 4 and S > 1:
                    p1.append(_biquad(G, blk, b - a, ch, S, co, 0, st1))  # (a block the scan does not take continues in mode 0: same state)
                else:
                    p1.append(_biquad(G, blk, b - a, ch, S, co, 1, st1))
                p0.append(_biquad(G, blk, b - a, ch, S, co, 0, st0))
            par, seq = torch.cat(p1, dim=1), torch.cat(p0, dim=1)
        else:
            par, seq = _biquad(G, x, frames, ch, S, co, 1), _biquad(G, x, frames, ch, S, co, 0)
    assert float((par - seq).abs().max()) <= TOL, (ch, S, frames, carry, kind, freq)
    # and against the oracle's BltFilter itself (blt.rs:397-560), not only against the reference-order kernel: first and last stream
    for s_ in sorted({0, S - 1}):
        src = O.TestSource(xs[s_], ch, 48000)
        ref = (src.low_pass(freq) if kind == "low_pass" else src.high_pass(freq)).collect()
        assert float(np.max(np.abs(par[s_].cpu().numpy() - ref))) <= TOL, (s_, ch, S, frames, carry, kind, freq)
    G.async_status()

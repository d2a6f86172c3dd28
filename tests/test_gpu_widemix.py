"""rh_wide_mix_block: a block of a mixer of any channel count in ONE launch -- per source Amplify -> UniformSourceIterator(channels, rate)
(amplify.rs:64, uniform.rs:58-67: SampleRateConverter, then ChannelCountConverter) and the ordered sum (mixer.rs:185-198).  The oracle is the
per-sample chain `mixer::mixer(ch, rate)` + `add(src.amplify(g))` over continuous sources (current_span_len() == None).  Bit-exact.
The block planner below is the one GpuMixer runs (include/rodio_hip.hpp, wide generations): what a block can emit, where its first taps lie."""
import ctypes as C
from math import gcd

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def lerp_ready(n, F, T):  # #m with floor(m F / T) <= n - 2: both taps of the lerp exist
    return 0 if n == 0 else ((n - 1) * T + F - 1) // F


def out_frames(n, F, T):  # ... plus the verbatim last frame, when an output frame lands on it (sample_rate.rs:193-200)
    if n == 0:
        return 0
    if F == T:
        return n
    c1 = lerp_ready(n, F, T)
    return c1 + (1 if c1 * F < n * T else 0)


class Pump:
    """Sources arrive block by block (`feed` frames a pull); a block emits what every live source has both taps for."""

    def __init__(self, rh, srcs, to_ch, to_rate):
        import torch

        self.torch = torch
        self.to_ch, self.to_rate = to_ch, to_rate
        self.s = []
        for x, ch, rate, gain in srcs:
            g = gcd(rate, to_rate)
            self.s.append(dict(x=x, ch=ch, rate=rate, gain=gain, F=rate // g, T=to_rate // g, n=len(x) // ch, fed=0, pos=0, dev=torch.from_numpy(x).cuda() if len(x) else torch.zeros(1, device="cuda"), ended=len(x) == 0))
        self.m = 0

    def block(self, feeds, cap):
        from rodio_amd import _lib, source

        source._ensure()
        for s, f in zip(self.s, feeds):
            if not s["ended"]:
                s["fed"] = min(s["n"], s["fed"] + f)
                if s["fed"] == s["n"]:
                    s["ended"] = True
        live = [lerp_ready(s["fed"], s["F"], s["T"]) if s["F"] != s["T"] else s["fed"] for s in self.s if not s["ended"]]
        m_end = min(live) if live else max([out_frames(s["n"], s["F"], s["T"]) for s in self.s] + [0])
        m_end = max(self.m, min(m_end, self.m + cap))
        out = m_end - self.m
        if out == 0:
            return np.zeros(0, np.float32), not live and m_end >= max([out_frames(s["n"], s["F"], s["T"]) for s in self.s] + [0])
        arr = (_lib.WideSrc * len(self.s))()
        for k, s in enumerate(self.s):
            F, T = s["F"], s["T"]
            end = out_frames(s["n"], F, T) if s["ended"] else m_end
            i0 = self.m * F // T
            arr[k].frames = max(0, min(end, m_end) - self.m)
            arr[k].data = s["dev"].data_ptr() + 4 * i0 * s["ch"]
            arr[k].channels, arr[k].from_rate = s["ch"], s["rate"]
            arr[k].phase = self.m * F % T
            arr[k].last = (s["n"] - 1 - i0) if s["ended"] and s["n"] - 1 >= i0 else (0 if s["ended"] else 0xFFFFFFFF)
            arr[k].gain = s["gain"]
            # what the planner promises: every tap of a live source lies in what has been fed
            if arr[k].frames and not s["ended"]:
                assert (m_end - 1) * F // T + (0 if F == T else 1) <= s["fed"] - 1
        dst = self.torch.full((out * self.to_ch,), float("nan"), device="cuda")
        _lib.check(_lib.lib.rh_wide_mix_block(C.c_void_p(dst.data_ptr()), self.to_ch, self.to_rate, out, arr, len(self.s), source._stream()), "rh_wide_mix_block")
        self.m = m_end
        done = not live and m_end >= max([out_frames(s["n"], s["F"], s["T"]) for s in self.s] + [0])
        return dst.cpu().numpy(), done


def _oracle(srcs, to_ch, to_rate):
    from oracle import rodio_oracle as O

    mx = O.Mixer(to_ch, to_rate)
    for x, ch, rate, gain in srcs:
        s = O.TestSource(x, ch, rate)
        if gain != 1.0:
            s = s.amplify(gain)
        mx.add(s)
    return mx.collect()


def _run(rh, srcs, to_ch, to_rate, rng, feed=4096, cap=1 << 20):
    p = Pump(rh, srcs, to_ch, to_rate)
    parts = []
    for _ in range(100000):
        feeds = [int(rng.integers(feed // 2, feed + 1)) for _ in srcs]
        o, done = p.block(feeds, cap)
        parts.append(o)
        if done:
            break
    else:
        raise AssertionError("the pump never finished")
    return np.concatenate(parts) if parts else np.zeros(0, np.float32)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("to_ch,to_rate", [(6, 48000), (4, 44100), (8, 96000), (3, 22050), (2, 48000), (1, 48000)])
def test_wide_mix_blocks_bit_exact(rh, to_ch, to_rate):
    rng = np.random.default_rng(600 + to_ch)
    layouts = [(6, 44100, 1.0), (2, 44100, 0.5), (1, 48000, 1.0), (8, 96000, 0.25), (to_ch, to_rate, 1.0), (6, 32000, 2.0), (4, 8000, 1.0), (3, 11025, -1.5)]
    srcs = []
    for ch, rate, gain in layouts:
        n = int(rng.integers(2000, 30000))
        srcs.append((rng.uniform(-1, 1, n * ch).astype(np.float32), ch, rate, gain))
    want = _oracle(srcs, to_ch, to_rate)
    got = _run(rh, srcs, to_ch, to_rate, rng)
    assert got.shape == want.shape
    assert np.array_equal(_bits(got), _bits(want))


def test_wide_mix_more_sources_than_a_launch_holds(rh):
    """40 sources: the table of a launch holds 32, the rest continue from the stored partial sum -- the same left-to-right additions."""
    rng = np.random.default_rng(77)
    srcs = [(rng.uniform(-1, 1, int(rng.integers(500, 6000)) * ch).astype(np.float32), ch, rate, float(rng.uniform(0.1, 2)))
            for ch, rate in [(6, 44100), (2, 48000), (1, 22050), (6, 48000)] * 10]
    want = _oracle(srcs, 6, 48000)
    got = _run(rh, srcs, 6, 48000, rng, feed=1500)
    assert np.array_equal(_bits(got), _bits(want))


def test_wide_mix_edges(rh):
    """Sources of one and two frames, an empty one, blocks capped at a few frames, special values; a source whose last frame an output lands on
    (verbatim) and one where none does."""
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, 6 * 37).astype(np.float32)
    x[5] = np.inf
    x[17] = -0.0
    x[6 * 36 + 2] = np.nan
    srcs = [(x, 6, 44100, 1.0), (np.float32([0.25, -0.5]), 2, 48000, 1.0), (np.float32([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]), 6, 32000, 1.0),
            (np.zeros(0, np.float32), 6, 44100, 1.0), (rng.uniform(-1, 1, 160 * 6).astype(np.float32), 6, 48001, 1.0), (rng.uniform(-1, 1, 147 * 2 + 2).astype(np.float32), 2, 44100, 1.0)]
    want = _oracle(srcs, 6, 48000)
    for feed, cap in ((16, 7), (1000, 1 << 20), (3, 1 << 20)):
        got = _run(rh, srcs, 6, 48000, np.random.default_rng(1), feed=feed, cap=cap)
        assert got.shape == want.shape
        assert np.array_equal(_bits(got), _bits(want)), (feed, cap)


def test_wide_mix_arguments(rh):
    import torch

    from rodio_amd import _lib

    d = torch.zeros(64, device="cuda")
    s = (_lib.WideSrc * 1)()
    s[0].data, s[0].channels, s[0].from_rate, s[0].phase, s[0].frames, s[0].last, s[0].gain = d.data_ptr(), 2, 44100, 0, 4, 0xFFFFFFFF, 1.0
    f = _lib.lib.rh_wide_mix_block
    assert f(C.c_void_p(d.data_ptr()), 6, 48000, 0, s, 1, None) == 0  # nothing to do
    assert f(None, 6, 48000, 4, s, 1, None) == 1  # RH_ERR_INVALID
    assert f(C.c_void_p(d.data_ptr()), 0, 48000, 4, s, 1, None) == 1
    s[0].frames = 5  # more than the block
    assert f(C.c_void_p(d.data_ptr()), 6, 48000, 4, s, 1, None) == 1
    s[0].frames, s[0].phase = 4, 160  # phase is a remainder mod T = 160
    assert f(C.c_void_p(d.data_ptr()), 6, 48000, 4, s, 1, None) == 1
    s[0].phase, s[0].from_rate = 0, 4294967291  # F * T beyond u32 (sample_rate.rs:157)
    assert f(C.c_void_p(d.data_ptr()), 6, 48000, 4, s, 1, None) == 3  # RH_ERR_UNSUPPORTED


def _one_shot(rh, srcs, to_ch, to_rate):
    """Every source whole and ended: ONE call of rh_wide_mix_block gives the whole mix."""
    rng = np.random.default_rng(0)
    return _run(rh, srcs, to_ch, to_rate, rng, feed=1 << 20)


def test_wide_mix_replays_the_references_own_vectors(rh):
    """The numbers rodio's tests hold for the three iterators this kernel folds into one -- through the kernel, not through the oracle:
    src/mixer.rs:208-256 (basic, channels_conv, rate_conv), src/conversions/sample_rate.rs:356-387, src/conversions/channels.rs:114-143
    (a SamplesBuffer of a few samples is one span: the same stream as a continuous source)."""
    f = np.float32
    a, b = f([10.0, -10.0, 10.0, -10.0]), f([5.0, 5.0, 5.0, 5.0])
    assert _one_shot(rh, [(a, 1, 48000, 1.0), (b, 1, 48000, 1.0)], 1, 48000).tolist() == [15.0, -5.0, 15.0, -5.0]                               # mixer.rs:208-222
    assert _one_shot(rh, [(a, 1, 48000, 1.0), (b, 1, 48000, 1.0)], 2, 48000).tolist() == [15.0, 15.0, -5.0, -5.0, 15.0, 15.0, -5.0, -5.0]       # :224-242
    assert _one_shot(rh, [(a, 1, 48000, 1.0), (b, 1, 48000, 1.0)], 1, 96000).tolist() == [15.0, 5.0, -5.0, 5.0, 15.0, 5.0, -5.0]                 # :244-256
    out = _one_shot(rh, [(f([2.0, 16.0, 4.0, 18.0, 6.0, 20.0, 8.0, 22.0]), 2, 2000, 1.0)], 2, 3000)                                               # sample_rate.rs:356-366
    assert np.trunc(out).tolist() == [2.0, 16.0, 3.0, 17.0, 4.0, 18.0, 6.0, 20.0, 7.0, 21.0, 8.0, 22.0]
    assert np.trunc(_one_shot(rh, [(f([1.0, 14.0]), 1, 1000, 1.0)], 1, 7000)).tolist() == [1.0, 2.0, 4.0, 6.0, 8.0, 10.0, 12.0, 14.0]             # :368-376
    assert _one_shot(rh, [(np.arange(17, dtype=f), 1, 12000, 1.0)], 1, 2400).tolist() == [0.0, 5.0, 10.0, 15.0]                                   # :378-387
    assert _one_shot(rh, [(f([1, 2, 3, 4, 5, 6]), 3, 1, 1.0)], 2, 1).tolist() == [1, 2, 4, 5]                                                     # channels.rs:114-143
    assert _one_shot(rh, [(f([1, 2, 3, 4, 5, 6, 7, 8]), 4, 1, 1.0)], 1, 1).tolist() == [1, 5]
    assert _one_shot(rh, [(f([1, 2, 3, 4]), 1, 1, 1.0)], 2, 1).tolist() == [1, 1, 2, 2, 3, 3, 4, 4]
    assert _one_shot(rh, [(f([1, 2]), 1, 1, 1.0)], 4, 1).tolist() == [1, 1, 0, 0, 2, 2, 0, 0]
    assert _one_shot(rh, [(f([1, 2, 3, 4]), 2, 1, 1.0)], 4, 1).tolist() == [1, 2, 0, 0, 3, 4, 0, 0]


def test_wide_mix_properties_at_full_block_sizes(rh):
    """Size-independent properties at a block the oracle would need minutes for (64 sources x 256 Ki frames of 5.1): from == to is the identity
    (sample_rate.rs:254-270), up by k then every k-th frame is the input (:318-334), the mix is linear in the gains -- and the order of the
    additions is the insertion order (a permutation of the sources changes the bits of a sum of many, the same order never does)."""
    import torch

    from rodio_amd import _lib, source

    source._ensure()
    rng = np.random.default_rng(9)
    S, M, Cc = 64, 1 << 18, 6
    rows = [torch.from_numpy(rng.uniform(-1, 1, (M + 2) * Cc).astype(np.float32)).cuda() for _ in range(S)]

    def mix(order, gains, rate, to_rate, frames):
        arr = (_lib.WideSrc * len(order))()
        for k, s in enumerate(order):
            arr[k].data, arr[k].channels, arr[k].from_rate, arr[k].phase, arr[k].frames, arr[k].last, arr[k].gain = rows[s].data_ptr(), Cc, rate, 0, frames, 0xFFFFFFFF, gains[k]
        dst = torch.empty(frames * Cc, device="cuda")
        _lib.check(_lib.lib.rh_wide_mix_block(C.c_void_p(dst.data_ptr()), Cc, to_rate, frames, arr, len(order), source._stream()), "rh_wide_mix_block")
        return dst

    one = mix([3], [1.0], 48000, 48000, M)
    assert torch.equal(one, rows[3][: M * Cc])
    up = mix([5], [1.0], 16000, 48000, 3 * (M // 4))
    assert torch.equal(up.view(-1, Cc)[::3], rows[5][: (M // 4) * Cc].view(-1, Cc))
    order = list(range(S))
    g = [0.5] * S
    m1 = mix(order, g, 44100, 48000, M)
    assert torch.equal(m1, mix(order, g, 44100, 48000, M))                       # run after run: the same bits
    m2 = mix(order, [1.0] * S, 44100, 48000, M)
    assert torch.equal(m1 * 2, m2)                                               # a power of two scales every term and every partial sum exactly
    rev = mix(order[::-1], g, 44100, 48000, M)
    assert not torch.equal(rev, m1) and float((rev - m1).abs().max()) < 1e-4     # another order: other roundings of the same sum
    # ... and against the f64 sum of the same taps: the ordered f32 sum stays within S ulps of a sum of 64 terms of size < 1
    i = (torch.arange(M, device="cuda", dtype=torch.int64) * 147) // 160
    w = ((torch.arange(M, device="cuda", dtype=torch.int64) * 147) % 160).to(torch.float64) / 160.0
    ref = torch.zeros(M, Cc, dtype=torch.float64, device="cuda")
    for s in order:
        x = rows[s].view(-1, Cc).to(torch.float64)
        ref += 0.5 * (x[i] + (x[i + 1] - x[i]) * w[:, None])
    assert float((m1.view(-1, Cc).to(torch.float64) - ref).abs().max()) < 64 * 4e-6


@pytest.mark.parametrize("ch,rate,to_rate", [(6, 44100, 48000), (2, 44100, 48000), (6, 48000, 48000), (4, 96000, 44100), (1, 22050, 48000)])
def test_wide_mix_alike_sources_take_the_uniform_kernel(rh, ch, rate, to_rate):
    """Sources of the mixer's own layout and one rate: while all of them are live a block is k_wide_mix_uniform's (tap offset and weight worked out
    once per lane); the blocks in which one of them ends go back to the general kernel.  40 sources (two launches a block), different lengths
    and gains: the oracle's mixer, bit for bit."""
    rng = np.random.default_rng(800 + ch)
    srcs = [(rng.uniform(-1, 1, int(rng.integers(9000, 30000)) * ch).astype(np.float32), ch, rate, float(np.float32(rng.uniform(0.2, 1.5)))) for _ in range(40)]
    want = _oracle(srcs, ch, to_rate)
    got = _run(rh, srcs, ch, to_rate, rng, feed=3000)
    assert got.shape == want.shape
    assert np.array_equal(_bits(got), _bits(want))

"""rh_uniform_segments: UniformSourceIterator span by span (src/source/uniform.rs:50-97) as independent segments.
The oracle is the per-sample chain `UniformSourceIterator::new(src, to_ch, to_rate)` over a source that reports spans
(SamplesBuffer: buffer.rs:76-82; a fixed span length like a decoder's packets).  Bit-exact."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
U64_MAX = (1 << 64) - 1


def _span_frames(lib, n, fr, to, complete):
    m = C.c_uint64(0)
    assert lib.rh_uniform_span_frames(n, fr, to, int(complete), C.byref(m)) == 0
    return m.value


def _first_tap(lib, m, fr, to):
    i = C.c_uint64(0)
    assert lib.rh_uniform_first_tap(m, fr, to, C.byref(i)) == 0
    return i.value


def _convert(rh, x, ch, rate, to_ch, to_rate, span_samples, cuts, gain=1.0, dev_table=False):
    """Converts x (interleaved, `ch` channels) span by span; every span is cut into pieces at the relative positions
    `cuts` (fractions of the span), each piece one segment -- the way a block-streaming caller meets them."""
    import torch

    from rodio_amd import _lib, source

    source._ensure()
    lib = _lib.lib
    frames = len(x) // ch
    span_f = frames if span_samples is None else min(span_samples, 32768) // ch
    d_in = torch.from_numpy(x).cuda()
    segs, off_out = [], 0
    spans = []
    f0 = 0
    while f0 < frames:
        n = min(span_f, frames - f0)
        spans.append((f0, n))
        f0 += n
    total = sum(_span_frames(lib, n, rate, to_rate, True) for _, n in spans)
    d_out = torch.full((max(total, 1) * to_ch,), float("nan"), device="cuda")
    for f0, n in spans:
        marks = sorted(set(min(n, max(1, int(round(c * n)))) for c in cuts) | {n})
        m_done = 0
        for k, upto in enumerate(marks):
            closed = upto == n
            ready = _span_frames(lib, upto, rate, to_rate, closed)
            if ready > m_done:
                first = min(_first_tap(lib, m_done, rate, to_rate), upto - 1)
                s = _lib.UniformSeg()
                s.src = d_in.data_ptr() + 4 * (f0 + first) * ch
                s.dst = d_out.data_ptr() + 4 * (off_out + m_done) * to_ch
                s.src_frame0, s.src_frames = first, upto - first
                s.m0, s.m1 = m_done, ready
                s.span_frames = n if closed else U64_MAX
                s.from_rate, s.to_rate, s.from_ch, s.to_ch, s.gain, s.reserved = rate, to_rate, ch, to_ch, gain, 0
                segs.append(s)
                m_done = ready
        off_out += m_done
    assert off_out == total
    arr = (_lib.UniformSeg * len(segs))(*segs)
    if dev_table:
        raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
        d_tab = torch.from_numpy(raw).cuda()
        most = max(s.m1 - s.m0 for s in segs)
        _lib.check(lib.rh_uniform_segments_dev(C.c_void_p(d_tab.data_ptr()), len(segs), most, source._stream()), "rh_uniform_segments_dev")
    else:
        _lib.check(lib.rh_uniform_segments(arr, len(segs), source._stream()), "rh_uniform_segments")
    torch.cuda.synchronize()
    return d_out[: total * to_ch].cpu().numpy()


CASES = [
    # ch, rate, to_ch, to_rate, frames, span (samples; None = one continuous stream)
    (2, 44100, 2, 48000, 100000, "buffer"),
    (1, 44100, 2, 48000, 70001, "buffer"),
    (2, 48000, 2, 44100, 50000, "buffer"),
    (2, 44100, 2, 48000, 40000, 2304),     # an MP3 decoder's packets: 1152 stereo frames
    (2, 22050, 1, 48000, 9000, 512),
    (4, 96000, 2, 8000, 30000, 4096),      # 12 : 1 down, surplus channels dropped
    (1, 8000, 6, 48000, 3000, 100),
    (2, 48000, 2, 48000, 20000, "buffer"),  # from == to: the converter passes through (sample_rate.rs:133-136)
    (3, 48000, 2, 48000, 9999, 30),
    (2, 44100, 2, 48000, 30000, None),
    (2, 44100, 2, 48000, 1, "buffer"),
    (2, 44100, 2, 48000, 2, 2),             # spans of a single frame: each is its own verbatim last frame
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("cuts", [(), (0.5,), (0.1, 0.11, 0.7, 0.99)])
def test_uniform_segments_equal_the_per_sample_chain(O, rh, case, cuts):
    ch, rate, to_ch, to_rate, frames, span = CASES[case]
    x = (np.random.default_rng(900 + case).uniform(-1, 1, frames * ch)).astype(np.float32)
    if span == "buffer":
        src, span_samples = O.SamplesBuffer(ch, rate, x), len(x)
    elif span is None:
        src, span_samples = O.TestSource(x, ch, rate), None
    else:
        src, span_samples = O.SpanSource(x, ch, rate, span), span
    ref = O.UniformSourceIterator(src, to_ch, to_rate).collect()
    got = _convert(rh, x, ch, rate, to_ch, to_rate, span_samples, cuts, dev_table=bool(case % 2))
    assert len(got) == len(ref), (len(got), len(ref))
    assert np.array_equal(got, ref), int(np.argmax(got != ref))


def test_uniform_segments_gain_sits_in_front_of_the_converter(O, rh):
    # mixer.add(src.amplify(g)): Amplify first, then Mixer::add's UniformSourceIterator (mixer.rs:58-66)
    x = np.random.default_rng(77).uniform(-1, 1, 2 * 50000).astype(np.float32)
    g = float(np.float32(0.37))
    ref = O.UniformSourceIterator(O.SamplesBuffer(2, 44100, x).amplify(g), 2, 48000).collect()
    got = _convert(rh, x, 2, 44100, 2, 48000, len(x), (0.3,), gain=g)
    assert np.array_equal(got, ref)


def test_uniform_segments_refuses_what_it_cannot_read(rh):
    import torch

    from rodio_amd import _lib, source

    source._ensure()
    d = torch.zeros(64, device="cuda")
    s = _lib.UniformSeg()
    s.src, s.dst = d.data_ptr(), d.data_ptr()
    s.src_frame0, s.src_frames, s.m0, s.m1, s.span_frames = 0, 4, 0, 8, U64_MAX  # output frame 7 of 44.1 -> 48 k reads input frames 6 and 7
    s.from_rate, s.to_rate, s.from_ch, s.to_ch, s.gain = 44100, 48000, 2, 2, 1.0
    assert _lib.lib.rh_uniform_segments(C.byref(s), 1, None) == 1  # RH_ERR_INVALID
    s.src_frames = 8
    assert _lib.lib.rh_uniform_segments(C.byref(s), 1, None) == 0
    s.from_ch = 0
    assert _lib.lib.rh_uniform_segments(C.byref(s), 1, None) == 1
    s.from_ch, s.from_rate, s.to_rate = 2, 4294967291, 4294967279  # F * T overflows u32 (sample_rate.rs:45-47)
    assert _lib.lib.rh_uniform_segments(C.byref(s), 1, None) == 3  # RH_ERR_UNSUPPORTED

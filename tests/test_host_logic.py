"""Host-side logic of the C ABI (no GPU needed) against the oracle: output-length closed form of
the resampler incl. span chunking, biquad coefficients, spatial gains, delay length."""
import ctypes as C

import numpy as np
import pytest


def _out_frames(rh, n, frm, to, ch, span):
    from rodio_amd import _lib

    o = C.c_uint64(0)
    st = _lib.lib.rh_resample_out_frames(n, frm, to, ch, span, C.byref(o))
    return st, o.value


RATES = [(44100, 48000), (48000, 44100), (8000, 48000), (48000, 8000), (44100, 40000), (11025, 48000),
         (2000, 3000), (1000, 7000), (12000, 2400), (48000, 96000), (22050, 22050)]


@pytest.mark.parametrize("frm,to", RATES)
@pytest.mark.parametrize("ch", [1, 2, 3])
def test_out_frames_unchunked(rh, O, frm, to, ch):
    for n in [0, 1, 2, 3, 4, 5, 17, 100, 147, 160, 161, 1000]:
        x = np.arange(n * ch, dtype=np.float32)
        ref = O.SampleRateConverter(O.TestSource(x, ch, frm), frm, to, ch).collect()
        st, m = _out_frames(rh, n, frm, to, ch, 0)
        assert st == 0
        assert m * ch == len(ref), (n, frm, to, ch)


@pytest.mark.parametrize("frm,to", [(44100, 48000), (48000, 44100), (8000, 48000), (48000, 8000)])
@pytest.mark.parametrize("span,ch", [(64, 2), (100, 2), (32768, 2), (96, 1), (10, 1), (1 << 20, 2)])
def test_out_frames_chunked(rh, O, frm, to, span, ch):
    # UniformSourceIterator restarts the converter every min(span, 32768) samples (uniform.rs:56)
    for n in [0, 1, 31, 32, 33, 50, 51, 99, 100, 101, 250, 16384, 16385, 40000]:
        x = np.arange(n * ch, dtype=np.float32)
        ref = O.UniformSourceIterator(O.SpanSource(x, ch, frm, span), ch, to).collect()
        st, m = _out_frames(rh, n, frm, to, ch, span)
        assert st == 0
        assert m * ch == len(ref), (n, frm, to, span, ch)


def test_out_frames_rejects(rh):
    assert _out_frames(rh, 10, 0, 48000, 2, 0)[0] == 1        # zero rate: rodio's NonZero
    assert _out_frames(rh, 10, 44100, 48000, 0, 0)[0] == 1
    assert _out_frames(rh, 10, 44100, 48000, 6, 32768)[0] == 3  # span that splits a frame
    assert _out_frames(rh, 10, 4294967291, 4294967279, 1, 0)[0] == 3  # from*to overflows u32 (sample_rate.rs:45-47)


@pytest.mark.parametrize("kind", ["low_pass", "high_pass"])
@pytest.mark.parametrize("freq,q,fs", [(200, 0.5, 48000), (1000, 0.5, 48000), (300, 0.5, 44100), (5000, 0.7071, 96000),
                                        (20, 0.5, 48000), (200, 2.0, 8000)])
def test_biquad_coeffs_bit_exact(rh, O, kind, freq, q, fs):
    assert np.array_equal(rh.biquad_coeffs(kind, freq, q, fs), O.blt_coeffs(kind, freq, q, fs))


def test_spatial_gains_bit_exact(rh, O):
    rng = np.random.default_rng(3)
    for _ in range(200):
        e, l, r = rng.uniform(-3, 3, (3, 3)).astype(np.float32)
        assert np.array_equal(rh.spatial_gains(e, l, r), O.spatial_gains(e, l, r))
    # BASELINE config 3 geometry
    for s in range(64):
        e = [0.5 + 0.01 * s, 0, 1]
        assert np.array_equal(rh.spatial_gains(e, [-1, 0, 0], [1, 0, 0]), O.spatial_gains(e, [-1, 0, 0], [1, 0, 0]))


def test_delay_samples(rh, O):
    # 682_666_667 ns -> 65536 samples at 48 kHz stereo; one ns less -> 65535 (odd!)
    assert rh.delay_samples(682_666_667, 48000, 2) == 65536 == O.delay_samples(682_666_667, 48000, 2)
    assert rh.delay_samples(682_666_666, 48000, 2) == 65535 == O.delay_samples(682_666_666, 48000, 2)
    rng = np.random.default_rng(5)
    for _ in range(100):
        ns, rate, ch = int(rng.integers(0, 10**10)), int(rng.integers(1, 400000)), int(rng.integers(1, 9))
        assert rh.delay_samples(ns, rate, ch) == O.delay_samples(ns, rate, ch)


# ------------------------------------------------------------------ WAV container (host side) ----
def _wav_bytes(ch, rate, bits, payload, fmt_tag=1, extensible=False, junk=False):
    import struct

    bps = (bits + 7) // 8
    if extensible:
        sub = struct.pack("<H", fmt_tag) + bytes.fromhex("000000001000800000aa00389b71")
        fmt = struct.pack("<HHIIHHHHI", 0xFFFE, ch, rate, rate * ch * bps, ch * bps, bits, 22, bits, 0) + sub
    else:
        fmt = struct.pack("<HHIIHH", fmt_tag, ch, rate, rate * ch * bps, ch * bps, bits)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if junk:
        chunks += b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\0"  # odd-sized chunk + pad byte
    chunks += b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks


def test_wav_probe_walks_chunks(rh):
    import numpy as np

    pay = np.arange(12, dtype="<i2").tobytes()
    w = rh.wav_probe(_wav_bytes(2, 44100, 16, pay))
    assert (w["channels"], w["sample_rate"], w["bits_per_sample"], w["is_float"], w["samples"], w["data_offset"]) == (2, 44100, 16, 0, 12, 44)
    w = rh.wav_probe(_wav_bytes(6, 48000, 24, b"\0" * 36, extensible=True, junk=True))
    assert (w["channels"], w["bits_per_sample"], w["samples"], w["is_float"]) == (6, 24, 12, 0)
    w = rh.wav_probe(_wav_bytes(1, 8000, 32, b"\0" * 16, fmt_tag=3))
    assert w["is_float"] == 1 and w["samples"] == 4
    # truncated data chunk: what is there
    full = _wav_bytes(2, 44100, 16, pay)
    assert rh.wav_probe(full[:-4])["samples"] == 10
    import pytest

    with pytest.raises(rh.RhError):
        rh.wav_probe(b"RIFX" + full[4:])
    with pytest.raises(rh.RhError):
        rh.wav_probe(_wav_bytes(2, 44100, 16, pay, fmt_tag=0x55))  # MP3-in-WAV: not PCM


def test_wav_header_f32_matches_the_spec(rh):
    import ctypes as C
    import struct

    from rodio_amd import _lib

    hdr = (C.c_uint8 * 44)()
    assert _lib.lib.rh_wav_header_f32_host(hdr, 44, 2, 48000, 7) == 44  # 7 samples: 3 whole stereo frames
    b = bytes(hdr)
    assert b[:4] == b"RIFF" and b[8:16] == b"WAVEfmt " and b[36:40] == b"data"
    assert struct.unpack("<I", b[4:8])[0] == 36 + 24 and struct.unpack("<I", b[40:44])[0] == 24
    assert struct.unpack("<HHIIHH", b[20:36]) == (3, 2, 48000, 48000 * 8, 8, 32)
    assert _lib.lib.rh_wav_header_f32_host(hdr, 40, 2, 48000, 8) == 0


# ---------------------------------------------------- property tests (the reference uses quickcheck) ----
def test_out_frames_property_random_rates(rh, O):
    """sample_rate.rs:252-334 checks the converter with quickcheck; here the closed form of the C ABI
    (rh_resample_out_frames) must agree with the oracle's iterator for arbitrary rates and lengths."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=300, deadline=None, derandomize=True, database=None)
    @given(frm=st.integers(1, 200000), to=st.integers(1, 200000), n=st.integers(0, 600), ch=st.integers(1, 4),
           span=st.sampled_from([0, 0, 0, 12, 24, 120, 32768, 65536]))
    def prop(frm, to, n, ch, span):
        span_eff = span - span % ch if span else 0  # spans hold whole frames
        if span_eff and min(span_eff, 32768) % ch:  # uniform.rs:56 would cut a frame in two: rejected (RH_ERR_UNSUPPORTED)
            assert _out_frames(rh, n, frm, to, ch, span_eff)[0] == 3
            return
        import math

        gg = math.gcd(frm, to)
        if (frm // gg) * (to // gg) > 0xFFFFFFFF:  # the reference's u32 position arithmetic would overflow (sample_rate.rs:45-47): rejected
            assert _out_frames(rh, n, frm, to, ch, span_eff)[0] == 3
            return
        x = np.arange(n * ch, dtype=np.float32)
        if span_eff:
            ref = O.UniformSourceIterator(O.SpanSource(x, ch, frm, span_eff), ch, to).collect()
        else:
            ref = O.SampleRateConverter(O.TestSource(x, ch, frm), frm, to, ch).collect()
        stt, m = _out_frames(rh, n, frm, to, ch, span_eff)
        assert stt == 0 and m * ch == len(ref), (frm, to, n, ch, span_eff, m, len(ref))

    prop()


def test_resampler_pending_frames_property(rh):
    """rh_resampler_pending_frames (streaming) sums to the one-pass length for any block split; no GPU
    needed: the count is host arithmetic -- but the handle needs rh_init, so this only checks it refuses."""
    from rodio_amd import _lib

    h = C.c_void_p()
    assert _lib.lib.rh_resampler_create(C.byref(h), 44100, 48000, 2) in (0, 6)  # RH_OK with a GPU, NOT_INITIALIZED without


def test_db_helpers_match_the_reference_table(rh, O):
    """math.rs:238-339: the 27-row dB table (within 1 %) and the round trip (<= 16 eps); and bit-equality with the
    oracle's restatement of the same expressions."""
    from rodio_amd import _lib

    L = _lib.lib
    table = [(100.0, 100000.0), (90.0, 31623.0), (80.0, 10000.0), (70.0, 3162.3), (60.0, 1000.0), (50.0, 316.23), (40.0, 100.0), (30.0, 31.623),
             (20.0, 10.0), (10.0, 3.1623), (5.0, 1.7783), (1.0, 1.1220), (0.0, 1.0), (-1.0, 0.89125), (-5.0, 0.56234), (-10.0, 0.31623), (-20.0, 0.1),
             (-30.0, 0.031623), (-40.0, 0.01), (-50.0, 0.0031623), (-60.0, 0.001), (-70.0, 0.00031623), (-80.0, 0.0001), (-90.0, 0.000031623), (-100.0, 0.00001)]
    for db, lin in table:
        assert abs(L.rh_db_to_linear(db) - lin) <= 0.01 * lin
        assert abs(L.rh_linear_to_db(lin) - db) <= max(0.01 * abs(db), 0.01)
        assert L.rh_db_to_linear(db) == O.db_to_linear(db) and L.rh_linear_to_db(lin) == O.linear_to_db(lin)
    eps = np.finfo(np.float32).eps
    for db in np.linspace(-60, 20, 161):
        back = L.rh_linear_to_db(L.rh_db_to_linear(float(db)))
        assert abs(back - db) <= 16 * eps * max(abs(db), 1.0)
    assert L.rh_linear_to_db(0.0) == -np.inf and np.isnan(L.rh_linear_to_db(-1.0))
    for ns, rate in [(5_000_000, 48000), (100_000_000, 44100), (4_000_000_000, 48000), (1, 8000)]:
        assert L.rh_duration_to_coefficient(ns, rate) == O.duration_to_coefficient(ns, rate)

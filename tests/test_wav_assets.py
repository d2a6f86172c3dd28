"""(f)4 WAV ingest pinned by the files the reference's own test reads (tests/wav_test.rs:1-33: six files under assets/, each decoded and
asserted to hold a non-zero sample) and by a decoder that shares nothing with this repo or its oracle: scipy.io.wavfile.

tests/golden/wav/ holds those six files byte for byte (data, not code: what `rodio::Decoder::try_from(file)` is handed).  They are what
hand-made RIFF images are not: written by Audacity and LMMS, with `LIST` / `fact` / `PEAK` chunks in front of the data chunk, which
therefore starts at file offsets 44, 80, 94, 102 and 138 -- 102 is not a multiple of four although the samples are 32-bit.  hound reads a
byte stream (src/decoder/wav.rs:94-172); a caller of the C ABI that uploads the file and passes `file + data_offset` hands rh_wav_decode
that address, so the decode must not care (it refused until this test existed).

Expected values: scipy's integers through dasp_sample 0.11.0's `to_sample::<f32>()` (i16 / 32768, i32 / 2^31, the 24-bit sample -- which
scipy returns shifted into the top of an i32 -- / 2^23; f32 verbatim), in numpy.  Bit for bit.
"""
import io
import os
import warnings

import numpy as np
import pytest

WAVS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wav")
# file, channels, bits, float?, data offset (read off the files with a hex dump, not with the code under test)
FILES = [("audacity16bit.wav", 1, 16, 0, 44), ("lmms16bit.wav", 2, 16, 0, 94), ("lmms24bit.wav", 2, 24, 0, 94),
         ("audacity32bit.wav", 1, 32, 1, 80), ("lmms32bit.wav", 2, 32, 1, 138), ("audacity32bit_int.wav", 2, 32, 0, 102)]


def _scipy(name):
    from scipy.io import wavfile

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # "chunk not understood": the chunks this test is about
        rate, d = wavfile.read(os.path.join(WAVS, name))
    return rate, d


def _expected(d, bits):
    flat = d.reshape(-1)
    if d.dtype == np.float32:
        return flat.copy()
    if bits == 16:
        return flat.astype(np.float32) / np.float32(32768.0)
    if bits == 24:
        assert d.dtype == np.int32 and not np.any(flat & 0xFF)  # scipy: the three bytes in the top of an i32
        return (flat >> 8).astype(np.float32) / np.float32(8388608.0)
    return flat.astype(np.float32) / np.float32(2147483648.0)  # i32 -> f32 rounds to nearest, as Rust's `as f32`


@pytest.mark.parametrize("name,ch,bits,is_float,offset", FILES)
def test_probe_walks_the_reference_assets(rh, name, ch, bits, is_float, offset):
    b = open(os.path.join(WAVS, name), "rb").read()
    assert b[offset - 8: offset - 4] == b"data"
    w = rh.wav_probe(b)
    rate, d = _scipy(name)
    assert (w["channels"], w["sample_rate"], w["bits_per_sample"], w["is_float"], w["data_offset"]) == (ch, rate, bits, is_float, offset)
    assert w["samples"] == d.size and w["data_bytes"] == d.size * bits // 8
    assert (d.ndim == 1) == (ch == 1) and (ch == 1 or d.shape[1] == ch)


def test_probe_survives_damaged_files(rh):
    """The RIFF walk on the host (rh_wav_probe_host) over what real files look like after an accident: the six assets cut at random places and with
    random bytes of their chunk headers overwritten -- it refuses or it answers, and what it answers lies inside the bytes it was given (no read
    past the end: the image is handed over in a buffer of exactly its size)."""
    rng = np.random.default_rng(2024)
    n_ok = n_refused = 0
    for name, *_ in FILES:
        b = open(os.path.join(WAVS, name), "rb").read()
        head = min(len(b), 256)
        for _ in range(300):
            img = bytearray(b[: int(rng.integers(0, len(b) + 1))] if rng.random() < 0.5 else b)
            for _ in range(int(rng.integers(0, 4))):
                if len(img):
                    img[int(rng.integers(0, min(head, len(img))))] = int(rng.integers(0, 256))
            try:
                w = rh.wav_probe(bytes(img))
            except rh.RhError:
                n_refused += 1
                continue
            n_ok += 1
            assert w["data_offset"] + w["data_bytes"] <= len(img) and w["channels"] >= 1 and w["sample_rate"] >= 1
            assert w["samples"] * ((w["bits_per_sample"] + 7) // 8) <= w["data_bytes"]
    assert n_ok > 100 and n_refused > 100


@pytest.mark.gpu
@pytest.mark.parametrize("name,ch,bits,is_float,offset", FILES)
def test_decode_of_the_reference_assets(rh, name, ch, bits, is_float, offset):
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    rh.init(0)
    b = open(os.path.join(WAVS, name), "rb").read()
    rate, d = _scipy(name)
    want = _expected(d, bits)
    for image in (False, True):  # the data chunk in a buffer of its own / where it lies in the uploaded file (offset 102: two bytes off)
        src = rh.WavDecoder(b, image=image)
        assert (src.channels(), src.sample_rate()) == (ch, rate)
        got = src.collect()
        assert np.any(got != 0.0)  # tests/wav_test.rs: `decoder.any(|x| x != 0.0)`
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), (name, image)
        # the one-launch form (decode + ChannelCountConverter), down to mono and up to 5.1 (channels.rs:59-70)
        for to in (1, 6):
            one = rh.WavDecoderChannels(b, to, image=image).collect().reshape(-1, to)
            fr = want.reshape(-1, ch)
            exp = np.zeros((len(fr), to), np.float32)
            k = min(ch, to)
            exp[:, :k] = fr[:, :k]
            if ch == 1 and to >= 2:
                exp[:, 1] = fr[:, 0]
            assert np.array_equal(one.view(np.uint32), exp.view(np.uint32)), (name, image, to)


@pytest.mark.gpu
def test_egress_read_back_by_scipy(rh):
    """src/wav_output.rs:62-96: what wav_to_writer leaves is a 32-bit float WAVE a third-party reader agrees with."""
    from scipy.io import wavfile

    rh.init(0)
    x = (np.random.default_rng(77).uniform(-1, 1, 2 * 4001)).astype(np.float32)
    out = rh.wav_to_bytes(rh.TestSource(x[:-1], 2, 48000))  # 8 001 samples: the half frame is dropped (wav_output.rs:98-140)
    rate, d = wavfile.read(io.BytesIO(out))
    assert rate == 48000 and d.dtype == np.float32 and d.shape == (4000, 2) and np.array_equal(d.reshape(-1), x[:8000])


@pytest.mark.gpu
def test_wav_to_file_as_the_reference_tests_it(rh):
    """src/wav_output.rs:144-181 `test_wav_to_file`: SineWave::new(745.0).amplify(0.1).take_duration(1 s) written by wav_to_file and read back
    (the reference reads with hound; here with scipy, and with this library's own decoder): the source's rate and channel count, as many
    samples as the source yields, the same samples.  SineWave is 48 kHz mono, sample n = sin(TAU * phase_n), phase stepping by f / rate
    (signal_generator.rs:51-53,130-135; sine.rs:51-58) -- built on the host here: the test is about the file, not about the sine."""
    from oracle import rodio_oracle as O
    from scipy.io import wavfile

    rh.init(0)
    rate, f = 48000, np.float32(745.0)
    phase, step, xs = np.float32(0.0), np.float32(f / np.float32(rate)), []
    for _ in range(rate + 1000):  # (a little more than the second take_duration keeps)
        xs.append(np.sin(np.float32(6.2831855) * phase, dtype=np.float32))
        phase = np.float32((phase + step) % np.float32(1.0))
    x = np.array(xs, np.float32)
    make = lambda M: M.TestSource(x, 1, rate).amplify(0.1).take_duration(1_000_000_000)  # noqa: E731
    out = rh.wav_to_bytes(make(rh))
    expected = make(O).collect()
    assert len(expected) == rate  # one second of a mono source: 1e9 / (1e9 / 48000) samples (take.rs:63-67)
    r, d = wavfile.read(io.BytesIO(out))
    assert r == rate and d.ndim == 1 and d.dtype == np.float32          # reference.sample_rate(), reference.channels(), f32 samples
    assert len(d) == len(expected) and np.array_equal(d.view(np.uint32), expected.view(np.uint32))
    back = rh.WavDecoder(out)
    assert (back.channels(), back.sample_rate()) == (1, rate) and np.array_equal(back.collect().view(np.uint32), expected.view(np.uint32))

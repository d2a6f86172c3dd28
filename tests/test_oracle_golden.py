"""Pins the CPU oracle against every golden vector the reference's own tests hold for the hot
path.  Each test cites the reference test it restates
(paths under /root/reference)."""
import numpy as np
import pytest


# ---------------------------------------------------------------- resampler ----
def test_resampler_upsample(O):
    # src/conversions/sample_rate.rs:356-367
    inp = [2.0, 16.0, 4.0, 18.0, 6.0, 20.0, 8.0, 22.0]
    out = O.SampleRateConverter(O.TestSource(inp, 2, 2000), 2000, 3000, 2).collect()
    assert len(out) == 12
    assert np.trunc(out).tolist() == [2.0, 16.0, 3.0, 17.0, 4.0, 18.0, 6.0, 20.0, 7.0, 21.0, 8.0, 22.0]


def test_resampler_upsample2(O):
    # src/conversions/sample_rate.rs:369-377
    out = O.SampleRateConverter(O.TestSource([1.0, 14.0], 1, 1000), 1000, 7000, 1).collect()
    assert np.trunc(out).tolist() == [1.0, 2.0, 4.0, 6.0, 8.0, 10.0, 12.0, 14.0]


def test_resampler_downsample(O):
    # src/conversions/sample_rate.rs:379-387
    inp = np.arange(17, dtype=np.float32)
    out = O.SampleRateConverter(O.TestSource(inp, 1, 12000), 12000, 2400, 1).collect()
    assert out.tolist() == [0.0, 5.0, 10.0, 15.0]


@pytest.mark.parametrize("frm,to,ch", [(44100, 48000, 2), (1, 7, 3), (96000, 8000, 1), (384000, 11025, 5)])
def test_resampler_empty(O, frm, to, ch):
    # quickcheck `empty`, src/conversions/sample_rate.rs:254-267
    assert len(O.SampleRateConverter(O.TestSource([], ch, frm), frm, to, ch).collect()) == 0


@pytest.mark.parametrize("seed", range(4))
def test_resampler_identity(O, seed):
    # quickcheck `identity`, src/conversions/sample_rate.rs:270-279
    rng = np.random.default_rng(seed)
    ch = int(rng.integers(1, 9))
    inp = rng.integers(-32768, 32767, size=int(rng.integers(0, 300))).astype(np.float32)
    rate = int(rng.integers(1, 200000))
    out = O.SampleRateConverter(O.TestSource(inp, ch, rate), rate, rate, ch).collect()
    assert np.array_equal(out, inp)


@pytest.mark.parametrize("seed", range(8))
def test_resampler_divide_sample_rate(O, seed):
    # quickcheck `divide_sample_rate`, src/conversions/sample_rate.rs:283-306
    rng = np.random.default_rng(100 + seed)
    ch = int(rng.integers(1, 6))
    k = int(rng.integers(1, 12))
    to = int(rng.integers(1, 48001))
    n = int(rng.integers(0, 400))
    inp = rng.integers(-32768, 32767, size=n).astype(np.float32)
    inp = inp[: ch * (len(inp) // ch)]
    out = O.SampleRateConverter(O.TestSource(inp, ch, to * k), to * k, to, ch).collect()
    expect = inp.reshape(-1, ch)[::k].reshape(-1)
    assert np.array_equal(out, expect)


@pytest.mark.parametrize("seed", range(8))
def test_resampler_multiply_sample_rate(O, seed):
    # quickcheck `multiply_sample_rate`, src/conversions/sample_rate.rs:310-333
    rng = np.random.default_rng(200 + seed)
    ch = int(rng.integers(1, 6))
    k = int(rng.integers(1, 12))
    frm = int(rng.integers(1, 65536))
    n = int(rng.integers(0, 300))
    inp = rng.integers(-32768, 32767, size=n).astype(np.float32)
    inp = inp[: ch * (len(inp) // ch)]
    out = O.SampleRateConverter(O.TestSource(inp, ch, frm), frm, frm * k, ch).collect()
    assert np.array_equal(out.reshape(-1, ch)[::k].reshape(-1), inp)


def test_lerp_random(O):
    # quickcheck `lerp_random`, src/math.rs:187-216
    rng = np.random.default_rng(7)
    from math import gcd

    checked = 0
    while checked < 2000:
        a, b = (np.float32(x) for x in rng.uniform(-1, 1, 2))
        num, den = int(rng.integers(0, 5000)), int(rng.integers(1, 5000))
        g = gcd(num, den)
        num, den = num // g, den // g
        c = num / den
        if num > 1000 or not (0.0 <= c <= 1.0):
            continue
        ref = float(a) * (1.0 - c) + float(b) * c
        assert abs(float(O.lerp(a, b, num, den)) - ref) < 1e-6
        checked += 1


# ----------------------------------------------------------------- channels ----
def test_channels_remove(O):
    # src/conversions/channels.rs:114-125
    out = O.ChannelCountConverter(O.TestSource([1, 2, 3, 4, 5, 6], 3, 1), 3, 2).collect()
    assert out.tolist() == [1.0, 2.0, 4.0, 5.0]
    out = O.ChannelCountConverter(O.TestSource([1, 2, 3, 4, 5, 6, 7, 8], 4, 1), 4, 1).collect()
    assert out.tolist() == [1.0, 5.0]


def test_channels_add(O):
    # src/conversions/channels.rs:127-143
    out = O.ChannelCountConverter(O.TestSource([1, 2, 3, 4], 1, 1), 1, 2).collect()
    assert out.tolist() == [1, 1, 2, 2, 3, 3, 4, 4]
    out = O.ChannelCountConverter(O.TestSource([1, 2], 1, 1), 1, 4).collect()
    assert out.tolist() == [1, 1, 0, 0, 2, 2, 0, 0]
    out = O.ChannelCountConverter(O.TestSource([1, 2, 3, 4], 2, 1), 2, 4).collect()
    assert out.tolist() == [1, 2, 0, 0, 3, 4, 0, 0]


def test_channels_len(O):
    # len_more / len_less, src/conversions/channels.rs:164-177 (and the size_hint cases :146-162,
    # restated as produced length)
    assert len(O.ChannelCountConverter(O.TestSource([1, 2, 3, 4], 2, 1), 2, 3).collect()) == 6
    assert len(O.ChannelCountConverter(O.TestSource([1, 2, 3, 4], 2, 1), 2, 1).collect()) == 2
    assert len(O.ChannelCountConverter(O.TestSource([1, 2, 3], 1, 1), 1, 2).collect()) == 6
    assert len(O.ChannelCountConverter(O.TestSource([1, 2, 3, 4, 5, 6], 3, 1), 3, 8).collect()) == 16
    assert len(O.ChannelCountConverter(O.TestSource(range(1, 9), 4, 1), 4, 1).collect()) == 2


# -------------------------------------------------------------------- mixer ----
def test_mixer_basic(O):
    # src/mixer.rs:208-230
    m = O.Mixer(1, 48000)
    m.add(O.SamplesBuffer(1, 48000, [10.0, -10.0, 10.0, -10.0]))
    m.add(O.SamplesBuffer(1, 48000, [5.0, 5.0, 5.0, 5.0]))
    assert [m.next() for _ in range(5)] == [15.0, -5.0, 15.0, -5.0, None]


def test_mixer_channels_conv(O):
    # src/mixer.rs:232-258
    m = O.Mixer(2, 48000)
    m.add(O.SamplesBuffer(1, 48000, [10.0, -10.0, 10.0, -10.0]))
    m.add(O.SamplesBuffer(1, 48000, [5.0, 5.0, 5.0, 5.0]))
    assert [m.next() for _ in range(9)] == [15.0, 15.0, -5.0, -5.0, 15.0, 15.0, -5.0, -5.0, None]


def test_mixer_rate_conv(O):
    # src/mixer.rs:260-285
    m = O.Mixer(1, 96000)
    m.add(O.SamplesBuffer(1, 48000, [10.0, -10.0, 10.0, -10.0]))
    m.add(O.SamplesBuffer(1, 48000, [5.0, 5.0, 5.0, 5.0]))
    assert [m.next() for _ in range(8)] == [15.0, 5.0, -5.0, 5.0, 15.0, 5.0, -5.0, None]


def test_mixer_start_afterwards(O):
    # src/mixer.rs:287-318
    m = O.Mixer(1, 48000)
    m.add(O.SamplesBuffer(1, 48000, [10.0, -10.0, 10.0, -10.0]))
    assert m.next() == 10.0
    assert m.next() == -10.0
    m.add(O.SamplesBuffer(1, 48000, [5.0, 5.0, 6.0, 6.0, 7.0, 7.0, 7.0]))
    assert m.next() == 15.0
    assert m.next() == -5.0
    assert m.next() == 6.0
    assert m.next() == 6.0
    m.add(O.SamplesBuffer(1, 48000, [2.0]))
    assert m.next() == 9.0
    assert m.next() == 7.0
    assert m.next() == 7.0
    assert m.next() is None


def test_mixer_added_taking_phase_into_account(O):
    # src/mixer.rs:320-341
    m = O.Mixer(2, 48000)
    m.add(O.SamplesBuffer(2, 48000, [10.0, -10.0, 10.0, -10.0]))
    assert m.next() == 10.0
    m.add(O.SamplesBuffer(2, 48000, [5.0, -5.0, 6.0, -6.0]))
    assert m.next() == -10.0  # not yet mixed (out of phase)
    assert m.next() == 15.0  # mixing starts


# ------------------------------------------------------------ channel volume ----
def test_channel_volume_mono_to_stereo(O):
    # src/source/channel_volume.rs:135-146
    out = O.ChannelVolume(O.TestSource([1.0, 2.0, 3.0], 1, 44100), [0.5, 0.8]).collect()
    f = np.float32
    assert out.tolist() == [f(1) * f(0.5), f(1) * f(0.8), f(2) * f(0.5), f(2) * f(0.8), f(3) * f(0.5), f(3) * f(0.8)]


def test_channel_volume_stereo_to_mono(O):
    # src/source/channel_volume.rs:148-155
    out = O.ChannelVolume(O.TestSource([1.0, 2.0, 3.0, 4.0], 2, 44100), [1.0]).collect()
    assert out.tolist() == [1.5, 3.5]


def test_channel_volume_stereo_to_stereo_with_mixing(O):
    # src/source/channel_volume.rs:157-166
    out = O.ChannelVolume(O.TestSource([1.0, 3.0, 2.0, 4.0], 2, 44100), [0.5, 2.0]).collect()
    assert out.tolist() == [1.0, 4.0, 1.5, 6.0]


def test_channel_volume_six_channels_from_stereo(O):
    # tests/channel_volume.rs:9-62: a stereo source through ChannelVolume [1,1,0,0,0,0] has output on the first two
    # of six channels only, and ends on a frame boundary (the reference decodes assets/music.mp3; any stereo stream does)
    x = (np.random.default_rng(11).uniform(-1, 1, 2 * 4001)).astype(np.float32)
    src = O.ChannelVolume(O.TestSource(x, 2, 44100), [1.0, 1.0, 0.0, 0.0, 0.0, 0.0])
    assert src.channels() == 6
    out = src.collect()
    assert len(out) % 6 == 0 and len(out) == 6 * 4001
    fr = out.reshape(-1, 6)
    assert np.all(fr[:, 2:] == 0.0) and np.any(fr[:, :2] != 0.0)
    assert np.array_equal(fr[:, 0], fr[:, 1])  # both carry the frame's mono mix (channel_volume.rs:71-88)


# ------------------------------------------------------------------ dB table ----
DECIBELS_LINEAR_TABLE = [
    (100.0, 100000.0), (90.0, 31623.0), (80.0, 10000.0), (70.0, 3162.0), (60.0, 1000.0), (50.0, 316.2),
    (40.0, 100.0), (30.0, 31.62), (20.0, 10.0), (10.0, 3.162), (5.998, 1.995), (3.003, 1.413),
    (1.002, 1.122), (0.0, 1.0), (-1.002, 0.891), (-3.003, 0.708), (-5.998, 0.501), (-10.0, 0.3162),
    (-20.0, 0.1), (-30.0, 0.03162), (-40.0, 0.01), (-50.0, 0.003162), (-60.0, 0.001),
    (-70.0, 0.0003162), (-80.0, 0.0001), (-90.0, 0.00003162), (-100.0, 0.00001),
]  # src/math.rs:238-266


def test_db_table(O):
    # src/math.rs:268-303
    for db, lin in DECIBELS_LINEAR_TABLE:
        assert 0.99 < O.db_to_linear(db) / lin < 1.01
        if abs(db) > 1e-5:
            assert 0.99 < O.linear_to_db(lin) / db < 1.01


def test_db_round_trip(O):
    # src/math.rs:305-339
    eps = float(np.finfo(np.float32).eps)
    for db in [-60.0, -20.0, -6.0, 0.0, 6.0, 20.0, 40.0]:
        assert abs(O.linear_to_db(O.db_to_linear(db)) - db) < 16 * eps
    for lin in [0.001, 0.1, 1.0, 10.0, 100.0]:
        lin32 = float(np.float32(lin))
        assert abs((O.db_to_linear(O.linear_to_db(lin)) - lin32) / lin32) < 16 * eps


# ------------------------------------------------------------------- limiter ----
def test_limiter_behaviour(O):
    # tests/limit.rs:7-155 range assertions: settled peak within +-0.1 of threshold, passthrough
    # below threshold.  (The reference pins ranges only, not values.)
    sr = 48000
    t = np.arange(sr // 2, dtype=np.float64) / sr
    sine = (np.sin(2 * np.pi * 440.0 * t) * 2.0).astype(np.float32)  # SineWave(440).amplify(2.0)
    for thr_db, expect in [(-1.0, 0.89), (-3.0, 0.71), (-6.0, 0.50)]:
        out = O.TestSource(sine, 1, sr).limit(threshold=thr_db, knee_width=4.0).collect()
        peak = float(np.max(np.abs(out[-4800:])))
        assert abs(peak - expect) < 0.1, (thr_db, peak)
    quiet = (np.sin(2 * np.pi * 440.0 * t) * 0.2).astype(np.float32)
    out = O.TestSource(quiet, 1, sr).limit().collect()
    assert float(np.max(np.abs(out - quiet))) < 0.01


# ------------------------------------------------- closed form of the resampler ----
def closed_form_resample(x, frm, to, ch):
    """SURVEY.md Appendix A.1 closed form -- an independent restatement used to cross-check the
    iterator state machine (and later the kernels' index math)."""
    from math import gcd

    g = gcd(frm, to)
    F, T = frm // g, to // g
    x = np.asarray(x, np.float32).reshape(-1, ch)
    if F == T:
        return x.reshape(-1).copy()
    N = len(x)
    out = []
    m = 0
    while True:
        i, num = (m * F) // T, (m * F) % T
        if i <= N - 2:
            a, b = x[i], x[i + 1]
            out.append(a + (b - a) * np.float32(num) / np.float32(T))
        elif i == N - 1:
            out.append(x[i].copy())
            break
        else:
            break
        m += 1
    return np.concatenate(out).astype(np.float32) if out else np.empty(0, np.float32)


@pytest.mark.parametrize("frm,to,ch,n", [(44100, 48000, 2, 1000), (48000, 44100, 2, 777), (8000, 48000, 1, 50),
                                         (44100, 40000, 2, 333), (11025, 48000, 6, 100), (48000, 8000, 3, 401),
                                         (44100, 48000, 2, 1), (44100, 48000, 2, 2), (48000, 44100, 1, 1)])
def test_resampler_matches_closed_form(O, frm, to, ch, n):
    rng = np.random.default_rng(n)
    x = rng.uniform(-1, 1, n * ch).astype(np.float32)
    out = O.SampleRateConverter(O.TestSource(x, ch, frm), frm, to, ch).collect()
    ref = closed_form_resample(x, frm, to, ch)
    assert np.array_equal(out, ref)


# ------------------------------------------ BASELINE config 5 fixtures (tests/golden) ----
def _golden(name):
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))


def test_golden_music_excerpt_conversions(O):
    # tests/golden/make_golden.py: 32 768 samples of the reference's assets/music.wav, expected arrays
    # computed from the cited formulas with numpy (sample.rs:42-44 -> dasp i16->f32; channels.rs:57-85)
    i16 = _golden("music_excerpt_i16.npy")
    f32 = _golden("music_excerpt_f32.npy")
    assert np.array_equal(O.convert("i16_to_f32", i16), f32)
    six = f32[: (len(f32) // 6) * 6]
    out = O.ChannelCountConverter(O.TestSource(six, 6, 44100), 6, 2).collect()
    assert np.array_equal(out, _golden("music_excerpt_6to2.npy"))


# ------------------------------------------------ LinearGainRamp ----
def test_linear_ramp_golden(O):
    # src/source/linear_ramp.rs:176-192 and :194-210: 10 samples of 1.0, 1 channel, 1 Hz, 4 s ramp
    ones = np.ones(10, np.float32)
    out = O.TestSource(ones, 1, 1).linear_gain_ramp(4_000_000_000, 0.0, 1.0, True).collect()
    assert out.tolist() == [0.0, 0.25, 0.5, 0.75, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]
    out = O.TestSource(ones, 1, 1).linear_gain_ramp(4_000_000_000, 0.0, 0.5, True).collect()
    assert out.tolist() == [0.0, 0.125, 0.25, 0.375, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5]


def test_linear_ramp_seek_values(O):
    # src/source/linear_ramp.rs:213-222 (the part before the first seek): cycle [0, .4, .8], 10 s ramp
    x = np.float32([0.0, 0.4, 0.8] * 7)[:20]
    out = O.TestSource(x, 1, 1).linear_gain_ramp(10_000_000_000, 0.0, 1.0, True).collect()
    assert np.allclose(out[:3], [0.0, 0.04, 0.16], atol=1e-6)
    assert np.allclose(out[10:13], x[10:13], atol=1e-6)  # ramp finished: gain 1.0


# ------------------------------------------------------ TakeDuration ----
def test_take_duration_golden(O):
    # src/source/take.rs:247-261: a duration of exactly 10 sample periods (mono, 48 kHz) yields 10 samples
    nps = int(np.float32(1_000_000_000) / np.float32(48000))
    x = np.ones(100, np.float32)
    assert len(O.TestSource(x, 1, 48000).take_duration(nps * 10).collect()) == 10
    # :263-280: stereo 44.1 kHz, 5 sample periods -> the cut frame is completed with silence
    nps = 1_000_000_000 // (44100 * 2)
    out = O.TestSource(np.ones(10, np.float32), 2, 44100).take_duration(nps * 5).collect()
    assert out.tolist() == [1.0, 1.0, 1.0, 1.0, 1.0, 0.0]
    # :241-245: zero duration -> nothing
    assert len(O.TestSource(x, 1, 48000).take_duration(0).collect()) == 0


def test_take_duration_and_delay_answer_current_span_len_as_the_reference_does(O):
    """take.rs:176-195 and delay.rs:94-98, by hand (the reference has no test for them): TakeDuration answers what its duration still admits
    unless the input's span is shorter -- over an input that says None too -- and Some(0) once it is spent (the silence that completes a cut
    frame lies behind that); Delay adds the silence it still owes to its input's answer, and hands None on.  A UniformSourceIterator
    behind a take therefore stops in front of that silence."""
    nps = 1_000_000_000 // (44100 * 2)
    t = O.TestSource(np.ones(100, np.float32), 2, 44100).take_duration(nps * 5)  # admits 5 samples: cuts the third frame
    answers = []
    while True:
        answers.append(t.current_span_len())
        if not len(t.pull(1)):
            break
    assert answers == [5, 4, 3, 2, 1, 0, 0]  # five samples, the sixth is the frame's silence: Some(0) in front of it, and behind
    assert O.SamplesBuffer(2, 44100, np.ones(4, np.float32)).take_duration(nps * 10).current_span_len() == 4  # the input's span is shorter (:192-194)
    assert O.SamplesBuffer(2, 44100, np.ones(40, np.float32)).take_duration(nps * 10).current_span_len() == 10
    assert O.TestSource(np.ones(8, np.float32), 2, 44100).take_duration(0).current_span_len() == 0  # remaining_nanos == 0 (:185-187)
    d = O.TestSource(np.ones(8, np.float32), 2, 44100).delay(1_000_000_000)
    assert d.current_span_len() is None
    d = O.SamplesBuffer(1, 1000, np.ones(8, np.float32)).delay(3_000_000)  # three samples of silence
    answers = []
    for _ in range(5):
        answers.append(d.current_span_len())
        d.pull(1)
    assert answers == [11, 10, 9, 8, 8]
    u = O.UniformSourceIterator(O.TestSource(np.ones(100, np.float32), 2, 44100).take_duration(nps * 5), 2, 44100)
    assert u.collect().tolist() == [1.0] * 5  # (the bare adapter returns six samples: test_take_duration_golden)


def test_sine_generator_restatement_matches_the_reference_vector():
    # signal_generator.rs:227-238 (TEST_EPSILON = 1e-6): the input of BASELINE config 1
    from conftest import sine_generator

    w = sine_generator(1000, 100.0, 7)
    for got, want in zip(w, [0.0, 0.58778525, 0.95105652, 0.95105652, 0.58778525, 0.0, -0.58778554]):
        assert abs(float(got) - want) <= 2e-6


def test_config1_plumbing_on_the_oracle(O):
    # BASELINE configs[0] (CPU plumbing): SineWave at 44.1 kHz -> SampleRateConverter to 48 kHz -> amplify(0.8)
    from conftest import sine_generator

    x = sine_generator(44100, 440.0, 44100)
    out = O.SampleRateConverter(O.TestSource(x, 1, 44100), 44100, 48000, 1).amplify(0.8).collect()
    assert len(out) == 48000 and np.max(np.abs(out)) <= 0.8 + 1e-6
    # linear interpolation of a 440 Hz sine sampled at 44.1 kHz stays within the chord error of the true 48 kHz sine
    t = np.arange(48000) / 48000.0
    assert np.max(np.abs(out - 0.8 * np.sin(2 * np.pi * 440.0 * t))) < 5e-3  # chord error + the f32 phase accumulator's drift over 1 s


# ------------------------------------------------------------------ crossfade = take(+fadeout) + fade_in + mix ----
def _crossfade(M, fadeout_src, fadein_src, ns):
    """source/crossfade.rs:10-23: input_fadeout.take_duration(d) with the fade-out filter, mixed with
    input_fadein.take_duration(d).fade_in(d)  (Mix == a two-source mixer, mix.rs:43-53)."""
    m = M.Mixer(1, 1)
    m.add(fadeout_src.take_duration(ns, True))
    m.add(fadein_src.take_duration(ns).fade_in(ns))
    return m.collect()


def test_crossfade_with_self(O):
    # crossfade.rs:45-64
    d = np.arange(1, 11, dtype=np.float32)
    out = _crossfade(O, O.TestSource(d, 1, 1), O.TestSource(d, 1, 1), 5_000_000_001)
    assert len(out) == 5 and np.all(np.abs(out - np.array([1, 2, 3, 4, 5], np.float32)) < 1e-6)


def test_crossfade_with_silence(O):
    # crossfade.rs:66-81
    d = np.arange(1, 11, dtype=np.float32)
    out = _crossfade(O, O.TestSource(d, 1, 1), O.TestSource(np.zeros(10, np.float32), 1, 1), 5_000_000_001)
    assert len(out) == 5 and np.all(np.abs(out - np.array([1.0, 2.0 * 0.8, 3.0 * 0.6, 4.0 * 0.4, 5.0 * 0.2], np.float32)) < 1e-6)


# ------------------------------------------------------------------ dither ----
def test_dither_reference_test_properties(O):
    """The reference's noise is entropy-seeded, so its tests pin properties, not samples (dither.rs:301-393):
    the dither stays within 2 LSB of the input (test_dither_adds_noise), HighPass noise has negative lag-1
    autocorrelation per channel and independent channels (test_highpass_dither_multichannel_independence).
    The counter-based restatement must have them, plus the distributions' moments (noise.rs:156,216,394)."""
    sr, bits = 44100, 16
    lsb = np.float32(1.0 / (1 << (bits - 1)))
    t = np.arange(441) / sr
    x = np.sin(2 * np.pi * 440 * t).astype(np.float32)
    for seed in (0, 1, 12345):
        y = O.TestSource(x, 1, sr).dither(bits, "TPDF", seed).collect()
        assert np.all(np.isfinite(y)) and np.max(np.abs(y - x)) <= 2 * lsb and np.any(y != x)
    z = np.zeros(2 * 200000, dtype=np.float32)
    for seed in (7, 8):
        hp = O.TestSource(z, 2, sr).dither(bits, "HighPass", seed).collect() / -lsb
        left, right = hp[0::2].astype(np.float64), hp[1::2].astype(np.float64)
        assert np.mean(left[:-1] * left[1:]) < 0 and np.mean(right[:-1] * right[1:]) < 0
        assert abs(np.mean(left * right)) < 0.01
        # lag-1 autocorrelation of white[k] - white[k-1] is -1/2 of its variance (2/3)
        assert abs(np.mean(left[:-1] * left[1:]) + 1.0 / 3.0) < 0.01
    n = 400000
    z = np.zeros(n, dtype=np.float32)
    # Triangular(-1, 1, mode 0) (noise.rs:206) has variance 1/6: sigma 0.408 LSB, the "optimal 0.408 LSB" of
    # noise.rs:387 (the 2/sqrt(6) that WhiteTriangular::std_dev() reports at :216-218 is the figure for (-2, 2))
    for alg, std, bound in (("TPDF", 1 / np.sqrt(6), 1.0), ("RPDF", np.sqrt(1 / 3), 1.0), ("GPDF", 0.6, 6 * 0.6)):
        w = (O.TestSource(z, 1, sr).dither(bits, alg, 99).collect() / -lsb).astype(np.float64)
        assert abs(np.mean(w)) < 0.005 and abs(np.std(w) - std) < 0.005 and np.max(np.abs(w)) <= bound
        assert abs(np.mean(w[:-1] * w[1:])) < 0.005  # white
    # different seeds give different noise; the same seed the same
    a = O.TestSource(z[:1000], 1, sr).dither(bits, "TPDF", 1).collect()
    b = O.TestSource(z[:1000], 1, sr).dither(bits, "TPDF", 2).collect()
    c = O.TestSource(z[:1000], 1, sr).dither(bits, "TPDF", 1).collect()
    assert np.array_equal(a, c) and not np.array_equal(a, b)
    # lsb for the depths rodio names (dither.rs:180)
    for bits2 in (8, 16, 24, 32):
        y = O.TestSource(z[:4096], 1, sr).dither(bits2, "RPDF", 5).collect()
        assert 0 < np.max(np.abs(y)) <= 1.0 / (1 << (bits2 - 1))


# ------------------------------------------------------------------ total_duration() and size_hint() (the rest of the trait) ----
def test_total_duration_and_size_hint_derived_by_hand(O):
    """The reference has one numeric test here (channels.rs:146-161, below); every other expectation is DERIVED BY HAND from the cited lines --
    none of these numbers was produced by running the oracle."""
    z = np.zeros(9600, np.float32)  # 4800 stereo frames at 48 kHz = 100 ms (buffer.rs:45-51: 1e9 * 9600 / 48000 / 2)
    buf = lambda: O.SamplesBuffer(2, 48000, z)  # noqa: E731
    # buffer.rs:134-137 counts the remaining samples; :95-97 the duration
    s = buf()
    assert s.size_hint() == (9600, 9600) and s.total_duration() == 100_000_000
    s.pull(3)
    assert s.size_hint() == (9597, 9597) and s.total_duration() == 100_000_000
    # benches/shared.rs:14-21: TestSource implements only next() -> the trait's default (0, None); :47-49: it is GIVEN its duration
    t = O.TestSource(z, 2, 48000, total_duration=5)
    assert t.size_hint() == (0, None) and t.total_duration() == 5 and O.TestSource(z, 2, 48000).total_duration() is None
    # the adapters that hand both on: amplify.rs:68-70,95-97, blt.rs:144-146,171-173, limit.rs:705-707,592-594, agc.rs:561-563,588-590,
    # distortion.rs:75-77,102-104, linear_ramp.rs:109-111,136-138, channel_volume.rs:91-93,119-121 (the INPUT's count, whatever the layouts)
    for wrap in (lambda s: s.amplify(0.5), lambda s: s.low_pass(200), lambda s: s.high_pass(300), lambda s: s.limit(), lambda s: s.automatic_gain_control(),
                 lambda s: s.distortion(2.0, 0.5), lambda s: s.fade_in(10_000_000), lambda s: O.ChannelVolume(s, [1.0, 0.5, 0.25])):
        a = wrap(buf())
        assert a.size_hint() == (9600, 9600) and a.total_duration() == 100_000_000
        a.pull(1)
    assert O.ChannelVolume(buf(), [1.0, 0.5, 0.25]).pull(1) is not None
    cv = O.ChannelVolume(buf(), [1.0, 0.5, 0.25])
    cv.pull(1)  # channel_volume.rs:71-79: the first output sample takes a whole input frame
    assert cv.size_hint() == (9598, 9598)
    # delay.rs:8-16: 10 000 001 ns * 2 * 48000 / 1e9 = 960.000096 -> 960 samples of silence; :78-84 adds what is still owed; :111-115 adds the
    # REQUESTED duration (not the rounded silence)
    d = buf().delay(10_000_001)
    assert d.size_hint() == (10560, 10560) and d.total_duration() == 110_000_001
    d.pull(100)
    assert d.size_hint() == (10460, 10460)
    d.pull(900)  # 1000 taken: the silence is through, 40 samples of the buffer gone
    assert d.size_hint() == (9560, 9560)
    assert O.TestSource(z, 2, 48000).delay(10_000_001).size_hint() == (960, None)
    # take.rs:63-67: duration_per_sample = 1e9 / (48000 * 2) = 10416 ns; :151-171: 50 ms / 10416 = 4800.3 -> 4800 samples, cut to the input's
    # bounds, an upper bound even over an input without one; :209-219: the shorter duration
    tk = buf().take_duration(50_000_000)
    assert tk.size_hint() == (4800, 4800) and tk.total_duration() == 50_000_000
    tk.pull(10)  # remaining = 50 000 000 - 10 * 10416 = 49 895 840 -> 4790 samples
    assert tk.size_hint() == (4790, 4790)
    assert O.TestSource(z, 2, 48000).take_duration(50_000_000).size_hint() == (0, 4800)
    assert buf().take_duration(500_000_000).total_duration() == 100_000_000 and buf().take_duration(500_000_000).size_hint() == (9600, 9600)
    assert O.TestSource(z, 2, 48000).take_duration(50_000_000).total_duration() is None
    whole = buf().take_duration(50_000_000)
    assert len(whole.collect()) == 4800 and whole.size_hint() == (0, 0)  # remaining 3200 ns < 10416: 0 samples
    # uniform.rs:37,131-133: the duration the input gave when the iterator was built; :100-108: (lower, None) -- of the pending input while no
    # chain exists (:105), of ChannelCountConverter(SampleRateConverter(Take(..))) afterwards
    u = O.UniformSourceIterator(O.SamplesBuffer(2, 44100, np.zeros(2000, np.float32)), 2, 48000)
    assert u.size_hint() == (2000, None) and u.total_duration() == 1_000_000_000 * 2000 // 44100 // 2
    u.pull(1)
    # Take admits min(2000, 32768) = 2000; SampleRateConverter::new took two frames (sample_rate.rs:58-71): 1996 left in Take and in the buffer ->
    # Take (uniform.rs:181-196): (1996, Some(1996)).  sample_rate.rs:204-238 with from = 147, to = 160, pos_in_chunk 0, out_pos 1, one sample in
    # the output buffer: 1996 - (147 - 2) * 2 = 1706; 1706 * 160 / 147 = 1856; + (160 - 1) * 2 + 1 = 2175.  channels.rs:88-102 with from = to = 2,
    # position 1: (2175 + 1) / 2 * 2 - 1 = 2175.
    assert u.size_hint() == (2175, None)
    # mixer.rs:139-166: nothing plays before the first call admits the sources; then the longest lower bound, and None since one source's is
    m = O.Mixer(2, 48000)
    assert m.rx.size_hint() == (0, 0) and m.rx.total_duration() is None  # :104-106
    m.add(O.SamplesBuffer(2, 44100, np.zeros(2000, np.float32)))
    m.add(O.SamplesBuffer(2, 48000, np.zeros(100, np.float32)))
    assert m.rx.size_hint() == (0, 0)  # (still pending)
    m.next()
    # the second source at the mixer's own format: Take 100 -> after one sample 99; the converter passes through (sample_rate.rs:232-233);
    # channels.rs:88-102: (99 + 1) / 2 * 2 - 1 = 99.  The first: 2175 as above.
    assert m.rx.size_hint() == (2175, None)
    # mix.rs:56-67,104-112 over buffered.rs:16,192-195 and delay.rs:78-84,111-115: reverb = Mix(Buffered, Delay(Amplify(Buffered))):
    # total = max(f1, f1 + d); lower = max(0, the silence the echo owes) before the first sample (uniform.rs:105 on both branches)
    r = buf().reverb(10_000_000, 0.5)  # delay.rs:8-16: 960 samples
    assert r.size_hint() == (960, None) and r.total_duration() == 110_000_000
    r.pull(2)  # the echo's chain: Take min(9600 + 960, 32768) - 2 left, 958 owed; channels.rs:88-102 at position 0: 958 / 2 * 2 = 958
    assert r.size_hint() == (958, None)
    r.pull(1)  # 957 owed, position 1: (957 + 1) / 2 * 2 - 1 = 957
    assert r.size_hint() == (957, None)
    assert O.TestSource(z, 2, 48000).reverb(10_000_000, 0.5).total_duration() is None


def test_channels_size_hint_reference_test(O):
    """channels.rs:146-161 `size_hint`: the converter over a plain Vec::into_iter() counts its remaining output exactly, down to (0, Some(0))."""
    def check(inp, f, t):
        count = len(O.ChannelCountConverter(O.TestSource(inp, 1, 1, exact_size_hint=True), f, t).collect())
        c = O.ChannelCountConverter(O.TestSource(inp, 1, 1, exact_size_hint=True), f, t)
        for left in range(count, -1, -1):
            assert c.size_hint() == (left, left), (f, t, left, c.size_hint())
            c.pull(1)
        assert c.size_hint() == (0, 0)

    check([1.0, 2.0, 3.0], 1, 2)
    check([1.0, 2.0, 3.0, 4.0], 2, 4)
    check([1.0, 2.0, 3.0, 4.0], 4, 2)

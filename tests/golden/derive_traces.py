"""A SECOND derivation of the rows of SURVEY.md 8(a) that the reference itself pins with no numbers (a4, a7, a8, a9, a11, a12): short traces
computed here, in plain Python over numpy float32 SCALARS (every operation rounds to f32, as Rust's does), each a straight-line reading of
the cited lines of /root/reference -- written from those lines, not from oracle/rodio_oracle.cpp, with which it shares no code and no
language.  VERDICT r05 weak #1: "half the oracle is pinned by nothing but care".  What comes out is committed as tests/golden/traces.npz;
tests/test_oracle_traces.py holds the oracle (CPU suite) and the HIP kernels (GPU suite) against it.

    python tests/golden/derive_traces.py        (rewrites tests/golden/traces.npz; needs nothing but numpy)

Arithmetic.  + - * / sqrt abs max min and comparisons on np.float32 scalars are IEEE-754 single precision: bit for bit what Rust computes
(Rust never contracts a*b+c).  The transcendental functions are where platforms may differ in the last bit: they are evaluated in float64 and
rounded once to f32 (the correctly rounded result, which glibc's expf / sinf / cosf / log2f / exp2f -- what both Rust's std and the oracle's
C++ call on this platform -- return in all but rare cases).  Traces whose every step is IEEE-exact given their coefficients (biquad, AGC,
reverb, amplify, conversions) are compared BIT FOR BIT; the limiter calls log2 and exp2 per sample and is compared to 2e-6.
"""
import os

import numpy as np

f32 = np.float32
HERE = os.path.dirname(os.path.abspath(__file__))


def rnd(seed, n, scale=1.0):
    return (np.random.default_rng(seed).uniform(-1, 1, n) * scale).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------- math.rs
def duration_to_float(ns):  # math.rs:118-122 = Duration::as_secs_f32: secs as f32 + nanos as f32 / 1e9
    return f32(ns // 1_000_000_000) + f32(ns % 1_000_000_000) / f32(1_000_000_000.0)


def duration_to_coefficient(ns, sample_rate):  # math.rs:110-113: Float::exp(-1.0 / (duration_to_float(duration) * sample_rate.get() as Float))
    with np.errstate(divide="ignore"):
        arg = f32(-1.0) / (duration_to_float(ns) * f32(sample_rate))
    return f32(np.exp(np.float64(arg)))


LOG2_10 = f32(3.32192809488736234787)   # std::f32::consts
LOG10_2 = f32(0.301029995663981195214)
PI = f32(3.14159265358979323846264338327950288)


def db_to_linear(db):  # math.rs:51-56: Float::powf(2.0, decibels * 0.05 * Float::LOG2_10)
    return f32(np.exp2(np.float64(db * f32(0.05) * LOG2_10)))


def linear_to_db(lin):  # math.rs:86-90: linear.log2() * Float::LOG10_2 * 20.0
    with np.errstate(divide="ignore"):
        return f32(np.log2(np.float64(lin))) * LOG10_2 * f32(20.0)


def fmax(a, b):  # f32::max / f32::min: if one argument is NaN the other is returned
    return b if a != a else a if b != b else (a if a > b else b)


def fmin(a, b):
    return b if a != a else a if b != b else (a if a < b else b)


# ---------------------------------------------------------------------------------------------------------------- a8: blt.rs
def blt_coefficients(high_pass, freq, q, fs):
    """blt.rs:502-544 `to_applier`: w0 = 2 pi f / fs; low pass: alpha = sin w0 / (2 q), b1 = 1 - cos w0, b0 = b2 = b1 / 2, a0 = 1 + alpha,
    a1 = -2 cos w0, a2 = 1 - alpha; high pass: b0 = b2 = (1 + cos w0) / 2, b1 = -1 - cos w0; everything divided by a0 (:545-556)."""
    w0 = f32(2.0) * PI * f32(freq) / f32(fs)
    sin_w0, cos_w0 = f32(np.sin(np.float64(w0))), f32(np.cos(np.float64(w0)))
    alpha = sin_w0 / (f32(2.0) * f32(q))
    if not high_pass:
        b1 = f32(1.0) - cos_w0
        b0 = b1 / f32(2.0)
        b2 = b0
    else:
        b0 = (f32(1.0) + cos_w0) / f32(2.0)
        b1 = f32(-1.0) - cos_w0
        b2 = b0
    a0 = f32(1.0) + alpha
    a1 = f32(-2.0) * cos_w0
    a2 = f32(1.0) - alpha
    return b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0


def blt(x, channels, co):
    """blt.rs:558-560 `apply`: b0*x_n + b1*x_n1 + b2*x_n2 - a1*y_n1 - a2*y_n2, left to right; :397-410 (mono), :431-451 (stereo: the two
    channels alternate), :472-492 (any layout: position mod channels): one history per channel, y_n2 = y_n1, x_n2 = x_n1, y_n1 = result, x_n1 = x."""
    b0, b1, b2, a1, a2 = co
    hist = [[f32(0.0)] * 4 for _ in range(channels)]  # x_n1, x_n2, y_n1, y_n2
    out = np.empty(len(x), np.float32)
    for n in range(len(x)):
        h = hist[n % channels]
        xn = f32(x[n])
        r = b0 * xn + b1 * h[0] + b2 * h[1] - a1 * h[2] - a2 * h[3]
        h[3], h[1], h[2], h[0] = h[2], h[0], r, xn
        out[n] = r
    return out


# ---------------------------------------------------------------------------------------------------------------- a9: reverb
def delay_samples(ns, rate, channels):  # delay.rs:8-16: (ns * channels * rate / 1e9) in u128, truncated -- INTERLEAVED samples
    return ns * channels * rate // 1_000_000_000


def reverb(x, ns, amplitude, rate, channels):
    """source/mod.rs:628-634: self.mix(self.clone().amplify(amplitude).delay(duration)).  mix.rs:43-53: both -> s1 + s2, one -> that one,
    none -> end.  delay.rs:68-75: `remaining_samples` zeros (Some(0.0)), then the input.  amplify.rs:64: x * factor.  (Both inputs of Mix sit in
    a UniformSourceIterator at the source's own format: samples pass through as they are.)  The delay counts SAMPLES: an odd one on stereo puts
    the left channel's echo under the right channel."""
    d = delay_samples(ns, rate, channels)
    out = np.empty(len(x) + d, np.float32)
    for n in range(len(x) + d):
        s1 = f32(x[n]) if n < len(x) else None
        if n < d:
            s2 = f32(0.0)
        elif n - d < len(x):
            s2 = f32(x[n - d]) * f32(amplitude)
        else:
            s2 = None
        out[n] = s1 + s2 if s1 is not None and s2 is not None else s1 if s1 is not None else s2
    return out


# ---------------------------------------------------------------------------------------------------------------- a11: limit.rs
def limit(x, channels, rate, threshold=-1.0, knee_width=4.0, attack_ns=5_000_000, release_ns=100_000_000):
    """limit.rs:94-130: attack / release = duration_to_coefficient(.., sample_rate); :876-885: inv_knee_8 = 1 / (8 * knee_width).
    :853-873 `process_sample`: bias = linear_to_db(|x| + f32::MIN_POSITIVE) - threshold; k = bias * 2; k < -knee -> 0; |k| <= knee ->
    (k + knee)^2 * inv_knee_8; else bias.  :903-916 `process_channel`: integrator = max(db, release * integrator + (1 - release) * db);
    peak = attack * peak + (1 - attack) * integrator.  Gain: x * db_to_linear(-max over the channels' CURRENT peaks) -- mono :927-935, stereo
    :941-960 (the channel toggles; the other channel's peak is the one of ITS last sample: L(n) meets R(n-1)), any layout :966-988 (fold from 0.0)."""
    threshold, knee_width = f32(threshold), f32(knee_width)
    attack, release = duration_to_coefficient(attack_ns, rate), duration_to_coefficient(release_ns, rate)
    inv_knee_8 = f32(1.0) / (f32(8.0) * knee_width)
    integ = [f32(0.0)] * channels
    peak = [f32(0.0)] * channels
    min_positive = f32(np.finfo(np.float32).tiny)
    out = np.empty(len(x), np.float32)
    for n in range(len(x)):
        c = n % channels
        s = f32(x[n])
        bias_db = linear_to_db(abs(s) + min_positive) - threshold
        knee_boundary_db = bias_db * f32(2.0)
        if knee_boundary_db < -knee_width:
            db = f32(0.0)
        elif abs(knee_boundary_db) <= knee_width:
            t = knee_boundary_db + knee_width
            db = t * t * inv_knee_8
        else:
            db = bias_db
        integ[c] = fmax(db, release * integ[c] + (f32(1.0) - release) * db)
        peak[c] = attack * peak[c] + (f32(1.0) - attack) * integ[c]
        if channels == 1:
            max_peak = peak[0]
        elif channels == 2:
            max_peak = fmax(peak[0], peak[1])
        else:
            max_peak = f32(0.0)
            for p in peak:
                max_peak = fmax(max_peak, p)
        out[n] = s * db_to_linear(-max_peak)
    return out


# ---------------------------------------------------------------------------------------------------------------- a12: agc.rs
def agc(x, rate, target_level=1.0, attack_ns=4_000_000_000, release_ns=0, absolute_max_gain=7.0, floor=0.0, window=8192):
    """agc.rs:183-236: attack / release coefficients from durations clamped to 10 s by the trait method (source/mod.rs:432-433); current_gain 1.0,
    peak_level 0.0, floor.  ONE state for all interleaved channels.  :433-504 `process_sample`: v = |x|; :397-407 peak = peak * c + v * (1 - c)
    with c = 0 if v > peak else release; :413-417 + :145-156: old = buffer[i]; sum = sum - old + v*v; buffer[i] = v*v; i = (i + 1) & 8191;
    rms = sqrt(sum / 8192); rms_gain = target / rms if rms > 0 else max_gain; :421-427 peak_gain = min(target / peak, max_gain) if peak > 0 else
    max_gain; desired = max(min(rms_gain, peak_gain), floor); speed = attack if desired > gain else release; gain = gain * speed + desired *
    (1 - speed); gain = clamp(gain, 0.1, max_gain) (f32::clamp: NaN stays NaN); out = x * gain."""
    ten_s = 10_000_000_000
    attack = duration_to_coefficient(min(attack_ns, ten_s), rate)
    release = duration_to_coefficient(min(release_ns, ten_s), rate)
    target, max_gain, floor = f32(target_level), f32(absolute_max_gain), f32(floor)
    buf = [f32(0.0)] * window
    total, idx, peak, gain = f32(0.0), 0, f32(0.0), f32(1.0)
    out = np.empty(len(x), np.float32)
    gains = np.empty(len(x), np.float32)
    for n in range(len(x)):
        s = f32(x[n])
        v = abs(s)
        c = f32(0.0) if v > peak else release
        peak = peak * c + v * (f32(1.0) - c)
        sq = v * v
        old = buf[idx]
        total = total - old + sq
        buf[idx] = sq
        idx = (idx + 1) & (window - 1)
        rms = np.sqrt(total / f32(window))
        rms_gain = target / rms if rms > 0 else max_gain
        peak_gain = fmin(target / peak, max_gain) if peak > 0 else max_gain
        desired = fmax(fmin(rms_gain, peak_gain), floor)
        speed = attack if desired > gain else release
        gain = gain * speed + desired * (f32(1.0) - speed)
        if gain < f32(0.1):  # f32::clamp
            gain = f32(0.1)
        elif gain > max_gain:
            gain = max_gain
        gains[n] = gain
        out[n] = s * gain
    return out, gains


# ---------------------------------------------------------------------------------------------------------------- a4 / (f)2: dasp_sample 0.11.0
def dasp_known_answers():
    """dasp_sample 0.11.0 `conv` (Cargo.lock:317-318; the crate is not vendored: the published formulas, SURVEY.md 8(a) a4, evaluated BY HAND at
    the codes where a mistake would show): signed -> f32 divides by 2^(bits-1); unsigned first moves to signed by subtracting 2^(bits-1);
    f32 -> iN multiplies by 2^(bits-1) and casts with Rust's `as` (saturating, NaN -> 0); f32 -> uN goes through iN and adds 2^(bits-1)."""
    return {
        "i16_to_f32": (np.array([-32768, -1, 0, 1, 16384, 32767], np.int16), np.array([-1.0, -1 / 32768, 0.0, 1 / 32768, 0.5, 32767 / 32768], np.float32)),
        "u16_to_f32": (np.array([0, 1, 32767, 32768, 49152, 65535], np.uint16), np.array([-1.0, -32767 / 32768, -1 / 32768, 0.0, 0.5, 32767 / 32768], np.float32)),
        "i8_to_f32": (np.array([-128, -1, 0, 64, 127], np.int8), np.array([-1.0, -1 / 128, 0.0, 0.5, 127 / 128], np.float32)),
        "u8_to_f32": (np.array([0, 127, 128, 192, 255], np.uint8), np.array([-1.0, -1 / 128, 0.0, 0.5, 127 / 128], np.float32)),
        "i24_to_f32": (np.array([-8388608, -1, 0, 4194304, 8388607], np.int32), np.array([-1.0, -1 / 8388608, 0.0, 0.5, 8388607 / 8388608], np.float32)),
        "i32_to_f32": (np.array([-2147483648, 0, 1073741824, 2147483647], np.int32), np.array([-1.0, 0.0, 0.5, 1.0], np.float32)),  # (2^31 - 1) / 2^31 rounds to 1.0 in f32
        # f32 -> i16: 1.0 * 32768 = 32768 saturates to 32767; -1.0 -> -32768; 0.5 -> 16384; -0.99997 * 32768 = -32767.017 truncates to -32767; NaN -> 0; +-inf saturate; 2.0 saturates
        "f32_to_i16": (np.array([1.0, -1.0, 0.5, -0.99997, 0.0, np.nan, np.inf, -np.inf, 2.0, 3.0518e-5, -3.0518e-5], np.float32),
                       np.array([32767, -32768, 16384, -32767, 0, 0, 32767, -32768, 32767, 1, -1], np.int16)),
        "f32_to_u16": (np.array([1.0, -1.0, 0.0, 0.5, np.nan, -2.0], np.float32), np.array([65535, 0, 32768, 49152, 32768, 0], np.uint16)),
        "f32_to_i8": (np.array([1.0, -1.0, 0.5, np.nan, 0.0078125], np.float32), np.array([127, -128, 64, 0, 1], np.int8)),
        "f32_to_i32": (np.array([1.0, -1.0, 0.5, np.nan, -0.25], np.float32), np.array([2147483647, -2147483648, 1073741824, 0, -536870912], np.int32)),
        "f32_to_i24": (np.array([0.5, -1.0, 0.25, np.nan], np.float32), np.array([4194304, -8388608, 2097152, 0], np.int32)),  # (an i32 container, unchecked: 1.0 -> 8388608)
    }


def main():
    out = {}
    # a8: low_pass(200) / high_pass(300) / low_pass(1000) at q = 0.5, stereo 48 kHz and mono 44.1 kHz, three channels
    x2 = rnd(101, 2 * 64, 0.9)
    x1 = rnd(102, 48, 0.9)
    x3 = rnd(103, 3 * 20, 0.9)
    out["blt_x2"], out["blt_x1"], out["blt_x3"] = x2, x1, x3
    for name, hp, freq, fs in (("lp200_48k", False, 200, 48000), ("hp300_48k", True, 300, 48000), ("lp1000_44k", False, 1000, 44100), ("hp1000_44k", True, 1000, 44100)):
        co = blt_coefficients(hp, freq, 0.5, fs)
        out[f"blt_co_{name}"] = np.array(co, np.float32)
        out[f"blt_y2_{name}"] = blt(x2, 2, co)
        out[f"blt_y1_{name}"] = blt(x1, 1, co)
        out[f"blt_y3_{name}"] = blt(x3, 3, co)
    # a9: reverb on stereo with an odd delay (delay.rs:14: 333 333 ns * 2 * 48000 / 1e9 = 31.99 -> 31 samples: L's echo under R) and an even one; mono
    xr = rnd(104, 2 * 40, 0.8)
    out["reverb_x"] = xr
    out["reverb_odd_ns"], out["reverb_even_ns"] = np.int64(333_333), np.int64(250_000)
    assert delay_samples(333_333, 48000, 2) == 31 and delay_samples(250_000, 48000, 2) == 24
    out["reverb_odd"] = reverb(xr, 333_333, 0.3, 48000, 2)
    out["reverb_even"] = reverb(xr, 250_000, 0.3, 48000, 2)
    out["reverb_mono"] = reverb(xr, 250_000, 0.5, 48000, 1)
    # a7: amplify (amplify.rs:64) and amplify_decibel (:33-35) on a few values
    xa = np.array([0.0, 1.0, -1.0, 0.3, -0.7, 1e-30, 3.0e38], np.float32)
    out["amplify_x"] = xa
    out["amplify_0p8"] = np.array([f32(v) * f32(0.8) for v in xa], np.float32)
    # a11: the limiter on loud stereo (L and R differ: the coupling L(n) / R(n-1) shows), mono, and three channels; a burst over silence
    xl2 = rnd(105, 2 * 1200, 1.0)   # 25 ms of stereo at 48 kHz: five attack times
    xl2[:1600] *= f32(2.0)   # loud, then quieter: attack, then release
    xl2[1600:] *= f32(0.5)
    xl2[0::2] *= f32(1.5)
    xl2[1::2] *= f32(0.25)
    xl1 = (rnd(106, 900, 2.0)).astype(np.float32)
    xl1[300:600] = 0.0
    xl3 = rnd(107, 3 * 300, 1.8)
    out["limit_x2"], out["limit_x1"], out["limit_x3"] = xl2, xl1, xl3
    out["limit_y2"] = limit(xl2, 2, 48000)
    out["limit_y1"] = limit(xl1, 1, 44100)
    out["limit_y3"] = limit(xl3, 3, 48000)
    out["limit_y2_m6"] = limit(xl2, 2, 48000, threshold=-6.0, knee_width=2.0, attack_ns=1_000_000, release_ns=20_000_000)
    # a12: the AGC with rodio's defaults (agc.rs:73-82: target 1.0, attack 4 s, release 0 s, max gain 7.0, floor 0) past the wrap of its window
    # (9000 > 8192 samples: the sum starts to give back what it took), silence -> burst -> silence, stereo interleaved through ONE state; and with
    # a release time (the general kernel), a floor and a lower ceiling
    xg = rnd(108, 9000, 0.05)
    xg[3000:3400] = rnd(109, 400, 3.0)
    xg[5000:5600] = 0.0
    out["agc_x"] = xg
    y, g = agc(xg, 48000)
    out["agc_y"], out["agc_gain"] = y, g
    y, g = agc(xg, 44100, target_level=0.5, attack_ns=500_000_000, release_ns=50_000_000, absolute_max_gain=4.0, floor=0.2)
    out["agc_y_rel"], out["agc_gain_rel"] = y, g
    # ... and a NaN in the input.  The sample itself comes out NaN (NaN * gain).  The window sum and the peak level are NaN from there on
    # (sum - old + NaN, agc.rs:150; peak * c + NaN * (1 - c), then NaN * 0 + v, :406) -- but the GAIN never is: `rms > 0.0` and `peak_level > 0.0`
    # are false for NaN (:453-457, :422-426), so both gains fall back to absolute_max_gain, desired = max_gain, and the gain climbs to the
    # ceiling and stays there: after one NaN rodio's AGC plays everything at absolute_max_gain until the adapter is rebuilt.
    xn = rnd(110, 64, 0.3)
    xn[20] = np.nan
    out["agc_nan_x"] = xn
    with np.errstate(invalid="ignore"):
        y, g = agc(xn, 48000, attack_ns=1_000_000)
    out["agc_nan_y"], out["agc_nan_gain"] = y, g
    assert np.isnan(y[20]) and np.all(np.isfinite(y[:20])) and np.all(np.isfinite(y[21:])) and np.all(np.isfinite(g)) and g[63] > 3.0
    for k, (src, dst) in dasp_known_answers().items():
        out[f"dasp_{k}_in"], out[f"dasp_{k}_out"] = src, dst
    np.savez(os.path.join(HERE, "traces.npz"), **out)
    print("wrote", os.path.join(HERE, "traces.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()

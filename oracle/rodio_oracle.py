"""ctypes front-end of the CPU oracle (oracle/rodio_oracle.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (rodio_amd/) never imports it.

The wrapper mirrors rodio's adapter spelling so parity tests read like the
reference's own tests, e.g.

    src = O.SamplesBuffer(1, 48000, [10., -10., 10., -10.])
    out = O.SampleRateConverter(src, 2000, 3000, 2).collect()
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librodio_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "rodio_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    vp, f32p = C.c_void_p, C.POINTER(C.c_float)
    sigs = {
        "orc_vec_source": (vp, [f32p, C.c_size_t, C.c_int, C.c_uint, C.c_long]),
        "orc_seq_source": (vp, []),
        "orc_seq_add": (None, [vp, f32p, C.c_size_t, C.c_int, C.c_uint]),
        "orc_current_span_len": (C.c_long, [vp]),
        "orc_size_hint": (C.c_int, [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
        "orc_total_duration_ns": (C.c_longlong, [vp]),
        "orc_vec_set_total_duration": (None, [vp, C.c_longlong]),
        "orc_vec_set_exact_hint": (None, [vp, C.c_int]),
        "orc_buffered": (vp, [vp]),
        "orc_sample_rate_converter": (vp, [vp, C.c_uint, C.c_uint, C.c_int]),
        "orc_channel_count_converter": (vp, [vp, C.c_int, C.c_int]),
        "orc_uniform": (vp, [vp, C.c_int, C.c_uint]),
        "orc_amplify": (vp, [vp, C.c_float]),
        "orc_distortion": (vp, [vp, C.c_float, C.c_float]),
        "orc_dither": (vp, [vp, C.c_uint, C.c_int, C.c_ulonglong]),
        "orc_take_duration": (vp, [vp, C.c_ulonglong, C.c_int]),
        "orc_take_duration_sought": (vp, [vp, C.c_ulonglong, C.c_ulonglong, C.c_int]),
        "orc_linear_gain_ramp": (vp, [vp, C.c_ulonglong, C.c_float, C.c_float, C.c_int]),
        "orc_low_pass": (vp, [vp, C.c_uint, C.c_float]),
        "orc_high_pass": (vp, [vp, C.c_uint, C.c_float]),
        "orc_delay": (vp, [vp, C.c_ulonglong]),
        "orc_speed": (vp, [vp, C.c_float]),
        "orc_reverb": (vp, [vp, C.c_ulonglong, C.c_float]),
        "orc_channel_volume": (vp, [vp, f32p, C.c_int]),
        "orc_spatial": (vp, [vp, f32p, f32p, f32p]),
        "orc_spatial_gains": (None, [f32p, f32p, f32p, f32p]),
        "orc_limit": (vp, [vp, C.c_float, C.c_float, C.c_ulonglong, C.c_ulonglong]),
        "orc_agc": (vp, [vp, C.c_float, C.c_ulonglong, C.c_ulonglong, C.c_float, C.c_float]),
        "orc_mixer": (vp, [C.c_int, C.c_uint]),
        "orc_mixer_add": (None, [vp, vp]),
        "orc_pull": (C.c_size_t, [vp, f32p, C.c_size_t]),
        "orc_drain": (C.c_size_t, [vp, C.POINTER(C.c_double)]),
        "orc_channels": (C.c_int, [vp]),
        "orc_sample_rate": (C.c_uint, [vp]),
        "orc_free": (None, [vp]),
        "orc_lerp": (C.c_float, [C.c_float, C.c_float, C.c_uint, C.c_uint]),
        "orc_db_to_linear": (C.c_float, [C.c_float]),
        "orc_linear_to_db": (C.c_float, [C.c_float]),
        "orc_duration_to_coefficient": (C.c_float, [C.c_ulonglong, C.c_uint]),
        "orc_blt_coeffs": (None, [C.c_int, C.c_uint, C.c_float, C.c_uint, f32p]),
        "orc_delay_samples": (C.c_ulonglong, [C.c_ulonglong, C.c_uint, C.c_int]),
        "orc_pipeline_resample_lowpass_mix": (
            C.c_size_t,
            [f32p, C.c_int, C.c_size_t, C.c_int, C.c_uint, C.c_uint, C.c_long, C.c_uint,
             C.c_float, f32p, C.c_size_t],
        ),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for name in ("i8_to_f32", "i16_to_f32", "u16_to_f32", "u8_to_f32", "i24_to_f32", "i32_to_f32",
                 "f32_to_i16", "f32_to_i8", "f32_to_i32", "f32_to_u16",
                 "f32_to_u8", "f32_to_i24", "f32_to_u24", "f32_to_u32", "f32_to_i64", "f32_to_u64", "f32_to_f64",
                 "u24_to_f32", "u32_to_f32", "i64_to_f32", "u64_to_f32", "f64_to_f32", "i64_to_f32_via_f64", "u64_to_f32_via_f64"):
        fn = getattr(lib, "orc_" + name)
        fn.restype = None
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    return lib


_lib = _load()


def _f32p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


SPAN_NONE = -1          # current_span_len() == None   (benches' TestSource, SineWave, ...)
SPAN_SAMPLES_BUFFER = -2  # SamplesBuffer rule: Some(len) until exhausted (src/buffer.rs:76-82)


class Source:
    """Owning handle of one oracle adapter.  Wrapping a Source in another adapter moves it
    (like Rust): the inner handle must not be used afterwards."""

    def __init__(self, ptr):
        self._p = ptr

    def _take(self):
        p, self._p = self._p, None
        if p is None:
            raise RuntimeError("source was moved")
        return p

    def __del__(self):
        if getattr(self, "_p", None):
            _lib.orc_free(self._p)
            self._p = None

    # -- iterator -----------------------------------------------------------
    def pull(self, n: int) -> np.ndarray:
        out = np.empty(max(n, 1), dtype=np.float32)
        got = _lib.orc_pull(self._p, _f32p(out), n)
        return out[:got].copy()

    def collect(self, chunk: int = 1 << 20) -> np.ndarray:
        parts = []
        while True:
            p = self.pull(chunk)
            parts.append(p)
            if len(p) < chunk:
                break
        return np.concatenate(parts) if parts else np.empty(0, np.float32)

    def drain(self):
        chk = C.c_double(0)
        n = _lib.orc_drain(self._p, C.byref(chk))
        return n, chk.value

    def channels(self) -> int:
        return _lib.orc_channels(self._p)

    def sample_rate(self) -> int:
        return _lib.orc_sample_rate(self._p)

    def current_span_len(self):
        v = _lib.orc_current_span_len(self._p)
        return None if v < 0 else v

    def size_hint(self):
        """Iterator::size_hint(): (lower, upper) with upper None for "no upper bound"."""
        lo, hi = C.c_ulonglong(0), C.c_ulonglong(0)
        bounded = _lib.orc_size_hint(self._p, C.byref(lo), C.byref(hi))
        return int(lo.value), (int(hi.value) if bounded else None)

    def total_duration(self):
        """Source::total_duration() in nanoseconds, or None (source/mod.rs:209-213)."""
        v = _lib.orc_total_duration_ns(self._p)
        return None if v < 0 else int(v)

    def buffered(self):  # buffered.rs:11-24
        return Source(_lib.orc_buffered(self._take()))

    # -- rodio's builder methods (src/source/mod.rs:255-731) ------------------
    def amplify(self, factor):
        return Source(_lib.orc_amplify(self._take(), factor))

    def take_duration(self, duration_ns, fade_out=False):  # take.rs; fade_out = set_filter_fadeout()
        return Source(_lib.orc_take_duration(self._take(), duration_ns, int(fade_out)))

    def take_duration_sought(self, duration_ns, pos_ns, fade_out=False):  # take.rs:222-231: the adapter as try_seek(pos) leaves it
        return Source(_lib.orc_take_duration_sought(self._take(), duration_ns, pos_ns, int(fade_out)))

    def dither(self, target_bits, algorithm="TPDF", seed=0):  # dither.rs:217-242 with the counter-based noise contract
        return Source(_lib.orc_dither(self._take(), target_bits, {"GPDF": 0, "HighPass": 1, "RPDF": 2, "TPDF": 3}[algorithm], seed))

    def distortion(self, gain, threshold):
        return Source(_lib.orc_distortion(self._take(), gain, threshold))

    def linear_gain_ramp(self, duration_ns, start_gain, end_gain, clamp_end):
        return Source(_lib.orc_linear_gain_ramp(self._take(), duration_ns, start_gain, end_gain, int(clamp_end)))

    def fade_in(self, duration_ns):  # fadein.rs:11-13
        return self.linear_gain_ramp(duration_ns, 0.0, 1.0, False)

    def fade_out(self, duration_ns):  # fadeout.rs:13
        return self.linear_gain_ramp(duration_ns, 1.0, 0.0, True)

    def low_pass(self, freq, q=0.5):
        return Source(_lib.orc_low_pass(self._take(), freq, q))

    def high_pass(self, freq, q=0.5):
        return Source(_lib.orc_high_pass(self._take(), freq, q))

    def speed(self, factor):  # speed.rs: only the reported sample rate changes
        return Source(_lib.orc_speed(self._take(), factor))

    def delay(self, ns):
        return Source(_lib.orc_delay(self._take(), ns))

    def reverb(self, ns, amplitude):
        return Source(_lib.orc_reverb(self._take(), ns, amplitude))

    def limit(self, threshold=-1.0, knee_width=4.0, attack_ns=5_000_000, release_ns=100_000_000):
        return Source(_lib.orc_limit(self._take(), threshold, knee_width, attack_ns, release_ns))

    def automatic_gain_control(self, target_level=1.0, attack_ns=4_000_000_000, release_ns=0,
                               absolute_max_gain=7.0, floor=0.0):
        return Source(_lib.orc_agc(self._take(), target_level, attack_ns, release_ns,
                                   absolute_max_gain, floor))


def TestSource(samples, channels, sample_rate, total_duration=None, exact_size_hint=False) -> Source:
    """benches/shared.rs:6-46, src/source/mod.rs:865-930: current_span_len() == None.  The benches' TestSource is GIVEN its
    total_duration (shared.rs:11,47-49: nanoseconds here) and answers the trait's default size_hint() (0, None) (it implements only
    next()); exact_size_hint=True makes it the plain `Vec::into_iter()` of the reference's converter tests, which counts its samples."""
    a = np.ascontiguousarray(samples, dtype=np.float32)
    p = _lib.orc_vec_source(_f32p(a), a.size, channels, sample_rate, SPAN_NONE)
    if total_duration is not None:
        _lib.orc_vec_set_total_duration(p, int(total_duration))
    if exact_size_hint:
        _lib.orc_vec_set_exact_hint(p, 1)
    return Source(p)


def SamplesBuffer(channels, sample_rate, samples) -> Source:
    """src/buffer.rs:23-140 (argument order as in rodio)."""
    a = np.ascontiguousarray(samples, dtype=np.float32)
    return Source(_lib.orc_vec_source(_f32p(a), a.size, channels, sample_rate, SPAN_SAMPLES_BUFFER))


def SpanSource(samples, channels, sample_rate, span_len) -> Source:
    """A source reporting a constant current_span_len() == Some(span_len) (decoder packets)."""
    a = np.ascontiguousarray(samples, dtype=np.float32)
    return Source(_lib.orc_vec_source(_f32p(a), a.size, channels, sample_rate, int(span_len)))


def SeqSource(parts) -> Source:
    """A source of several spans, each with a format of its own: parts = [(samples, channels, sample_rate), ...] (what a queue of sounds
    of different formats looks like to the adapters behind it, src/queue.rs:140-172).  current_span_len() = Some(len of the current part)."""
    p = _lib.orc_seq_source()
    for samples, channels, sample_rate in parts:
        a = np.ascontiguousarray(samples, dtype=np.float32)
        _lib.orc_seq_add(p, _f32p(a), a.size, channels, sample_rate)
    return Source(p)


def SampleRateConverter(inp: Source, from_rate, to_rate, channels) -> Source:
    return Source(_lib.orc_sample_rate_converter(inp._take(), from_rate, to_rate, channels))


def ChannelCountConverter(inp: Source, from_ch, to_ch) -> Source:
    return Source(_lib.orc_channel_count_converter(inp._take(), from_ch, to_ch))


def UniformSourceIterator(inp: Source, channels, sample_rate) -> Source:
    return Source(_lib.orc_uniform(inp._take(), channels, sample_rate))


def ChannelVolume(inp: Source, gains) -> Source:
    g = np.ascontiguousarray(gains, dtype=np.float32)
    return Source(_lib.orc_channel_volume(inp._take(), _f32p(g), g.size))


def spatial_gains(emitter, left, right) -> np.ndarray:
    e, l, r = (np.ascontiguousarray(x, dtype=np.float32) for x in (emitter, left, right))
    out = np.zeros(2, np.float32)
    _lib.orc_spatial_gains(_f32p(e), _f32p(l), _f32p(r), _f32p(out))
    return out


def Spatial(inp: Source, emitter, left, right) -> Source:
    e, l, r = (np.ascontiguousarray(x, dtype=np.float32) for x in (emitter, left, right))
    return Source(_lib.orc_spatial(inp._take(), _f32p(e), _f32p(l), _f32p(r)))


class Mixer:
    """`let (tx, rx) = mixer::mixer(ch, rate)` rolled into one object: add() is tx.add,
    next()/pull()/collect() are rx."""

    def __init__(self, channels, sample_rate):
        self.rx = Source(_lib.orc_mixer(channels, sample_rate))

    def add(self, src: Source):
        _lib.orc_mixer_add(self.rx._p, src._take())

    def next(self):
        p = self.rx.pull(1)
        return float(p[0]) if len(p) else None

    def pull(self, n):
        return self.rx.pull(n)

    def collect(self):
        return self.rx.collect()


# ---- scalar helpers ------------------------------------------------------------
def lerp(a, b, num, den):
    return _lib.orc_lerp(a, b, num, den)


def db_to_linear(db):
    return _lib.orc_db_to_linear(db)


def linear_to_db(lin):
    return _lib.orc_linear_to_db(lin)


def duration_to_coefficient(ns, rate):
    return _lib.orc_duration_to_coefficient(ns, rate)


def blt_coeffs(kind, freq, q, fs) -> np.ndarray:
    out = np.zeros(5, np.float32)
    _lib.orc_blt_coeffs(1 if kind in ("high_pass", 1, True) else 0, freq, q, fs, _f32p(out))
    return out


def delay_samples(ns, rate, ch) -> int:
    return int(_lib.orc_delay_samples(ns, rate, ch))


def convert(name: str, a: np.ndarray) -> np.ndarray:
    """SampleTypeConverter restatement, e.g. convert('i16_to_f32', int16_array)."""
    src_t, dst_t = name.split("_to_")
    dst_t = dst_t.split("_via_")[0]
    np_t = {"i8": np.int8, "u8": np.uint8, "i16": np.int16, "u16": np.uint16, "i24": np.int32, "u24": np.int32,
            "i32": np.int32, "u32": np.uint32, "i64": np.int64, "u64": np.uint64, "f32": np.float32, "f64": np.float64}
    a = np.ascontiguousarray(a, dtype=np_t[src_t])
    out = np.empty(a.shape, dtype=np_t[dst_t])
    getattr(_lib, "orc_" + name)(a.ctypes.data, out.ctypes.data, a.size)
    return out


def pipeline_resample_lowpass_mix(data: np.ndarray, from_rate, to_rate, span=SPAN_NONE, freq=200,
                                  q=0.5, want_output=True):
    """cfg-2 chain on the CPU.  data: [S, frames, ch] f32.  Returns the mixed stream (or just the
    produced-sample count when want_output is False: the timing leg of bench.py)."""
    data = np.ascontiguousarray(data, dtype=np.float32)
    S, frames, ch = data.shape
    if want_output:
        cap = int(frames * ch * (to_rate / from_rate + 1) + 64 * ch)
        out = np.empty(cap, np.float32)
        n = _lib.orc_pipeline_resample_lowpass_mix(_f32p(data), S, frames, ch, from_rate, to_rate,
                                                   span, freq, q, _f32p(out), cap)
        return out[:n].copy()
    return _lib.orc_pipeline_resample_lowpass_mix(_f32p(data), S, frames, ch, from_rate, to_rate,
                                                  span, freq, q, None, 0)

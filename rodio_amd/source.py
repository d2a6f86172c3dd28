"""Host-side mirror of rodio's `Source` adapter interface over the HIP C ABI.

rodio composes per-sample pull iterators (`trait Source: Iterator<Item = f32>`,
/root/reference/src/source/mod.rs:179-218).  Here a source is a whole block resident in HBM and
every adapter is one (or one fused) kernel launch through librodio_hip.so; names, argument
order and end-of-stream behaviour follow the reference so the parity tests read like rodio's
own tests:

    out = rh.SampleRateConverter(rh.TestSource(x, 2, 44100), 44100, 48000, 2).collect()
    mixer = rh.Mixer(2, 48000); mixer.add(rh.SamplesBuffer(1, 48000, x)); mixer.next()

Device memory, streams and (in bench.py) torch.distributed come from PyTorch-ROCm; no torch op
computes audio.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import AgcParams, LimitParams, RhError, RlmConfig, RlmGeometry, check, lib

_torch = None
_initialized = False


def _t():
    global _torch
    if _torch is None:
        import torch

        _torch = torch
    return _torch


def init(device: int = 0):
    """rh_init(): binds the library to a gfx950 device.  Raises RhError if there is none."""
    global _initialized
    torch = _t()
    if torch.cuda.is_available():
        torch.cuda.set_device(device)
    check(lib.rh_init(device), "rh_init")
    _initialized = True


def async_status(synchronize: bool = True):
    """rh_async_status(): raises RhError(RH_ERR_TIMEOUT) if a hand-off inside a handle-less scan kernel (`limit`, `biquad_batch`
    mode 1) expired in a launch that has completed -- by default after a device synchronise."""
    if synchronize:
        _t().cuda.synchronize()
    check(lib.rh_async_status(), "rh_async_status")


def _ensure():
    if not _initialized:
        init(_t().cuda.current_device() if _t().cuda.is_available() else 0)


def _stream():
    return C.c_void_p(_t().cuda.current_stream().cuda_stream)


def _dev_empty(n, dtype=None):
    torch = _t()
    return torch.empty(int(n), dtype=dtype or torch.float32, device="cuda")


def _ptr(t):
    return C.c_void_p(t.data_ptr())


SPAN_NONE = None


class GpuSource:
    """A block-resident `Source`: interleaved f32 samples in HBM + (channels, sample_rate,
    current_span_len).  Adapters return new GpuSources; `collect()` copies to the host."""

    def __init__(self, samples, channels: int, sample_rate: int, span_len=None):
        if channels <= 0 or sample_rate <= 0:
            raise ValueError("channels and sample_rate are NonZero in rodio")
        self.samples = samples  # torch.float32 CUDA tensor, 1-D
        self._channels = int(channels)
        self._sample_rate = int(sample_rate)
        self.span_len = span_len  # None, or int (samples)

    # -- Source trait ---------------------------------------------------------------------
    def channels(self) -> int:
        return self._channels

    def sample_rate(self) -> int:
        return self._sample_rate

    def current_span_len(self):
        return self.span_len

    def __len__(self):
        return int(self.samples.numel())

    def collect(self) -> np.ndarray:
        _t().cuda.current_stream().synchronize()
        return self.samples.detach().cpu().numpy().copy()

    # -- builder methods (src/source/mod.rs:255-731) ---------------------------------------
    def amplify(self, factor: float) -> "GpuSource":
        _ensure()
        out = _dev_empty(len(self))
        check(lib.rh_amplify(_ptr(out), _ptr(self.samples), len(self), factor, _stream()), "rh_amplify")
        return GpuSource(out, self._channels, self._sample_rate, self.span_len)

    def amplify_decibel(self, db: float) -> "GpuSource":
        """source/mod.rs amplify_decibel / Amplify::set_log_factor (amplify.rs:33-35): factor = db_to_linear(dB)."""
        return self.amplify(float(lib.rh_db_to_linear(db)))

    def speed(self, factor: float) -> "GpuSource":
        """src/source/speed.rs:104-133: the samples are untouched, the reported rate is scaled."""
        r = np.float32(self._sample_rate) * np.float32(factor)
        return GpuSource(self.samples, self._channels, int(max(r, np.float32(1.0))), self.span_len)

    def delay(self, duration_ns: int) -> "GpuSource":
        """src/source/delay.rs:8-16,68-75."""
        _ensure()
        d = delay_samples(duration_ns, self._sample_rate, self._channels)
        out = _dev_empty(len(self) + d)
        check(lib.rh_delay(_ptr(out), _ptr(self.samples), len(self), d, _stream()), "rh_delay")
        return GpuSource(out, self._channels, self._sample_rate, None if self.span_len is None else self.span_len + d)

    def take_duration(self, duration_ns: int, fade_out: bool = False) -> "GpuSource":
        """src/source/take.rs:96-148.  current_span_len() behind it is what the duration still admits unless the input's span is
        shorter -- Some(..) over an input that says None too, Some(0) once it is spent (take.rs:176-195): a UniformSourceIterator
        behind it converts in chains of 32768 samples and ends in front of the silence that completes a cut frame (`_admits`)."""
        _ensure()
        out = _dev_empty(len(self) + self._channels)
        m, ended = C.c_uint64(0), C.c_int32(0)
        check(lib.rh_take_duration(_ptr(out), _ptr(self.samples), len(self), 0, self._channels, self._sample_rate, duration_ns,
                                   int(fade_out), C.byref(m), C.byref(ended), _stream()), "rh_take_duration")
        per_sample = 1_000_000_000 // (self._sample_rate * self._channels)
        admits = duration_ns // per_sample if per_sample else 0
        span = self.span_len if (self.span_len is not None and self.span_len < admits) else admits
        res = GpuSource(out[: m.value], self._channels, self._sample_rate, span)
        res._admits = min(admits, m.value)
        return res

    def distortion(self, gain: float, threshold: float) -> "GpuSource":
        _ensure()
        out = _dev_empty(len(self))
        check(lib.rh_distortion(_ptr(out), _ptr(self.samples), len(self), gain, threshold, _stream()), "rh_distortion")
        return GpuSource(out, self._channels, self._sample_rate, self.span_len)

    DITHER = {"GPDF": 0, "HighPass": 1, "RPDF": 2, "TPDF": 3}

    def dither(self, target_bits: int, algorithm: str = "TPDF", seed: int = 0, sample_offset: int = 0) -> "GpuSource":
        """src/source/dither.rs:217-242 with a counter-based noise generator (see rh_dither)."""
        _ensure()
        out = _dev_empty(len(self))
        check(lib.rh_dither(_ptr(out), _ptr(self.samples), len(self), sample_offset, self._channels, target_bits, self.DITHER[algorithm], seed, _stream()), "rh_dither")
        return GpuSource(out, self._channels, self._sample_rate, self.span_len)

    def linear_gain_ramp(self, duration_ns: int, start_gain: float, end_gain: float, clamp_end: bool, sample_offset: int = 0) -> "GpuSource":
        _ensure()
        out = _dev_empty(len(self))
        check(lib.rh_linear_gain_ramp(_ptr(out), _ptr(self.samples), len(self), sample_offset, self._channels, self._sample_rate,
                                      duration_ns, start_gain, end_gain, int(clamp_end), _stream()), "rh_linear_gain_ramp")
        return GpuSource(out, self._channels, self._sample_rate, self.span_len)

    def fade_in(self, duration_ns: int) -> "GpuSource":  # fadein.rs:11-13
        return self.linear_gain_ramp(duration_ns, 0.0, 1.0, False)

    def fade_out(self, duration_ns: int) -> "GpuSource":  # fadeout.rs:13
        return self.linear_gain_ramp(duration_ns, 1.0, 0.0, True)

    def _blt(self, kind: int, freq: int, q: float, mode: int) -> "GpuSource":
        _ensure()
        co = biquad_coeffs(kind, freq, q, self._sample_rate)
        out = _dev_empty(len(self))
        frames = len(self) // self._channels
        check(lib.rh_biquad(_ptr(out), _ptr(self.samples), frames, self._channels, 1,
                            co.ctypes.data_as(_lib.f32p), None, mode, _stream()), "rh_biquad")
        return GpuSource(out[: frames * self._channels], self._channels, self._sample_rate, self.span_len)

    def low_pass(self, freq: int, q: float = 0.5, mode: int = 0) -> "GpuSource":
        return self._blt(0, freq, q, mode)

    def high_pass(self, freq: int, q: float = 0.5, mode: int = 0) -> "GpuSource":
        return self._blt(1, freq, q, mode)

    def reverb(self, duration_ns: int, amplitude: float) -> "GpuSource":
        _ensure()
        d = delay_samples(duration_ns, self._sample_rate, self._channels)
        out = _dev_empty(len(self) + d)
        check(lib.rh_echo_mix(_ptr(out), _ptr(self.samples), len(self), d, amplitude, _stream()), "rh_echo_mix")
        return GpuSource(out, self._channels, self._sample_rate, None)

    def limit(self, threshold=-1.0, knee_width=4.0, attack_ns=5_000_000, release_ns=100_000_000) -> "GpuSource":
        """src/source/limit.rs:94-130,853-988.  A stream that ends inside a frame (rodio's own reverb with an odd delay
        makes one) is still limited sample by sample (limit.rs:927-988 advances one channel per sample): the block is
        padded to a whole frame -- the limiter is causal, the padding cannot reach the real samples -- and trimmed."""
        _ensure()
        torch = _t()
        p = LimitParams(threshold, knee_width, attack_ns, release_ns)
        n = len(self)
        frames = -(-n // self._channels)
        src = self.samples
        if frames * self._channels != n:
            src = torch.zeros(frames * self._channels, dtype=torch.float32, device="cuda")
            src[:n] = self.samples
        out = _dev_empty(frames * self._channels)
        check(lib.rh_limit(_ptr(out), _ptr(src), frames, self._channels, self._sample_rate, 1,
                           C.byref(p), None, _stream()), "rh_limit")
        return GpuSource(out[:n], self._channels, self._sample_rate, self.span_len)

    def automatic_gain_control(self, target_level=1.0, attack_ns=4_000_000_000, release_ns=0,
                               absolute_max_gain=7.0, floor=0.0) -> "GpuSource":
        _ensure()
        p = AgcParams(target_level, attack_ns, release_ns, absolute_max_gain, floor)
        out = _dev_empty(len(self))
        check(lib.rh_agc(_ptr(out), _ptr(self.samples), len(self), self._sample_rate, 1, C.byref(p), None,
                         _stream()), "rh_agc")
        return GpuSource(out, self._channels, self._sample_rate, self.span_len)


def _upload(samples):
    torch = _t()
    a = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1)
    if a.size == 0:
        return torch.empty(0, dtype=torch.float32, device="cuda")
    return torch.from_numpy(a).to("cuda")


def TestSource(samples, channels, sample_rate) -> GpuSource:
    """benches/shared.rs:6-46: a Vec<f32> source with current_span_len() == None."""
    return GpuSource(_upload(samples), channels, sample_rate, None)


def SamplesBuffer(channels, sample_rate, samples) -> GpuSource:
    """src/buffer.rs:23-140: current_span_len() == Some(len)."""
    s = _upload(samples)
    return GpuSource(s, channels, sample_rate, int(s.numel()))


def SpanSource(samples, channels, sample_rate, span_len) -> GpuSource:
    return GpuSource(_upload(samples), channels, sample_rate, int(span_len))


# ---- conversions ---------------------------------------------------------------------------
def SampleRateConverter(inp: GpuSource, from_rate: int, to_rate: int, channels: int, span_len=0) -> GpuSource:
    """src/conversions/sample_rate.rs:52-57 (same argument order).  span_len applies
    UniformSourceIterator's chunking (0 = one continuous stream)."""
    _ensure()
    torch = _t()
    frames, rem = divmod(len(inp), channels)
    if from_rate == to_rate and rem:  # sample_rate.rs:133-136: pure pass-through, half frames included
        return GpuSource(inp.samples.clone(), channels, to_rate, None)
    m = C.c_uint64(0)
    check(lib.rh_resample_out_frames(frames, from_rate, to_rate, channels, span_len, C.byref(m)), "rh_resample_out_frames")
    out = _dev_empty(m.value * channels)
    check(lib.rh_resample_linear(_ptr(out), _ptr(inp.samples), frames, from_rate, to_rate, channels, span_len,
                                 _stream()), "rh_resample_linear")
    if rem and span_len == 0:
        # A stream that ends mid-frame (rodio's own `reverb` with an odd delay produces one; the sources'
        # contract source/mod.rs:169-178 forbids it).  sample_rate.rs:174-200: frames are Vecs, the zip of
        # the last whole frame with the partial one yields only `rem` samples per output position, then
        # the partial frame is drained verbatim.  A handful of samples: done here, on the host, in f32.
        g = np.gcd(from_rate, to_rate)
        F, T = from_rate // g, to_rate // g
        tail_in = inp.samples[max(frames - 1, 0) * channels:].cpu().numpy()
        part = tail_in[-rem:]
        if frames == 0:
            return GpuSource(_upload(part), channels, to_rate, None)
        last = tail_in[:channels]
        c1 = -((-(frames - 1) * T) // F)  # output positions whose two taps are whole frames
        tail, mm = [], c1
        while (mm * F) // T == frames - 1:
            num = np.float32((mm * F) % T)
            tail += [np.float32(last[c] + (part[c] - last[c]) * num / np.float32(T)) for c in range(rem)]  # math.rs:23-26
            mm += 1
        if (mm * F) // T == frames:
            tail += [np.float32(v) for v in part]
        out = torch.cat([out[: c1 * channels], _upload(np.asarray(tail, np.float32))]) if tail else out[: c1 * channels]
    return GpuSource(out, channels, to_rate, None)


def ChannelCountConverter(inp: GpuSource, from_ch: int, to_ch: int) -> GpuSource:
    """src/conversions/channels.rs:28."""
    _ensure()
    if from_ch == to_ch:  # channels.rs:57-85 degenerates to a pass-through, half frames included
        return GpuSource(inp.samples, to_ch, inp.sample_rate(), inp.span_len)
    frames = len(inp) // from_ch
    out = _dev_empty(frames * to_ch)
    check(lib.rh_channels_convert(_ptr(out), _ptr(inp.samples), frames, from_ch, to_ch, _stream()), "rh_channels_convert")
    return GpuSource(out, to_ch, inp.sample_rate(), inp.span_len)


def UniformSourceIterator(inp: GpuSource, channels: int, sample_rate: int) -> GpuSource:
    """src/source/uniform.rs:50-97: ChannelCountConverter<SampleRateConverter<Take<I>>> restarted
    every min(current_span_len, 32768) samples."""
    span = inp.current_span_len() or 0
    admits = getattr(inp, "_admits", None)
    if admits is not None and admits < len(inp):  # (a TakeDuration answers Some(0) there: the iterator never asks for the frame's padding)
        inp = GpuSource(inp.samples[:admits], inp.channels(), inp.sample_rate(), inp.span_len)
    if min(span, 32768) >= len(inp):  # one chain for everything: the same as a continuous stream (and the form that knows a cut last frame)
        span = 0
    r = SampleRateConverter(inp, inp.sample_rate(), sample_rate, inp.channels(), span)
    return ChannelCountConverter(r, inp.channels(), channels)


_CONV = {
    ("i8", "f32"): ("rh_convert_i8_to_f32", np.int8, np.float32),
    ("u8", "f32"): ("rh_convert_u8_to_f32", np.uint8, np.float32),
    ("i16", "f32"): ("rh_convert_i16_to_f32", np.int16, np.float32),
    ("u16", "f32"): ("rh_convert_u16_to_f32", np.uint16, np.float32),
    ("i24", "f32"): ("rh_convert_i24_to_f32", np.int32, np.float32),
    ("i32", "f32"): ("rh_convert_i32_to_f32", np.int32, np.float32),
    ("f32", "i8"): ("rh_convert_f32_to_i8", np.float32, np.int8),
    ("f32", "i16"): ("rh_convert_f32_to_i16", np.float32, np.int16),
    ("f32", "u16"): ("rh_convert_f32_to_u16", np.float32, np.uint16),
    ("f32", "i32"): ("rh_convert_f32_to_i32", np.float32, np.int32),
    ("f32", "u8"): ("rh_convert_f32_to_u8", np.float32, np.uint8),
    ("f32", "i24"): ("rh_convert_f32_to_i24", np.float32, np.int32),
    ("f32", "u24"): ("rh_convert_f32_to_u24", np.float32, np.int32),
    ("f32", "u32"): ("rh_convert_f32_to_u32", np.float32, np.uint32),
    ("f32", "i64"): ("rh_convert_f32_to_i64", np.float32, np.int64),
    ("f32", "u64"): ("rh_convert_f32_to_u64", np.float32, np.uint64),
    ("f32", "f64"): ("rh_convert_f32_to_f64", np.float32, np.float64),
    ("u24", "f32"): ("rh_convert_u24_to_f32", np.int32, np.float32),
    ("u32", "f32"): ("rh_convert_u32_to_f32", np.uint32, np.float32),
    ("i64", "f32"): ("rh_convert_i64_to_f32", np.int64, np.float32),
    ("u64", "f32"): ("rh_convert_u64_to_f32", np.uint64, np.float32),
    ("f64", "f32"): ("rh_convert_f64_to_f32", np.float64, np.float32),
}


def SampleTypeConverter(samples, src: str, dst: str) -> np.ndarray:
    """src/conversions/sample.rs:6-44 (`SampleTypeConverter<I, O>`; north_star's "DataConverter").
    Host array in, host array out; the conversion itself runs on the GPU."""
    _ensure()
    torch = _t()
    fn, st, dt = _CONV[(src, dst)]
    a = np.ascontiguousarray(samples, dtype=st)
    # torch has no uint16 arithmetic but can carry the bytes
    d_in = torch.from_numpy(a.view(np.uint8).reshape(-1)).to("cuda") if a.size else torch.empty(0, dtype=torch.uint8, device="cuda")
    d_out = torch.empty(a.size * np.dtype(dt).itemsize, dtype=torch.uint8, device="cuda")
    check(getattr(lib, fn)(_ptr(d_out), _ptr(d_in), a.size, _stream()), fn)
    torch.cuda.current_stream().synchronize()
    return d_out.cpu().numpy().view(dt).copy()


# ---- WAV either side of the path -----------------------------------------------------------------
def wav_probe(file_bytes: bytes):
    """RIFF chunk walk on the host (no GPU needed): dict of the fmt fields + data chunk position."""
    info = _lib.WavInfo()
    buf = (C.c_uint8 * len(file_bytes)).from_buffer_copy(file_bytes)
    check(lib.rh_wav_probe_host(buf, len(file_bytes), C.byref(info)), "rh_wav_probe_host")
    return {n: getattr(info, n) for n, _ in _lib.WavInfo._fields_}


def _wav_data_on_device(file_bytes: bytes, w, image: bool):
    """The data chunk on the device: a copy of its own, or (image=True) the whole file uploaded as it is and the chunk taken where it lies
    -- `file + data_offset`, whatever byte address that is."""
    torch = _t()
    if image and w["data_bytes"]:
        whole = torch.from_numpy(np.frombuffer(file_bytes, dtype=np.uint8).copy()).to("cuda")
        return whole[w["data_offset"]: w["data_offset"] + w["data_bytes"]]
    raw = np.frombuffer(file_bytes, dtype=np.uint8, count=w["data_bytes"], offset=w["data_offset"])
    return torch.from_numpy(raw.copy()).to("cuda") if raw.size else None


def WavDecoder(file_bytes: bytes, image: bool = False) -> "GpuSource":
    """src/decoder/wav.rs: the data chunk goes to the device as bytes and is converted there."""
    _ensure()
    torch = _t()
    w = wav_probe(file_bytes)
    d_in = _wav_data_on_device(file_bytes, w, image)
    raw = np.empty(0 if d_in is None else 1, np.uint8)
    if d_in is None:
        d_in = torch.empty(0, dtype=torch.uint8, device="cuda")
    out = _dev_empty(w["samples"] + w["channels"])
    m = C.c_uint64(0)
    check(lib.rh_wav_decode(_ptr(out), _ptr(d_in) if raw.size else None, w["samples"], w["channels"], w["bits_per_sample"],
                            w["is_float"], C.byref(m), _stream()), "rh_wav_decode")
    return GpuSource(out[: m.value], w["channels"], w["sample_rate"], None)


def WavDecoderChannels(file_bytes: bytes, to_channels: int, image: bool = False) -> "GpuSource":
    """`UniformSourceIterator::new(decoder, to_channels, the file's rate)`: src/decoder/wav.rs + src/conversions/channels.rs:57-85 in ONE
    launch (rh_wav_decode_channels) -- the decoded block in the file's own layout never exists."""
    _ensure()
    torch = _t()
    w = wav_probe(file_bytes)
    d_in = _wav_data_on_device(file_bytes, w, image)
    if d_in is None:
        d_in = torch.empty(1, dtype=torch.uint8, device="cuda")
    frames = (w["samples"] + w["channels"] - 1) // w["channels"]
    out = _dev_empty(frames * to_channels + 4)
    m = C.c_uint64(0)
    check(lib.rh_wav_decode_channels(_ptr(out), _ptr(d_in), w["samples"], w["channels"], w["bits_per_sample"], w["is_float"], to_channels, C.byref(m), _stream()),
          "rh_wav_decode_channels")
    return GpuSource(out[: m.value], to_channels, w["sample_rate"], None)


def wav_to_bytes(source: "GpuSource") -> bytes:
    """src/wav_output.rs:62-96: 32-bit float WAVE of the source's whole frames."""
    n = len(source)
    hdr = (C.c_uint8 * 44)()
    k = lib.rh_wav_header_f32_host(hdr, 44, source.channels(), source.sample_rate(), n)
    if k != 44:
        raise RhError(1, "rh_wav_header_f32_host")
    whole = n - n % source.channels()
    return bytes(hdr) + source.samples[:whole].cpu().numpy().astype("<f4").tobytes()


# ---- effects -------------------------------------------------------------------------------
def ChannelVolume(inp: GpuSource, gains) -> GpuSource:
    """src/source/channel_volume.rs:29-37,71-88."""
    _ensure()
    g = np.ascontiguousarray(gains, dtype=np.float32)
    frames = len(inp) // inp.channels()
    out = _dev_empty(frames * g.size)
    check(lib.rh_channel_volume(_ptr(out), _ptr(inp.samples), frames, inp.channels(), g.ctypes.data_as(_lib.f32p),
                                g.size, _stream()), "rh_channel_volume")
    return GpuSource(out, g.size, inp.sample_rate(), inp.span_len)


def spatial_gains(emitter, left, right) -> np.ndarray:
    e, l, r = (np.ascontiguousarray(x, dtype=np.float32) for x in (emitter, left, right))
    out = np.zeros(2, np.float32)
    P = _lib.f32p
    check(lib.rh_spatial_gains(e.ctypes.data_as(P), l.ctypes.data_as(P), r.ctypes.data_as(P), out.ctypes.data_as(P)),
          "rh_spatial_gains")
    return out


def Spatial(inp: GpuSource, emitter, left, right) -> GpuSource:
    """src/source/spatial.rs:26-46."""
    return ChannelVolume(inp, spatial_gains(emitter, left, right))


def spatial_gains_batch(emitters, left, right):
    """Device tensor [S, 2] of Spatial gains (one rh_spatial_gains per emitter)."""
    torch = _t()
    gains = np.stack([spatial_gains(e, left, right) for e in emitters]).astype(np.float32)
    return torch.from_numpy(gains).to("cuda")


def reverb_spatial_batch(x, sample_rate, duration_ns, amplitude, emitters, left, right, out=None, gains_dev=None):
    """BASELINE config 3 in one kernel: for every row s of the device tensor x [S, n] (interleaved stereo)
    `Spatial(reverb(x_s, duration, amplitude), emitters[s], left, right)`.  Returns [S, n_out] on device."""
    _ensure()
    torch = _t()
    S, n = x.shape
    d = delay_samples(duration_ns, sample_rate, 2)
    n_out = 2 * ((n + d) // 2)
    g_dev = gains_dev if gains_dev is not None else spatial_gains_batch(emitters, left, right)
    if out is None:
        stride = (n_out + 3) // 4 * 4
        out = torch.empty((S, stride), device="cuda", dtype=torch.float32)
    check(lib.rh_reverb_spatial(_ptr(out), _ptr(x), n, d, amplitude, _ptr(g_dev), S, x.stride(0), out.stride(0), _stream()),
          "rh_reverb_spatial")
    return out[:, :n_out]


def biquad_batch(x, coeffs5, mode=1, out=None):
    """rh_biquad over the rows of the device tensor x [S, frames*2] (stereo streams laid out back to
    back).  mode 0 = sequential (bit-exact), mode 1 = time-parallel scan."""
    _ensure()
    torch = _t()
    assert x.is_contiguous() and x.dim() == 2
    S, n = x.shape
    if out is None:
        out = torch.empty_like(x)
    co = np.ascontiguousarray(coeffs5, dtype=np.float32)
    check(lib.rh_biquad(_ptr(out), _ptr(x), n // 2, 2, S, co.ctypes.data_as(_lib.f32p), None, mode, _stream()), "rh_biquad")
    return out


def _check_state(state, shape, who):
    """A carried state is a contiguous float32 device tensor of exactly the elements the kernel will read and write."""
    if state is None:
        return
    torch = _t()
    want = int(np.prod(shape))
    if state.dtype != torch.float32 or not state.is_cuda or not state.is_contiguous() or state.numel() != want:
        raise ValueError(f"{who}: state must be a contiguous float32 CUDA tensor of {want} elements (shape {tuple(shape)}), got {state.dtype} {tuple(state.shape)} on {state.device}")


def limit_batch(x, channels, sample_rate, threshold=-1.0, knee_width=4.0, attack_ns=5_000_000, release_ns=100_000_000, state=None, out=None):
    """rh_limit over the rows of the device tensor x [S, frames*channels] (limit.rs:853-988, one limiter per row).
    state: optional device tensor [S, 2*channels] {integrator, peak} per channel, carried across blocks (updated in place)."""
    _ensure()
    torch = _t()
    assert x.is_contiguous() and x.dim() == 2
    S, n = x.shape
    if n % channels:
        raise ValueError(f"rows of {n} samples are not whole frames of {channels} channels (the kernel's row stride is frames * channels)")
    _check_state(state, (S, 2 * channels), "limit_batch")
    if out is None:
        out = torch.empty_like(x)
    p = LimitParams(threshold, knee_width, attack_ns, release_ns)
    check(lib.rh_limit(_ptr(out), _ptr(x), n // channels, channels, sample_rate, S, C.byref(p), _ptr(state) if state is not None else None, _stream()), "rh_limit")
    return out


def agc_state(n_streams):
    """Fresh AutomaticGainControl states (agc.rs:209-236) for n_streams rows, on the device."""
    _ensure()
    st = _dev_empty(int(lib.rh_agc_state_floats()) * n_streams)
    check(lib.rh_agc_state_init(_ptr(st), n_streams, _stream()), "rh_agc_state_init")
    return st


def agc_batch(x, sample_rate, target_level=1.0, attack_ns=4_000_000_000, release_ns=0, absolute_max_gain=7.0, floor=0.0, state=None, out=None):
    """rh_agc over the rows of the device tensor x [S, n_samples] (agc.rs:397-504, one AGC per row; all interleaved channels of a
    row share it).  state: agc_state(S), carried across blocks."""
    _ensure()
    torch = _t()
    assert x.is_contiguous() and x.dim() == 2
    S, n = x.shape
    _check_state(state, (S * int(lib.rh_agc_state_floats()),), "agc_batch")
    if out is None:
        out = torch.empty_like(x)
    p = AgcParams(target_level, attack_ns, release_ns, absolute_max_gain, floor)
    check(lib.rh_agc(_ptr(out), _ptr(x), n, sample_rate, S, C.byref(p), _ptr(state) if state is not None else None, _stream()), "rh_agc")
    return out


def biquad_coeffs(kind, freq, q, fs) -> np.ndarray:
    out = np.zeros(5, np.float32)
    k = 1 if kind in (1, "high_pass") else 0
    check(lib.rh_biquad_coeffs(k, freq, q, fs, out.ctypes.data_as(_lib.f32p)), "rh_biquad_coeffs")
    return out


def filter_scan_ok(kind, freq, q, fs) -> bool:
    """The filter contract (rodio_hip.h, rh_filter_scan_ok): does the time-parallel evaluation of this low_pass / high_pass stay within
    1e-5 of rodio's own f32 recurrence for a full-scale source?"""
    return bool(lib.rh_filter_scan_ok(1 if kind in (1, "high_pass") else 0, int(freq), float(q), int(fs)))


def delay_samples(ns, rate, ch) -> int:
    return int(lib.rh_delay_samples(ns, rate, ch))


# ---- mixer ---------------------------------------------------------------------------------
class Mixer:
    """`let (tx, rx) = mixer::mixer(channels, rate)` as one object (src/mixer.rs:25-43).

    add() converts the source with UniformSourceIterator (mixer.rs:58-66) and admits it at the
    next frame boundary of the output position (mixer.rs:175-183); next()/pull() serve the
    ordered sum (mixer.rs:185-198) computed by rh_mix_sum."""

    def __init__(self, channels: int, sample_rate: int):
        self._channels, self._rate = int(channels), int(sample_rate)
        self._srcs: list[tuple[GpuSource, int]] = []
        self._pos = 0
        self._mixed = None

    def channels(self):
        return self._channels

    def sample_rate(self):
        return self._rate

    def add(self, src: GpuSource):
        u = UniformSourceIterator(src, self._channels, self._rate)
        start = -(-self._pos // self._channels) * self._channels  # next frame boundary
        self._srcs.append((u, start))
        self._mixed = None

    def _mix(self):
        if self._mixed is None:
            _ensure()
            n = len(self._srcs)
            out_len = max((st + len(u) for u, st in self._srcs), default=0)
            out = _dev_empty(out_len)
            if out_len:
                ptrs = (C.c_void_p * n)(*[u.samples.data_ptr() if len(u) else None for u, _ in self._srcs])
                starts = (C.c_uint64 * n)(*[st for _, st in self._srcs])
                lens = (C.c_uint64 * n)(*[len(u) for u, _ in self._srcs])
                check(lib.rh_mix_sum(_ptr(out), out_len, ptrs, starts, lens, n, _stream()), "rh_mix_sum")
            _t().cuda.current_stream().synchronize()
            self._mixed = out.cpu().numpy()
        return self._mixed

    def pull(self, n: int) -> np.ndarray:
        m = self._mix()
        out = m[self._pos: self._pos + n].copy()
        self._pos += len(out)
        return out

    def next(self):
        """MixerSource::next (mixer.rs:120-136): the channel position advances on EVERY call, also on the ones that return None
        (an empty mixer); a source added meanwhile starts at the next frame boundary, so an odd number of None calls is
        followed by one more None before its first sample."""
        m = self._mix()
        p = self._pos
        self._pos += 1
        if any(st <= p < st + len(u) for u, st in self._srcs):
            return float(m[p])
        return None

    def collect(self) -> np.ndarray:
        return self.pull(1 << 62)


# ---- block streaming -----------------------------------------------------------------------------
class StreamingResampler:
    """SampleRateConverter over a stream that arrives in blocks (rh_resampler_*): feed(block) returns the
    output frames that became computable; feed(block, flush=True) ends the stream (rodio's None)."""

    def __init__(self, from_rate, to_rate, channels):
        _ensure()
        self._h = C.c_void_p()
        self.channels = channels
        check(lib.rh_resampler_create(C.byref(self._h), from_rate, to_rate, channels), "rh_resampler_create")

    def feed(self, block, flush=False):
        frames = block.numel() // self.channels
        n = C.c_uint64(0)
        check(lib.rh_resampler_pending_frames(self._h, frames, int(flush), C.byref(n)), "rh_resampler_pending_frames")
        out = _dev_empty(max(n.value * self.channels, 1))
        m = C.c_uint64(0)
        check(lib.rh_resampler_process(self._h, _ptr(out), n.value, _ptr(block) if frames else None, frames, int(flush),
                                       C.byref(m), _stream()), "rh_resampler_process")
        return out[: m.value * self.channels]

    def close(self):
        if self._h:
            lib.rh_resampler_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StreamingReverb:
    """reverb(duration, amplitude) over a stream that arrives in blocks (rh_echo_*)."""

    def __init__(self, duration_ns, amplitude, sample_rate, channels):
        _ensure()
        self.delay = delay_samples(duration_ns, sample_rate, channels)
        self._h = C.c_void_p()
        check(lib.rh_echo_create(C.byref(self._h), self.delay, amplitude), "rh_echo_create")

    def feed(self, block):
        out = _dev_empty(max(block.numel(), 1))
        check(lib.rh_echo_process(self._h, _ptr(out), _ptr(block) if block.numel() else None, block.numel(), _stream()), "rh_echo_process")
        return out[: block.numel()]

    def flush(self):
        out = _dev_empty(max(self.delay, 1))
        check(lib.rh_echo_flush(self._h, _ptr(out), _stream()), "rh_echo_flush")
        return out[: self.delay]

    def close(self):
        if self._h:
            lib.rh_echo_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- fused headline pipeline ------------------------------------------------------------------
class ResampleLowpassMix:
    """BASELINE config 2 as one kernel: for every source
    `mixer.add(UniformSourceIterator::new(src, ch, to_rate).low_pass(freq))`, then the mixer sum.

    Sources are device tensors ([frames, 2] or flat interleaved f32).  `span_len` = the sources'
    current_span_len() (None/0 = continuous)."""

    def __init__(self, from_rate, to_rate, channels=2, span_len=None, filter="low_pass", freq=200, q=0.5,
                 max_sources=256, max_in_frames=1 << 20, frames_per_lane=0, ring_stages=0, no_balance=0, force_general=0, filter_first=False):
        _ensure()
        kind = {"low_pass": 0, "high_pass": 1, None: -1, "none": -1}[filter]
        self.cfg = RlmConfig(from_rate, to_rate, channels, int(span_len or 0), kind, freq, q, max_sources,
                             max_in_frames, frames_per_lane, ring_stages, no_balance, force_general)
        self.cfg.filter_first = 1 if filter_first else 0  # `mixer.add(src.low_pass(f))`: filter at from_rate, then convert (rodio_hip.h)
        self._h = C.c_void_p()
        check(lib.rh_rlm_create(C.byref(self._h), C.byref(self.cfg)), "rh_rlm_create")
        self.channels = channels
        self._keep = None
        self._keep_prev = None
        self.out_frames = 0

    def geometry(self):
        g = RlmGeometry()
        check(lib.rh_rlm_geometry(self._h, C.byref(g)), "rh_rlm_geometry")
        return {n: getattr(g, n) for n, _ in RlmGeometry._fields_}

    def set_sources(self, tensors):
        n = len(tensors)
        self._keep = list(tensors)
        ptrs = (C.c_void_p * n)(*[t.data_ptr() if t.numel() else None for t in tensors])
        frames = (C.c_uint64 * n)(*[t.numel() // self.channels for t in tensors])
        check(lib.rh_rlm_set_sources(self._h, ptrs, frames, n), "rh_rlm_set_sources")
        # run() reports out_frames too; it is needed here to size the output allocation
        mx = 0
        for t in tensors:
            o = C.c_uint64(0)
            check(lib.rh_resample_out_frames(t.numel() // self.channels, self.cfg.from_rate, self.cfg.to_rate,
                                             self.channels, self.cfg.span_len, C.byref(o)), "rh_resample_out_frames")
            mx = max(mx, o.value)
        self.out_frames = mx

    def run(self, out=None):
        """Enqueue one pass on the current stream; returns the mixed [out_frames*channels] tensor."""
        if out is None:
            out = _dev_empty(max(self.out_frames * self.channels, 4))
        m = C.c_uint64(0)
        check(lib.rh_rlm_run(self._h, _ptr(out), out.numel() // self.channels, C.byref(m), _stream()), "rh_rlm_run")
        return out[: m.value * self.channels]

    def set_gains(self, gains):
        """Per-source Amplify factors (`src.amplify(g)` / Player::set_volume): folded into the fused kernel."""
        g = np.ascontiguousarray(gains, dtype=np.float32)
        check(lib.rh_rlm_set_gains(self._h, g.ctypes.data_as(_lib.f32p), g.size), "rh_rlm_set_gains")

    def set_filters(self, filters):
        """A filter per source, rodio's `mixer.add(a.low_pass(200)); mixer.add(b.high_pass(300)); mixer.add(c)`: a list of
        ("low_pass" | "high_pass" | None, freq[, q = 0.5]) entries, one per source; [] returns to the handle's one filter.
        Before set_sources()."""
        n = len(filters)
        kinds = (C.c_int32 * max(n, 1))()
        freqs = (C.c_uint32 * max(n, 1))()
        qs = (C.c_float * max(n, 1))()
        for i, f in enumerate(filters):
            kind = f[0] if f is not None else None
            kinds[i] = {"low_pass": 0, "high_pass": 1, None: -1, "none": -1}[kind]
            freqs[i] = int(f[1]) if kind not in (None, "none") else 0
            qs[i] = float(f[2]) if (kind not in (None, "none") and len(f) > 2) else 0.5
        check(lib.rh_rlm_set_filters(self._h, kinds, freqs, C.cast(qs, _lib.f32p), n), "rh_rlm_set_filters")

    def set_exclusive(self, exclusive=True):
        """False: other work shares the CUs while this handle runs (a collective on a second stream, copy launches): tiles by
        ticket instead of by workgroup index (rodio_hip.h)."""
        check(lib.rh_rlm_set_exclusive(self._h, 1 if exclusive else 0), "rh_rlm_set_exclusive")

    def set_mix_first(self, enable=True):
        """False: every source goes through the converter and the filter on its own (the general path)."""
        check(lib.rh_rlm_set_mix_first(self._h, 1 if enable else 0), "rh_rlm_set_mix_first")

    def run_subset(self, first, count, out=None):
        """Mix of the sources [first, first+count) only."""
        if out is None:
            out = _dev_empty(max(self.out_frames * self.channels, 4))
        m = C.c_uint64(0)
        check(lib.rh_rlm_run_subset(self._h, first, count, _ptr(out), out.numel() // self.channels, C.byref(m), _stream()), "rh_rlm_run_subset")
        return out[: m.value * self.channels]

    # -- block streaming: feed(blocks) returns the mixed frames that became computable ------------------
    def stream_begin(self, keep_history=False):
        """keep_history: stream_feed_v() may run on the summed state while its sources run together (rh_rlm_stream_keep_history): the
        mirror then keeps the rows of the block before alive, as the recovery needs them."""
        check(lib.rh_rlm_stream_begin(self._h), "rh_rlm_stream_begin")
        check(lib.rh_rlm_stream_keep_history(self._h, 1 if keep_history else 0), "rh_rlm_stream_keep_history")
        self._left = None
        self._keep_prev = None

    def stream_stats(self):
        """(blocks on the summed state, blocks with one state per source, recoveries) of the current stream."""
        a, b, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        check(lib.rh_rlm_stream_stats(self._h, C.byref(a), C.byref(b), C.byref(c)), "rh_rlm_stream_stats")
        return a.value, b.value, c.value

    def stream_feed(self, blocks, flush=False):
        """blocks: one device tensor of NEW interleaved samples per source (equal lengths).  The unconsumed
        tail of the previous call is kept here, as a Rust shim would keep it in its own buffers."""
        torch = _t()
        ch = self.channels
        bufs = list(blocks) if self._left is None else [torch.cat([l, b]) for l, b in zip(self._left, blocks)]
        n = len(bufs)
        avail = bufs[0].numel() // ch
        ptrs = (C.c_void_p * n)(*[b.data_ptr() if b.numel() else 0 for b in bufs])
        cap = int(avail * (self.cfg.to_rate / self.cfg.from_rate + 1)) + 64
        out = _dev_empty(max(cap * ch, 4))
        m, c = C.c_uint64(0), C.c_uint64(0)
        check(lib.rh_rlm_stream_block(self._h, ptrs, n, avail, int(flush), _ptr(out), cap, C.byref(m), C.byref(c), _stream()),
              "rh_rlm_stream_block")
        self._keep = bufs  # the launch reads them asynchronously
        self._left = [b[c.value * ch:].clone() for b in bufs]
        return out[: m.value * ch]

    def stream_feed_v(self, blocks, ended):
        """Per-source states: blocks[s] = NEW interleaved samples of source s (any length, also empty), ended[s] = it
        will deliver no more.  Returns the mixed frames this block could emit (whole tiles until all have ended)."""
        torch = _t()
        ch = self.channels
        bufs = list(blocks) if self._left is None else [torch.cat([l, b]) if b.numel() else l for l, b in zip(self._left, blocks)]
        n = len(bufs)
        avail = (C.c_uint64 * n)(*[b.numel() // ch for b in bufs])
        end = (C.c_uint8 * n)(*[1 if e else 0 for e in ended])
        ptrs = (C.c_void_p * n)(*[b.data_ptr() if b.numel() else 0 for b in bufs])
        cap = int(max(avail) * (self.cfg.to_rate / self.cfg.from_rate + 1)) + 64
        out = _dev_empty(max(cap * ch, 4))
        m, c = C.c_uint64(0), C.c_uint64(0)
        check(lib.rh_rlm_stream_block_v(self._h, ptrs, avail, end, n, _ptr(out), cap, C.byref(m), C.byref(c), _stream()), "rh_rlm_stream_block_v")
        self._keep_prev = self._keep  # (keep_history: the next call may replay the end of this block's rows)
        self._keep = bufs  # the launch reads them asynchronously
        self._left = [b[min(c.value * ch, b.numel()):].clone() for b in bufs]
        return out[: m.value * ch]

    def run_batch(self):
        """No mixing: returns [S, out_frames*channels], row s = UniformSourceIterator(src_s).low_pass(...)."""
        torch = _t()
        S = len(self._keep)
        stride = (self.out_frames + 1) // 2 * 2
        out = torch.empty((S, stride * self.channels), device="cuda", dtype=torch.float32)
        m = C.c_uint64(0)
        check(lib.rh_rlm_run_batch(self._h, _ptr(out), stride, C.byref(m), _stream()), "rh_rlm_run_batch")
        return out[:, : m.value * self.channels]

    def autotune(self, out=None):
        """Time the candidate geometries on the current sources and keep the fastest (synchronises)."""
        if out is None:
            out = _dev_empty(max(self.out_frames * self.channels, 4))
        r, ns = C.c_uint32(0), C.c_uint32(0)
        check(lib.rh_rlm_autotune(self._h, _ptr(out), out.numel() // self.channels, _stream(), C.byref(r), C.byref(ns)), "rh_rlm_autotune")
        return r.value, ns.value

    def phase_cycles(self):
        out = (C.c_double * 8)()
        st = lib.rh_rlm_phase_cycles(self._h, out)
        return None if st != 0 else list(out)

    def late_carries(self):
        n = C.c_uint64(0)
        check(lib.rh_rlm_late_carries(self._h, C.byref(n)), "rh_rlm_late_carries")
        return n.value

    def check_status(self):
        _t().cuda.current_stream().synchronize()
        check(lib.rh_rlm_last_status(self._h), "rh_rlm_last_status")

    def close(self):
        if self._h:
            lib.rh_rlm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

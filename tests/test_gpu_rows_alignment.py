"""The one-row launches at rows that start ANYWHERE: the vector / tile kernels of round 6's last session (rh::map4, rh::ld4_at, k_resample_tile,
k_pcm_to_channels_tile, k_channel_volume_tile, the four-a-lane converters) read aligned 16-byte vectors around whatever pointer they are given
and store 16 bytes only where the row allows it -- a `GpuSource` chain hands them rows inside larger buffers (`buffer + offset`).  Every entry,
src and dst each shifted by 0..3 samples, odd lengths: the same bits as the call on rows of their own (which test_gpu_parity.py holds against the
oracle).  Through the C ABI."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(rh):
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    rh.init(0)
    return rh


def _run(G, call, x, out_len, so, do, dtype_out="float32", in_dtype=None):
    """call(dst_ptr, src_ptr) on src = buffer + so elements, dst = buffer + do elements; returns the out_len outputs and checks the guard zones."""
    import torch

    from rodio_amd import source

    xin = torch.from_numpy(x)
    isz = xin.element_size()
    src_buf = torch.zeros(x.size + 16, dtype=xin.dtype, device="cuda")
    src_buf[so: so + x.size] = xin.cuda()
    tdt = getattr(torch, dtype_out)
    dst_buf = torch.full((out_len + 24,), 7, dtype=tdt, device="cuda")
    osz = dst_buf.element_size()
    call(C.c_void_p(dst_buf.data_ptr() + (8 + do) * osz), C.c_void_p(src_buf.data_ptr() + so * isz), source._stream())
    torch.cuda.synchronize()
    h = dst_buf.cpu().numpy()
    assert np.all(h[: 8 + do] == 7) and np.all(h[8 + do + out_len:] == 7), "wrote outside its row"
    return h[8 + do: 8 + do + out_len].copy()


OFFS = [(0, 0), (1, 0), (0, 1), (3, 2), (2, 3), (1, 1)]


@pytest.mark.parametrize("n", [1, 5, 1023, 40003])
def test_elementwise_rows_anywhere(G, n):
    from rodio_amd import _lib

    lib = _lib.lib
    x = (np.random.default_rng(n).uniform(-1, 1, n)).astype(np.float32)
    D = 37
    cases = {
        "amplify": (n, lambda d, s, st: lib.rh_amplify(d, s, n, 0.37, st)),
        "distortion": (n, lambda d, s, st: lib.rh_distortion(d, s, n, 3.0, 0.6, st)),
        "dither": (n, lambda d, s, st: lib.rh_dither(d, s, n, 11, 2, 16, 1, 99, st)),
        "ramp": (n, lambda d, s, st: lib.rh_linear_gain_ramp(d, s, n, 5, 3, 48000, 200_000_000, 0.1, 0.9, 1, st)),
        "delay": (n + D, lambda d, s, st: lib.rh_delay(d, s, n, D, st)),
        "echo_mix": (n + D, lambda d, s, st: lib.rh_echo_mix(d, s, n, D, 0.7, st)),
        "echo_mix_long": (n + 4 * n + 2, lambda d, s, st: lib.rh_echo_mix(d, s, n, 4 * n + 2, 0.7, st)),  # a delay longer than the row (mix.rs:47-52: the gap is Delay's zeros)
    }
    for name, (m, call) in cases.items():
        outs = [_run(G, lambda d, s, st: _lib.check(call(d, s, st), name), x, m, so, do) for so, do in OFFS]
        for o in outs[1:]:
            assert np.array_equal(o.view(np.uint32), outs[0].view(np.uint32)), name
    # take_duration: the samples the duration admits, then the zeros that complete the frame
    for fade in (0, 1):
        m, e = C.c_uint64(0), C.c_int32(0)
        call = lambda d, s, st: lib.rh_take_duration(d, s, n, 3, 3, 48000, 150_000_000, fade, C.byref(m), C.byref(e), st)  # noqa: E731
        outs = []
        for so, do in OFFS:
            full = _run(G, lambda d, s, st: _lib.check(call(d, s, st), "take"), x, n + 3, so, do)
            outs.append(full[: m.value].copy())
            assert np.all(full[m.value:] == 7)
        for o in outs[1:]:
            assert np.array_equal(o.view(np.uint32), outs[0].view(np.uint32))


@pytest.mark.parametrize("frames", [1, 7, 2049, 30011])
def test_layout_rows_anywhere(G, frames):
    from rodio_amd import _lib

    lib = _lib.lib
    for frm_ch, to_ch in [(6, 2), (2, 6), (1, 2), (2, 1), (3, 5), (2, 2)]:
        x = (np.random.default_rng(frames + frm_ch).uniform(-1, 1, frames * frm_ch)).astype(np.float32)
        gains = np.linspace(0.2, 1.1, to_ch).astype(np.float32)
        m = C.c_uint64(0)
        cases = {
            "channels_convert": lambda d, s, st: lib.rh_channels_convert(d, s, frames, frm_ch, to_ch, st),
            "channel_volume": lambda d, s, st: lib.rh_channel_volume(d, s, frames, frm_ch, gains.ctypes.data_as(_lib.f32p), to_ch, st),
            "decode_channels_f32": lambda d, s, st: lib.rh_wav_decode_channels(d, s, frames * frm_ch - (1 if frm_ch > 1 else 0), frm_ch, 32, 1, to_ch, C.byref(m), st),
        }
        for name, call in cases.items():
            outs = [_run(G, lambda d, s, st: _lib.check(call(d, s, st), name), x, frames * to_ch, so, do) for so, do in OFFS]
            for o in outs[1:]:
                assert np.array_equal(o.view(np.uint32), outs[0].view(np.uint32)), (name, frm_ch, to_ch)
    # PCM16 at odd BYTE addresses (file + data_offset): u8 buffer, offsets in bytes
    pcm = np.random.default_rng(5).integers(-32768, 32768, frames * 6, dtype=np.int64).astype("<i2")
    raw = pcm.view(np.uint8)
    m = C.c_uint64(0)
    outs = []
    for so in (0, 1, 2, 3, 5, 15):
        outs.append(_run(G, lambda d, s, st: _lib.check(lib.rh_wav_decode_channels(d, s, frames * 6, 6, 16, 0, 2, C.byref(m), st), "pcm"), raw, frames * 2, so, so % 4))
        outs.append(_run(G, lambda d, s, st: _lib.check(lib.rh_wav_decode(d, s, frames * 6, 6, 16, 0, C.byref(m), st), "pcm"), raw, frames * 6, so, so % 4).reshape(-1, 6)[:, :2].reshape(-1))
    want = (pcm.astype(np.float32) / np.float32(32768)).reshape(-1, 6)[:, :2].reshape(-1)
    for o in outs:
        assert np.array_equal(o.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("frm,to,ch,span", [(44100, 48000, 2, 0), (44100, 48000, 1, 0), (48000, 44100, 6, 0), (8000, 48000, 3, 0), (44100, 48000, 2, 96), (48000, 8000, 2, 32768)])
def test_resampler_rows_anywhere(G, frm, to, ch, span):
    from rodio_amd import _lib

    lib = _lib.lib
    for frames in (1, 2, 50, 20011):
        if span and frames * ch < span:
            continue
        x = (np.random.default_rng(frames).uniform(-1, 1, frames * ch)).astype(np.float32)
        m = C.c_uint64(0)
        _lib.check(lib.rh_resample_out_frames(frames, frm, to, ch, span, C.byref(m)), "out_frames")
        outs = [_run(G, lambda d, s, st: _lib.check(lib.rh_resample_linear(d, s, frames, frm, to, ch, span, st), "resample"), x, m.value * ch, so, do) for so, do in OFFS]
        for o in outs[1:]:
            assert np.array_equal(o.view(np.uint32), outs[0].view(np.uint32)), (frames, frm, to, ch, span)


def test_converter_rows_anywhere(G):
    from rodio_amd import _lib

    lib = _lib.lib
    n = 10007
    x = (np.random.default_rng(3).uniform(-1.2, 1.2, n)).astype(np.float32)
    for fn, dt in [("rh_convert_f32_to_i16", "int16"), ("rh_convert_f32_to_i32", "int32"), ("rh_convert_f32_to_u8", "uint8"), ("rh_convert_f32_to_f64", "float64"), ("rh_convert_f32_to_i64", "int64")]:
        outs = [_run(G, lambda d, s, st: _lib.check(getattr(lib, fn)(d, s, n, st), fn), x, n, so, do, dtype_out=dt) for so, do in OFFS]
        for o in outs[1:]:
            assert np.array_equal(o, outs[0]), fn
    for fn, src in [("rh_convert_i16_to_f32", np.random.default_rng(4).integers(-32768, 32768, n).astype(np.int16)), ("rh_convert_i32_to_f32", np.random.default_rng(4).integers(-2 ** 31, 2 ** 31, n).astype(np.int32)),
                    ("rh_convert_f64_to_f32", np.random.default_rng(4).uniform(-1, 1, n)), ("rh_convert_u8_to_f32", np.random.default_rng(4).integers(0, 256, n).astype(np.uint8))]:
        outs = [_run(G, lambda d, s, st: _lib.check(getattr(lib, fn)(d, s, n, st), fn), src, n, so, do) for so, do in OFFS]
        for o in outs[1:]:
            assert np.array_equal(o.view(np.uint32), outs[0].view(np.uint32)), fn

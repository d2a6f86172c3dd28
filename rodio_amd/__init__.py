"""rodio_amd -- MI355X (gfx950) implementation of rodio's per-sample DSP hot path.

The product is the C-ABI shared library `librodio_hip.so` (include/rodio_hip.h).  This Python
package is the host-side mirror of rodio's adapter interface over that ABI, used by the parity
tests and the benchmark; device memory comes from PyTorch-ROCm (plumbing only).

Importing the package loads the HIP library and FAILS if it is missing -- there is no CPU path.
"""
from . import _lib
from ._lib import LIB_PATH, RhError, lib
from .source import (  # noqa: F401
    ChannelCountConverter,
    ChannelVolume,
    GpuSource,
    Mixer,
    ResampleLowpassMix,
    SampleRateConverter,
    SampleTypeConverter,
    SamplesBuffer,
    SpanSource,
    StreamingResampler,
    StreamingReverb,
    Spatial,
    TestSource,
    UniformSourceIterator,
    WavDecoder,
    WavDecoderChannels,
    agc_batch,
    async_status,
    agc_state,
    biquad_batch,
    limit_batch,
    biquad_coeffs,
    delay_samples,
    filter_scan_ok,
    init,
    reverb_spatial_batch,
    spatial_gains,
    wav_probe,
    wav_to_bytes,
    spatial_gains_batch,
)

__all__ = [n for n in dir() if not n.startswith("_")]

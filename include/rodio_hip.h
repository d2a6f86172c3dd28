/* rodio_hip.h -- C ABI of the MI355X (gfx950) implementation of rodio's per-sample DSP hot path.
 *
 * rodio has no FFI: its extension point is the Rust trait `Source: Iterator<Item = f32>`
 * (/root/reference/src/source/mod.rs:179-218).  This header is the boundary a Rust shim
 * (`struct GpuSource<I: Source>` / `GpuMixer`) binds with `extern "C"`:
 * the shim pre-pulls a block of samples from the upstream iterator, hands it to one of the
 * block functions below and serves `next()` from the returned block.  Every entry point cites
 * the reference iterator it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - Audio is interleaved f32 (`Sample = f32`, src/common.rs:36,48; src/source/mod.rs:131-135).
 *   - All data pointers are DEVICE pointers owned by the caller unless the name ends in _host.
 *     Small parameter arrays (gains, coefficients) are HOST pointers and are copied by value.
 *   - Every function enqueues on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and returns an rh_status; nothing throws or aborts.  End of stream (`None`) is
 *     expressed through returned lengths, never through an error.
 *   - One handle is used from one thread at a time (mirrors `Send + !Sync`).
 *   - There is NO CPU fallback: without a usable HIP device rh_init() fails and every other
 *     call returns RH_ERR_NOT_INITIALIZED / RH_ERR_HIP.
 */
#ifndef RODIO_HIP_H
#define RODIO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t rh_status;
enum {
    RH_OK = 0,
    RH_ERR_INVALID = 1,         /* bad argument (zero rate/channels: rodio's NonZero types) */
    RH_ERR_HIP = 2,             /* a HIP runtime call failed; see rh_last_hip_error() */
    RH_ERR_UNSUPPORTED = 3,     /* valid in rodio, not covered by this kernel set */
    RH_ERR_NOMEM = 4,
    RH_ERR_TIMEOUT = 5,         /* an in-kernel bounded wait expired (fused pipeline) */
    RH_ERR_NOT_INITIALIZED = 6,
    RH_ERR_CAPACITY = 7         /* output buffer too small */
};

typedef void *rh_stream; /* hipStream_t */

/* ---- runtime ------------------------------------------------------------------------- */
int32_t rh_version(void);
const char *rh_status_string(rh_status s);
const char *rh_last_hip_error(void);
/* Binds the library to a gfx950 device.  Also (re-)reads the library's diagnostic / tuning environment variables (DESIGN.md
 * 7.1): nothing reads the environment on a call's way to a launch. */
rh_status rh_init(int32_t device);
rh_status rh_device_name(char *buf, size_t cap);
/* Makes the device rh_init() bound current on the CALLING thread.  HIP's current device is per thread: a helper thread of the host
 * (the reaper that frees retired streams, the pool that pulls sources) calls this once before its first call into the library, so
 * that what it frees, records or synchronises belongs to the right device in a multi-GPU process. */
rh_status rh_bind_thread(void);
/* The handle-less time-parallel kernels (rh_limit, rh_biquad mode 1) wait for hand-offs between their tiles with a bound.
 * A wait that expires (never seen on a healthy device) poisons the tile with NaN and sets a sticky word on the device:
 * RH_ERR_TIMEOUT here, once, then RH_OK again.  Covers the launches that have COMPLETED (synchronise the stream or an event
 * recorded behind them first); does not wait for anything itself.  (Handles have their own: rh_rlm_last_status.)  One word per
 * device (the one rh_init last bound), shared by every chain on it: the call that reads it first gets the report. */
rh_status rh_async_status(void);
rh_status rh_malloc(void **out, size_t bytes);
rh_status rh_free(void *p);
rh_status rh_memset(void *p, int32_t value, size_t bytes, rh_stream stream);
rh_status rh_memcpy_h2d(void *dst, const void *src_host, size_t bytes, rh_stream stream);
/* `rows` rows of `width_bytes` each, `pitch_bytes` apart on both sides: a block of staged rows without the unused ends. */
rh_status rh_memcpy_h2d_rows(void *dst, const void *src_host, size_t pitch_bytes, size_t width_bytes, size_t rows, rh_stream stream);
rh_status rh_memcpy_d2h(void *dst_host, const void *src, size_t bytes, rh_stream stream);
/* For a pull-model shim (include/rodio_hip.hpp): page-locked host blocks, copies that do not synchronise. */
rh_status rh_memcpy_d2h_async(void *dst_host, const void *src, size_t bytes, rh_stream stream);
rh_status rh_memcpy_d2d(void *dst, const void *src, size_t bytes, rh_stream stream);
rh_status rh_host_alloc(void **out, size_t bytes);
rh_status rh_host_free(void *p);
rh_status rh_stream_create(rh_stream *out);
rh_status rh_stream_destroy(rh_stream s);
rh_status rh_stream_synchronize(rh_stream s);
/* The scan kernels (rh_limit, rh_biquad mode 1, rh_agc) keep a scratch buffer per stream, grown on demand and reused by every
 * later launch on that stream; rh_stream_destroy frees it for the library's own streams.  For a stream of the caller's (a
 * PyTorch stream, ...) that is about to be destroyed: synchronises it and frees its buffer (a handle value the runtime
 * recycles must not inherit one). */
rh_status rh_stream_release_scratch(rh_stream s);
/* HIP-event timing on `stream` (bench.py measures the kernels on the stream they run on). */
rh_status rh_event_create(void **out);
rh_status rh_event_destroy(void *ev);
rh_status rh_event_record(void *ev, rh_stream stream);
rh_status rh_event_synchronize(void *ev);
/* Work queued on `stream` after this call starts once the work recorded by `ev` has completed (no host wait). */
rh_status rh_stream_wait_event(rh_stream stream, void *ev);
rh_status rh_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on stop */

/* ---- SampleTypeConverter ("DataConverter") ---------------------------------------------
 * src/conversions/sample.rs:42-44 -> dasp_sample 0.11.0 `ToSample` (Cargo.lock:317-318).
 * Call sites: src/decoder/wav.rs:119-136, src/stream.rs:538-545 (egress).  Bit-exact. */
rh_status rh_convert_i8_to_f32(float *dst, const int8_t *src, size_t n, rh_stream stream);
rh_status rh_convert_u8_to_f32(float *dst, const uint8_t *src, size_t n, rh_stream stream);
rh_status rh_convert_i16_to_f32(float *dst, const int16_t *src, size_t n, rh_stream stream);
rh_status rh_convert_u16_to_f32(float *dst, const uint16_t *src, size_t n, rh_stream stream);
rh_status rh_convert_i24_to_f32(float *dst, const int32_t *src, size_t n, rh_stream stream); /* I24 in i32 */
rh_status rh_convert_i32_to_f32(float *dst, const int32_t *src, size_t n, rh_stream stream);
rh_status rh_convert_f32_to_i8(int8_t *dst, const float *src, size_t n, rh_stream stream);
rh_status rh_convert_f32_to_i16(int16_t *dst, const float *src, size_t n, rh_stream stream);
rh_status rh_convert_f32_to_u16(uint16_t *dst, const float *src, size_t n, rh_stream stream);
rh_status rh_convert_f32_to_i32(int32_t *dst, const float *src, size_t n, rh_stream stream);
/* The rest of cpal's device formats (src/stream.rs:555-568 egress, src/microphone.rs:280-291 ingress).
 * I24 / U24 travel in 32-bit containers like cpal's (unchecked: f32 1.0 -> I24 8388608); unsigned
 * formats go through the signed one (dasp `iN::to_uN`).  64-bit formats lose nothing that f32 has. */
rh_status rh_convert_f32_to_u8(uint8_t *dst, const float *src, size_t n, rh_stream stream);
rh_status rh_convert_f32_to_i24(int32_t *dst, const float *src, size_t n, rh_stream stream);
rh_status rh_convert_f32_to_u24(int32_t *dst, const float *src, size_t n, rh_stream stream);
rh_status rh_convert_f32_to_u32(uint32_t *dst, const float *src, size_t n, rh_stream stream);
rh_status rh_convert_f32_to_i64(int64_t *dst, const float *src, size_t n, rh_stream stream);
rh_status rh_convert_f32_to_u64(uint64_t *dst, const float *src, size_t n, rh_stream stream);
rh_status rh_convert_f32_to_f64(double *dst, const float *src, size_t n, rh_stream stream);
rh_status rh_convert_u24_to_f32(float *dst, const int32_t *src, size_t n, rh_stream stream);
rh_status rh_convert_u32_to_f32(float *dst, const uint32_t *src, size_t n, rh_stream stream);
rh_status rh_convert_i64_to_f32(float *dst, const int64_t *src, size_t n, rh_stream stream);
rh_status rh_convert_u64_to_f32(float *dst, const uint64_t *src, size_t n, rh_stream stream);
rh_status rh_convert_f64_to_f32(float *dst, const double *src, size_t n, rh_stream stream);

/* ---- WAV / PCM either side of the path (src/decoder/wav.rs:94-172 ingest, src/wav_output.rs:62-96 egress).
 * probe (host, no GPU): walks the RIFF chunks of a file image.  decode: `data` is a DEVICE copy of the data
 * chunk (8-bit unsigned, 16/24/32-bit signed LE or 32-bit float) AT ANY BYTE ADDRESS -- `file + data_offset` of an uploaded file image
 * is fine: the chunks in front of `data` decide where it starts (the reference's own assets/audacity32bit_int.wav: 32-bit samples at
 * offset 102; tests/wav_test.rs) and hound reads a byte stream; dst receives *out_samples f32 samples --
 * n_samples plus the silence that completes a cut frame (wav.rs:161-169).  header (host): the 44-byte
 * 32-bit-float header wav_to_writer produces, for the whole frames of n_samples (wav_output.rs:98-140);
 * the payload is the f32 block itself.  Returns the bytes written (0 = does not fit). */
typedef struct rh_wav_info {
    uint32_t channels, sample_rate, bits_per_sample;
    int32_t is_float;
    uint64_t data_offset, data_bytes, samples;
} rh_wav_info;
rh_status rh_wav_probe_host(const uint8_t *bytes, size_t size, rh_wav_info *info);
rh_status rh_wav_decode(float *dst, const uint8_t *data, uint64_t n_samples, uint32_t channels,
                        uint32_t bits_per_sample, int32_t is_float, uint64_t *out_samples, rh_stream stream);
/* rh_wav_decode and ChannelCountConverter(channels -> to_channels) (src/conversions/channels.rs:57-85) in ONE pass: what
 * `UniformSourceIterator::new(decoder, to_channels, same rate)` makes of the file (uniform.rs:58-67 -- BASELINE config 5's chain).  dst
 * receives *out_samples = ceil(n_samples / channels) * to_channels samples, bit-identical to rh_wav_decode followed by
 * rh_channels_convert; the decoded block in between (4 bytes a sample written and read again) never exists. */
rh_status rh_wav_decode_channels(float *dst, const uint8_t *data, uint64_t n_samples, uint32_t channels,
                                 uint32_t bits_per_sample, int32_t is_float, uint32_t to_channels,
                                 uint64_t *out_samples, rh_stream stream);
size_t rh_wav_header_f32_host(uint8_t *out, size_t cap, uint32_t channels, uint32_t sample_rate, uint64_t n_samples);

/* ---- ChannelCountConverter: src/conversions/channels.rs:57-85.  Bit-exact.
 * dst holds frames*to_ch samples. */
rh_status rh_channels_convert(float *dst, const float *src, size_t frames, uint32_t from_ch,
                              uint32_t to_ch, rh_stream stream);

/* ---- Amplify: src/source/amplify.rs:64 */
rh_status rh_amplify(float *dst, const float *src, size_t n, float factor, rh_stream stream);

/* ---- Distortion: src/source/distortion.rs:66-72.  threshold < 0 or NaN: RH_ERR_INVALID (f32::clamp panics). */
rh_status rh_distortion(float *dst, const float *src, size_t n, float gain, float threshold, rh_stream stream);
/* Dither: src/source/dither.rs:217-242.  out = x - noise * lsb with lsb = 1 / 2^(target_bits-1) (:180).
 * algorithm follows the reference's enum order (dither.rs:40-69): 0 GPDF (normal, sigma 0.6), 1 HighPass (blue:
 * white[k] - white[k-channels]), 2 RPDF (uniform [-1,1]), 3 TPDF (triangular (-1,1), the default).
 * The reference seeds a SmallRng from system entropy, so its samples are not reproducible; here the noise of
 * sample k = sample_offset + i is a pure function of (seed, k):
 *   h = mix(seed ^ mix(k + 1)), mix = the splitmix64 finaliser; u1 = (int(h >> 40) - 2^23) / 2^23,
 *   u2 = (int((h >> 16) & 0xffffff) - 2^23) / 2^23; TPDF (u1+u2)/2, RPDF u1, GPDF Box-Muller on the same fields.
 * Stateless: any split of a stream into blocks (with the running sample_offset) equals one pass. */
rh_status rh_dither(float *dst, const float *src, size_t n, uint64_t sample_offset, uint32_t channels,
                    uint32_t target_bits, int32_t algorithm, uint64_t seed, rh_stream stream);
/* ---- LinearGainRamp, fade_in (0 -> 1, then 1.0), fade_out (1 -> 0, clamp_end): src/source/linear_ramp.rs:79-110,
 * fadein.rs:11-13, fadeout.rs:13.  Stateless: sample_offset = samples of the stream already processed. */
rh_status rh_linear_gain_ramp(float *dst, const float *src, size_t n, uint64_t sample_offset, uint32_t channels,
                              uint32_t sample_rate, uint64_t duration_ns, float start_gain, float end_gain,
                              int32_t clamp_end, rh_stream stream);

/* ---- Delay: src/source/delay.rs:8-16,68-75: rh_delay_samples() zeros, then the n input samples (dst holds both). */
rh_status rh_delay(float *dst, const float *src, uint64_t n, uint64_t delay_samples, rh_stream stream);
/* ---- TakeDuration (+ its fade-out filter): src/source/take.rs:96-148.  src holds n samples of the stream that
 * start at sample_offset; dst (capacity n + channels) receives *out_samples: the samples the duration
 * still admits, then the zeros that complete a cut frame.  *ended = 1 when the duration expires in this
 * block (rodio's None follows).  An upstream that ends first simply stops calling. */
rh_status rh_take_duration(float *dst, const float *src, uint64_t n, uint64_t sample_offset, uint32_t channels,
                           uint32_t sample_rate, uint64_t duration_ns, int32_t fade_out,
                           uint64_t *out_samples, int32_t *ended, rh_stream stream);
/* ... the same from ANY state of the adapter (after `try_seek`, take.rs:222-231: remaining = requested.saturating_sub(pos), the frame
 * position 0): remaining_ns = what is left of the duration at src[0], requested_ns = the adapter's whole duration (the fade-out
 * filter's denominator, :33-38), frame_phase = samples of the current frame already emitted (decides the silence that completes
 * a cut frame, :107-115).  *remaining_after_ns (may be NULL) = what is left behind the samples taken. */
rh_status rh_take_duration_from(float *dst, const float *src, uint64_t n, uint64_t remaining_ns, uint64_t requested_ns,
                                uint32_t frame_phase, uint32_t channels, uint32_t sample_rate, int32_t fade_out,
                                uint64_t *out_samples, int32_t *ended, uint64_t *remaining_after_ns, rh_stream stream);

/* ---- ChannelVolume / Spatial: src/source/channel_volume.rs:71-88, src/source/spatial.rs:48-69.
 * gains_host has out_ch entries (out_ch <= 16).  dst holds frames*out_ch samples. */
rh_status rh_channel_volume(float *dst, const float *src, size_t frames, uint32_t in_ch,
                            const float *gains_host, uint32_t out_ch, rh_stream stream);
/* Host-side: gains[2] from emitter / ear positions (spatial.rs:48-69). */
rh_status rh_spatial_gains(const float emitter[3], const float left_ear[3],
                           const float right_ear[3], float out_gains[2]);

/* ---- reverb = Mix(x, Delay(Amplify(x))): src/source/mod.rs:628-634, delay.rs:8-16,68-75,
 * mix.rs:43-53.  delay_samples counts INTERLEAVED samples (delay.rs:14).  dst holds
 * n + delay_samples samples. */
uint64_t rh_delay_samples(uint64_t delay_ns, uint32_t sample_rate, uint32_t channels);
rh_status rh_echo_mix(float *dst, const float *src, size_t n, size_t delay_samples, float gain,
                      rh_stream stream);

/* ---- fused, batched reverb -> Spatial (BASELINE config 3): for every stream s < n_streams
 *        Spatial(reverb(x_s, delay, gain), ...) = ChannelVolume(reverb(x_s), gains[s])
 * Replaces source/mod.rs:628-634 + channel_volume.rs:71-88 in one pass over the input.  x_s = src +
 * s*src_stride holds n interleaved STEREO samples; row s of dst (dst + s*dst_stride) receives
 * 2*floor((n + delay_samples)/2) samples.  gains_dev: DEVICE array [n_streams][2] (rh_spatial_gains
 * per stream).  Bit-exact with the two-step reference chain. */
rh_status rh_reverb_spatial(float *dst, const float *src, size_t n, size_t delay_samples, float gain,
                            const float *gains_dev, uint32_t n_streams, size_t src_stride,
                            size_t dst_stride, rh_stream stream);

/* ---- SampleRateConverter (+ UniformSourceIterator span chunking):
 * src/conversions/sample_rate.rs:52-90,110-122,131-201, src/math.rs:23-26,
 * src/source/uniform.rs:50-97.  span_len = 0 means current_span_len() == None; otherwise the
 * converter restarts every min(span_len, 32768) SAMPLES like uniform.rs:56.
 * Bit-exact with the reference's lerp (mul, IEEE divide, add; no FMA). */
rh_status rh_resample_out_frames(uint64_t in_frames, uint32_t from_rate, uint32_t to_rate,
                                 uint32_t channels, uint64_t span_len, uint64_t *out_frames);
rh_status rh_resample_linear(float *dst, const float *src, uint64_t in_frames, uint32_t from_rate,
                             uint32_t to_rate, uint32_t channels, uint64_t span_len,
                             rh_stream stream);

/* ---- UniformSourceIterator span by span: src/source/uniform.rs:50-97.  rodio re-builds its converter chain
 *        Take{n: current_span_len().min(32768)} -> SampleRateConverter(from_rate -> to_rate, from_ch) -> ChannelCountConverter(from_ch -> to_ch)
 * whenever the current one runs dry, so a source that reports spans (SamplesBuffer buffer.rs:76-82, Buffered buffered.rs:109,
 * the decoders symphonia.rs:199-201) is converted span by span: each span starts a fresh converter and ends with its last frame
 * verbatim (sample_rate.rs:193-200); rate and layout may change from span to span.  Spans are independent work items, and so
 * are the pieces a span is cut into when it arrives in blocks.  A SEGMENT = output frames [m0, m1) (span-relative) of one span:
 *   src          device pointer to input frame `src_frame0` of the span (src_frames frames of from_ch samples are there)
 *   dst          device pointer: output frame m0 lands at dst[0 .. to_ch)
 *   span_frames  input frames of the WHOLE span once it is complete -- its last frame is then emitted verbatim and nothing
 *                follows; UINT64_MAX while the span is still open (more input will come, or current_span_len() == None)
 * Output frame m reads input frames floor(m*F/T) and +1 (F/T = from/to reduced); the caller keeps the frames the next
 * segment's first tap needs (rh_uniform_first_tap).  Bit-exact with the reference (lerp as mul, IEEE divide, add).
 * A span that ends INSIDE a frame -- uniform.rs:56's `.min(32768)` cuts frames of 3, 5, 6, 7 channels, and a source may return None inside
 * a frame -- is what source/mod.rs:196-200 asks sources not to produce, and rodio plays it anyway: the SampleRateConverter meets a
 * short frame, every output frame that lerps towards it is cut to its length (zip, sample_rate.rs:174-179), the short frame itself
 * comes out verbatim when an output lands on it (:193-200), and the ChannelCountConverter behind regroups those runs into frames of
 * from_ch samples (channels.rs:57-85).  The TAIL of such a span is a segment of its own, marked by reserved = t (the cut frame's
 * samples, 0 < t < from_ch):
 *   span_frames  q, the WHOLE frames of the span
 *   src          device pointer to input frame src_frame0 (= q - src_frames, src_frames in {0, 1}: the frame in front of the cut is
 *                passed whenever an output frame lerps towards the cut frame), followed by the t samples of the cut frame
 *   m0, m1       output SAMPLES [m0, m1) of the tail, 0 <= m0 <= m1 <= rh_uniform_cut_tail_samples(); dst = where sample m0 lands
 * The whole frames in front of the tail convert as the frames of an open span (span_frames = UINT64_MAX: no verbatim last frame).
 * What follows a cut span starts at the sample behind the cut: the caller's stream is a stream of samples, not of frames.
 * rh_uniform_span_frames: output frames computable from the first span_in_frames input frames of a span; complete != 0 adds
 * the verbatim last frame.  rh_uniform_segments validates and launches a host table (any number of segments, sources, formats
 * in one call); the _dev form takes the table from DEVICE memory unvalidated (it can travel in the caller's staging copy);
 * max_out_frames = the largest m1 - m0 in it. */
typedef struct rh_uniform_seg {
    const float *src;
    float *dst;
    uint64_t src_frame0, src_frames;
    uint64_t m0, m1;
    uint64_t span_frames;
    uint32_t from_rate, to_rate;
    uint32_t from_ch, to_ch;
    float gain;          /* Amplify in FRONT of the converter (mixer.add(src.amplify(g)), amplify.rs:64): both taps are scaled
                          * before the lerp, which is the reference's order of operations; 1.0 = none (x * 1.0 == x) */
    uint32_t reserved;   /* 0; t > 0 marks the tail segment of a span that ends t samples into a frame (see above) */
} rh_uniform_seg;
rh_status rh_uniform_span_frames(uint64_t span_in_frames, uint32_t from_rate, uint32_t to_rate, int32_t complete,
                                 uint64_t *out_frames);
rh_status rh_uniform_first_tap(uint64_t out_frame, uint32_t from_rate, uint32_t to_rate, uint64_t *in_frame);
/* Output samples of the tail of a span of span_whole_frames whole frames + tail_samples samples (0 < tail_samples < from_ch). */
rh_status rh_uniform_cut_tail_samples(uint64_t span_whole_frames, uint32_t tail_samples, uint32_t from_rate, uint32_t to_rate,
                                      uint32_t from_ch, uint32_t to_ch, uint64_t *out_samples);
rh_status rh_uniform_segments(const rh_uniform_seg *segs_host, uint32_t n_segs, rh_stream stream);
rh_status rh_uniform_segments_dev(const rh_uniform_seg *segs_dev, uint32_t n_segs, uint64_t max_out_frames,
                                  rh_stream stream);

/* ---- block streaming: the same two adapters when the stream arrives in blocks (what a `Source` shim
 * does: pull a block upstream, process, serve next() from it).  The handle keeps what the reference's
 * iterator keeps between samples; ANY split of a stream into blocks gives the bits of one pass.
 * (rh_biquad / rh_limit / rh_agc carry their state through their `state` argument.)
 *
 * SampleRateConverter, continuous stream (current_span_len() == None; a spanned source is converted span
 * by span with rh_resample_linear, uniform.rs:56-67).  process() consumes in_frames new frames and emits
 * every output frame whose two taps have arrived; flush != 0 marks the end of the stream (rodio's None):
 * the last input frame is then emitted verbatim (sample_rate.rs:193-200).  pending_frames() tells the
 * caller how large dst must be. */
typedef struct rh_resampler rh_resampler;
rh_status rh_resampler_create(rh_resampler **out, uint32_t from_rate, uint32_t to_rate, uint32_t channels);
rh_status rh_resampler_reset(rh_resampler *p);
rh_status rh_resampler_destroy(rh_resampler *p);
rh_status rh_resampler_pending_frames(rh_resampler *p, uint64_t in_frames, int32_t flush, uint64_t *out_frames);
rh_status rh_resampler_process(rh_resampler *p, float *dst, uint64_t dst_capacity_frames, const float *src,
                               uint64_t in_frames, int32_t flush, uint64_t *out_frames, rh_stream stream);
/* reverb (delay_samples interleaved samples, gain): process() maps n samples to n samples; after the last
 * block flush() emits the delay_samples samples of the delayed clone that outlive the source. */
typedef struct rh_echo rh_echo;
rh_status rh_echo_create(rh_echo **out, uint64_t delay_samples, float gain);
rh_status rh_echo_reset(rh_echo *p);
rh_status rh_echo_destroy(rh_echo *p);
rh_status rh_echo_process(rh_echo *p, float *dst, const float *src, uint64_t n, rh_stream stream);
rh_status rh_echo_flush(rh_echo *p, float *dst, rh_stream stream);

/* ---- Mixer: src/mixer.rs:58-66,120-136,175-198.  out[t] = ((0+v0[t])+v1[t])+... over the live
 * sources in insertion order (bit-identical rounding sequence).  Source s contributes samples
 * [start[s], start[s]+len[s]) (start = frame-aligned admission, mixer.rs:175-183).
 * srcs_host / start_host / len_host are host arrays of n_sources entries; dst holds out_len
 * samples = max(start+len). */
rh_status rh_mix_sum(float *dst, size_t out_len, const float *const *srcs_host,
                     const uint64_t *start_host, const uint64_t *len_host, uint32_t n_sources,
                     rh_stream stream);

/* ---- A block of a mixer of ANY channel count in ONE launch (mixer::mixer(nz!(6), rate): src/mixer.rs:25,185-198), for continuous
 * sources (current_span_len() == None) of any rate and layout: per source Amplify (amplify.rs:64) -> SampleRateConverter
 * (sample_rate.rs:131-201) -> ChannelCountConverter (channels.rs:57-85), i.e. UniformSourceIterator (uniform.rs:58-67), and the
 * ordered sum.  Bit-identical to the chain of stand-alone calls (rh_amplify, rh_resample_linear / rh_uniform_segments,
 * rh_channels_convert, rh_mix_sum) it replaces.  dst receives out_frames frames of `channels` floats: the frames m0 .. m0+out_frames-1
 * of the mix, whatever m0 is -- the caller describes, per source, where those frames lie in what it holds:
 *   data      device pointer to the source frame that holds the FIRST tap of output frame m0: frame floor(m0 F / T) of the source's
 *             stream (F / T = from_rate / to_rate reduced), `channels` interleaved floats a frame
 *   phase     (m0 F) mod T: where frame m0 lies between its taps
 *   frames    output frames of this block the source reaches (<= out_frames): it is silent behind them (it has ended).  Both taps of
 *             every one of them -- frames floor((phase + j F) / T) and the next, counted from data -- must be readable, except
 *   last      ... that a source which has ENDED passes the index (counted from data) of its LAST frame: an output frame whose first
 *             tap is that frame is emitted verbatim (sample_rate.rs:193-200) and its second tap is not read.  Live: UINT32_MAX.
 * 4-byte alignment is all the rows need.  srcs_host is a host array, read before the call returns.  include/rodio_hip.hpp
 * (GpuMixer, mixers of more than two channels) plans the blocks. */
typedef struct rh_wide_src {
    const float *data;
    uint32_t channels, from_rate;
    uint32_t phase;
    uint64_t frames;
    uint32_t last;
    float gain;
} rh_wide_src;
rh_status rh_wide_mix_block(float *dst, uint32_t channels, uint32_t to_rate, uint64_t out_frames,
                            const rh_wide_src *srcs_host, uint32_t n_sources, rh_stream stream);

/* ---- BltFilter (low_pass / high_pass): src/source/blt.rs:502-544,558-560,397-492.
 * kind: 0 = low_pass, 1 = high_pass.  coeffs5 = {b0,b1,b2,a1,a2} (already divided by a0).
 * state (optional device pointer, 4*channels floats {x1,x2,y1,y2} per channel) carries the
 * filter across blocks; NULL = zero state, not written back.
 * mode 0 = sequential per (source,channel) stream, same op order as blt.rs:559 (bit-exact);
 * mode 1 = time-parallel scan (<=1e-5 abs): 1 to 8 channels, the same state as mode 0 (a stream may
 *          change modes between blocks).  A call the scan kernel does not take -- rows that do not start on 16-byte boundaries
 *          (dst, src, and frames*channels % 4 == 0 for n_streams > 1), more than 8 channels, a filter that does not forget
 *          within 64 tiles, dst == src (the scan reads the two frames in front of every share after a neighbour may have
 *          overwritten them) -- runs in mode 0 instead: never an error, the exact bits, slower.
 * The batch form filters n_streams equally shaped blocks laid out back to back.  Mode 0 works in place. */
rh_status rh_biquad_coeffs(int32_t kind, uint32_t freq, float q, uint32_t sample_rate,
                           float out_coeffs5[5]);
rh_status rh_biquad(float *dst, const float *src, uint64_t frames, uint32_t channels,
                    uint32_t n_streams, const float coeffs5_host[5], float *state, int32_t mode,
                    rh_stream stream);
/* THE FILTER CONTRACT.  Every time-parallel evaluation of a low_pass / high_pass in this library (rh_biquad mode 1, the fused rh_rlm_*
 * path) is CLOSER to the exact response than rodio's own f32 recurrence is, so its distance from rodio is rodio's rounding noise -- which
 * the recurrence amplifies by ~1 / (1 - r)^2 for poles of radius r.  Returns 1 where that distance stays <= 1e-5 for a FULL-SCALE source
 * (|x| <= 1; it scales with the peak): low_pass with 1 - r >= 0.0125 (>= 100 Hz at 48 kHz), high_pass with 1 - r >= 0.075 (>= 600 Hz at
 * 48 kHz); measured table: profiles/r04_filter_contract.txt.  0 outside: a drop-in for rodio's samples then takes rh_biquad mode 0 (the
 * reference's order, bit for bit) -- include/rodio_hip.hpp does so on its own (GpuSource: per filter; GpuMixer: the source gets a chain
 * amplify -> uniform -> filter in mode 0 of its own and enters the mixer unfiltered). */
int32_t rh_filter_scan_ok(int32_t kind, uint32_t freq, float q, uint32_t sample_rate);

/* ---- src/math.rs:51-56,86-90,110-113 (host): dB <-> linear as the reference spells them (2^(dB*0.05*log2 10),
 * log2(x)*log10(2)*20) and the smoothing coefficient exp(-1/(seconds*rate)) of the limiter and the AGC.
 * Amplify::set_log_factor / Source::amplify_decibel (amplify.rs:33-35) is rh_amplify with rh_db_to_linear(dB). */
float rh_db_to_linear(float decibels);
float rh_linear_to_db(float linear);
float rh_duration_to_coefficient(uint64_t duration_ns, uint32_t sample_rate);

/* ---- Limit: src/source/limit.rs:94-130,853-988.  state: 2*channels floats
 * {integrator, peak} per channel (optional).  Works in place (dst == src).  A hand-off that expires inside the
 * kernel poisons the output with NaN and is reported by rh_async_status().
 * Parity is by TOLERANCE only (<= 1e-5 abs, tested): the time-parallel kernel evaluates log2 / exp2 with the device's instructions, composes the
 * integrator and peak recurrences as max-affine maps and, inside a run, spells them mul + FMA where limit.rs:909-913 has mul, mul, add (one
 * rounding fewer per step).  RH_LIMIT_SEQ=1 takes the one-lane-per-stream kernel in the reference's operation order throughout. */
typedef struct rh_limit_params {
    float threshold_db; /* LimitSettings::threshold  (default -1) */
    float knee_width_db;/* LimitSettings::knee_width (default 4)  */
    uint64_t attack_ns; /* default 5 ms   */
    uint64_t release_ns;/* default 100 ms */
} rh_limit_params;
rh_status rh_limit(float *dst, const float *src, uint64_t frames, uint32_t channels,
                   uint32_t sample_rate, uint32_t n_streams, const rh_limit_params *params_host,
                   float *state, rh_stream stream);

/* ---- AutomaticGainControl: src/source/agc.rs:133-171,397-504.  One state for all interleaved
 * channels of a stream.  state (optional): rh_agc_state_floats() floats per stream.
 * NaN / Inf input: as the reference (agc.rs:150,406,422-426,453-457, derived in tests/golden/derive_traces.py): the sample comes out NaN, the
 * window sum and the peak level stay NaN, both `> 0.0` tests fail from then on and the gain climbs to absolute_max_gain and stays there. */
typedef struct rh_agc_params {
    float target_level;      /* default 1.0 */
    uint64_t attack_ns;      /* default 4 s; clamped to 10 s like source/mod.rs:432-433 */
    uint64_t release_ns;     /* default 0 */
    float absolute_max_gain; /* default 7.0 */
    float floor;             /* default 0.0 */
} rh_agc_params;
size_t rh_agc_state_floats(void);
/* Resets n_streams states to a fresh AGC (gain 1.0, empty RMS window): agc.rs:209-236. */
rh_status rh_agc_state_init(float *state, uint32_t n_streams, rh_stream stream);
rh_status rh_agc(float *dst, const float *src, uint64_t n_samples, uint32_t sample_rate,
                 uint32_t n_streams, const rh_agc_params *params_host, float *state,
                 rh_stream stream);

/* ---- fused pipeline (BASELINE config 2): for every source
 *        mixer.add(UniformSourceIterator::new(src, ch, to_rate).low_pass(freq))
 *      then the ordered mixer sum -- one kernel, each input byte read once.
 * Replaces uniform.rs:78-97 + sample_rate.rs:131-201 + blt.rs:397-451 + mixer.rs:185-198. */
typedef struct rh_rlm_config {
    uint32_t from_rate, to_rate;
    uint32_t channels;     /* 1 (mono) or 2 (stereo): the frames of every source and of the mix (other layouts: rh_channels_convert / rh_uniform_segments in front) */
    uint64_t span_len;     /* 0 = None; else chunk of min(span_len, 32768) samples */
    int32_t filter_kind;   /* 0 = low_pass, 1 = high_pass, -1 = no filter, 2 = custom_coeffs; from_rate == to_rate: the converter passes through */
    uint32_t filter_freq;
    float filter_q;        /* rodio's low_pass() uses 0.5 (blt.rs:11-16) */
    uint32_t max_sources;
    uint64_t max_in_frames;
    uint32_t frames_per_lane; /* output frames per lane of a 64-lane tile; 0 = auto */
    uint32_t ring_stages;     /* LDS stages of the source prefetch ring (2..4); 0 = auto */
    uint32_t no_balance;      /* diagnostics: 1 = do not pad the LDS request to even out waves per CU */
    uint32_t force_general;   /* diagnostics/tests: 1 = use the ragged-batch kernel even for equal lengths */
    float custom_coeffs[5];   /* filter_kind 2: {b0,b1,b2,a1,a2}, already divided by a0 */
    uint32_t filter_first;    /* 1 = `mixer.add(src.low_pass(f))`: the filter runs on every source BEFORE the converter, at from_rate (source/mod.rs:255-275:
                               * a filter takes its sample rate from its input; mixer.rs:58-66 converts what it is given).  0 = the benchmark's spelling,
                               * `mixer.add(UniformSourceIterator::new(src, ..).low_pass(f))`: convert, then filter at to_rate.  With 1, one-shot runs of
                               * equal-length batches (sum the sources, filter that one stream, convert it: three linear stages, compared at 1e-5);
                               * rh_rlm_run on sources of different lengths and the rh_rlm_stream_* calls return RH_ERR_UNSUPPORTED. */
} rh_rlm_config;
typedef struct rh_rlm rh_rlm;
rh_status rh_rlm_create(rh_rlm **out, const rh_rlm_config *cfg);
rh_status rh_rlm_destroy(rh_rlm *p);
/* srcs_host[s] = device pointer of source s ([in_frames_host[s]][channels] f32, 16-byte
 * aligned).  All sources start at mixer time 0.  dst holds out_capacity_frames*channels
 * samples; *out_frames = max over sources of the resampled length. */
rh_status rh_rlm_set_sources(rh_rlm *p, const float *const *srcs_host,
                             const uint64_t *in_frames_host, uint32_t n_sources);
/* Optional per-source Amplify factors (src/source/amplify.rs:64; Player::set_volume): source s contributes
 * gains[s] * (its converted, filtered stream).  The chain is linear, so it does not matter where in it rodio
 * applies the factor; without a filter the result stays bit-identical to amplify-then-convert.  Sources beyond
 * n keep 1.0.  Takes effect for the sources that are set and for every later set_sources / stream block. */
rh_status rh_rlm_set_gains(rh_rlm *p, const float *gains_host, uint32_t n);
/* A filter PER SOURCE: `mixer.add(a.low_pass(200)); mixer.add(b.high_pass(300)); mixer.add(c)` -- in rodio every source carries its own
 * adapters into Mixer::add (src/source/mod.rs:686-721, src/mixer.rs:58-66).  kinds/freqs/qs are host arrays of n entries (kind as
 * rh_rlm_config.filter_kind: -1 none, 0 low_pass, 1 high_pass; q 0.5 is rodio's low_pass()/high_pass()); sources beyond n keep the
 * handle's own filter; n == 0 returns to it for all.  Sources with the same (kind, freq, q) form a class; every class runs as a fused
 * launch of its own -- summed at the input rate first where its sources share a length, the bit-exact ordered sum
 * where it has no filter -- and the classes' mixes are added in order of first appearance (f32; <= 1e-5 from rodio's per-sample order
 * like every filtered path).  Call before rh_rlm_set_sources (which deals the sources over the classes); rh_rlm_set_gains may follow
 * either.  One-shot runs (rh_rlm_run, rh_rlm_autotune); rh_rlm_run_subset / _batch and the stream entries return RH_ERR_UNSUPPORTED on
 * such a handle (a streaming host keeps one handle per filter: include/rodio_hip.hpp GpuMixer::add(src, gain, filter)). */
rh_status rh_rlm_set_filters(rh_rlm *p, const int32_t *kinds_host, const uint32_t *freqs_host, const float *qs_host, uint32_t n);
/* May the launches of this handle assume that they have the device to themselves?  exclusive != 0 (the default: a one-shot job on
 * its own): a launch whose tiles are all resident at once numbers them by workgroup index.  exclusive == 0: other work shares the
 * CUs while the handle runs -- a collective on a second stream (the N > 1 ranks of bench.py: the all-reduce of block k overlaps the
 * kernel of block k+1), copy launches (GpuMixer), another process -- and tiles are handed out by ticket, which needs neither full
 * residency nor in-order dispatch: a tile only ever waits for tiles that already hold a wave slot.  Same results either way. */
rh_status rh_rlm_set_exclusive(rh_rlm *p, int32_t exclusive);
/* "Mix first" (DESIGN.md 4.6): filtered sources of one length and one filter are summed at the input rate and converted and
 * filtered once (the converter and the filter are linear).  enable == 0 keeps every source on its own through the converter and
 * the filter (what a batch of per-source filters or lengths takes anyway); results agree within the 1e-5 of the filtered path.
 * Takes effect from the next run / stream block. */
rh_status rh_rlm_set_mix_first(rh_rlm *p, int32_t enable);
rh_status rh_rlm_run(rh_rlm *p, float *dst, uint64_t out_capacity_frames, uint64_t *out_frames,
                     rh_stream stream);
/* The same over the sources [first, first+count) only (a sub-mix; count = 1: one filtered stream). */
rh_status rh_rlm_run_subset(rh_rlm *p, uint32_t first, uint32_t count, float *dst,
                            uint64_t out_capacity_frames, uint64_t *out_frames, rh_stream stream);
/* Block streaming of the fused path: the same sources arrive block by block (what a `GpuMixer` shim does
 * with its upstream iterators).  begin() starts a stream; every block() call passes, per source, a device
 * pointer to `avail_frames` input frames that start where the previous call's *consumed_frames left off
 * (the caller keeps the unconsumed frames in front of the new ones), and receives *out_frames mixed frames
 * in dst.  flush != 0 ends the stream (rodio's None): the remaining frames and the verbatim last frame are
 * emitted.  Across blocks the handle carries the converter's position and the summed filter state; the
 * concatenated output equals one rh_rlm_run over the whole stream to f32 rounding (<= 1e-6 in the tests).
 * *consumed_frames is always a whole number of 16-byte vectors (a multiple of 2 stereo / 4 mono frames): a caller whose sources are
 * RESIDENT in device memory passes `row + consumed so far` as the next block's pointer -- no staging copy at all (bench.py --config stream).
 * cfg.span_len != 0: every source reports spans of that many samples (a SamplesBuffer longer than 32 768 samples: any
 * span_len >= 32768): the converter restarts every min(span_len, 32768) samples at the same frames of every source
 * (uniform.rs:56-67), each span's last frame verbatim.  Sources with other span patterns: rh_uniform_segments in front of a
 * pass-through stream (from_rate == to_rate), which is what include/rodio_hip.hpp does.  One set of sources per stream. */
rh_status rh_rlm_stream_begin(rh_rlm *p);
rh_status rh_rlm_stream_block(rh_rlm *p, const float *const *srcs_host, uint32_t n_sources,
                              uint64_t avail_frames, int32_t flush, float *dst,
                              uint64_t out_capacity_frames, uint64_t *out_frames,
                              uint64_t *consumed_frames, rh_stream stream);
/* The same for sources that end at different times (what a mixer usually holds): source s passes
 * avail_frames_host[s] frames; ended_host[s] != 0 says that these are its last ones (rodio's None; sticky).
 * All sources run on one clock (they start with the stream).  Live sources bound what a block can emit (whole
 * tiles of 64*frames_per_lane output frames; everything once all sources have ended, which ends the stream);
 * *consumed_frames is common to all sources: each drops min(consumed, what it holds).  The handle keeps one
 * filter state PER SOURCE across blocks (this entry always takes the ragged-batch kernel).  begin() as above;
 * a stream uses one of the two block entries throughout. */
rh_status rh_rlm_stream_block_v(rh_rlm *p, const float *const *srcs_host, const uint64_t *avail_frames_host,
                                const uint8_t *ended_host, uint32_t n_sources, float *dst,
                                uint64_t out_capacity_frames, uint64_t *out_frames,
                                uint64_t *consumed_frames, rh_stream stream);
/* rh_rlm_stream_block_v, sources that RUN TOGETHER: while every source is live and passes the same number of frames -- what a mixer's
 * sources do from the moment they are added until the first of them ends -- their filter states need not be told apart: the stream
 * carries their SUM (as rh_rlm_stream_block does) and every block is summed at the input rate first and converted and filtered once
 * (DESIGN.md 4.6).  When a source ends or falls behind, the states of the others are recovered from the rows of the block BEFORE
 * (a replay of its last few tiles through the per-source kernel: a stable filter has forgotten what lies further back) and the stream
 * goes on with one state per source -- until its sources run together AGAIN: every source either gone (ended, and everything it had
 * emitted) or live with the same frames as the other live ones.  Then the sum of the live states becomes the stream's summed state and
 * the blocks are summed first once more (a block with a state per source costs a wave per tile walking every source: ~360 us for 256
 * sources whatever the block's length; a mixer whose sounds end one after the other would otherwise spend its life there).
 * The caller opts in by promising what the recovery needs: on != 0 = "the rows I pass to a block
 * stay valid and unchanged until the work of the NEXT block call has run" (three row sets in rotation do: include/rodio_hip.hpp).
 * Between rh_rlm_stream_begin and the stream's first block.  Without it -- the default -- every block takes the per-source kernel. */
rh_status rh_rlm_stream_keep_history(rh_rlm *p, int32_t on);
/* Blocks side by side.  A block of a stream on the summed state is ONE launch where the library can make it one (k_rlm_sblk: the sum over
 * the sources, the conversion and the filter in one kernel), and a short kernel spends a third of its life filling and draining the chip.
 * on != 0 = "the rows I pass to a block call are COMPLETE in device memory when I make the call, and nothing I queued on `stream` since the
 * block in front is something this block has to wait for" (decoded assets resident in HBM; a caller that has synchronised its producer).
 * The library then launches a block WITHOUT a barrier behind the block in front, on the same stream (hipExtAnyOrderLaunch): its workgroups
 * start on every XCD that has finished the block in front while the others still work on it, and the few tiles the stream's filter state
 * still reaches wait for it inside the kernel (tagged words, left by the last tile of the block in front).  Nothing changes for the output:
 * a block's dst is complete in the order of the stream for every later launch that carries a barrier -- every launch but these.  Off (the
 * default: rows that a copy on the caller's stream is still filling) every block starts behind everything in front of it.
 * rh_rlm_set_exclusive(0) switches it off (the blocks' workgroups must fit the chip at once). */
rh_status rh_rlm_stream_overlap(rh_rlm *p, int32_t on);
/* Diagnostics: blocks of the current stream that ran as one launch. */
rh_status rh_rlm_stream_one_launch_blocks(rh_rlm *p, uint32_t *blocks);
/* Diagnostics: ... and how many of those started while the block in front still ran (rh_rlm_stream_overlap). */
rh_status rh_rlm_stream_overlapped_blocks(rh_rlm *p, uint32_t *blocks);
/* Diagnostics: blocks of the current stream that ran on the summed state / on one state per source, and recoveries in between. */
rh_status rh_rlm_stream_stats(rh_rlm *p, uint32_t *summed_blocks, uint32_t *per_source_blocks, uint32_t *recoveries);
/* No mixer: every source is converted and filtered into its own row, dst + s*dst_stride_frames*channels
 * (equal-length sources only: RH_ERR_UNSUPPORTED otherwise).  One launch for all sources. */
rh_status rh_rlm_run_batch(rh_rlm *p, float *dst, uint64_t dst_stride_frames, uint64_t *out_frames,
                           rh_stream stream);
/* Optional: time the candidate launch geometries of the equal-length kernel on the sources that are set
 * (a few runs each into dst, which is overwritten) and keep the fastest -- like a GEMM library's
 * find step.  Synchronises.  Works on whichever kernel the current sources take.  Reports the geometry kept. */
rh_status rh_rlm_autotune(rh_rlm *p, float *dst, uint64_t out_capacity_frames, rh_stream stream,
                          uint32_t *frames_per_lane, uint32_t *ring_stages);
/* After a synchronise: 0 if the last run completed, RH_ERR_TIMEOUT if a bounded wait expired. */
rh_status rh_rlm_last_status(rh_rlm *p);
/* Diagnostics: number of (tile, source) carries since the last call that had not been published by
 * the neighbouring tile when they were due and had to be polled (synchronises with the device). */
rh_status rh_rlm_late_carries(rh_rlm *p, uint64_t *count);
/* Diagnostics, only in builds with -DRH_PHASE_PROFILE (RH_ERR_UNSUPPORTED otherwise): mean shader
 * cycles per wave spent in each phase of the source loop during the last run. */
rh_status rh_rlm_phase_cycles(rh_rlm *p, double out8[8]);
/* Launch geometry chosen by create (for the bench's roofline report). */
typedef struct rh_rlm_geometry_info {
    uint32_t threads;          /* lanes per workgroup (one wave64 = one time tile) */
    uint32_t frames_per_lane;  /* tile = 64 * frames_per_lane output frames */
    uint32_t ring_stages;      /* sources prefetched ahead + 1 */
    uint32_t stage_kib;        /* KiB of LDS per stage */
    uint32_t lds_bytes;        /* dynamic LDS per workgroup */
    uint32_t lookback_tiles;   /* predecessor tiles whose aggregates form a tile's carry */
    uint32_t resident_waves_per_cu;
    uint32_t n_tiles;          /* grid of the last set_sources */
    uint32_t general_kernel;   /* 1: ragged-batch kernel (per-source carries), 0: equal-length kernel */
    uint32_t ragged_pair;      /* 1: one-shot runs of this batch take k_rlm_fast<RAG> (stable sources summed first, the pairs of ending sources behind them) instead of the ragged-batch kernel */
    uint32_t mix_first;        /* one-shot runs of this batch (filtered, equal lengths) sum the sources at the input rate first and convert + filter
                                * that one stream: 1 = two launches (k_mix_ring or k_mix_rows, then the fused kernel on one source),
                                * 2 = one (k_rlm_chunk: long stereo rows); 3 = a handle with a filter per source (rh_rlm_set_filters) whose classes all take
                                * k_rlm_chunk and whose LAST run walked them in one launch (k_rlm_chunk_multi), the classes' mixes added behind it */
} rh_rlm_geometry_info;
rh_status rh_rlm_geometry(rh_rlm *p, rh_rlm_geometry_info *info);

/* ---- multi-GPU: one process per GPU, sources sharded over the ranks, the mixer sum (src/mixer.rs:185-198 is the
 * only place rodio's streams meet) completed by ONE collective per mixed block over RCCL / xGMI.  Rank 0 calls
 * rh_comm_unique_id and hands the 128 bytes to the other ranks (any out-of-band channel); every rank then calls
 * rh_comm_init on its own device.  all-reduce leaves the full mix on every rank; reduce only on `root` (the
 * rank that owns the sink).  Both run in place on `stream`, behind the kernel that produced the block. */
typedef struct rh_comm rh_comm;
rh_status rh_comm_unique_id(uint8_t out128[128]);
rh_status rh_comm_init(rh_comm **out, int32_t rank, int32_t nranks, const uint8_t uid128[128]);
rh_status rh_comm_destroy(rh_comm *c);
rh_status rh_allreduce_sum_f32(rh_comm *c, float *buf, size_t n, rh_stream stream);
rh_status rh_reduce_sum_f32(rh_comm *c, float *buf, size_t n, int32_t root, rh_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* RODIO_HIP_H */

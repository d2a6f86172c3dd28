// rh_formats.hip -- the rest of cpal's device sample formats, either side of the path:
// egress  f32 -> {u8, I24, U24, u32, i64, u64, f64}   src/stream.rs:538-545,555-568 (`Sample::from_sample`)
// ingress {U24, u32, i64, u64, f64} -> f32            src/microphone.rs:280-291
// (i8/u8/i16/u16/I24/i32 -> f32 and f32 -> i8/i16/u16/i32 live in rh_elementwise.hip.)
// Arithmetic: dasp_sample 0.11.0 conv.rs (Cargo.lock:317-318; un-vendored, restated -- parity unpinned):
// float -> signed is `(s * 2^(bits-1)) as iN` (Rust `as`: truncate, saturate, NaN -> 0), unsigned goes
// THROUGH the signed type (`iN::to_uN`: add 2^(bits-1) with wrap-around), I24/U24 are unchecked i32
// containers, signed -> float is `s as f32 / 2^(bits-1)`.
// Streaming kernels, HBM-bound: one element per lane per grid-stride step (the 8-byte types move
// 512 B per wave-instruction).
#include "rh_common.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int32_t f32_as_i32(float v) {  // Rust `v as i32`
    if (v != v) return 0;
    if (v <= -2147483648.0f) return INT32_MIN;
    if (v >= 2147483648.0f) return INT32_MAX;
    return (int32_t)v;
}
__device__ __forceinline__ int64_t f32_as_i64(float v) {  // Rust `v as i64`
    if (v != v) return 0;
    if (v <= -9223372036854775808.0f) return INT64_MIN;
    if (v >= 9223372036854775808.0f) return INT64_MAX;
    return (int64_t)v;
}
__device__ __forceinline__ int32_t f32_as_i8(float v) {
    if (v != v) return 0;
    if (v <= -128.0f) return -128;
    if (v >= 128.0f) return 127;
    return (int32_t)v;
}

struct F32ToU8 { typedef float In; typedef uint8_t Out; static __device__ __forceinline__ Out cvt(In s) { return (uint8_t)(f32_as_i8(s * 128.0f) + 128); } };
struct F32ToI24 { typedef float In; typedef int32_t Out; static __device__ __forceinline__ Out cvt(In s) { return f32_as_i32(s * 8388608.0f); } };
struct F32ToU24 { typedef float In; typedef int32_t Out; static __device__ __forceinline__ Out cvt(In s) { return (int32_t)((uint32_t)f32_as_i32(s * 8388608.0f) + 8388608u); } };
struct F32ToU32 { typedef float In; typedef uint32_t Out; static __device__ __forceinline__ Out cvt(In s) { return (uint32_t)f32_as_i32(s * 2147483648.0f) + 0x80000000u; } };
struct F32ToI64 { typedef float In; typedef int64_t Out; static __device__ __forceinline__ Out cvt(In s) { return f32_as_i64(s * 9223372036854775808.0f); } };
struct F32ToU64 { typedef float In; typedef uint64_t Out; static __device__ __forceinline__ Out cvt(In s) { return (uint64_t)f32_as_i64(s * 9223372036854775808.0f) + 0x8000000000000000ull; } };
struct F32ToF64 { typedef float In; typedef double Out; static __device__ __forceinline__ Out cvt(In s) { return (double)s; } };
struct U24ToF32 { typedef int32_t In; typedef float Out; static __device__ __forceinline__ Out cvt(In s) { return (float)(s - 8388608) / 8388608.0f; } };
struct U32ToF32 { typedef uint32_t In; typedef float Out; static __device__ __forceinline__ Out cvt(In s) { return (float)(int32_t)(s - 0x80000000u) / 2147483648.0f; } };
struct I64ToF32 { typedef int64_t In; typedef float Out; static __device__ __forceinline__ Out cvt(In s) { return (float)s / 9223372036854775808.0f; } };
struct U64ToF32 { typedef uint64_t In; typedef float Out; static __device__ __forceinline__ Out cvt(In s) { return (float)(int64_t)(s - 0x8000000000000000ull) / 9223372036854775808.0f; } };
// dasp_sample 0.11.0 is not under /root/reference, so the i64 / u64 -> f32 formula above is restated from memory of the crate: ONE rounding,
// `s as f32 / 2^63`.  If the crate instead goes through f64 -- `(s as f64 / 2^63) as f32`, TWO roundings -- values such as 2^62 + 2^38 + 1 come out one
// ulp apart.  Not decidable offline; the other reading is kept behind RH_DASP_I64_VIA_F64=1 (read by rh_init) so that it can be
// flipped the day someone checks the crate: tests/test_gpu_parity.py runs both against the oracle's matching restatement.
struct I64ToF32ViaF64 { typedef int64_t In; typedef float Out; static __device__ __forceinline__ Out cvt(In s) { return (float)((double)s / 9223372036854775808.0); } };
struct U64ToF32ViaF64 { typedef uint64_t In; typedef float Out; static __device__ __forceinline__ Out cvt(In s) { return (float)((double)(int64_t)(s - 0x8000000000000000ull) / 9223372036854775808.0); } };
struct F64ToF32 { typedef double In; typedef float Out; static __device__ __forceinline__ Out cvt(In s) { return (float)s; } };

// Distortion: src/source/distortion.rs:66-72  (v = x*gain; v.clamp(-t, t); NaN stays NaN)
__global__ __launch_bounds__(kBlock) void k_distortion(float *__restrict__ dst, const float *__restrict__ src, size_t n, float gain, float threshold, int vec_ok) {
    rh::map4<kBlock>(dst, src, n, vec_ok, [=](size_t, float x) {
        float v = x * gain;
        v = v < -threshold ? -threshold : v;
        v = v > threshold ? threshold : v;
        return v;
    });
}
// Dither: src/source/dither.rs:217-242  out = x - noise * lsb, lsb = 1 / 2^(bits-1).  The reference draws the
// noise from a SmallRng seeded from system entropy (noise.rs:137,198,378,554), so no two runs of it agree; the
// contract here is a counter-based generator: the noise of sample k is a pure function of (seed, k), which makes
// the op stateless and any block split reproduce one pass.  Distributions as in noise.rs: uniform [-1,1] (:146),
// triangular (-1,1) mode 0 (:206), normal sigma 0.6 (:394), blue = white[k] - white[k - channels] per channel (:579).
__device__ __forceinline__ uint64_t dither_bits(uint64_t seed, uint64_t k) {
    auto mix = [](uint64_t z) {  // splitmix64 finaliser
        z ^= z >> 30;
        z *= 0xbf58476d1ce4e5b9ull;
        z ^= z >> 27;
        z *= 0x94d049bb133111ebull;
        z ^= z >> 31;
        return z;
    };
    return mix(seed ^ mix(k + 1));
}
__device__ __forceinline__ float dither_u1(uint64_t h) { return (float)((int32_t)(h >> 40) - 8388608) * 1.1920928955078125e-07f; }              // 24 bits -> [-1, 1)
__device__ __forceinline__ float dither_u2(uint64_t h) { return (float)((int32_t)((h >> 16) & 0xffffffu) - 8388608) * 1.1920928955078125e-07f; }
__global__ __launch_bounds__(kBlock) void k_dither(float *__restrict__ dst, const float *__restrict__ src, size_t n, uint64_t k0, uint32_t channels, float lsb, int32_t algorithm, uint64_t seed, int vec_ok) {
    rh::map4<kBlock>(dst, src, n, vec_ok, [=](size_t i, float x) {
        const uint64_t k = k0 + i;
        const uint64_t h = dither_bits(seed, k);
        float noise;
        if (algorithm == 3) {  // TPDF
            noise = (dither_u1(h) + dither_u2(h)) * 0.5f;
        } else if (algorithm == 2) {  // RPDF
            noise = dither_u1(h);
        } else if (algorithm == 1) {  // HighPass: the channel's previous white sample is the one a frame earlier
            const float prev = k >= channels ? dither_u1(dither_bits(seed, k - channels)) : 0.0f;
            noise = dither_u1(h) - prev;
        } else {  // GPDF: Box-Muller on the two 24-bit fields
            const float a = (float)((uint32_t)(h >> 40) + 1u) * 5.9604644775390625e-08f;       // (0, 1]
            const float b = (float)((uint32_t)(h >> 16) & 0xffffffu) * 5.9604644775390625e-08f;  // [0, 1)
            noise = sqrtf(-2.0f * logf(a)) * cosf(6.2831853071795864769f * b) * 0.6f;
        }
        return x - noise * lsb;
    });
}
// LinearGainRamp (fade_in / fade_out): src/source/linear_ramp.rs:79-110.  The iterator's `elapsed` is a
// pure function of the frame index while the ramp runs (f * (1e9 / rate) ns), so the op is stateless:
// sample k0+i of the stream is in frame (k0+i)/channels.
__device__ __forceinline__ float secs_f32(uint64_t ns) { return (float)(ns / 1000000000ull) + (float)(uint32_t)(ns % 1000000000ull) / 1000000000.0f; }
// (a lane's four samples: ONE 64-bit division finds the first one's frame, the others follow by counting channels)
__global__ __launch_bounds__(kBlock) void k_linear_gain_ramp(float *__restrict__ dst, const float *__restrict__ src, size_t n, uint64_t k0, uint32_t channels, uint64_t step_ns,
                                                             uint64_t done_frame, float total_s, float start_gain, float end_gain, float after, int vec_ok) {
    const size_t nvec = (n + 3) / 4, stride = (size_t)gridDim.x * kBlock;
    for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const size_t i = 4 * v;
        uint64_t frame = (k0 + i) / channels;
        uint32_t c = (uint32_t)((k0 + i) - frame * channels);
        auto factor_of = [&](uint64_t fr) {
            // elapsed >= total  <=>  frame >= ceil(total / step) (done_frame; never when the step is 0)
            if (step_ns != 0 && fr >= done_frame) return after;
            const float p = secs_f32(fr * step_ns) / total_s;
            return start_gain * (1.0f - p) + end_gain * p;
        };
        float factor = factor_of(frame);
        float x[4], y[4];
        const bool whole = i + 4 <= n, vec = (vec_ok & 2) && whole;  // (vec_ok: rh::rows_vec_bits)
        if (whole) {
            const float4 t = (vec_ok & 1) ? rh::ld_nt(reinterpret_cast<const float4 *>(src) + v) : rh::ld4_at(src, (int64_t)i, n);
            x[0] = t.x, x[1] = t.y, x[2] = t.z, x[3] = t.w;
        } else {
            for (int j = 0; j < 4; ++j) x[j] = i + j < n ? src[i + j] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            y[j] = x[j] * factor;
            if (++c == channels) c = 0, factor = factor_of(++frame);
        }
        if (vec) {
            rh::st_nt(reinterpret_cast<float4 *>(dst) + v, make_float4(y[0], y[1], y[2], y[3]));
        } else {
            for (int j = 0; j < 4; ++j)
                if (i + j < n) dst[i + j] = y[j];
        }
    }
}

// Delay: src/source/delay.rs:68-75 -- `delay` samples of silence, then the input.
__global__ __launch_bounds__(kBlock) void k_delay(float *__restrict__ dst, const float *__restrict__ src, uint64_t n, uint64_t delay, int vec_ok) {
    const uint64_t total = n + delay, nvec = (total + 3) / 4, stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const uint64_t i = 4 * v;
        const float4 x = rh::ld4_at(src, (int64_t)i - (int64_t)delay, n);  // 0.0 in front of the row: Delay's silence
        if (vec_ok && i + 4 <= total) {
            rh::st_nt(reinterpret_cast<float4 *>(dst) + v, x);
        } else {
            const float e[4] = {x.x, x.y, x.z, x.w};
            for (int j = 0; j < 4; ++j)
                if (i + j < total) dst[i + j] = e[j];
        }
    }
}
// TakeDuration: src/source/take.rs:96-148.  out[i] = x[i] (optionally * remaining_ms / total_ms, the
// fade-out filter of :33-38) for the `take` samples the duration admits, then `pad` zeros that complete
// the frame.  remaining at sample i of the block = rem0 - i * (1e9 / (rate*channels)).
__global__ __launch_bounds__(kBlock) void k_take_duration(float *__restrict__ dst, const float *__restrict__ src, uint64_t take, uint64_t pad, uint64_t rem0_ns, uint64_t dps_ns,
                                                          uint64_t requested_ns, int fade, int vec_ok) {
    const float total = (float)(requested_ns / 1000000ull);
    auto one = [=](size_t i, float x) {
        float v = x;
        if (fade) v = v * (float)((rem0_ns - i * dps_ns) / 1000000ull) / total;
        return v;
    };
    rh::map4<kBlock>(dst, src, (size_t)take, vec_ok, one);
    if (pad && blockIdx.x == 0 && threadIdx.x < pad) dst[take + threadIdx.x] = 0.0f;  // the zeros that complete a cut frame (fewer than `channels`)
    if (pad > kBlock && blockIdx.x == 0)
        for (uint64_t i = kBlock + threadIdx.x; i < pad; i += kBlock) dst[take + i] = 0.0f;
}

// E consecutive samples a lane, one load and one store, E such that the WIDER side is 16 bytes (four for the 4-byte formats, two where i64 / u64 /
// f64 are involved: a lane that stores 32 bytes fills half of every line a store instruction touches -- i16 -> f32, rh_elementwise.hip); a
// vector a lane (rh::grid_tiles).  A sample a lane under the capped grid-stride loop was 0.3-0.6 of 8 TB/s.
template <typename T, int E>
struct alignas(E * sizeof(T)) Pack {
    T e[E];
};
template <typename Op>
__global__ __launch_bounds__(kBlock) void k_convert(typename Op::Out *__restrict__ dst, const typename Op::In *__restrict__ src, size_t n, int vec_ok) {
    constexpr int E = 16 / (sizeof(typename Op::In) > sizeof(typename Op::Out) ? sizeof(typename Op::In) : sizeof(typename Op::Out));
    typedef Pack<typename Op::In, E> QI;
    typedef Pack<typename Op::Out, E> QO;
    const size_t nvec = (n + E - 1) / E, stride = (size_t)gridDim.x * kBlock;
    for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const size_t i = E * v;
        if (vec_ok && i + E <= n) {
            const QI x = reinterpret_cast<const QI *>(src)[v];
            QO y;
#pragma unroll
            for (int j = 0; j < E; ++j) y.e[j] = Op::cvt(x.e[j]);
            reinterpret_cast<QO *>(dst)[v] = y;
        } else {
            for (int j = 0; j < E; ++j)
                if (i + j < n) dst[i + j] = Op::cvt(src[i + j]);
        }
    }
}
template <typename Op>
rh_status launch(typename Op::Out *dst, const typename Op::In *src, size_t n, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (n == 0) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    constexpr int E = 16 / (sizeof(typename Op::In) > sizeof(typename Op::Out) ? sizeof(typename Op::In) : sizeof(typename Op::Out));
    const int vec_ok = reinterpret_cast<uintptr_t>(dst) % sizeof(Pack<typename Op::Out, E>) == 0 && reinterpret_cast<uintptr_t>(src) % sizeof(Pack<typename Op::In, E>) == 0;
    hipLaunchKernelGGL(k_convert<Op>, dim3(rh::grid_tiles((n + E - 1) / E)), dim3(kBlock), 0, rh::as_stream(stream), dst, src, n, vec_ok);
    RH_CHECK_LAUNCH();
    return RH_OK;
}

}  // namespace

extern "C" {
rh_status rh_distortion(float *dst, const float *src, size_t n, float gain, float threshold, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (!(threshold >= 0.0f)) return RH_ERR_INVALID;  // f32::clamp panics when min > max or NaN
    if (n == 0) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    hipLaunchKernelGGL(k_distortion, dim3(rh::grid_tiles((n + 3) / 4)), dim3(kBlock), 0, rh::as_stream(stream), dst, src, n, gain, threshold, rh::rows_vec_bits(dst, src));
    RH_CHECK_LAUNCH();
    return RH_OK;
}
rh_status rh_dither(float *dst, const float *src, size_t n, uint64_t sample_offset, uint32_t channels, uint32_t target_bits, int32_t algorithm, uint64_t seed, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (channels == 0 || target_bits == 0 || target_bits > 64 || algorithm < 0 || algorithm > 3) return RH_ERR_INVALID;  // BitDepth is NonZero (dither.rs:176)
    if (n == 0) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    const float lsb = (float)(1.0 / (double)(1ull << (target_bits - 1)));  // dither.rs:180
    hipLaunchKernelGGL(k_dither, dim3(rh::grid_tiles((n + 3) / 4)), dim3(kBlock), 0, rh::as_stream(stream), dst, src, n, sample_offset, channels, lsb, algorithm, seed, rh::rows_vec_bits(dst, src));
    RH_CHECK_LAUNCH();
    return RH_OK;
}
rh_status rh_linear_gain_ramp(float *dst, const float *src, size_t n, uint64_t sample_offset, uint32_t channels, uint32_t sample_rate, uint64_t duration_ns, float start_gain,
                              float end_gain, int32_t clamp_end, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (channels == 0 || sample_rate == 0 || duration_ns == 0) return RH_ERR_INVALID;  // linear_ramp.rs:34 asserts a non-zero duration
    if (n == 0) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    const uint64_t step_ns = 1000000000ull / sample_rate;  // linear_ramp.rs:98-100 (0 above 1 GHz: the ramp never advances)
    const float total_s = (float)(duration_ns / 1000000000ull) + (float)(uint32_t)(duration_ns % 1000000000ull) / 1000000000.0f;
    const uint64_t done_frame = step_ns ? (duration_ns + step_ns - 1) / step_ns : 0;  // elapsed >= total from this frame on
    hipLaunchKernelGGL(k_linear_gain_ramp, dim3(rh::grid_tiles((n + 3) / 4)), dim3(kBlock), 0, rh::as_stream(stream), dst, src, n, sample_offset, channels, step_ns, done_frame, total_s, start_gain,
                       end_gain, clamp_end ? end_gain : 1.0f, rh::rows_vec_bits(dst, src));
    RH_CHECK_LAUNCH();
    return RH_OK;
}
rh_status rh_delay(float *dst, const float *src, uint64_t n, uint64_t delay_samples, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (n + delay_samples == 0) return RH_OK;
    if (!dst || (n && !src)) return RH_ERR_INVALID;
    hipLaunchKernelGGL(k_delay, dim3(rh::grid_tiles((n + delay_samples + 3) / 4)), dim3(kBlock), 0, rh::as_stream(stream), dst, src, n, delay_samples, (int)((reinterpret_cast<uintptr_t>(dst) & 15u) == 0));
    RH_CHECK_LAUNCH();
    return RH_OK;
}
rh_status rh_take_duration_from(float *dst, const float *src, uint64_t n, uint64_t remaining_ns, uint64_t requested_ns, uint32_t frame_phase, uint32_t channels, uint32_t sample_rate,
                                int32_t fade_out, uint64_t *out_samples, int32_t *ended, uint64_t *remaining_after_ns, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (channels == 0 || sample_rate == 0 || !out_samples || frame_phase >= channels) return RH_ERR_INVALID;
    const uint64_t dps = 1000000000ull / ((uint64_t)sample_rate * channels);  // take.rs:63-67
    if (dps == 0) return RH_ERR_UNSUPPORTED;  // above 1 GHz*channel the reference never expires
    const uint64_t left = remaining_ns / dps;  // samples the remaining duration admits (:107: `remaining < duration_per_sample` ends it)
    const uint64_t take = n < left ? n : left;
    const bool expires_here = left <= n;       // the duration runs out inside this block (or exactly at its end)
    const uint64_t phase_end = (frame_phase + take) % channels;
    const uint64_t pad = expires_here && phase_end ? channels - phase_end : 0;  // :107-115
    *out_samples = take + pad;
    if (ended) *ended = expires_here ? 1 : 0;
    if (remaining_after_ns) *remaining_after_ns = remaining_ns - take * dps;
    if (take + pad == 0) return RH_OK;
    if (!dst || (take && !src)) return RH_ERR_INVALID;
    hipLaunchKernelGGL(k_take_duration, dim3(rh::grid_tiles((take + 3) / 4)), dim3(kBlock), 0, rh::as_stream(stream), dst, src, take, pad, remaining_ns, dps, requested_ns, fade_out ? 1 : 0, rh::rows_vec_bits(dst, src));
    RH_CHECK_LAUNCH();
    return RH_OK;
}
rh_status rh_take_duration(float *dst, const float *src, uint64_t n, uint64_t sample_offset, uint32_t channels, uint32_t sample_rate, uint64_t duration_ns, int32_t fade_out,
                           uint64_t *out_samples, int32_t *ended, rh_stream stream) {
    if (channels == 0 || sample_rate == 0) return RH_ERR_INVALID;
    const uint64_t dps = 1000000000ull / ((uint64_t)sample_rate * channels);
    if (dps == 0) {
        RH_REQUIRE_INIT();
        return RH_ERR_UNSUPPORTED;
    }
    const uint64_t K = duration_ns / dps;  // samples of the stream the duration admits
    const uint64_t done = sample_offset < K ? sample_offset : K;
    return rh_take_duration_from(dst, src, n, duration_ns - done * dps, duration_ns, (uint32_t)(done % channels), channels, sample_rate, fade_out, out_samples, ended, nullptr, stream);
}
rh_status rh_convert_f32_to_u8(uint8_t *dst, const float *src, size_t n, rh_stream s) { return launch<F32ToU8>(dst, src, n, s); }
rh_status rh_convert_f32_to_i24(int32_t *dst, const float *src, size_t n, rh_stream s) { return launch<F32ToI24>(dst, src, n, s); }
rh_status rh_convert_f32_to_u24(int32_t *dst, const float *src, size_t n, rh_stream s) { return launch<F32ToU24>(dst, src, n, s); }
rh_status rh_convert_f32_to_u32(uint32_t *dst, const float *src, size_t n, rh_stream s) { return launch<F32ToU32>(dst, src, n, s); }
rh_status rh_convert_f32_to_i64(int64_t *dst, const float *src, size_t n, rh_stream s) { return launch<F32ToI64>(dst, src, n, s); }
rh_status rh_convert_f32_to_u64(uint64_t *dst, const float *src, size_t n, rh_stream s) { return launch<F32ToU64>(dst, src, n, s); }
rh_status rh_convert_f32_to_f64(double *dst, const float *src, size_t n, rh_stream s) { return launch<F32ToF64>(dst, src, n, s); }
rh_status rh_convert_u24_to_f32(float *dst, const int32_t *src, size_t n, rh_stream s) { return launch<U24ToF32>(dst, src, n, s); }
rh_status rh_convert_u32_to_f32(float *dst, const uint32_t *src, size_t n, rh_stream s) { return launch<U32ToF32>(dst, src, n, s); }
rh_status rh_convert_i64_to_f32(float *dst, const int64_t *src, size_t n, rh_stream s) { return rh::knob(rh::K_DASP_I64_VIA_F64) ? launch<I64ToF32ViaF64>(dst, src, n, s) : launch<I64ToF32>(dst, src, n, s); }
rh_status rh_convert_u64_to_f32(float *dst, const uint64_t *src, size_t n, rh_stream s) { return rh::knob(rh::K_DASP_I64_VIA_F64) ? launch<U64ToF32ViaF64>(dst, src, n, s) : launch<U64ToF32>(dst, src, n, s); }
rh_status rh_convert_f64_to_f32(float *dst, const double *src, size_t n, rh_stream s) { return launch<F64ToF32>(dst, src, n, s); }
}

// rh_elementwise.hip -- the stateless, bit-exact block ops:
//   SampleTypeConverter  (src/conversions/sample.rs:42-44 -> dasp_sample 0.11.0)
//   ChannelCountConverter (src/conversions/channels.rs:57-85)
//   Amplify              (src/source/amplify.rs:64)
//   ChannelVolume        (src/source/channel_volume.rs:71-88)
//   reverb echo-mix      (src/source/mod.rs:628-634, delay.rs:68-75, mix.rs:43-53)
// All are HBM-bound streaming kernels: 16 B per lane per access where alignment allows,
// grid-stride over <= 2048 workgroups of 256 threads (4 waves of 64).
#include <cmath>

#include <cstdlib>
#include <type_traits>

#include "rh_common.h"

namespace {

constexpr int kBlock = 256;

// ------------------------------------------------------------ int -> f32 ----
template <typename T>
struct ToF32;
template <>
struct ToF32<int8_t> {
    static __device__ __forceinline__ float cvt(int8_t s) { return (float)s / 128.0f; }
};
template <>
struct ToF32<uint8_t> {  // u8 -> i8 (s - 128) -> f32
    static __device__ __forceinline__ float cvt(uint8_t s) { return (float)((int)s - 128) / 128.0f; }
};
template <>
struct ToF32<int16_t> {
    static __device__ __forceinline__ float cvt(int16_t s) { return (float)s / 32768.0f; }
};
template <>
struct ToF32<uint16_t> {  // u16 -> i16 (s - 32768) -> f32
    static __device__ __forceinline__ float cvt(uint16_t s) { return (float)((int)s - 32768) / 32768.0f; }
};
struct I24Tag {};
struct I32Tag {};

// 16 bytes of input per lane per iteration: VEC = 16 / sizeof(T) samples, VEC/4 float4 stores.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_int_to_f32(float *__restrict__ dst, const T *__restrict__ src, size_t n) {
    constexpr int VEC = 16 / sizeof(T);
    const size_t nvec = n / VEC;
    const size_t stride = (size_t)gridDim.x * kBlock;
    const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const uint4 *src4 = reinterpret_cast<const uint4 *>(src);
    auto emit = [&](size_t v, const uint4 &raw) {
        T vals[VEC];
        __builtin_memcpy(vals, &raw, 16);
        float out[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) out[k] = ToF32<T>::cvt(vals[k]);
        float4 *d4 = reinterpret_cast<float4 *>(dst + v * VEC);
#pragma unroll
        for (int k = 0; k < VEC / 4; ++k) d4[k] = make_float4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
    };
    size_t v = tid;
    for (; v + stride < nvec; v += 2 * stride) {  // two independent 16-byte loads in flight per lane
        const uint4 r0 = rh::ld_nt(src4 + v), r1 = rh::ld_nt(src4 + v + stride);
        emit(v, r0);
        emit(v + stride, r1);
    }
    if (v < nvec) emit(v, rh::ld_nt(src4 + v));
    for (size_t i = nvec * VEC + tid; i < n; i += stride) dst[i] = ToF32<T>::cvt(src[i]);
}

// 1- and 2-byte samples, the form whose STORES are whole lines: a lane takes four samples (4 or 8 bytes in, one 16-byte store), twice, 256
// vectors apart -- every store instruction of a wave writes 1 KiB in a row.  The kernel above gives a lane 16 bytes in and 32 (64) out: each of
// its stores fills half (a quarter) of every line it touches -- i16 -> f32 on 512 MiB: 0.73-0.75 of 8 TB/s against 0.805 (tools/bench_rows.py),
// which is BASELINE config 5's first step.  A workgroup owns 512 output vectors.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_int_to_f32_lines(float *__restrict__ dst, const T *__restrict__ src, size_t n) {
    static_assert(sizeof(T) <= 2, "one- and two-byte samples");
    typedef unsigned raw_t __attribute__((ext_vector_type(sizeof(T) == 2 ? 2 : 1)));
    const size_t nq = n / 4;  // groups of four samples
    const raw_t *s4 = reinterpret_cast<const raw_t *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    const size_t stride = (size_t)gridDim.x * (2 * kBlock);
    for (size_t base = (size_t)blockIdx.x * (2 * kBlock); base < nq; base += stride) {
        const size_t q0 = base + threadIdx.x, q1 = q0 + kBlock;
        raw_t r0 = {}, r1 = {};
        if (q0 < nq) r0 = __builtin_nontemporal_load(s4 + q0);
        if (q1 < nq) r1 = __builtin_nontemporal_load(s4 + q1);
        auto emit = [&](size_t q, raw_t raw) {
            T v[4];
            __builtin_memcpy(v, &raw, 4 * sizeof(T));
            rh::st_nt(d4 + q, make_float4(ToF32<T>::cvt(v[0]), ToF32<T>::cvt(v[1]), ToF32<T>::cvt(v[2]), ToF32<T>::cvt(v[3])));
        };
        if (q0 < nq) emit(q0, r0);
        if (q1 < nq) emit(q1, r1);
    }
    if (blockIdx.x == 0)
        for (size_t i = nq * 4 + threadIdx.x; i < n; i += kBlock) dst[i] = ToF32<T>::cvt(src[i]);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_int_to_f32_scalar(float *__restrict__ dst, const T *__restrict__ src, size_t n) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) dst[i] = ToF32<T>::cvt(src[i]);
}

// i32-carried samples: scale = 2^23 (I24) or 2^31 (i32); float4 in, float4 out.
__global__ __launch_bounds__(kBlock) void k_i32_to_f32(float *__restrict__ dst, const int32_t *__restrict__ src, size_t n, float scale, int vec_ok) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
    size_t done = 0;
    if (vec_ok) {
        const size_t nvec = n / 4;
        const int4 *s4 = reinterpret_cast<const int4 *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (size_t v = tid; v < nvec; v += stride) {
            int4 r = s4[v];
            d4[v] = make_float4((float)r.x / scale, (float)r.y / scale, (float)r.z / scale, (float)r.w / scale);
        }
        done = nvec * 4;
    }
    for (size_t i = done + tid; i < n; i += stride) dst[i] = (float)src[i] / scale;
}

// ------------------------------------------------------------ f32 -> int ----
// Rust `as`: truncate toward zero, saturate, NaN -> 0.
template <typename T>
__device__ __forceinline__ T sat_cast(float v, float lo, float hi_excl, T tmin, T tmax) {
    if (v != v) return (T)0;
    if (v <= lo) return tmin;
    if (v >= hi_excl) return tmax;
    return (T)(int)v;
}
template <typename T>
struct FromF32;
template <>
struct FromF32<int8_t> {
    static __device__ __forceinline__ int8_t cvt(float s) { return sat_cast<int8_t>(s * 128.0f, -128.0f, 128.0f, (int8_t)-128, (int8_t)127); }
};
template <>
struct FromF32<int16_t> {
    static __device__ __forceinline__ int16_t cvt(float s) { return sat_cast<int16_t>(s * 32768.0f, -32768.0f, 32768.0f, (int16_t)-32768, (int16_t)32767); }
};
template <>
struct FromF32<uint16_t> {  // f32 -> i16 -> u16 (s + 32768)
    static __device__ __forceinline__ uint16_t cvt(float s) { return (uint16_t)((int)FromF32<int16_t>::cvt(s) + 32768); }
};
template <>
struct FromF32<int32_t> {
    static __device__ __forceinline__ int32_t cvt(float s) {
        float v = s * 2147483648.0f;
        if (v != v) return 0;
        if (v <= -2147483648.0f) return INT32_MIN;
        if (v >= 2147483648.0f) return INT32_MAX;
        return (int32_t)v;
    }
};
template <typename T>
__global__ __launch_bounds__(kBlock) void k_f32_to_int(T *__restrict__ dst, const float *__restrict__ src, size_t n, int vec_ok) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
    size_t done = 0;
    if (vec_ok) {  // 4 samples per lane: float4 in, 4*sizeof(T) out
        const size_t nvec = n / 4;
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        for (size_t v = tid; v < nvec; v += stride) {
            float4 r = rh::ld_nt(s4 + v);
            T o[4] = {FromF32<T>::cvt(r.x), FromF32<T>::cvt(r.y), FromF32<T>::cvt(r.z), FromF32<T>::cvt(r.w)};
            __builtin_memcpy(dst + v * 4, o, sizeof(o));
        }
        done = nvec * 4;
    }
    for (size_t i = done + tid; i < n; i += stride) dst[i] = FromF32<T>::cvt(src[i]);
}

// ------------------------------------------------ ChannelCountConverter ----
// One lane per OUTPUT sample (coalesced stores); channels.rs:59-70 as a pure function of
// (frame, k): k < from -> in[k]; k == 1 -> the frame's first sample; else 0.0.
__global__ __launch_bounds__(kBlock) void k_channels_convert(float *__restrict__ dst, const float *__restrict__ src, size_t frames, uint32_t from, uint32_t to) {
    const size_t total = frames * to;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t o = (size_t)blockIdx.x * kBlock + threadIdx.x; o < total; o += stride) {
        const size_t f = o / to;
        const uint32_t k = (uint32_t)(o - f * to);
        float v;
        if (k < from) v = rh::ld_nt(src + f * from + k);
        else if (k == 1) v = rh::ld_nt(src + f * from);
        else v = 0.0f;
        rh::st_nt(dst + o, v);
    }
}
// Stereo <-> N fast paths are not needed for correctness; the generic kernel already writes
// coalesced and its reads hit each input line once.

// --------------------------------------------------------------- Amplify ----
__global__ __launch_bounds__(kBlock) void k_amplify(float *__restrict__ dst, const float *__restrict__ src, size_t n, float factor, int vec) {
    rh::map4<kBlock>(dst, src, n, vec, [=](size_t, float x) { return x * factor; });
}

// ---------------------------------------------------------- ChannelVolume ----
struct Gains {
    float g[16];
};
// One lane per frame: ordered channel sum from EQUILIBRIUM (0.0), / C_in, x gain[k].
__global__ __launch_bounds__(kBlock) void k_channel_volume(float *__restrict__ dst, const float *__restrict__ src, size_t frames, uint32_t in_ch, Gains gains, uint32_t out_ch) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t f = (size_t)blockIdx.x * kBlock + threadIdx.x; f < frames; f += stride) {
        float m = 0.0f;
        for (uint32_t c = 0; c < in_ch; ++c) m = m + src[f * in_ch + c];
        m = m / (float)in_ch;
        for (uint32_t k = 0; k < out_ch; ++k) dst[f * out_ch + k] = m * gains.g[k];
    }
}
// The same, a TILE of frames per workgroup (round 6; rh_wav.hip's k_pcm_to_channels_tile is the shape): the tile's samples come in once as
// aligned 16-byte vectors and are parked in LDS, a lane per frame takes the mean there (the reference's order: from 0.0, channel after
// channel, then / C_in), and a lane per four output SAMPLES multiplies by the channel's gain and stores 16 bytes.  A lane per frame that
// loops over both layouts measured 0.48 (6 -> 2) and 0.39 (2 -> 6) of 8 TB/s.
__global__ __launch_bounds__(kBlock) void k_channel_volume_tile(float *__restrict__ dst, const float *__restrict__ src, size_t frames, uint32_t in_ch, Gains gains, uint32_t out_ch,
                                                                 uint32_t tile_frames, uint32_t in_floats, int vec_ok) {
    extern __shared__ uint4 cv_tile[];
    float *lds = reinterpret_cast<float *>(cv_tile);
    float *means = lds + in_floats, *g = means + tile_frames;
    const size_t f0 = (size_t)blockIdx.x * tile_frames;  // (a multiple of 4 frames: the tile's first output sample starts a 16-byte vector)
    const uint32_t nf = (uint32_t)(frames - f0 < tile_frames ? frames - f0 : tile_frames);
    const uintptr_t p0 = reinterpret_cast<uintptr_t>(src + f0 * in_ch), a0 = p0 & ~(uintptr_t)15;
    const uint32_t shift = (uint32_t)(p0 - a0) / 4u, nvec = (shift + nf * in_ch + 3u) / 4u;
    for (uint32_t v = threadIdx.x; v < nvec; v += kBlock) cv_tile[v] = rh::ld_nt(reinterpret_cast<const uint4 *>(a0) + v);
    if (threadIdx.x < out_ch) g[threadIdx.x] = gains.g[threadIdx.x];
    __syncthreads();
    for (uint32_t f = threadIdx.x; f < nf; f += kBlock) {
        const float *x = lds + shift + f * in_ch;
        float m = 0.0f;
        for (uint32_t c = 0; c < in_ch; ++c) m = m + x[c];
        means[f] = m / (float)in_ch;
    }
    __syncthreads();
    const uint32_t total = nf * out_ch, nv = (total + 3u) / 4u;
    float *out = dst + f0 * out_ch;
    for (uint32_t v = threadIdx.x; v < nv; v += kBlock) {
        const uint32_t o0 = 4u * v;
        uint32_t f = o0 / out_ch, k = o0 - f * out_ch;
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            e[j] = o0 + j < total ? means[f] * g[k] : 0.0f;
            if (++k == out_ch) k = 0, ++f;
        }
        if (vec_ok && o0 + 4u <= total) {
            rh::st_nt(reinterpret_cast<float4 *>(out + o0), make_float4(e[0], e[1], e[2], e[3]));
        } else {
            for (int j = 0; j < 4; ++j)
                if (o0 + j < total) out[o0 + j] = e[j];
        }
    }
}
// Stereo in / stereo out (Spatial): one float2 load + one float2 store per lane.
__global__ __launch_bounds__(kBlock) void k_channel_volume_2x2(float *__restrict__ dst, const float *__restrict__ src, size_t frames, float g0, float g1, int vec_ok) {
    // two frames a lane (one 16-byte load, one 16-byte store), a vector a lane
    const size_t nvec = (frames + 1) / 2, stride = (size_t)gridDim.x * kBlock;
    auto one = [=](float l, float r, float &ol, float &orr) {
        float m = (0.0f + l) + r;
        m = m / 2.0f;
        ol = m * g0, orr = m * g1;
    };
    for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        if (vec_ok && 2 * v + 2 <= frames) {
            const float4 x = rh::ld_nt(reinterpret_cast<const float4 *>(src) + v);
            float4 y;
            one(x.x, x.y, y.x, y.y);
            one(x.z, x.w, y.z, y.w);
            rh::st_nt(reinterpret_cast<float4 *>(dst) + v, y);
        } else {
            for (size_t f = 2 * v; f < 2 * v + 2 && f < frames; ++f) one(src[2 * f], src[2 * f + 1], dst[2 * f], dst[2 * f + 1]);
        }
    }
}

// ------------------------------------------------------------ reverb mix ----
// y[n] = x[n] + 0.0 (n < D); x[n] + a*x[n-D] (D <= n < L); a*x[n-D] (L <= n < L+D); when D > L
// the gap [L, D) is Delay's zeros alone.  The second tap is 4*D bytes behind the first, i.e.
// an L2 / Infinity-Cache hit for any realistic D, so HBM sees one read and one write.
__global__ __launch_bounds__(kBlock) void k_echo_mix(float *__restrict__ dst, const float *__restrict__ src, size_t n, size_t delay, float gain, int vec_ok) {
    // four consecutive samples a lane (round 6: a sample a lane measured 0.37 of 8 TB/s, tools/bench_rows.py); the operations per sample are
    // the reference's: Delay(Amplify(x)) = x[i - D] * gain, or Delay's 0.0; mix.rs:47-52 adds where the input still runs
    const size_t total = n + delay, nvec = (total + 3) / 4, stride = (size_t)gridDim.x * kBlock;
    for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const size_t i = 4 * v;
        const float4 xv = rh::ld4_at(src, (int64_t)i, n), dv = rh::ld4_at(src, (int64_t)i - (int64_t)delay, n);
        const float x[4] = {xv.x, xv.y, xv.z, xv.w}, d[4] = {dv.x, dv.y, dv.z, dv.w};
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s2 = (i + j < delay) ? 0.0f : d[j] * gain;
            e[j] = (i + j < n) ? (x[j] + s2) : s2;
        }
        if (vec_ok && i + 4 <= total) {
            rh::st_nt(reinterpret_cast<float4 *>(dst) + v, make_float4(e[0], e[1], e[2], e[3]));
        } else {
            for (int j = 0; j < 4; ++j)
                if (i + j < total) dst[i + j] = e[j];
        }
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename T>
void launch_lines(float *dst, const T *src, size_t n, hipStream_t s) {
    if constexpr (sizeof(T) <= 2) hipLaunchKernelGGL(k_int_to_f32_lines<T>, dim3(rh::grid_tiles((n / 4 + 1) / 2 + 1)), dim3(kBlock), 0, s, dst, src, n);
}
template <typename T>
rh_status launch_int_to_f32(float *dst, const T *src, size_t n, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (n == 0) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    constexpr int VEC = 16 / sizeof(T);
    if (sizeof(T) <= 2 && aligned16(dst) && (reinterpret_cast<uintptr_t>(src) & (4 * sizeof(T) - 1)) == 0 && !rh::knob(rh::K_PCM_NO_TILE)) {
        launch_lines(dst, src, n, rh::as_stream(stream));
    } else if (aligned16(dst) && aligned16(src)) {
        hipLaunchKernelGGL(k_int_to_f32<T>, dim3(rh::grid_tiles(n / VEC + 1)), dim3(kBlock), 0, rh::as_stream(stream), dst, src, n);
    } else if ((std::is_same<T, int16_t>::value || std::is_same<T, uint8_t>::value) && (reinterpret_cast<uintptr_t>(src) & (sizeof(T) - 1)) == 0 &&
               rh::pcm_tile_try(dst, reinterpret_cast<const uint8_t *>(src), n, n, 1, 1, std::is_same<T, int16_t>::value ? 1 : 0, rh::as_stream(stream))) {
        // a row that starts off a vector boundary (inside a larger buffer): rh_wav.hip's tile kernel, PCM16 / unsigned 8-bit are two of its formats
    } else {
        hipLaunchKernelGGL(k_int_to_f32_scalar<T>, dim3(rh::grid_for(n)), dim3(kBlock), 0, rh::as_stream(stream), dst, src, n);
    }
    RH_CHECK_LAUNCH();
    return RH_OK;
}
template <typename T>
rh_status launch_f32_to_int(T *dst, const float *src, size_t n, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (n == 0) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    const int vec_ok = aligned16(src) && (reinterpret_cast<uintptr_t>(dst) % (4 * sizeof(T)) == 0);
    hipLaunchKernelGGL(k_f32_to_int<T>, dim3(rh::grid_tiles(n / 4 + 1)), dim3(kBlock), 0, rh::as_stream(stream), dst, src, n, vec_ok);
    RH_CHECK_LAUNCH();
    return RH_OK;
}


// ---------------------------------------------- reverb -> Spatial, fused and batched ----
// BASELINE config 3: for every stream  Spatial(reverb(x, D, a), emitter, ears)  in one pass:
//   r[i] = x[i] + 0 (i < D) | x[i] + a*x[i-D] (D <= i < L) | a*x[i-D] (L <= i < L+D)      source/mod.rs:628-634
//   m = ((0 + r[2f]) + r[2f+1]) / 2 ;  out[2f+k] = m * g[k]                               channel_volume.rs:71-88
// x is read once from HBM (the echo tap is 4*D bytes behind: L2 / Infinity Cache), the reverb
// intermediate never exists.  One lane = two output frames (16-byte accesses) when D % 4 == 0.
__device__ __forceinline__ float rev_at(const float *__restrict__ x, size_t i, size_t n, size_t delay, float gain) {
    const float s2 = (i < delay) ? 0.0f : x[i - delay] * gain;  // Delay(Amplify(x))
    return (i < n) ? (x[i] + s2) : s2;                             // mix.rs:47-52
}
struct StreamGains {
    const float *g;  // device [n_streams][2]
};
// Long delays (config 3: D = 65 536 samples): walk the stream in steps of D.  A lane owns the 4-sample
// column c and visits i = c, c+D, c+2D, ...: the direct tap of one step is the echo tap of the next, so it
// stays in registers and every input byte is loaded exactly once -- no reliance on a cache holding 4*D
// bytes per stream between the two uses (measured before: FETCH_SIZE = 1.77x the input with the schedule
// below, which re-reads the echo tap through L2 / Infinity Cache).  Same arithmetic, same bits.
__global__ __launch_bounds__(kBlock) void k_reverb_spatial_cols(float *__restrict__ dst, const float *__restrict__ src, size_t n, size_t delay, float gain,
                                                                const float *__restrict__ gains, size_t src_stride, size_t dst_stride, uint32_t n_streams) {
    // n % 4 == 0, delay % 4 == 0, 16-byte aligned rows; columns = delay / 4
    const size_t cols = delay / 4;
    const size_t total = n + delay;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t w = (size_t)blockIdx.x * kBlock + threadIdx.x; w < cols * n_streams; w += stride) {
        const uint32_t stream = (uint32_t)(w / cols);
        const size_t c = (w - (size_t)stream * cols) * 4;
        const float *x = src + (size_t)stream * src_stride;
        float *o = dst + (size_t)stream * dst_stride;
        const float g0 = gains[2 * stream], g1 = gains[2 * stream + 1];
        float4 prev = make_float4(0.f, 0.f, 0.f, 0.f);
        for (size_t i0 = c; i0 < total; i0 += 4 * delay) {
            float4 a4[4];  // four steps of loads in flight before the first one is consumed
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t i = i0 + (size_t)u * delay;
                a4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < n) a4[u] = rh::ld_nt(reinterpret_cast<const float4 *>(x + i));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t i = i0 + (size_t)u * delay;
                if (i >= total) break;
                const float4 a = a4[u];
                float r[4];
                const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {prev.x, prev.y, prev.z, prev.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float s2 = (i >= delay) ? bv[k] * gain : 0.0f;
                    r[k] = (i < n) ? (av[k] + s2) : s2;
                }
                float m0 = (0.0f + r[0]) + r[1], m1 = (0.0f + r[2]) + r[3];
                m0 = m0 / 2.0f;
                m1 = m1 / 2.0f;
                rh::st_nt(reinterpret_cast<float4 *>(o + i), make_float4(m0 * g0, m0 * g1, m1 * g0, m1 * g1));
                prev = a;
            }
        }
    }
}

// The same walk with the NEXT U steps' loads in flight while the current U are worked on (two register sets): a lane never sits
// between batches with nothing outstanding.  W: 16-byte columns per lane, 64 lanes apart -- a wave's step covers W KiB of one row
// with W fully coalesced instructions (W = 1: a lane per column).
template <int U, int W = 1>
__global__ __launch_bounds__(kBlock) void k_reverb_spatial_cols_p(float *__restrict__ dst, const float *__restrict__ src, size_t n, size_t delay, float gain,
                                                                  const float *__restrict__ gains, size_t src_stride, size_t dst_stride, uint32_t n_streams) {
    const size_t groups = delay / (4 * 64 * W);  // wave-sized column groups per stream (host: delay % (256 * W) == 0)
    const size_t total = n + delay;
    const size_t wv = ((size_t)blockIdx.x * kBlock + threadIdx.x) / 64;  // one wave per group: the host launches exactly enough workgroups
    if (wv >= groups * n_streams) return;
    const uint32_t stream = (uint32_t)(wv / groups);
    const size_t c = ((wv - (size_t)stream * groups) * W * 64 + (threadIdx.x & 63u)) * 4;  // the lane's first column; the others 256 samples apart
    const float *x = src + (size_t)stream * src_stride;
    float *o = dst + (size_t)stream * dst_stride;
    const float g0 = gains[2 * stream], g1 = gains[2 * stream + 1];
    auto fetch = [&](float4 (&a4)[U][W], size_t i0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = i0 + (size_t)u * delay;
#pragma unroll
            for (int v = 0; v < W; ++v) {
                a4[u][v] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < n) a4[u][v] = rh::ld_nt(reinterpret_cast<const float4 *>(x + i + 256 * v));  // (n % (256 * W) == 0: a group is inside the row or past it)
            }
        }
    };
    float4 prev[W];
#pragma unroll
    for (int v = 0; v < W; ++v) prev[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 cur[U][W], nxt[U][W];
    fetch(cur, c);
    for (size_t i0 = c; i0 < total; i0 += (size_t)U * delay) {
        const size_t i1 = i0 + (size_t)U * delay;
        if (i1 < total) fetch(nxt, i1);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = i0 + (size_t)u * delay;
            if (i >= total) break;
#pragma unroll
            for (int v = 0; v < W; ++v) {
                const float4 a = cur[u][v];
                float r[4];
                const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {prev[v].x, prev[v].y, prev[v].z, prev[v].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float s2 = (i >= delay) ? bv[k] * gain : 0.0f;
                    r[k] = (i < n) ? (av[k] + s2) : s2;
                }
                float m0 = (0.0f + r[0]) + r[1], m1 = (0.0f + r[2]) + r[3];
                m0 = m0 / 2.0f;
                m1 = m1 / 2.0f;
                rh::st_nt(reinterpret_cast<float4 *>(o + i + 256 * v), make_float4(m0 * g0, m0 * g1, m1 * g0, m1 * g1));
                prev[v] = a;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int v = 0; v < W; ++v) cur[u][v] = nxt[u][v];
    }
}

// XCD-aware schedule: the echo tap x[i-D] is 4*D bytes (256 KiB in config 3) behind the direct tap.
// Block b runs on XCD b % 8 (observed placement; speed only), whose 4 MiB L2 is private, so the
// delayed read is an L2 hit only if the SAME XCD read that line a moment ago: each XCD therefore walks
// whole streams (s = xcd, xcd+8, ...) one after the other with all of its blocks side by side --
// its L2 then sees D*4 bytes * 2 between a line's two uses instead of that times the number of
// concurrent streams (measured on config 3: 0.268 ms with all 64 streams in flight on every XCD).
template <bool VEC4>
__global__ __launch_bounds__(kBlock) void k_reverb_spatial(float *__restrict__ dst, const float *__restrict__ src, size_t n, size_t delay, float gain,
                                                           const float *__restrict__ gains, size_t src_stride, size_t dst_stride, size_t frames_out, uint32_t n_streams) {
    const uint32_t xcd = blockIdx.x & 7u, lb = blockIdx.x >> 3, nbx = gridDim.x >> 3;  // gridDim.x is a multiple of 8
    const size_t stride = (size_t)nbx * kBlock;
    for (uint32_t stream = xcd; stream < n_streams; stream += 8) {
        const float *x = src + (size_t)stream * src_stride;
        float *o = dst + (size_t)stream * dst_stride;
        const float g0 = gains[2 * stream], g1 = gains[2 * stream + 1];
        if (VEC4) {  // n % 4 == 0, delay % 4 == 0, 16-byte aligned rows
            const size_t quads = (frames_out + 1) / 2;
            for (size_t q = (size_t)lb * kBlock + threadIdx.x; q < quads; q += stride) {
                const size_t i = 4 * q;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
                if (i < n) a = *reinterpret_cast<const float4 *>(x + i);
                if (i >= delay) b = *reinterpret_cast<const float4 *>(x + i - delay);  // i - delay < n because i < n + delay
                float r[4];
                const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float s2 = (i >= delay) ? bv[k] * gain : 0.0f;
                    r[k] = (i < n) ? (av[k] + s2) : s2;
                }
                float m0 = (0.0f + r[0]) + r[1], m1 = (0.0f + r[2]) + r[3];
                m0 = m0 / 2.0f;
                m1 = m1 / 2.0f;
                *reinterpret_cast<float4 *>(o + i) = make_float4(m0 * g0, m0 * g1, m1 * g0, m1 * g1);
            }
        } else {
            for (size_t f = (size_t)lb * kBlock + threadIdx.x; f < frames_out; f += stride) {
                const float r0 = rev_at(x, 2 * f, n, delay, gain), r1 = rev_at(x, 2 * f + 1, n, delay, gain);
                float m = (0.0f + r0) + r1;
                m = m / 2.0f;
                *reinterpret_cast<float2 *>(o + 2 * f) = make_float2(m * g0, m * g1);
            }
        }
    }
}
}  // namespace

extern "C" {

rh_status rh_convert_i8_to_f32(float *dst, const int8_t *src, size_t n, rh_stream s) { return launch_int_to_f32(dst, src, n, s); }
rh_status rh_convert_u8_to_f32(float *dst, const uint8_t *src, size_t n, rh_stream s) { return launch_int_to_f32(dst, src, n, s); }
rh_status rh_convert_i16_to_f32(float *dst, const int16_t *src, size_t n, rh_stream s) { return launch_int_to_f32(dst, src, n, s); }
rh_status rh_convert_u16_to_f32(float *dst, const uint16_t *src, size_t n, rh_stream s) { return launch_int_to_f32(dst, src, n, s); }
static rh_status i32_like(float *dst, const int32_t *src, size_t n, float scale, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (n == 0) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    hipLaunchKernelGGL(k_i32_to_f32, dim3(rh::grid_tiles(n / 4 + 1)), dim3(kBlock), 0, rh::as_stream(stream), dst, src, n, scale, (int)(aligned16(dst) && aligned16(src)));
    RH_CHECK_LAUNCH();
    return RH_OK;
}
rh_status rh_convert_i24_to_f32(float *dst, const int32_t *src, size_t n, rh_stream s) { return i32_like(dst, src, n, 8388608.0f, s); }
rh_status rh_convert_i32_to_f32(float *dst, const int32_t *src, size_t n, rh_stream s) { return i32_like(dst, src, n, 2147483648.0f, s); }
rh_status rh_convert_f32_to_i8(int8_t *dst, const float *src, size_t n, rh_stream s) { return launch_f32_to_int(dst, src, n, s); }
rh_status rh_convert_f32_to_i16(int16_t *dst, const float *src, size_t n, rh_stream s) { return launch_f32_to_int(dst, src, n, s); }
rh_status rh_convert_f32_to_u16(uint16_t *dst, const float *src, size_t n, rh_stream s) { return launch_f32_to_int(dst, src, n, s); }
rh_status rh_convert_f32_to_i32(int32_t *dst, const float *src, size_t n, rh_stream s) { return launch_f32_to_int(dst, src, n, s); }

rh_status rh_channels_convert(float *dst, const float *src, size_t frames, uint32_t from_ch, uint32_t to_ch, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (from_ch == 0 || to_ch == 0 || from_ch > 65535 || to_ch > 65535) return RH_ERR_INVALID;
    if (frames == 0) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    // a tile of frames per workgroup through LDS (rh_wav.hip: f32 frames are the fifth sample format there; values pass as bits) -- the lane-per-
    // output kernel below reaches 0.38-0.59 of the roofline where the output is the wider side (2 -> 6, 1 -> 2), the tile 0.72-0.83 everywhere
    if (!rh::pcm_tile_try(dst, reinterpret_cast<const uint8_t *>(src), (uint64_t)frames * from_ch, frames, from_ch, to_ch, 4, rh::as_stream(stream)))
        hipLaunchKernelGGL(k_channels_convert, dim3(rh::grid_for(frames * to_ch)), dim3(kBlock), 0, rh::as_stream(stream), dst, src, frames, from_ch, to_ch);
    RH_CHECK_LAUNCH();
    return RH_OK;
}

rh_status rh_amplify(float *dst, const float *src, size_t n, float factor, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (n == 0) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    hipLaunchKernelGGL(k_amplify, dim3(rh::grid_tiles((n + 3) / 4)), dim3(kBlock), 0, rh::as_stream(stream), dst, src, n, factor, rh::rows_vec_bits(dst, src));
    RH_CHECK_LAUNCH();
    return RH_OK;
}

rh_status rh_channel_volume(float *dst, const float *src, size_t frames, uint32_t in_ch, const float *gains_host, uint32_t out_ch, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (in_ch == 0 || out_ch == 0 || !gains_host) return RH_ERR_INVALID;
    if (out_ch > 16) return RH_ERR_UNSUPPORTED;
    if (frames == 0) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    if (in_ch == 2 && out_ch == 2) {
        hipLaunchKernelGGL(k_channel_volume_2x2, dim3(rh::grid_tiles((frames + 1) / 2)), dim3(kBlock), 0, rh::as_stream(stream), dst, src, frames, gains_host[0], gains_host[1],
                           (int)(aligned16(dst) && aligned16(src)));
    } else {
        Gains g{};
        for (uint32_t k = 0; k < out_ch; ++k) g.g[k] = gains_host[k];
        // ~10 KiB in + out a tile (rh_wav.hip's measurement of this shape); frames of hundreds of channels keep the lane per frame
        const uint64_t tf = (10240ull / (4ull * (in_ch + out_ch))) & ~3ull;
        if (tf >= 8 && !rh::knob(rh::K_PCM_NO_TILE) && (frames + tf - 1) / tf <= 0x7fffffffull) {
            const uint32_t in_floats = (uint32_t)((tf * in_ch + 4 + 3) & ~3ull);
            const size_t lds = ((size_t)in_floats + tf + 16) * 4;
            hipLaunchKernelGGL(k_channel_volume_tile, dim3((unsigned)((frames + tf - 1) / tf)), dim3(kBlock), lds, rh::as_stream(stream), dst, src, frames, in_ch, g, out_ch, (uint32_t)tf, in_floats,
                               (int)aligned16(dst));
        } else {
            hipLaunchKernelGGL(k_channel_volume, dim3(rh::grid_for(frames)), dim3(kBlock), 0, rh::as_stream(stream), dst, src, frames, in_ch, g, out_ch);
        }
    }
    RH_CHECK_LAUNCH();
    return RH_OK;
}

// spatial.rs:19-24 (dist_sq), :48-69 (set_positions): two gains for ChannelVolume, f32 throughout.
static float dist_sq3(const float a[3], const float b[3]) {
    float s = 0.0f;
    for (int k = 0; k < 3; ++k) s += (a[k] - b[k]) * (a[k] - b[k]);
    return s;
}
rh_status rh_spatial_gains(const float emitter[3], const float left_ear[3], const float right_ear[3], float out_gains[2]) {
    if (!emitter || !left_ear || !right_ear || !out_gains) return RH_ERR_INVALID;
    const float left_dist_sq = dist_sq3(left_ear, emitter);
    const float right_dist_sq = dist_sq3(right_ear, emitter);
    const float max_diff = sqrtf(dist_sq3(left_ear, right_ear));
    const float left_dist = sqrtf(left_dist_sq);
    const float right_dist = sqrtf(right_dist_sq);
    const float left_diff_modifier = fminf(((left_dist - right_dist) / max_diff + 1.0f) / 4.0f + 0.5f, 1.0f);
    const float right_diff_modifier = fminf(((right_dist - left_dist) / max_diff + 1.0f) / 4.0f + 0.5f, 1.0f);
    const float left_dist_modifier = fminf(1.0f / left_dist_sq, 1.0f);
    const float right_dist_modifier = fminf(1.0f / right_dist_sq, 1.0f);
    out_gains[0] = left_diff_modifier * left_dist_modifier;
    out_gains[1] = right_diff_modifier * right_dist_modifier;
    return RH_OK;
}

uint64_t rh_delay_samples(uint64_t delay_ns, uint32_t sample_rate, uint32_t channels) {
    // delay.rs:8-16: ns * channels * rate / 1e9 in u128
    unsigned __int128 s = (unsigned __int128)delay_ns * channels * sample_rate / 1000000000ull;
    return (uint64_t)s;
}

rh_status rh_reverb_spatial(float *dst, const float *src, size_t n, size_t delay_samples, float gain, const float *gains_dev, uint32_t n_streams,
                            size_t src_stride, size_t dst_stride, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (n_streams == 0) return RH_OK;
    if (n % 2 != 0) return RH_ERR_INVALID;  // stereo frames (source/mod.rs:169-178: sources end on frame boundaries)
    const size_t frames_out = (n + delay_samples) / 2;  // an odd total leaves half a frame, which ChannelVolume drops (channel_volume.rs:76)
    if (frames_out == 0) return RH_OK;
    if (!dst || !src || !gains_dev) return RH_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 7u) return RH_ERR_INVALID;
    const bool vec4 = n % 4 == 0 && delay_samples % 4 == 0 && src_stride % 4 == 0 && dst_stride % 4 == 0 &&
                      ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) == 0;
    // 8 workgroups per CU, split evenly over the 8 XCDs; small jobs get fewer (still a multiple of 8)
    const size_t items = vec4 ? (frames_out + 1) / 2 : frames_out;
    const unsigned per_xcd_want = rh::grid_for(items, kBlock, 256u);  // blocks that have work on one stream
    const unsigned gx = 8u * (per_xcd_want < 256u ? per_xcd_want : 256u);
    const dim3 grid(gx);
    if (vec4 && delay_samples >= 4096 && (delay_samples / 4) * n_streams >= 64u * 1024u) {
        // enough independent columns to fill the chip: every input byte once
        const size_t lanes = (delay_samples / 4) * n_streams;
        // the walk with the next batch of loads in flight behind the current one, two columns per lane (a wave's step covers 2 KiB of a
        // row in two coalesced instructions).  Measured on config 3, same sessions: 0.2077-0.2097 ms for a lane per column and batches
        // of 8 steps (the round's first pipelined form; without the second register set: 0.211), 0.197-0.204 for two columns and batches
        // of 4 (default), 0.200 for four columns and 2; three columns / batches of 8: slower.  RH_RS_PIPE = 10 W + U; 0: no pipeline
        const char *pk = rh::knob(rh::K_RS_PIPE);
        const int pipe = pk ? atoi(pk) : 24;
        if (pipe > 0 && delay_samples % 256 == 0 && n % 256 == 0) {
            // RH_RS_PIPE = 10 * W + U (W = 2 or 4 columns per lane, a wave's step then covers W KiB) or U alone (a lane per column)
            int W = pipe >= 10 ? pipe / 10 : 1;
            const int U = pipe % 10 ? pipe % 10 : 8;
            if (W > 1 && (delay_samples % (256 * (size_t)W) || n % (256 * (size_t)W))) W = 1;
            const size_t waves = lanes / 64 / (size_t)W;
            const dim3 g((unsigned)((waves * 64 + kBlock - 1) / kBlock));
#define RH_RS_LAUNCH(u, w) hipLaunchKernelGGL((k_reverb_spatial_cols_p<u, w>), g, dim3(kBlock), 0, rh::as_stream(stream), dst, src, n, delay_samples, gain, gains_dev, src_stride, dst_stride, n_streams)
            if (W == 4 && U <= 2) RH_RS_LAUNCH(2, 4);
            else if (W == 4) RH_RS_LAUNCH(4, 4);
            else if (W == 2 && U <= 2) RH_RS_LAUNCH(2, 2);
            else if (W == 2 && U <= 4) RH_RS_LAUNCH(4, 2);
            else if (W == 2 && U <= 6) RH_RS_LAUNCH(6, 2);
            else if (W == 2) RH_RS_LAUNCH(8, 2);
            else if (W == 3) RH_RS_LAUNCH(4, 3);
            else if (U == 2) RH_RS_LAUNCH(2, 1);
            else if (U == 4) RH_RS_LAUNCH(4, 1);
            else if (U == 6) RH_RS_LAUNCH(6, 1);
            else RH_RS_LAUNCH(8, 1);
#undef RH_RS_LAUNCH
            RH_CHECK_LAUNCH();
            return RH_OK;
        }
        hipLaunchKernelGGL(k_reverb_spatial_cols, dim3(rh::grid_for(lanes, kBlock, 256u * 16u)), dim3(kBlock), 0, rh::as_stream(stream), dst, src, n, delay_samples, gain, gains_dev, src_stride, dst_stride, n_streams);
    } else if (vec4)
        hipLaunchKernelGGL(k_reverb_spatial<true>, grid, dim3(kBlock), 0, rh::as_stream(stream), dst, src, n, delay_samples, gain, gains_dev, src_stride, dst_stride, frames_out, n_streams);
    else hipLaunchKernelGGL(k_reverb_spatial<false>, grid, dim3(kBlock), 0, rh::as_stream(stream), dst, src, n, delay_samples, gain, gains_dev, src_stride, dst_stride, frames_out, n_streams);
    RH_CHECK_LAUNCH();
    return RH_OK;
}

rh_status rh_echo_mix(float *dst, const float *src, size_t n, size_t delay_samples, float gain, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (n + delay_samples == 0) return RH_OK;
    if (!dst || (!src && n)) return RH_ERR_INVALID;
    hipLaunchKernelGGL(k_echo_mix, dim3(rh::grid_tiles((n + delay_samples + 3) / 4)), dim3(kBlock), 0, rh::as_stream(stream), dst, src, n, delay_samples, gain, (int)aligned16(dst));
    RH_CHECK_LAUNCH();
    return RH_OK;
}

}  // extern "C"

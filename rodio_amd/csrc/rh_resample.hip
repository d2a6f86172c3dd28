// rh_resample.hip -- SampleRateConverter + UniformSourceIterator span chunking, and the Mixer sum.
//
//   src/conversions/sample_rate.rs:52-90,110-122,131-201   (the streaming lerp state machine)
//   src/math.rs:23-26                                        (lerp: mul, divide, add -- in that order)
//   src/source/uniform.rs:50-97                              (restart every min(span,32768) samples)
//   src/mixer.rs:185-198                                     (ordered f32 sum)
//
// The reference's state machine is a pure function of the output frame index m:
//     i = floor(m*F/T), num = (m*F) mod T        (F/T = from/to reduced by gcd, sample_rate.rs:74)
//     i <= N-2 : x[i] + (x[i+1]-x[i]) * num / T  (lerp)
//     i == N-1 : x[N-1] verbatim, then the stream ends (the drain at :193-200)
// applied independently per chunk of min(span_len,32768) samples.  That closed form is what
// the kernel evaluates -- one lane per output frame, both taps read straight from HBM/L2
// (adjacent lanes read adjacent frames, so every input line is fetched once).
// This TU is compiled with -ffp-contract=off: the lerp must not become an FMA.
#include <numeric>

#include <cstring>

#include "rh_common.h"

namespace rh {

struct ResampleGeom {
    uint32_t F, T;
    uint64_t in_frames;
    uint64_t chunk_in;   // input frames per full chunk (== in_frames when unchunked)
    uint64_t chunk_out;  // output frames per full chunk
    uint64_t n_chunks;
    uint64_t last_in;    // input frames of the last chunk
    uint64_t out_frames;
    int fits32;          // (chunk_out * F) < 2^32: index math in u32 like the reference
};

// Output frames of one independently converted run of n input frames:
// every m with floor(mF/T) <= n-2, plus one verbatim frame if some m lands on i == n-1.
static uint64_t run_out_frames(uint64_t n, uint64_t F, uint64_t T) {
    if (n == 0) return 0;
    if (F == T) return n;
    const unsigned __int128 num = (unsigned __int128)(n - 1) * T;
    const uint64_t c1 = (uint64_t)((num + F - 1) / F);  // #m with i(m) <= n-2
    const bool lands = (unsigned __int128)c1 * F < (unsigned __int128)n * T;
    return c1 + (lands ? 1 : 0);
}

rh_status make_resample_geom(uint64_t in_frames, uint32_t from_rate, uint32_t to_rate, uint32_t channels,
                             uint64_t span_len, ResampleGeom *g) {
    if (from_rate == 0 || to_rate == 0 || channels == 0) return RH_ERR_INVALID;
    const uint32_t gc = std::gcd(from_rate, to_rate);
    g->F = from_rate / gc;
    g->T = to_rate / gc;
    // The reference multiplies from*pos in u32 (sample_rate.rs:157,173; doc :45-47).
    if ((uint64_t)g->F * g->T > 0xffffffffull) return RH_ERR_UNSUPPORTED;
    g->in_frames = in_frames;
    uint64_t chunk_in = in_frames;
    if (span_len != 0) {
        const uint64_t span = span_len < 32768 ? span_len : 32768;  // uniform.rs:56
        if (span % channels != 0) return RH_ERR_UNSUPPORTED;        // a span that splits a frame
        chunk_in = span / channels;
        if (chunk_in == 0) return RH_ERR_INVALID;
    }
    if (chunk_in >= in_frames || in_frames == 0) {
        g->chunk_in = in_frames;
        g->n_chunks = in_frames ? 1 : 0;
        g->last_in = in_frames;
        g->chunk_out = run_out_frames(in_frames, g->F, g->T);
        g->out_frames = g->chunk_out;
    } else {
        g->chunk_in = chunk_in;
        g->n_chunks = (in_frames + chunk_in - 1) / chunk_in;
        g->last_in = in_frames - (g->n_chunks - 1) * chunk_in;
        g->chunk_out = run_out_frames(chunk_in, g->F, g->T);
        g->out_frames = (g->n_chunks - 1) * g->chunk_out + run_out_frames(g->last_in, g->F, g->T);
    }
    const unsigned __int128 pmax = (unsigned __int128)(g->chunk_out + 1) * g->F;
    g->fits32 = pmax < ((unsigned __int128)1 << 32);
    return RH_OK;
}

}  // namespace rh

namespace {

constexpr int kBlock = 256;

template <int C, bool P32>
__global__ __launch_bounds__(kBlock) void k_resample_linear(float *__restrict__ dst, const float *__restrict__ src, rh::ResampleGeom g, uint32_t channels) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const float Tf = (float)g.T;
    for (uint64_t m = (uint64_t)blockIdx.x * kBlock + threadIdx.x; m < g.out_frames; m += stride) {
        uint64_t k = 0, ml = m;
        if (g.n_chunks > 1) {
            k = m / g.chunk_out;
            if (k > g.n_chunks - 1) k = g.n_chunks - 1;
            ml = m - k * g.chunk_out;
        }
        uint64_t il;
        uint32_t num;
        if (P32) {
            const uint32_t p = (uint32_t)ml * g.F;
            il = p / g.T;
            num = p - (uint32_t)il * g.T;
        } else {
            const uint64_t p = ml * g.F;
            il = p / g.T;
            num = (uint32_t)(p - il * g.T);
        }
        const uint64_t nc = (k == g.n_chunks - 1) ? g.last_in : g.chunk_in;
        const bool verbatim = il + 1 >= nc;  // i == nc-1: the drained last frame
        const uint64_t i = k * g.chunk_in + (verbatim ? nc - 1 : il);
        const float numf = (float)num;
        if (C == 2) {
            const float2 a = reinterpret_cast<const float2 *>(src)[i];
            float2 o = a;
            if (!verbatim) {
                const float2 b = reinterpret_cast<const float2 *>(src)[i + 1];
                o.x = a.x + (b.x - a.x) * numf / Tf;
                o.y = a.y + (b.y - a.y) * numf / Tf;
            }
            reinterpret_cast<float2 *>(dst)[m] = o;
        } else {
            const uint32_t ch = (C == 0) ? channels : (uint32_t)C;
            for (uint32_t c = 0; c < ch; ++c) {
                const float a = src[i * ch + c];
                float o = a;
                if (!verbatim) {
                    const float b = src[(i + 1) * ch + c];
                    o = a + (b - a) * numf / Tf;
                }
                dst[m * ch + c] = o;
            }
        }
    }
}

struct MixDesc {
    const float *data;
    uint64_t start;
    uint64_t len;
};

// One lane per output sample, sources visited in insertion order: the rounding sequence of
// mixer.rs:185-198 (`sum = 0.0; sum += v_s`).  No atomics, no tree: bit-identical to the CPU.
// The source table travels BY VALUE as a kernel argument, kMixChunk sources per launch: nothing to upload, nothing
// to synchronise, nothing that a later call could overwrite while this launch is still queued.  More sources take
// further launches that continue from the stored partial sum -- the same left-to-right sequence of f32 additions.
constexpr uint32_t kMixChunk = 32;
struct MixTable {
    MixDesc d[kMixChunk];
};
template <bool CONT>
__global__ __launch_bounds__(kBlock) void k_mix_sum(float *__restrict__ dst, uint64_t out_len, const MixTable tbl, uint32_t n_sources) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t o = (uint64_t)blockIdx.x * kBlock + threadIdx.x; o < out_len; o += stride) {
        float acc = CONT ? dst[o] : 0.0f;
        for (uint32_t s = 0; s < n_sources; ++s) {
            const MixDesc d = tbl.d[s];
            const uint64_t rel = o - d.start;  // wraps to huge when o < start
            if (rel < d.len) acc += d.data[rel];
        }
        dst[o] = acc;
    }
}

// Same, four consecutive samples per lane (float4 loads) when every start is a multiple of 4
// samples and every pointer is 16-byte aligned -- the common "all sources start together" case.
template <bool CONT>
__global__ __launch_bounds__(kBlock) void k_mix_sum_v4(float *__restrict__ dst, uint64_t out_len, const MixTable tbl, uint32_t n_sources) {
    const uint64_t nvec = (out_len + 3) / 4;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        const uint64_t o = v * 4;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (CONT) {
            for (int k = 0; k < 4; ++k)
                if (o + k < out_len) acc[k] = dst[o + k];
        }
        for (uint32_t s = 0; s < n_sources; ++s) {
            const MixDesc d = tbl.d[s];
            const uint64_t rel = o - d.start;
            if (rel < d.len) {
                if (rel + 4 <= d.len) {
                    const float4 x = *reinterpret_cast<const float4 *>(d.data + rel);
                    acc[0] += x.x;
                    acc[1] += x.y;
                    acc[2] += x.z;
                    acc[3] += x.w;
                } else {
                    for (int k = 0; k < 4; ++k)
                        if (rel + k < d.len) acc[k] += d.data[rel + k];
                }
            }
        }
        if (o + 4 <= out_len) {
            *reinterpret_cast<float4 *>(dst + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        } else {
            for (int k = 0; k < 4; ++k)
                if (o + k < out_len) dst[o + k] = acc[k];
        }
    }
}

}  // namespace

extern "C" {

rh_status rh_resample_out_frames(uint64_t in_frames, uint32_t from_rate, uint32_t to_rate, uint32_t channels, uint64_t span_len, uint64_t *out_frames) {
    if (!out_frames) return RH_ERR_INVALID;
    rh::ResampleGeom g;
    rh_status st = rh::make_resample_geom(in_frames, from_rate, to_rate, channels, span_len, &g);
    if (st != RH_OK) return st;
    *out_frames = g.out_frames;
    return RH_OK;
}

rh_status rh_resample_linear(float *dst, const float *src, uint64_t in_frames, uint32_t from_rate, uint32_t to_rate, uint32_t channels, uint64_t span_len, rh_stream stream) {
    RH_REQUIRE_INIT();
    rh::ResampleGeom g;
    rh_status st = rh::make_resample_geom(in_frames, from_rate, to_rate, channels, span_len, &g);
    if (st != RH_OK) return st;
    if (g.out_frames == 0) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    hipStream_t s = rh::as_stream(stream);
    if (g.F == g.T) {  // sample_rate.rs:133-136 passthrough
        RH_HIP_TRY(rh::copy_d2d(dst, src, in_frames * channels * sizeof(float), s));
        return RH_OK;
    }
    const unsigned grid = rh::grid_for(g.out_frames);
    const bool f2 = channels == 2 && (reinterpret_cast<uintptr_t>(dst) % 8 == 0) && (reinterpret_cast<uintptr_t>(src) % 8 == 0);
    if (f2) {
        if (g.fits32) hipLaunchKernelGGL((k_resample_linear<2, true>), dim3(grid), dim3(kBlock), 0, s, dst, src, g, channels);
        else hipLaunchKernelGGL((k_resample_linear<2, false>), dim3(grid), dim3(kBlock), 0, s, dst, src, g, channels);
    } else if (channels == 1) {
        if (g.fits32) hipLaunchKernelGGL((k_resample_linear<1, true>), dim3(grid), dim3(kBlock), 0, s, dst, src, g, channels);
        else hipLaunchKernelGGL((k_resample_linear<1, false>), dim3(grid), dim3(kBlock), 0, s, dst, src, g, channels);
    } else {
        if (g.fits32) hipLaunchKernelGGL((k_resample_linear<0, true>), dim3(grid), dim3(kBlock), 0, s, dst, src, g, channels);
        else hipLaunchKernelGGL((k_resample_linear<0, false>), dim3(grid), dim3(kBlock), 0, s, dst, src, g, channels);
    }
    RH_CHECK_LAUNCH();
    return RH_OK;
}

rh_status rh_mix_sum(float *dst, size_t out_len, const float *const *srcs_host, const uint64_t *start_host, const uint64_t *len_host, uint32_t n_sources, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (out_len == 0) return RH_OK;
    if (!dst || (n_sources && (!srcs_host || !start_host || !len_host))) return RH_ERR_INVALID;
    hipStream_t s = rh::as_stream(stream);
    if (n_sources == 0) {  // an empty mixer yields nothing (mixer.rs:131-135); caller asked for zeros
        RH_HIP_TRY(rh::fill_async(dst, 0, out_len * sizeof(float), s));
        return RH_OK;
    }
    bool vec_ok = (reinterpret_cast<uintptr_t>(dst) % 16 == 0);
    for (uint32_t i = 0; i < n_sources; ++i) {
        if (len_host[i] && !srcs_host[i]) return RH_ERR_INVALID;
        vec_ok = vec_ok && (start_host[i] % 4 == 0) && (reinterpret_cast<uintptr_t>(srcs_host[i]) % 16 == 0);
    }
    for (uint32_t first = 0; first < n_sources; first += kMixChunk) {
        const uint32_t n = n_sources - first < kMixChunk ? n_sources - first : kMixChunk;
        MixTable t;
        std::memset(&t, 0, sizeof(t));
        for (uint32_t i = 0; i < n; ++i) t.d[i] = MixDesc{srcs_host[first + i], start_host[first + i], len_host[first + i]};
        if (vec_ok) {
            if (first) hipLaunchKernelGGL(k_mix_sum_v4<true>, dim3(rh::grid_tiles((out_len + 3) / 4)), dim3(kBlock), 0, s, dst, (uint64_t)out_len, t, n);
            else hipLaunchKernelGGL(k_mix_sum_v4<false>, dim3(rh::grid_tiles((out_len + 3) / 4)), dim3(kBlock), 0, s, dst, (uint64_t)out_len, t, n);
        } else {
            if (first) hipLaunchKernelGGL(k_mix_sum<true>, dim3(rh::grid_for(out_len)), dim3(kBlock), 0, s, dst, (uint64_t)out_len, t, n);
            else hipLaunchKernelGGL(k_mix_sum<false>, dim3(rh::grid_for(out_len)), dim3(kBlock), 0, s, dst, (uint64_t)out_len, t, n);
        }
        RH_CHECK_LAUNCH();
    }
    return RH_OK;
}

}  // extern "C"

// rh_uniform.hip -- UniformSourceIterator span by span (src/source/uniform.rs:50-97).
//
// rodio re-builds its converter chain whenever the current one runs dry:
//     bootstrap():  span_len = input.current_span_len().map(|x| x.min(32768))          uniform.rs:56
//                   Take{n: span_len} -> SampleRateConverter(from_rate -> to_rate, from_channels)
//                                     -> ChannelCountConverter(from_channels -> to_channels)   uniform.rs:58-67
// so a spanned source (SamplesBuffer: buffer.rs:76-82, Buffered: buffered.rs:109, the decoders: symphonia.rs:199-201)
// is converted SPAN BY SPAN: every span starts a fresh converter (position 0, no history) and ends with its last frame
// emitted verbatim (sample_rate.rs:193-200); format changes take effect at span boundaries.  Spans are independent work
// items, and so are the pieces ("segments") a span is cut into when it arrives in blocks: output frame m of a span is
//     i = floor(m*F/T), num = (m*F) mod T   (F/T = from/to reduced, sample_rate.rs:74; the reference counts both per chunk
//                                            of T outputs in u32, which is the same numbers as long as F*T < 2^32)
//     i <= N-2 : x[i] + (x[i+1]-x[i]) * num / T     (math.rs:25, in that order: this TU is built with -ffp-contract=off)
//     i == N-1 : x[N-1] verbatim, and the span is over
// followed by the channel rule of channels.rs:57-85 (k < from: the input channel; k == 1 of a mono source: channel 0 again;
// otherwise 0.0; surplus input channels are dropped).  One launch converts a whole table of segments -- any number of
// sources, rates and layouts -- which is what GpuMixer / GpuSource::uniform of include/rodio_hip.hpp put in front of the
// fused filter + mix launch for sources that report spans.
#include <numeric>

#include <cstring>

#include "rh_common.h"

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kTile = 2048;        // output frames per workgroup of a large launch;
constexpr uint32_t kTileSmall = 256;    // ... of a small one (a chain's block: a frame per lane, eight times the workgroups -- 8 workgroups walking 2048 frames each took 20 us)
inline uint32_t tile_for(uint64_t most, uint32_t n_segs) { return ((most + kTile - 1) / kTile) * n_segs >= 1024 ? kTile : kTileSmall; }
constexpr uint32_t kSegsPerLaunch = 48; // by-value table: 48 x 80 B of kernel arguments

struct SegTable {
    rh_uniform_seg s[kSegsPerLaunch];
};

// #m with floor(m*F/T) <= n-2 (both taps of the lerp exist)
__host__ __device__ inline uint64_t lerp_ready(uint64_t n, uint64_t F, uint64_t T) {
    if (n == 0) return 0;
    return (uint64_t)((((unsigned __int128)(n - 1) * T) + F - 1) / F);
}
// The tail of a span of q whole frames that ends inside a frame (rodio_hip.h): the output frames m_first .. m_first + k - 1 have their
// first tap on the last whole frame and lerp towards the cut frame (k may be 0: downsampling can skip that frame, and a span without
// a whole frame has none); e = 1 when an output lands on the cut frame itself, which then comes out verbatim.
__host__ __device__ inline void cut_counts(uint64_t q, uint64_t F, uint64_t T, uint64_t &m_first, uint64_t &k, uint64_t &e) {
    if (F == T) {  // sample_rate.rs:133-136: the converter passes every sample through
        m_first = q, k = 0, e = 1;
        return;
    }
    m_first = lerp_ready(q, F, T);
    k = q >= 1 ? lerp_ready(q + 1, F, T) - lerp_ready(q, F, T) : 0;
    e = lerp_ready(q + 2, F, T) - lerp_ready(q + 1, F, T) >= 1 ? 1 : 0;
}

__device__ __forceinline__ uint32_t gcd_u32(uint32_t a, uint32_t b) {
    while (b) {
        const uint32_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

// One output SAMPLE per lane: sample j of the tail is position j % to_ch of the (j / to_ch)-th group of from_ch converter samples
// (channels.rs:57-85), and converter sample idx of the tail is channel idx % t of its (idx / t)-th short frame.
__device__ __forceinline__ void convert_cut_tail(const rh_uniform_seg &g, uint32_t tile, const uint32_t kTile) {  // (the launch's tile length shadows the constant)
    const uint64_t j0 = g.m0 + (uint64_t)tile * kTile;
    if (j0 >= g.m1) return;
    const uint64_t j1 = j0 + kTile < g.m1 ? j0 + kTile : g.m1;
    const uint32_t gc = gcd_u32(g.from_rate, g.to_rate);
    const uint32_t F = g.from_rate / gc, T = g.to_rate / gc;
    const uint32_t fc = g.from_ch, tc = g.to_ch, t = g.reserved;
    uint64_t mf, k, e;
    cut_counts(g.span_frames, F, T, mf, k, e);
    const float *last = g.src;                         // the frame in front of the cut (read only when k > 0: then it is there)
    const float *p = g.src + g.src_frames * fc;        // the cut frame's t samples
    const float Tf = (float)T;
    for (uint64_t j = j0 + threadIdx.x; j < j1; j += kBlock) {
        const uint64_t grp = j / tc;
        const uint32_t pos = (uint32_t)(j - grp * tc);
        float v = 0.0f;  // channels the source does not have (a mono source cannot be cut: channels.rs:64-73's repeat never applies)
        if (pos < fc) {
            const uint64_t idx = grp * fc + pos, r = idx / t;
            const uint32_t c = (uint32_t)(idx - r * t);
            const float pv = p[c] * g.gain;
            v = pv;
            if (r < k) {
                const unsigned __int128 pp = (unsigned __int128)(mf + r) * F;
                const uint32_t num = (uint32_t)(pp % T);
                const float av = last[c] * g.gain;
                v = av + (pv - av) * (float)num / Tf;  // math.rs:25
            }
        }
        g.dst[j - g.m0] = v;
    }
}

__device__ __forceinline__ void convert_frames(const rh_uniform_seg &g, uint32_t tile, const uint32_t kTile) {  // (the launch's tile length shadows the constant)
    if (g.reserved) return convert_cut_tail(g, tile, kTile);
    const uint64_t mt0 = g.m0 + (uint64_t)tile * kTile;
    if (mt0 >= g.m1) return;
    const uint64_t mt1 = mt0 + kTile < g.m1 ? mt0 + kTile : g.m1;
    const uint32_t gc = gcd_u32(g.from_rate, g.to_rate);  // a handful of scalar-ish iterations, the same in every lane
    const uint32_t F = g.from_rate / gc, T = g.to_rate / gc;
    const uint32_t fc = g.from_ch, tc = g.to_ch;
    const uint32_t nc = fc < tc ? fc : tc;  // channels that carry input
    const float Tf = (float)T;
    const bool small = g.m1 <= (1ull << 31);  // F < 2^32 (F*T < 2^32 is checked on the host): m*F fits 64 bits
    for (uint64_t m = mt0 + threadIdx.x; m < mt1; m += kBlock) {
        float *o = g.dst + (m - g.m0) * tc;
        uint64_t i;
        uint32_t num;
        if (F == T) {  // sample_rate.rs:133-136: the converter passes through
            i = m;
            num = 0;
        } else if (small) {
            const uint64_t pp = m * F;
            i = pp / T;
            num = (uint32_t)(pp - i * T);
        } else {
            const unsigned __int128 pp = (unsigned __int128)m * F;
            i = (uint64_t)(pp / T);
            num = (uint32_t)(pp - (unsigned __int128)i * T);
        }
        const bool verbatim = F == T || i + 1 >= g.span_frames;  // the span's last frame (sample_rate.rs:193-200)
        const float *a = g.src + (i - g.src_frame0) * fc;
        const float numf = (float)num;
        for (uint32_t k = 0; k < nc; ++k) {
            const float av = a[k] * g.gain;  // amplify.rs:64 in front of the converter
            float v = av;
            if (!verbatim) {
                const float bv = a[fc + k] * g.gain;
                v = av + (bv - av) * numf / Tf;
            }
            o[k] = v;
        }
        if (tc > fc) {  // channels.rs:64-73
            if (fc == 1) o[1] = o[0];
            for (uint32_t k = (fc == 1 ? 2u : fc); k < tc; ++k) o[k] = 0.0f;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_uniform_segs_val(const SegTable tbl, uint32_t n, uint32_t tile_frames) {
    if (blockIdx.y < n) convert_frames(tbl.s[blockIdx.y], blockIdx.x, tile_frames);
}
__global__ __launch_bounds__(kBlock) void k_uniform_segs_dev(const rh_uniform_seg *__restrict__ segs, uint32_t n, uint32_t tile_frames) {
    if (blockIdx.y >= n) return;
    __shared__ rh_uniform_seg g;
    if (threadIdx.x < sizeof(rh_uniform_seg) / 8) reinterpret_cast<uint64_t *>(&g)[threadIdx.x] = reinterpret_cast<const uint64_t *>(segs + blockIdx.y)[threadIdx.x];
    __syncthreads();
    convert_frames(g, blockIdx.x, tile_frames);
}

}  // namespace

extern "C" {

rh_status rh_uniform_span_frames(uint64_t span_in_frames, uint32_t from_rate, uint32_t to_rate, int32_t complete, uint64_t *out_frames) {
    if (!out_frames || from_rate == 0 || to_rate == 0) return RH_ERR_INVALID;
    const uint32_t gc = std::gcd(from_rate, to_rate);
    const uint64_t F = from_rate / gc, T = to_rate / gc;
    if (F * T > 0xffffffffull) return RH_ERR_UNSUPPORTED;  // the reference's u32 products overflow (sample_rate.rs:45-47)
    if (F == T) {
        *out_frames = span_in_frames;
        return RH_OK;
    }
    uint64_t c = lerp_ready(span_in_frames, F, T);
    // a complete span also emits the m that lands on its last frame (upsampling: always; downsampling: if one does)
    if (complete && span_in_frames && (unsigned __int128)c * F < (unsigned __int128)span_in_frames * T) c += 1;
    *out_frames = c;
    return RH_OK;
}

rh_status rh_uniform_first_tap(uint64_t out_frame, uint32_t from_rate, uint32_t to_rate, uint64_t *in_frame) {
    if (!in_frame || from_rate == 0 || to_rate == 0) return RH_ERR_INVALID;
    const uint32_t gc = std::gcd(from_rate, to_rate);
    const uint64_t F = from_rate / gc, T = to_rate / gc;
    *in_frame = (uint64_t)(((unsigned __int128)out_frame * F) / T);
    return RH_OK;
}

rh_status rh_uniform_cut_tail_samples(uint64_t span_whole_frames, uint32_t tail_samples, uint32_t from_rate, uint32_t to_rate, uint32_t from_ch, uint32_t to_ch, uint64_t *out_samples) {
    if (!out_samples || from_rate == 0 || to_rate == 0 || from_ch == 0 || to_ch == 0 || tail_samples == 0 || tail_samples >= from_ch) return RH_ERR_INVALID;
    const uint32_t gc = std::gcd(from_rate, to_rate);
    const uint64_t F = from_rate / gc, T = to_rate / gc;
    if (F * T > 0xffffffffull) return RH_ERR_UNSUPPORTED;
    uint64_t mf, k, e;
    cut_counts(span_whole_frames, F, T, mf, k, e);
    // (k + e) runs of tail_samples converter samples, regrouped by from_ch: every complete group gives an output frame, the rest
    // the channels it covers (channels.rs:57-85: the None behind the last sample ends the chain)
    const uint64_t u = (k + e) * tail_samples, groups = u / from_ch, rest = u % from_ch;
    *out_samples = groups * to_ch + (rest < to_ch ? rest : to_ch);
    return RH_OK;
}

static rh_status check_seg(const rh_uniform_seg &g) {
    if (g.from_rate == 0 || g.to_rate == 0 || g.from_ch == 0 || g.to_ch == 0) return RH_ERR_INVALID;  // NonZero in rodio
    if (g.m1 < g.m0) return RH_ERR_INVALID;
    if (g.m1 == g.m0) return RH_OK;
    if (g.reserved) {  // the tail of a span that ends inside a frame
        if (!g.src || !g.dst || g.reserved >= g.from_ch || g.src_frames > 1 || g.src_frames > g.span_frames || g.src_frame0 + g.src_frames != g.span_frames) return RH_ERR_INVALID;
        uint64_t total = 0;
        const rh_status st = rh_uniform_cut_tail_samples(g.span_frames, g.reserved, g.from_rate, g.to_rate, g.from_ch, g.to_ch, &total);
        if (st != RH_OK) return st;
        if (g.m1 > total) return RH_ERR_INVALID;
        const uint32_t gc = std::gcd(g.from_rate, g.to_rate);
        uint64_t mf, k, e;
        cut_counts(g.span_frames, g.from_rate / gc, g.to_rate / gc, mf, k, e);
        if (k > 0 && g.src_frames != 1) return RH_ERR_INVALID;  // an output lerps towards the cut frame: the frame in front of it must be there
        return RH_OK;
    }
    if (!g.src || !g.dst || g.src_frames == 0) return RH_ERR_INVALID;
    const uint32_t gc = std::gcd(g.from_rate, g.to_rate);
    const uint64_t F = g.from_rate / gc, T = g.to_rate / gc;
    if (F * T > 0xffffffffull) return RH_ERR_UNSUPPORTED;
    // every tap the segment reads lies inside [src_frame0, src_frame0 + src_frames)
    const uint64_t i_first = (uint64_t)(((unsigned __int128)g.m0 * F) / T), i_last = (uint64_t)(((unsigned __int128)(g.m1 - 1) * F) / T);
    if (i_first < g.src_frame0) return RH_ERR_INVALID;
    const bool last_verbatim = F == T || i_last + 1 >= g.span_frames;
    const uint64_t top = last_verbatim ? (F == T ? i_last : g.span_frames - 1) : i_last + 1;
    if (top >= g.src_frame0 + g.src_frames) return RH_ERR_INVALID;
    if (F != T && g.span_frames != UINT64_MAX) {  // a complete span has no output frame past its verbatim one
        uint64_t total = 0;
        (void)rh_uniform_span_frames(g.span_frames, g.from_rate, g.to_rate, 1, &total);
        if (g.m1 > total) return RH_ERR_INVALID;
    }
    return RH_OK;
}

rh_status rh_uniform_segments(const rh_uniform_seg *segs_host, uint32_t n_segs, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (n_segs && !segs_host) return RH_ERR_INVALID;
    for (uint32_t k = 0; k < n_segs; ++k) {
        const rh_status st = check_seg(segs_host[k]);
        if (st != RH_OK) return st;
    }
    hipStream_t s = rh::as_stream(stream);
    for (uint32_t first = 0; first < n_segs; first += kSegsPerLaunch) {
        const uint32_t n = n_segs - first < kSegsPerLaunch ? n_segs - first : kSegsPerLaunch;
        SegTable t;
        std::memset(&t, 0, sizeof(t));
        uint64_t most = 0;
        for (uint32_t k = 0; k < n; ++k) {
            t.s[k] = segs_host[first + k];
            most = std::max<uint64_t>(most, t.s[k].m1 - t.s[k].m0);
        }
        if (!most) continue;
        const uint32_t tf = tile_for(most, n);
        const uint64_t tiles = (most + tf - 1) / tf;
        if (tiles > 0x7fffffffull) return RH_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(k_uniform_segs_val, dim3((uint32_t)tiles, n), dim3(kBlock), 0, s, t, n, tf);
        RH_CHECK_LAUNCH();
    }
    return RH_OK;
}

rh_status rh_uniform_segments_dev(const rh_uniform_seg *segs_dev, uint32_t n_segs, uint64_t max_out_frames, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (!n_segs || !max_out_frames) return RH_OK;
    if (!segs_dev || n_segs > 65535u) return RH_ERR_INVALID;
    const uint32_t tf = tile_for(max_out_frames, n_segs);
    const uint64_t tiles = (max_out_frames + tf - 1) / tf;
    if (tiles > 0x7fffffffull) return RH_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_uniform_segs_dev, dim3((uint32_t)tiles, n_segs), dim3(kBlock), 0, rh::as_stream(stream), segs_dev, n_segs, tf);
    RH_CHECK_LAUNCH();
    return RH_OK;
}

}  // extern "C"

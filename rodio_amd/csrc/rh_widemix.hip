// rh_widemix.hip -- a block of a mixer of ANY channel count in one launch (round 6):
//     out[m][c] = ((0.0 + v_0[m][c]) + v_1[m][c]) + ...                                        mixer.rs:185-198, insertion order
//     v_s       = ChannelCountConverter(SampleRateConverter(Amplify(x_s)))                     uniform.rs:58-67, amplify.rs:64
// for continuous sources (current_span_len() == None: one converter for the whole stream, uniform.rs:56) of any rate and layout.
// Until now a mixer of more than two channels (mixer::mixer(nz!(6), rate), mixer.rs:25) ran every source as a chain of its own --
// an amplify launch, a converter launch, a device copy per source and block -- and rh_mix_sum added the rows.  Here the converter
// is a pure function of the output frame, as everywhere in this library:
//     i = floor(m F / T), num = (m F) mod T     (F / T = from / to reduced, sample_rate.rs:74)
//     i <= N - 2 :  a + (b - a) * num / T       (math.rs:25, in that order; a = g x[i][k], b = g x[i+1][k])
//     i == N - 1 :  a, verbatim, and the source is over (sample_rate.rs:193-200)
//     F == T     :  a (sample_rate.rs:133-136: the converter passes every sample through)
// followed by channels.rs:59-70 (k < from: the input channel; k == 1 of a narrower source: its first channel; otherwise 0.0,
// which leaves an f32 sum that started at +0.0 as it is) -- so one lane per output SAMPLE walks the sources in insertion order
// and adds: the reference's rounding sequence, bit for bit, with fully coalesced loads and stores.  HBM-bound: every input byte
// is read once from memory (the second tap of a frame is the first tap of the next: L1 / L2), 4 S C_s N_s bytes in, 4 C M out.
// The source table travels by value as a kernel argument (32 sources a launch; more continue from the stored partial sum).
#include <numeric>
#include <vector>

#include "rh_common.h"

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kWideChunk = 32;  // sources a launch
constexpr int kWideGroup = 4;        // ... walked in groups of this many, whose tap loads leave together
constexpr uint32_t kWideRates = 4;   // ... of at most this many different (rate, phase) pairs: the position of an output frame between its taps is worked out once per pair

struct WideRate {
    uint32_t F, T;  // reduced rates (an unused slot: F = 0, T = 1)
    uint32_t r0;    // (m0 F) mod T
    float Tf;
};
struct WideDesc {
    const float *data;  // the frame that holds the first tap of the launch's first output frame
    uint32_t ch;        // channels of the source
    uint32_t frames;    // output frames of this launch the source reaches (0: a slot that pads the table to whole groups)
    uint32_t last;      // ended sources: index (relative to data) of the LAST frame; live ones: 0xffffffff
    uint32_t rate;      // index into WideTable::r
    float gain;
    float Tf;           // T of its rate as a float; 0: the converter passes through (F == T, sample_rate.rs:133-136)
};
struct WideTable {
    WideRate r[kWideRates];
    WideDesc d[kWideChunk];
};

// The sources are walked in groups of kWideGroup whose loads leave together: a lane's additions stay in insertion order (the reference's rounding
// sequence), but nothing about one source's taps depends on the sum so far.  A source that does not reach the sample -- it has ended, or the
// channel is one it does not have -- reads its own first float and adds +0.0, which leaves a sum that started at +0.0 as it is.
template <bool CONT>
__global__ __launch_bounds__(kBlock) void k_wide_mix(float *__restrict__ dst, uint32_t to_ch, uint32_t out_frames, const WideTable tbl, uint32_t n_sources) {
    const uint64_t total = (uint64_t)out_frames * to_ch;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    {
        // The table lies in the kernel-argument segment and is read through the scalar cache, descriptor after descriptor: the first wave of
        // a CU would miss on every line of it in turn (measured: 0.39 us per source, 12.5 us for a table of 32 -- whatever the block's length).
        // One word of every line is asked for HERE, all requests in flight at once: one round trip, and the walk below hits.
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&tbl);
        constexpr uint32_t kLines = (sizeof(WideTable) + 63) / 64;
        uint32_t t[kLines], touch = 0;
#pragma unroll
        for (uint32_t q = 0; q < kLines; ++q) t[q] = w[q * 16u];
#pragma unroll
        for (uint32_t q = 0; q < kLines; ++q) touch |= t[q];
        asm volatile("" ::"s"(touch));
    }
    for (uint64_t o = (uint64_t)blockIdx.x * kBlock + threadIdx.x; o < total; o += stride) {
        const uint32_t j = (uint32_t)(o / to_ch);
        const uint32_t c = (uint32_t)(o - (uint64_t)j * to_ch);
        uint32_t il[kWideRates];
        float w[kWideRates];  // num as a float (math.rs:25 multiplies by it, then divides by T)
#pragma unroll
        for (uint32_t q = 0; q < kWideRates; ++q) {
            const uint32_t p = tbl.r[q].r0 + j * tbl.r[q].F;  // (host: fits 32 bits)
            il[q] = p / tbl.r[q].T;
            w[q] = (float)(p - il[q] * tbl.r[q].T);
        }
        float acc = CONT ? dst[o] : 0.0f;
        // (two groups in flight -- the taps of the group behind requested before the group in front is added -- and groups of eight were
        //  measured: no faster.  At block sizes a launch is a few waves per CU walking the table; 0.27 us per source is their instruction stream.)
        for (uint32_t s0 = 0; s0 < n_sources; s0 += kWideGroup) {  // (host: n_sources is a whole number of groups)
            float a[kWideGroup], b[kWideGroup], wv[kWideGroup], Tf[kWideGroup], g[kWideGroup];
            bool on[kWideGroup], lerp[kWideGroup];
#pragma unroll
            for (int u = 0; u < kWideGroup; ++u) {
                const WideDesc d = tbl.d[s0 + u];
                const uint32_t q = d.rate;
                const uint32_t i = q == 0 ? il[0] : (q == 1 ? il[1] : (q == 2 ? il[2] : il[3]));
                wv[u] = q == 0 ? w[0] : (q == 1 ? w[1] : (q == 2 ? w[2] : w[3]));
                Tf[u] = d.Tf;
                g[u] = d.gain;
                const uint32_t k = c < d.ch ? c : 0u;                         // channels.rs:59-70: k < from: the input channel; k == 1 of a mono source: its only one;
                on[u] = j < d.frames && (c < d.ch || (c == 1u && d.ch == 1u));  // otherwise 0.0
                lerp[u] = on[u] && d.Tf != 0.0f && i < d.last;
                const float *pa = d.data + (on[u] ? (uint64_t)i * d.ch + k : 0ull);
                a[u] = *pa;
                b[u] = pa[lerp[u] ? d.ch : 0u];
            }
#pragma unroll
            for (int u = 0; u < kWideGroup; ++u) {
                const float x = a[u] * g[u];  // Amplify (amplify.rs:64) in front of the converter
                float v = x;
                if (lerp[u]) {
                    const float y = b[u] * g[u];
                    v = x + (y - x) * wv[u] / Tf[u];
                }
                acc += on[u] ? v : 0.0f;
            }
        }
        dst[o] = acc;
    }
}

// ---- the uniform case: every source of the launch has the mixer's channel count and ONE rate, and reaches every frame of the launch with both
// taps (no source ends inside it).  Then the tap offset and the weight are the lane's own, once, and a source costs two loads, the lerp and
// the add: what a mixer of many alike sources runs (k_wide_mix spends ~35 vector instructions per source and sample on the cases it must
// tell apart, which is what bounds it at scale: 0.41 of the roofline for 256 stereo sources).  The same operations in the same order.
struct WideUniSrc {
    const float *data;
    float gain;
    uint32_t pad;
};
struct WideUniTable {
    uint32_t F, T, r0, ch;
    float Tf;
    uint32_t pad[3];
    WideUniSrc d[kWideChunk];
};
constexpr int kUniGroup = 8;  // sources whose taps are in flight together (sixteen loads a lane; four: 0.44 of the roofline at 64 x 64 Ki frames of 5.1, eight: see profiles/r06_bench_wide.txt)
template <bool CONT, bool LERP>
__global__ __launch_bounds__(kBlock) void k_wide_mix_uniform(float *__restrict__ dst, uint32_t out_frames, const WideUniTable tbl, uint32_t n_sources) {
    const uint32_t ch = tbl.ch;
    const uint64_t total = (uint64_t)out_frames * ch;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t o = (uint64_t)blockIdx.x * kBlock + threadIdx.x; o < total; o += stride) {
        const uint32_t j = (uint32_t)(o / ch);
        const uint32_t c = (uint32_t)(o - (uint64_t)j * ch);
        const uint32_t p = tbl.r0 + j * tbl.F;
        const uint32_t il = LERP ? p / tbl.T : j;
        const float w = LERP ? (float)(p - il * tbl.T) : 0.0f;
        const uint64_t off = (uint64_t)il * ch + c;
        float acc = CONT ? dst[o] : 0.0f;
        uint32_t nv = n_sources;
        asm volatile("" : "+v"(nv));  // (a vector register: "is this slot a source" below is a select, not a branch that would split the group's loads)
        for (uint32_t s0 = 0; s0 < n_sources; s0 += kUniGroup) {  // (the slots behind the last source of a short last group repeat the first source: read, not added)
            float a[kUniGroup], b[kUniGroup], g[kUniGroup];
#pragma unroll
            for (int u = 0; u < kUniGroup; ++u) {
                const WideUniSrc d = tbl.d[s0 + u];
                g[u] = d.gain;
                a[u] = d.data[off];
                b[u] = LERP ? d.data[off + ch] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < kUniGroup; ++u) {
                const float x = a[u] * g[u];
                float v = x;
                if (LERP) {
                    const float y = b[u] * g[u];
                    v = x + (y - x) * w / tbl.Tf;
                }
                acc += s0 + u < nv ? v : 0.0f;  // (+ 0.0 leaves a sum that started at + 0.0 as it is)
            }
        }
        dst[o] = acc;
    }
}

}  // namespace

rh_status rh_wide_mix_block(float *dst, uint32_t channels, uint32_t to_rate, uint64_t out_frames, const rh_wide_src *srcs_host, uint32_t n_sources, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (out_frames == 0) return RH_OK;
    if (!dst || channels == 0 || to_rate == 0 || (n_sources && !srcs_host)) return RH_ERR_INVALID;
    if (out_frames > 0x7fffffffull) return RH_ERR_UNSUPPORTED;
    struct Red {
        uint32_t F, T;
    };
    std::vector<Red> red(n_sources);
    uint64_t step = out_frames;  // output frames per launch: r0 + j F must fit 32 bits for every source
    for (uint32_t s = 0; s < n_sources; ++s) {
        const rh_wide_src &x = srcs_host[s];
        if (x.frames == 0) continue;
        if (!x.data || x.channels == 0 || x.from_rate == 0 || x.frames > out_frames) return RH_ERR_INVALID;
        const uint32_t g = std::gcd(x.from_rate, to_rate);
        red[s] = Red{x.from_rate / g, to_rate / g};
        if ((uint64_t)red[s].F * red[s].T > 0xffffffffull) return RH_ERR_UNSUPPORTED;  // the reference multiplies in u32 (sample_rate.rs:157,173)
        if (x.phase >= red[s].T) return RH_ERR_INVALID;
        const uint64_t fit = (0xffffffffull - red[s].T) / red[s].F;
        if (fit < step) step = fit;
    }
    if (step == 0) return RH_ERR_UNSUPPORTED;
    // Alike sources (one rate and phase, the mixer's layout): the frames in front of the first one that ends -- or whose last frame a tap reaches --
    // go to the uniform kernel as a launch of their own, the rest to the general one.
    uint64_t j_uni = 0;
    {
        bool alike = true;
        int first = -1;
        uint64_t upto = out_frames;
        for (uint32_t s = 0; s < n_sources && alike; ++s) {
            const rh_wide_src &x = srcs_host[s];
            if (x.frames == 0) continue;
            if (first < 0) first = (int)s;
            alike = x.channels == channels && red[s].F == red[(size_t)first].F && red[s].T == red[(size_t)first].T && x.phase == srcs_host[first].phase;
            uint64_t u = x.frames;
            if (x.last != 0xffffffffu && red[s].F != red[s].T) {  // il(j) >= last  <=>  phase + j F >= last T
                const uint64_t need = (uint64_t)x.last * red[s].T;
                const uint64_t jl = need > x.phase ? (need - x.phase + red[s].F - 1) / red[s].F : 0;
                if (jl < u) u = jl;
            }
            if (u < upto) upto = u;
        }
        if (alike && first >= 0 && upto >= 1024 && upto < out_frames) j_uni = upto;
    }
    for (uint64_t j0 = 0; j0 < out_frames;) {
        const uint64_t seg_end = j0 < j_uni ? j_uni : out_frames;
        const uint32_t nf = (uint32_t)(seg_end - j0 < step ? seg_end - j0 : step);
        WideTable tbl;
        uint32_t k = 0, nr = 0;
        bool cont = false;
        auto clear_rates = [&]() {
            for (uint32_t q = 0; q < kWideRates; ++q) tbl.r[q] = WideRate{0u, 1u, 0u, 1.0f};
            nr = 0;
        };
        clear_rates();
        auto launch = [&]() {
            {  // the uniform case (k_wide_mix_uniform): one rate, the mixer's layout, no source that ends inside the launch
                bool uni = k > 0 && nr == 1;
                const uint64_t il_max = ((uint64_t)tbl.r[0].r0 + (uint64_t)(nf - 1) * tbl.r[0].F) / tbl.r[0].T;
                const bool lerp = tbl.r[0].F != tbl.r[0].T;
                for (uint32_t q = 0; q < k && uni; ++q) uni = tbl.d[q].ch == channels && tbl.d[q].frames == nf && (!lerp || il_max < tbl.d[q].last);
                if (uni && !rh::knob(rh::K_WIDE_GENERAL)) {
                    WideUniTable u;
                    u.F = tbl.r[0].F, u.T = tbl.r[0].T, u.r0 = tbl.r[0].r0, u.ch = channels, u.Tf = tbl.r[0].Tf;
                    u.pad[0] = u.pad[1] = u.pad[2] = 0;
                    for (uint32_t q = 0; q < kWideChunk; ++q) u.d[q] = WideUniSrc{tbl.d[q < k ? q : 0].data, q < k ? tbl.d[q].gain : 0.0f, 0u};
                    const uint64_t total = (uint64_t)nf * channels;
                    const dim3 grid(rh::grid_for((size_t)total, kBlock, 256u * 16u));
                    float *d = dst + j0 * channels;
                    hipStream_t st = rh::as_stream(stream);
                    if (cont && lerp) hipLaunchKernelGGL((k_wide_mix_uniform<true, true>), grid, dim3(kBlock), 0, st, d, nf, u, k);
                    else if (cont) hipLaunchKernelGGL((k_wide_mix_uniform<true, false>), grid, dim3(kBlock), 0, st, d, nf, u, k);
                    else if (lerp) hipLaunchKernelGGL((k_wide_mix_uniform<false, true>), grid, dim3(kBlock), 0, st, d, nf, u, k);
                    else hipLaunchKernelGGL((k_wide_mix_uniform<false, false>), grid, dim3(kBlock), 0, st, d, nf, u, k);
                    cont = true;
                    k = 0;
                    clear_rates();
                    return;
                }
            }
            while (k % kWideGroup) {  // whole groups: slots that reach no frame (and point at something readable: the first source's first tap)
                WideDesc &d = tbl.d[k++];
                d = WideDesc{tbl.d[0].data, 1u, 0u, 0u, 0u, 0.0f, 0.0f};
            }
            const uint64_t total = (uint64_t)nf * channels;
            const dim3 grid(rh::grid_for((size_t)total, kBlock, 256u * 16u));
            float *d = dst + j0 * channels;
            if (cont) hipLaunchKernelGGL(k_wide_mix<true>, grid, dim3(kBlock), 0, rh::as_stream(stream), d, channels, nf, tbl, k);
            else hipLaunchKernelGGL(k_wide_mix<false>, grid, dim3(kBlock), 0, rh::as_stream(stream), d, channels, nf, tbl, k);
            cont = true;
            k = 0;
            clear_rates();
        };
        for (uint32_t s = 0; s < n_sources; ++s) {
            const rh_wide_src &x = srcs_host[s];
            if (x.frames <= j0) continue;
            const uint64_t p = (uint64_t)x.phase + j0 * red[s].F;  // position of frame j0 between its taps, from the block's first tap on
            const uint64_t i0 = p / red[s].T;
            const uint32_t r0 = (uint32_t)(p - i0 * red[s].T);
            uint32_t q = 0;
            while (q < nr && !(tbl.r[q].F == red[s].F && tbl.r[q].T == red[s].T && tbl.r[q].r0 == r0)) ++q;
            if (q == nr && nr == kWideRates) {  // a fifth rate: what has been gathered goes first (the sum continues from the stored partial sum)
                launch();
                q = 0;
            }
            if (q == nr) tbl.r[nr++] = WideRate{red[s].F, red[s].T, r0, (float)red[s].T};
            WideDesc &d = tbl.d[k++];
            d.data = x.data + i0 * x.channels;
            d.ch = x.channels;
            d.frames = (uint32_t)(x.frames - j0 < nf ? x.frames - j0 : nf);
            d.last = x.last == 0xffffffffu ? 0xffffffffu : (x.last >= i0 ? (uint32_t)(x.last - i0) : 0u);
            d.rate = q;
            d.gain = x.gain;
            d.Tf = red[s].F == red[s].T ? 0.0f : (float)red[s].T;
            if (k == kWideChunk) launch();
        }
        if (k || !cont) launch();  // (no source reaches these frames: the mix is +0.0 there)
        RH_CHECK_LAUNCH();
        j0 += nf;
    }
    return RH_OK;
}

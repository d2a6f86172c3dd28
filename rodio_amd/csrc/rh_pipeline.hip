// rh_pipeline.hip -- the headline kernel (BASELINE config 2): for every source
//       mixer.add(UniformSourceIterator::new(src, 2, to_rate).low_pass(freq))
// followed by the ordered mixer sum, as ONE launch that reads each input byte once.
//
// Replaces, per output sample, the reference call stack of SURVEY.md 3.1:
//   MixerSource::next            src/mixer.rs:120-136, :185-198   (ordered f32 sum)
//   BltFilter::next              src/source/blt.rs:397-451, :558-560 (biquad, Direct Form I)
//   UniformSourceIterator::next  src/source/uniform.rs:78-97        (span chunking)
//   SampleRateConverter::next    src/conversions/sample_rate.rs:131-201, src/math.rs:23-26
//
// Work decomposition (wave64, no MFMA -- there is no contraction here):
//   * workgroup = one tile of L = threads*R consecutive OUTPUT frames, for ALL sources;
//     lane = a run of R consecutive frames.  The workgroup walks the sources in insertion
//     order and keeps the mix accumulators (R stereo frames) in registers, so the mixer sum
//     costs no memory traffic and keeps the reference's source order.
//   * input frames of (source, tile) are contiguous in HBM: they are fetched with 16-byte
//     coalesced loads one source ahead (register staged) into a double-buffered LDS tile;
//     the lerp taps are LDS reads.  HBM traffic = input once + mixed output once.
//   * the biquad is a linear recurrence along time.  Each lane runs it over its run from a
//     zero y-state, the end states are combined with a wave64 Kogge-Stone scan over the
//     2x2 companion-matrix powers A^(R*2^k) (host-computed in f64), waves are chained
//     through LDS, and tiles are chained through HBM "granules" ({epoch,value} 8-byte
//     words written with one agent-scope relaxed store each; cdna_hip_programming.md G16
//     form R2).  A tile needs only the zero-state aggregates of its J predecessors, where
//     J is the number of tiles after which ||A^(L*J)|| < 2^-40 (the filter is stable, so
//     older history is below f32 resolution): no chained inclusive prefix, hence no
//     serial dependency along the 500+ tiles.  The correction g1[r]*S1 + g2[r]*S2
//     (homogeneous response to the true start state S) is added D sources later, which
//     hides the hand-off latency behind the next sources' streaming.
//   * tiles are numbered by an atomic ticket, so a tile only ever waits for tiles that
//     already hold a CU: progress does not depend on dispatch order or residency.
#include <cmath>
#include <cstring>
#include <numeric>
#include <type_traits>
#include <vector>

#include "rh_common.h"

namespace rh {
struct ResampleGeom {
    uint32_t F, T;
    uint64_t in_frames, chunk_in, chunk_out, n_chunks, last_in, out_frames;
    int fits32;
};
rh_status make_resample_geom(uint64_t in_frames, uint32_t from_rate, uint32_t to_rate, uint32_t channels, uint64_t span_len, ResampleGeom *g);
}  // namespace rh

namespace {

constexpr int kMaxR = 32;
constexpr int kMaxThreads = 512;
constexpr int kMaxLook = 64;
constexpr uint32_t kSpinLimit = 1u << 16;  // x (~1 us load + s_sleep): ~0.1 s, then give up for good
constexpr int kHeaderBytes = 320;  // wagg[2][8][4] f32 (256 B) + cbuf[2][4] f32 (32 B) + misc (32 B)

struct SrcDesc {
    const float *data;
    uint64_t frames;      // N_s
    uint64_t out_frames;  // M_s
};

// ---- the biquad as data -----------------------------------------------------------------------
// H(z) = b0 + (c1 z^-1 + c2 z^-2)/A(z) with c1 = b1 - b0*a1, c2 = b2 - b0*a2.  The recursive
// part w = y - b0*x is what is scanned along time: w is smooth whenever the poles sit near
// z = 1 (also for a high-pass, whose y is not), so its zero-state run and its homogeneous
// correction stay of the magnitude of w instead of cancelling large terms.
//
// The scan works on z = Tm * (w[n-1], w[n-2]) rather than on the companion state itself.  For
// the double real pole p of rodio's default q = 0.5 (blt.rs:11-16) Tm = [[1,-p],[0,1]] turns the
// companion matrix into [[p,0],[1,p]], whose powers [[p^n,0],[n p^(n-1),p^n]] multiply the SMALL
// component z1 = w1 - p*w2 by the large entry; for complex poles rho*e^(+-j*theta),
// Tm = [[1,-rho cos],[0,rho sin]] gives rho*Rotation(theta), a normal matrix.  In that basis
// every table below is benign in f32; in the companion basis the same algebra needs f64 (measured:
// 10-30x the reference's own f32 error).  All tables are powers of B = Tm A Tm^-1 computed on the
// host in f64 and rounded once.
struct Uniforms {
    float b0, c1, c2, a1, a2;
    float Tm[4];                  // (w1,w2) -> z
    float scanM[4][4];            // B^(R*2^k), k = 0..3   (row_shr 1,2,4,8)
    float waveM[4];               // B^(64R)
};
struct Tables {                    // per-lane tables (loaded once per lane)
    float g[kMaxR][2];             // row 0 of A^(r+1) Tm^-1: homogeneous response of w inside a run
    float bc15M[64][4];            // B^(R*((lane&15)+1))   (row_bcast:15 step)
    float bc31M[64][4];            // B^(R*((lane&31)+1))   (row_bcast:31 step)
    float laneM[64][4];            // B^(R*lane)
    float carryM[kMaxThreads][4];  // B^(R*tid)
    float lookM[kMaxLook][4];      // B^(L*j)
};

struct Params {
    const SrcDesc *srcs;
    const Tables *tabs;
    float *out;
    unsigned long long *gran;  // [S][tiles][4] {epoch, f32 bits}
    uint32_t *ticket;
    uint32_t *status;
    uint64_t out_frames;
    uint64_t chunk_in, chunk_out;  // chunk_out == 0: unchunked
    uint32_t n_sources, n_tiles;
    uint32_t F, T, qF, rF;
    float Tf, rcpT;
    uint32_t epoch, J;
    uint32_t stage_bytes;  // bytes of one LDS input stage
    uint32_t ticket_base;  // value of *ticket when this launch starts (the counter is never reset)
    Uniforms u;
};

struct Cursor {
    uint64_t k, ml, il;
    uint32_t num;
};
__device__ __forceinline__ Cursor cursor_at(uint64_t m, const Params &p) {
    Cursor c;
    c.k = p.chunk_out ? m / p.chunk_out : 0;
    c.ml = m - c.k * p.chunk_out;
    const uint64_t pp = c.ml * p.F;
    c.il = pp / p.T;
    c.num = (uint32_t)(pp - c.il * p.T);
    return c;
}
__device__ __forceinline__ void cursor_next(Cursor &c, const Params &p) {
    c.ml += 1;
    if (p.chunk_out && c.ml == p.chunk_out) {  // uniform.rs:56-67: the converter restarts
        c.k += 1;
        c.ml = 0;
        c.il = 0;
        c.num = 0;
    } else {
        c.il += p.qF;
        c.num += p.rF;
        if (c.num >= p.T) {
            c.num -= p.T;
            c.il += 1;
        }
    }
}
// Global input frame index + lerp numerator.  At the last frame of a chunk the reference
// emits the frame verbatim (sample_rate.rs:193-200): numerator 0 gives exactly that.
__device__ __forceinline__ void cursor_resolve(const Cursor &c, const Params &p, uint64_t &i, uint32_t &num) {
    uint64_t il = c.il;
    num = c.num;
    if (p.chunk_out && il + 1 >= p.chunk_in) {
        il = p.chunk_in - 1;
        num = 0;
    }
    i = c.k * p.chunk_in + il;
}

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Correctly rounded t / T with a host-side correctly rounded reciprocal (Markstein): q0 within
// 1 ulp, exact residual by FMA, one correction.  Same value as the IEEE divide in math.rs:25.
__device__ __forceinline__ float div_T(float t, float Tf, float rcpT) {
    const float q0 = t * rcpT;
    const float rem = fma_(-q0, Tf, t);
    return fma_(rem, rcpT, q0);
}

// y += M * x for a row-major 2x2
__device__ __forceinline__ void mat_acc(const float *M, float x1, float x2, float &y1, float &y2) {
    y1 = fma_(M[0], x1, fma_(M[1], x2, y1));
    y2 = fma_(M[2], x1, fma_(M[3], x2, y2));
}

// Cross-lane moves on the VALU data path (DPP), no LDS round trip.  Lanes whose source is out of
// range, or whose row is masked off, read 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
constexpr int kDppRowShr = 0x110;    // row_shr:n  = 0x110 + n
constexpr int kDppWaveShr1 = 0x138;  // wave_shr:1
constexpr int kDppBcast15 = 0x142;   // lane 15 of each row -> the next row
constexpr int kDppBcast31 = 0x143;   // lane 31 -> rows 2 and 3

#define RH_LDS __attribute__((address_space(3)))
typedef RH_LDS unsigned char lds_u8;
typedef RH_LDS float lds_f32;
typedef RH_LDS uint32_t lds_u32;
typedef float v2f __attribute__((ext_vector_type(2)));  // native vectors: HIP's float2/float4 classes
typedef float v4f __attribute__((ext_vector_type(4)));  // cannot be read through address-space pointers
typedef RH_LDS v2f lds_f2;
typedef RH_LDS v4f lds_f4;
#define RH_GLB __attribute__((address_space(1)))
typedef RH_GLB const float glb_cf32;  // a pointer loaded from a descriptor is generic: say it is global,
typedef RH_GLB const v2f glb_cf2;     // or every source load is a flat_load that also blocks lgkmcnt
typedef RH_GLB const v4f glb_cf4;

template <int R, int KV, int D, bool FILT>
__global__ __launch_bounds__(kMaxThreads) void k_rlm_stereo(const Params p) {
    static_assert(R % 2 == 0 && R <= kMaxR, "R must be even");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // explicit LDS address space: a generic pointer here turns every tap into a flat_load
    lds_u8 *const lds = (lds_u8 *)smem;
    lds_f32 *const wagg = (lds_f32 *)lds;                // [2][8][4]
    lds_f32 *const cbuf = (lds_f32 *)(lds + 256);        // [2][4]
    lds_u32 *const misc = (lds_u32 *)(lds + 288);        // ticket
    lds_u8 *const inbuf = lds + kHeaderBytes;            // [2][stage_bytes]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = blockDim.x, W = NT >> 6;

    if (tid == 0) misc[0] = atomicAdd(p.ticket, 1u) - p.ticket_base;
    __syncthreads();
    const uint32_t tile = misc[0];
    const uint64_t L = (uint64_t)NT * R;
    const uint64_t m_tile0 = (uint64_t)tile * L;
    const uint64_t m0 = m_tile0 + (uint64_t)tid * R;
    const bool first = (m0 == 0);  // stream start: x'[-1] = x'[-2] = 0

    // ---- input span of this tile (identical for every source) -------------------------------
    uint64_t i_base, i_end;
    {
        uint32_t nn;
        cursor_resolve(cursor_at(m_tile0 >= 2 ? m_tile0 - 2 : 0, p), p, i_base, nn);
        i_base &= ~1ull;  // 16-byte aligned vectors
        cursor_resolve(cursor_at(m_tile0 + L - 1, p), p, i_end, nn);
        i_end += 1;
    }
    uint32_t nvec = (uint32_t)((i_end - i_base + 2) / 2);
    if (nvec > (uint32_t)(KV * NT)) nvec = KV * NT;  // host sizes KV so this never bites

    // ---- per-lane tap table: LDS byte offset of frame i(m) and the lerp weight ----------------
    // FILT: weight = num/T (the lerp becomes one FMA, <= 1.5 ulp from math.rs:25 -- far inside
    // the 1e-5 budget of the filtered pipeline); !FILT: weight = num, divided exactly per sample
    // so that resample+mix alone stays bit-identical to the reference.
    int offA[R + 2];
    float wgt[R + 2];
    {
        Cursor c = cursor_at(first ? 0 : m0 - 2, p);
#pragma unroll
        for (int rr = 0; rr < R + 2; ++rr) {
            const bool dummy = first && rr < 2;
            uint64_t i;
            uint32_t num;
            cursor_resolve(c, p, i, num);
            offA[rr] = dummy ? 0 : (int)((i - i_base) * 8);
            wgt[rr] = dummy ? 0.0f : (FILT ? (float)num / p.Tf : (float)num);
            if (!dummy) cursor_next(c, p);
        }
    }

    const Tables *__restrict__ tb = p.tabs;
    float lM[4], cM[4], b15[4], b31[4];
    float g1v[R], g2v[R];  // homogeneous response of a run, in VGPRs (2R scalars would not fit the SGPR file)
    if (FILT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            lM[q] = tb->laneM[lane][q];
            cM[q] = tb->carryM[tid][q];
            b15[q] = tb->bc15M[lane][q];
            b31[q] = tb->bc31M[lane][q];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            g1v[r] = tb->g[r][0];
            g2v[r] = tb->g[r][1];
        }
    }
    const float b0 = p.u.b0, c1 = p.u.c1, c2 = p.u.c2, na1 = -p.u.a1, na2 = -p.u.a2;

    v2f acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = v2f{0.0f, 0.0f};
    float Qr[D][4];  // start-of-run states (zero tile carry) of the D sources in flight
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int q = 0; q < 4; ++q) Qr[d][q] = 0.0f;

    bool dead = false;  // a bounded wait expired: never spin again in this workgroup

    // ---- staging: source frames [i_base, i_base + 2*nvec) -> LDS stage, 16 bytes per lane -------
    // Fast path (the span lies inside the source): global_load_lds DMA, no VGPR round trip, issued
    // a whole source ahead.  Slow path (the tile touches the end of the source): register staged,
    // replicating the last frame, which is what makes the resampler's "last frame verbatim" rule
    // (sample_rate.rs:193-200) fall out of the plain lerp.
    auto stage_source = [&](const SrcDesc &sd, uint32_t stage) {
        lds_u8 *dstb = inbuf + stage * p.stage_bytes;
        glb_cf32 *base = (glb_cf32 *)sd.data;
        if (i_base + 2ull * nvec <= sd.frames) {
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                uint32_t j = tid + k * NT;
                j = j < nvec ? j : nvec - 1;  // surplus lanes re-fetch the last vector into unused slots
                __builtin_amdgcn_global_load_lds(base + (i_base + 2ull * j) * 2,
                                                 (RH_LDS void *)(dstb + (wave * 64 + k * NT) * 16), 16, 0, 0);
            }
        } else {
#pragma unroll 1
            for (int k = 0; k < KV; ++k) {
                const uint32_t j = tid + k * NT;
                const uint64_t f = i_base + 2ull * j;
                if (j < nvec) {
                    v4f v = {0.f, 0.f, 0.f, 0.f};
                    if (sd.frames) {
                        const uint64_t f0 = f < sd.frames ? f : sd.frames - 1;
                        const uint64_t f1 = f + 1 < sd.frames ? f + 1 : sd.frames - 1;
                        const v2f a = *(glb_cf2 *)(base + f0 * 2);
                        const v2f b = *(glb_cf2 *)(base + f1 * 2);
                        v = v4f{a.x, a.y, b.x, b.y};
                    }
                    *(lds_f4 *)(dstb + j * 16) = v;
                }
            }
        }
    };

    const uint32_t S = p.n_sources;
    SrcDesc sd_cur{nullptr, 0, 0}, sd_nxt{nullptr, 0, 0}, sd_nn{nullptr, 0, 0};  // descriptors of sources s, s+1, s+2
    if (S > 0) {
        sd_cur = p.srcs[0];
        if (sd_cur.out_frames > m_tile0) stage_source(sd_cur, 0);
        if (S > 1) {
            sd_nxt = p.srcs[1];
            if (sd_nxt.out_frames > m_tile0) stage_source(sd_nxt, 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    uint64_t Ms_hist[D];  // out_frames of the D sources in flight
#pragma unroll
    for (int d = 0; d < D; ++d) Ms_hist[d] = 0;

    const uint32_t n_iter = FILT ? S + D : S;
    for (uint32_t s = 0; s < n_iter; ++s) {
        float Qnew[4] = {0.f, 0.f, 0.f, 0.f};
        if (s + 2 < S) sd_nn = p.srcs[s + 2];  // needed right after the barrier
        const uint64_t Ms = s < S ? sd_cur.out_frames : 0;
        const bool active = Ms > m_tile0;  // this tile still holds frames of source s
        const uint64_t Mp = Ms_hist[0];
        const bool pactive = FILT && s >= (uint32_t)D && Mp > m_tile0;  // source s-D is finished in this iteration

        // ---- wave 0: start fetching the carry granules of source s-D (consumed after the run) --
        unsigned long long gv[4] = {0, 0, 0, 0};
        bool need = false;
        const unsigned long long *gp = p.gran;
        if (FILT && wave == 0 && pactive && tile > 0) {
            need = (uint32_t)lane < p.J && (uint32_t)lane < tile;
            gp = p.gran + ((uint64_t)(s - D) * p.n_tiles + (tile - 1 - (need ? lane : 0))) * 4;
            if (need && !dead) {
#pragma unroll
                for (int q = 0; q < 4; ++q) gv[q] = __hip_atomic_load(gp + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }

        if (active) {
            const lds_u8 *buf = inbuf + (s & 1) * p.stage_bytes;
            auto tap = [&](int rr) -> v2f {
                const v2f a = *(const lds_f2 *)(buf + offA[rr]);
                const v2f b = *(const lds_f2 *)(buf + offA[rr] + 8);
                v2f x;
                if (FILT) {
                    x.x = fma_(b.x - a.x, wgt[rr], a.x);
                    x.y = fma_(b.y - a.y, wgt[rr], a.y);
                } else {  // math.rs:25: first + (second - first) * num / den, exactly
                    x.x = a.x + div_T((b.x - a.x) * wgt[rr], p.Tf, p.rcpT);
                    x.y = a.y + div_T((b.y - a.y) * wgt[rr], p.Tf, p.rcpT);
                }
                return x;
            };
            // lanes past the end of the source contribute nothing (the reference's iterator ended)
            const int nvalid = Ms >= m0 + R ? R : (Ms > m0 ? (int)(Ms - m0) : 0);
            const bool full = Ms >= m_tile0 + L;  // uniform: no lane needs masking
            if (FILT) {
                v2f x2 = first ? v2f{0.f, 0.f} : tap(0);
                v2f x1 = first ? v2f{0.f, 0.f} : tap(1);
                v2f w1 = {0.f, 0.f}, w2 = {0.f, 0.f};
                auto run = [&](auto masked) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const v2f x = tap(r + 2);
                        v2f w;  // zero-state step of the recursive part; the w1 term goes last (shortest chain)
                        w.x = fma_(na1, w1.x, fma_(na2, w2.x, fma_(c2, x2.x, c1 * x1.x)));
                        w.y = fma_(na1, w1.y, fma_(na2, w2.y, fma_(c2, x2.y, c1 * x1.y)));
                        float yx = fma_(b0, x.x, w.x), yy = fma_(b0, x.y, w.y);
                        if (decltype(masked)::value) {
                            const bool v = r < nvalid;
                            yx = v ? yx : 0.0f;
                            yy = v ? yy : 0.0f;
                        }
                        acc[r].x += yx;
                        acc[r].y += yy;
                        w2 = w1;
                        w1 = w;
                        x2 = x1;
                        x1 = x;
                    }
                };
                if (full) run(std::false_type{});
                else run(std::true_type{});
                // ---- run end state in the scan basis, then the wave64 inclusive scan ----
                float P[4] = {0.f, 0.f, 0.f, 0.f};
                mat_acc(p.u.Tm, w1.x, w2.x, P[0], P[1]);
                mat_acc(p.u.Tm, w1.y, w2.y, P[2], P[3]);
#define RH_SCAN_STEP(K, N)                                                                          \
    {                                                                                               \
        const float q0 = dpp0<kDppRowShr + N, 0xf>(P[0]), q1 = dpp0<kDppRowShr + N, 0xf>(P[1]);     \
        const float q2 = dpp0<kDppRowShr + N, 0xf>(P[2]), q3 = dpp0<kDppRowShr + N, 0xf>(P[3]);     \
        mat_acc(p.u.scanM[K], q0, q1, P[0], P[1]);                                                  \
        mat_acc(p.u.scanM[K], q2, q3, P[2], P[3]);                                                  \
    }
                RH_SCAN_STEP(0, 1)
                RH_SCAN_STEP(1, 2)
                RH_SCAN_STEP(2, 4)
                RH_SCAN_STEP(3, 8)
#undef RH_SCAN_STEP
                {  // rows 1 and 3 take the inclusive prefix of the row before them
                    const float q0 = dpp0<kDppBcast15, 0xa>(P[0]), q1 = dpp0<kDppBcast15, 0xa>(P[1]);
                    const float q2 = dpp0<kDppBcast15, 0xa>(P[2]), q3 = dpp0<kDppBcast15, 0xa>(P[3]);
                    mat_acc(b15, q0, q1, P[0], P[1]);
                    mat_acc(b15, q2, q3, P[2], P[3]);
                }
                {  // rows 2 and 3 take the inclusive prefix of lanes 0..31
                    const float q0 = dpp0<kDppBcast31, 0xc>(P[0]), q1 = dpp0<kDppBcast31, 0xc>(P[1]);
                    const float q2 = dpp0<kDppBcast31, 0xc>(P[2]), q3 = dpp0<kDppBcast31, 0xc>(P[3]);
                    mat_acc(b31, q0, q1, P[0], P[1]);
                    mat_acc(b31, q2, q3, P[2], P[3]);
                }
                if (lane == 63) *(lds_f4 *)(wagg + ((s & 1) * 8 + wave) * 4) = v4f{P[0], P[1], P[2], P[3]};
#pragma unroll
                for (int q = 0; q < 4; ++q) Qnew[q] = dpp0<kDppWaveShr1, 0xf>(P[q]);  // exclusive: lane 0 gets 0
            } else {
                auto run = [&](auto masked) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        v2f x = tap(r + 2);
                        if (decltype(masked)::value && !(r < nvalid)) continue;  // an ended source adds nothing, not even +0.0
                        acc[r].x += x.x;
                        acc[r].y += x.y;
                    }
                };
                if (full) run(std::false_type{});
                else run(std::true_type{});
            }
        }
        // ---- wave 0: finish the carry of source s-D: c = sum_j B^(L*j) * aggregate(tile-1-j) ------
        if (FILT && wave == 0) {
            float c[4] = {0.f, 0.f, 0.f, 0.f};
            if (pactive && tile > 0) {
                bool ok = !need;
                if (need && !dead) {
                    ok = true;
#pragma unroll
                    for (int q = 0; q < 4; ++q) ok = ok && ((uint32_t)(gv[q] >> 32) == p.epoch);
                }
                uint32_t spins = 0;
                while (!dead && !__all(ok)) {  // rare: the neighbour is more than D sources behind
                    if (++spins > kSpinLimit) {
                        if (lane == 0) atomicOr(p.status, 1u);
                        dead = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                    if (!ok) {
                        bool all = true;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            gv[q] = __hip_atomic_load(gp + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            all = all && ((uint32_t)(gv[q] >> 32) == p.epoch);
                        }
                        ok = all;
                    }
                }
                if (need && ok) {
                    const float *M = tb->lookM[lane];
                    mat_acc(M, __uint_as_float((uint32_t)gv[0]), __uint_as_float((uint32_t)gv[1]), c[0], c[1]);
                    mat_acc(M, __uint_as_float((uint32_t)gv[2]), __uint_as_float((uint32_t)gv[3]), c[2], c[3]);
                }
                if (p.J > 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int d = 32; d >= 1; d >>= 1) c[q] += __shfl_xor(c[q], d);
                    }
                }
            }
            if (lane == 0) *(lds_f4 *)(cbuf + (s & 1) * 4) = v4f{c[0], c[1], c[2], c[3]};
        }
        // the DMA of source s+1 (issued a whole iteration ago) must have landed before the barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // stage (s&1) is free now: start fetching source s+2 into it
        sd_cur = sd_nxt;
        sd_nxt = sd_nn;
        if (s + 2 < S && sd_nxt.out_frames > m_tile0) stage_source(sd_nxt, s & 1);
        if (FILT) {
            if (active) {
                // chain the waves: state at the start of this wave (zero tile carry)
                float Wst[4] = {0.f, 0.f, 0.f, 0.f};
                const lds_f32 *wa = wagg + (s & 1) * 32;
                for (int u = 0; u < wave; ++u) {
                    const v4f n = *(const lds_f4 *)(wa + u * 4);
                    float n0 = n.x, n1 = n.y, n2 = n.z, n3 = n.w;
                    mat_acc(p.u.waveM, Wst[0], Wst[1], n0, n1);
                    mat_acc(p.u.waveM, Wst[2], Wst[3], n2, n3);
                    Wst[0] = n0; Wst[1] = n1; Wst[2] = n2; Wst[3] = n3;
                }
                if (wave == W - 1 && lane < 4) {  // publish the tile aggregate: 4 granules
                    const v4f n = *(const lds_f4 *)(wa + wave * 4);
                    float e0 = n.x, e1 = n.y, e2 = n.z, e3 = n.w;
                    mat_acc(p.u.waveM, Wst[0], Wst[1], e0, e1);
                    mat_acc(p.u.waveM, Wst[2], Wst[3], e2, e3);
                    const float ev = lane == 0 ? e0 : lane == 1 ? e1 : lane == 2 ? e2 : e3;
                    const unsigned long long word = ((unsigned long long)p.epoch << 32) | __float_as_uint(ev);
                    __hip_atomic_store(p.gran + ((uint64_t)s * p.n_tiles + tile) * 4 + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                mat_acc(lM, Wst[0], Wst[1], Qnew[0], Qnew[1]);
                mat_acc(lM, Wst[2], Wst[3], Qnew[2], Qnew[3]);
            }
            if (pactive) {  // finish source s-D: homogeneous response to its true start state
                const int nvalid = Mp >= m0 + R ? R : (Mp > m0 ? (int)(Mp - m0) : 0);
                const bool full = Mp >= m_tile0 + L;
                const v4f cb = *(const lds_f4 *)(cbuf + (s & 1) * 4);
                float S0 = Qr[0][0], S1 = Qr[0][1], S2 = Qr[0][2], S3 = Qr[0][3];
                mat_acc(cM, cb.x, cb.y, S0, S1);
                mat_acc(cM, cb.z, cb.w, S2, S3);
                if (!full) {  // zero the start state of lanes with no valid frame; partial lanes are masked per frame
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const bool v = r < nvalid;
                        const float hx = fma_(g1v[r], S0, g2v[r] * S1), hy = fma_(g1v[r], S2, g2v[r] * S3);
                        acc[r].x += v ? hx : 0.0f;
                        acc[r].y += v ? hy : 0.0f;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        acc[r].x = fma_(g1v[r], S0, fma_(g2v[r], S1, acc[r].x));
                        acc[r].y = fma_(g1v[r], S2, fma_(g2v[r], S3, acc[r].y));
                    }
                }
            }
#pragma unroll
            for (int d = 0; d + 1 < D; ++d) {
                Ms_hist[d] = Ms_hist[d + 1];
#pragma unroll
                for (int q = 0; q < 4; ++q) Qr[d][q] = Qr[d + 1][q];
            }
            Ms_hist[D - 1] = Ms;
#pragma unroll
            for (int q = 0; q < 4; ++q) Qr[D - 1][q] = Qnew[q];
        }
    }

    // ---- mixed output: R stereo frames per lane, 16-byte stores ---------------------------------
    float *o = p.out + m0 * 2;
#pragma unroll
    for (int r = 0; r < R; r += 2) {
        const uint64_t m = m0 + r;
        if (m + 1 < p.out_frames) {
            *reinterpret_cast<float4 *>(o + r * 2) = make_float4(acc[r].x, acc[r].y, acc[r + 1].x, acc[r + 1].y);
        } else if (m < p.out_frames) {
            *reinterpret_cast<float2 *>(o + r * 2) = make_float2(acc[r].x, acc[r].y);
        }
    }
}

// ------------------------------------------------------------------ host side ----
struct M2 {
    double a, b, c, d;
};
M2 mul(const M2 &x, const M2 &y) { return {x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d}; }
M2 mpow(M2 base, uint64_t e) {
    M2 r{1, 0, 0, 1};
    while (e) {
        if (e & 1) r = mul(r, base);
        base = mul(base, base);
        e >>= 1;
    }
    return r;
}
void put(float *dst, const M2 &m) {
    dst[0] = (float)m.a;
    dst[1] = (float)m.b;
    dst[2] = (float)m.c;
    dst[3] = (float)m.d;
}
double norm(const M2 &m) { return std::fabs(m.a) + std::fabs(m.b) + std::fabs(m.c) + std::fabs(m.d); }

// Scan basis for the companion matrix of z^2 + a1 z + a2 (see the comment above Uniforms).
void scan_basis(double a1, double a2, M2 &T, M2 &Tinv) {
    const double disc = a1 * a1 - 4.0 * a2;
    const double re = -0.5 * a1;
    double mu, nu;
    if (disc < 0.0 && std::sqrt(-disc) * 0.5 > 1e-3) {  // complex pair rho e^(+-j theta)
        mu = re;                                          // rho cos(theta)
        nu = std::sqrt(-disc) * 0.5;                      // rho sin(theta)
    } else {  // real poles (or a numerically double one): peel off the smaller pole
        const double sq = disc > 0.0 ? std::sqrt(disc) * 0.5 : 0.0;
        const double l1 = re + sq, l2 = re - sq;
        mu = std::fabs(l1) < std::fabs(l2) ? l1 : l2;
        nu = 1.0;
    }
    T = {1.0, -mu, 0.0, nu};
    Tinv = {1.0, mu / nu, 0.0, 1.0 / nu};
}

constexpr int kD = 2;  // sources in flight between publishing an aggregate and consuming the carry

using KernelFn = void (*)(const Params);
struct Variant {
    int R, KV;
    KernelFn filt, plain;
};
#define RH_VARIANT(r, kv) Variant{r, kv, &k_rlm_stereo<r, kv, kD, true>, &k_rlm_stereo<r, kv, kD, false>}
// KV = R/2+1 covers from <= to (upsampling, staged span <= L+7 frames); KV = R+1 covers from <= 2*to.
const Variant kVariants[] = {
    RH_VARIANT(4, 3),  RH_VARIANT(4, 5),  RH_VARIANT(6, 4),  RH_VARIANT(6, 7),   RH_VARIANT(8, 5),
    RH_VARIANT(8, 9),  RH_VARIANT(12, 7), RH_VARIANT(12, 13), RH_VARIANT(16, 9), RH_VARIANT(16, 17),
};
#undef RH_VARIANT
const void *find_kernel(int R, int KV, bool filt) {
    for (const Variant &v : kVariants)
        if (v.R == R && v.KV == KV) return reinterpret_cast<const void *>(filt ? v.filt : v.plain);
    return nullptr;
}
// Workgroups of `threads` lanes + `lds` dynamic bytes the hardware co-schedules on one CU.
int blocks_per_cu(const void *fn, int threads, size_t lds) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess) return 0;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, threads, lds) != hipSuccess) return 0;
    return n;
}

}  // namespace

struct rh_rlm {
    rh_rlm_config cfg;
    uint32_t F, T;
    uint64_t chunk_in, chunk_out;  // chunk_out == 0: unchunked
    bool filt;
    float coeffs[5];
    int R, KV, threads;
    const void *kernel = nullptr;
    int resident_per_cu = 0;
    uint32_t stage_bytes, lds_bytes, J;
    Tables *d_tabs = nullptr;
    Uniforms uni;
    SrcDesc *d_srcs = nullptr;
    unsigned long long *d_gran = nullptr;
    size_t gran_words = 0;
    uint32_t *d_ctl = nullptr;  // [0] ticket, [1] status
    uint32_t n_sources = 0, n_tiles = 0;
    uint64_t out_frames = 0;
    uint32_t epoch = 0;
    uint32_t ticket_base = 0;
};

extern "C" {

rh_status rh_rlm_create(rh_rlm **out, const rh_rlm_config *cfg) {
    RH_REQUIRE_INIT();
    if (!out || !cfg || cfg->from_rate == 0 || cfg->to_rate == 0 || cfg->channels == 0 || cfg->max_sources == 0) return RH_ERR_INVALID;
    if (cfg->channels != 2) return RH_ERR_UNSUPPORTED;
    if (cfg->from_rate == cfg->to_rate) return RH_ERR_UNSUPPORTED;  // passthrough converter: use rh_biquad + rh_mix_sum
    rh::ResampleGeom g;
    rh_status st = rh::make_resample_geom(cfg->max_in_frames, cfg->from_rate, cfg->to_rate, cfg->channels, cfg->span_len, &g);
    if (st != RH_OK) return st;
    if (g.F > 2 * g.T) return RH_ERR_UNSUPPORTED;  // staging is sized for ratios <= 2 (unfused ops cover the rest)
    rh_rlm *p = new rh_rlm();
    p->cfg = *cfg;
    p->F = g.F;
    p->T = g.T;
    p->chunk_in = g.n_chunks > 1 ? g.chunk_in : 0;
    p->chunk_out = g.n_chunks > 1 ? g.chunk_out : 0;
    p->filt = cfg->filter_kind >= 0;
    if (p->filt) {
        st = rh_biquad_coeffs(cfg->filter_kind, cfg->filter_freq, cfg->filter_q, cfg->to_rate, p->coeffs);
        if (st != RH_OK) {
            delete p;
            return st;
        }
    } else {
        p->coeffs[0] = 1.f;
        p->coeffs[1] = p->coeffs[2] = p->coeffs[3] = p->coeffs[4] = 0.f;
    }
    // ---- launch geometry: tiles of L = threads*R output frames.  All tiles advance source by
    // source in near lock-step (they exchange carries), so the cost of a geometry is the most
    // loaded CU: ceil(tiles / CUs) * L.  Among equals prefer ~2 workgroups per CU.
    const uint64_t M = g.out_frames ? g.out_frames : 1;
    const int cus = rh::g_num_cus;
    const int Rs[] = {4, 6, 8, 12, 16};
    double best = 1e300;
    int bestR = 8, bestT = 256;
    for (int R : Rs) {
        if (cfg->frames_per_lane && (int)cfg->frames_per_lane != R) continue;
        for (int T = 128; T <= kMaxThreads; T += 64) {
            if (cfg->threads && (int)cfg->threads != T) continue;
            const uint64_t L = (uint64_t)R * T;
            const uint64_t tiles = (M + L - 1) / L;
            const uint64_t per_cu = (tiles + cus - 1) / cus;
            const int kv = (g.F <= g.T) ? R / 2 + 1 : R + 1;
            const size_t lds = kHeaderBytes + 2 * (size_t)kv * T * 16;
            if (lds > 150 * 1024) continue;
            const void *fn = find_kernel(R, kv, p->filt);
            if (!fn) continue;
            // every tile should hold a CU slot at once (they advance in lock-step)
            if ((int)per_cu > blocks_per_cu(fn, T, lds)) continue;
            double cost = (double)per_cu * (double)L;
            cost *= 1.0 + 0.02 * std::fabs((double)per_cu * T / 256.0 - 2.0);  // soft preference: 8 waves/CU
            cost *= 1.0 + 0.3 / R;                                             // scan overhead ~ 1/R
            if (cost < best) {
                best = cost;
                bestR = R;
                bestT = T;
            }
        }
    }
    if (best == 1e300) {
        bestR = cfg->frames_per_lane ? cfg->frames_per_lane : 8;
        bestT = cfg->threads ? cfg->threads : 256;
        bool okR = false;
        for (int R : Rs) okR = okR || R == bestR;
        if (!okR || bestT % 64 || bestT < 64 || bestT > kMaxThreads) {
            delete p;
            return RH_ERR_INVALID;
        }
    }
    p->R = bestR;
    p->threads = bestT;
    p->KV = (g.F <= g.T) ? bestR / 2 + 1 : bestR + 1;
    p->stage_bytes = (uint32_t)p->KV * bestT * 16;
    p->lds_bytes = kHeaderBytes + 2 * p->stage_bytes;
    p->kernel = find_kernel(p->R, p->KV, p->filt);
    if (!p->kernel || p->lds_bytes > 159 * 1024) {
        delete p;
        return RH_ERR_UNSUPPORTED;
    }
    p->resident_per_cu = blocks_per_cu(p->kernel, p->threads, p->lds_bytes);
    // ---- tables (all powers of B = Tm A Tm^-1, f64 on the host, rounded to f32 once) --------------
    Tables *h = new Tables();
    std::memset(h, 0, sizeof(Tables));
    Uniforms &U = p->uni;
    std::memset(&U, 0, sizeof(U));
    U.b0 = p->coeffs[0];
    U.c1 = (float)((double)p->coeffs[1] - (double)p->coeffs[0] * (double)p->coeffs[3]);
    U.c2 = (float)((double)p->coeffs[2] - (double)p->coeffs[0] * (double)p->coeffs[4]);
    U.a1 = p->coeffs[3];
    U.a2 = p->coeffs[4];
    const M2 A{-(double)p->coeffs[3], -(double)p->coeffs[4], 1.0, 0.0};
    M2 Tm, Ti;
    scan_basis((double)p->coeffs[3], (double)p->coeffs[4], Tm, Ti);
    const M2 B = mul(mul(Tm, A), Ti);
    put(U.Tm, Tm);
    const uint64_t R = bestR, L = (uint64_t)bestR * bestT;
    for (int k = 0; k < 4; ++k) put(U.scanM[k], mpow(B, R << k));
    put(U.waveM, mpow(B, 64 * R));
    for (int r = 0; r < bestR; ++r) {
        const M2 m = mul(mpow(A, r + 1), Ti);  // w[r] = row 0 of A^(r+1) applied to the companion state Ti*z
        h->g[r][0] = (float)m.a;
        h->g[r][1] = (float)m.b;
    }
    for (int l = 0; l < 64; ++l) {
        put(h->laneM[l], mpow(B, R * l));
        put(h->bc15M[l], mpow(B, R * ((l & 15) + 1)));
        put(h->bc31M[l], mpow(B, R * ((l & 31) + 1)));
    }
    for (int t = 0; t < bestT; ++t) put(h->carryM[t], mpow(B, R * t));
    const M2 BL = mpow(B, L);
    uint32_t J = 0;
    if (p->filt) {
        M2 cur{1, 0, 0, 1};
        for (int j = 0; j < kMaxLook; ++j) {
            put(h->lookM[j], cur);
            J = j + 1;
            cur = mul(cur, BL);
            if (norm(cur) < 0x1p-40) break;  // older tiles are below f32 resolution of the state
            if (j == kMaxLook - 1) {          // pole radius too close to 1 for this tile length
                delete h;
                delete p;
                return RH_ERR_UNSUPPORTED;
            }
        }
    }
    p->J = J;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p->d_tabs), sizeof(Tables));
    if (e == hipSuccess) e = hipMemcpy(p->d_tabs, h, sizeof(Tables), hipMemcpyHostToDevice);
    delete h;
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&p->d_srcs), sizeof(SrcDesc) * cfg->max_sources);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&p->d_ctl), 64);
    if (e == hipSuccess) e = hipMemset(p->d_ctl, 0, 64);
    if (e != hipSuccess) {
        rh::set_hip_error(e, "rh_rlm_create");
        rh_rlm_destroy(p);
        return e == hipErrorOutOfMemory ? RH_ERR_NOMEM : RH_ERR_HIP;
    }
    *out = p;
    return RH_OK;
}

rh_status rh_rlm_destroy(rh_rlm *p) {
    if (!p) return RH_OK;
    if (p->d_tabs) (void)hipFree(p->d_tabs);
    if (p->d_srcs) (void)hipFree(p->d_srcs);
    if (p->d_gran) (void)hipFree(p->d_gran);
    if (p->d_ctl) (void)hipFree(p->d_ctl);
    delete p;
    return RH_OK;
}

rh_status rh_rlm_set_sources(rh_rlm *p, const float *const *srcs_host, const uint64_t *in_frames_host, uint32_t n_sources) {
    RH_REQUIRE_INIT();
    if (!p || (n_sources && (!srcs_host || !in_frames_host))) return RH_ERR_INVALID;
    if (n_sources > p->cfg.max_sources) return RH_ERR_CAPACITY;
    std::vector<SrcDesc> h(n_sources);
    uint64_t M = 0;
    for (uint32_t s = 0; s < n_sources; ++s) {
        if (in_frames_host[s] > p->cfg.max_in_frames) return RH_ERR_CAPACITY;
        if (in_frames_host[s] && (!srcs_host[s] || (reinterpret_cast<uintptr_t>(srcs_host[s]) & 15u))) return RH_ERR_INVALID;
        rh::ResampleGeom g;
        rh_status st = rh::make_resample_geom(in_frames_host[s], p->cfg.from_rate, p->cfg.to_rate, p->cfg.channels, p->cfg.span_len, &g);
        if (st != RH_OK) return st;
        h[s] = SrcDesc{srcs_host[s], in_frames_host[s], g.out_frames};
        if (g.out_frames > M) M = g.out_frames;
    }
    const uint64_t L = (uint64_t)p->R * p->threads;
    const uint64_t tiles = (M + L - 1) / L;
    if (tiles > 0x7fffffffull) return RH_ERR_UNSUPPORTED;
    if (n_sources) RH_HIP_TRY(hipMemcpy(p->d_srcs, h.data(), sizeof(SrcDesc) * n_sources, hipMemcpyHostToDevice));
    const size_t words = (size_t)n_sources * tiles * 4;
    if (p->filt && words > p->gran_words) {
        if (p->d_gran) RH_HIP_TRY(hipFree(p->d_gran));
        p->d_gran = nullptr;
        RH_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_gran), words * 8));
        RH_HIP_TRY(hipMemset(p->d_gran, 0, words * 8));  // epoch 0 never matches a run
        p->gran_words = words;
    }
    p->n_sources = n_sources;
    p->n_tiles = (uint32_t)tiles;
    p->out_frames = M;
    return RH_OK;
}

rh_status rh_rlm_run(rh_rlm *p, float *dst, uint64_t out_capacity_frames, uint64_t *out_frames, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (!p) return RH_ERR_INVALID;
    if (out_frames) *out_frames = p->out_frames;
    if (p->out_frames == 0) return RH_OK;
    if (!dst || (reinterpret_cast<uintptr_t>(dst) & 15u)) return RH_ERR_INVALID;
    if (out_capacity_frames < p->out_frames) return RH_ERR_CAPACITY;
    hipStream_t s = rh::as_stream(stream);
    p->epoch += 1;
    if (p->epoch == 0) {  // tag wrap: old tags could alias, start over from a clean table
        if (p->d_gran) RH_HIP_TRY(hipMemsetAsync(p->d_gran, 0, p->gran_words * 8, s));
        p->epoch = 1;
    }
    Params k;
    k.srcs = p->d_srcs;
    k.tabs = p->d_tabs;
    k.out = dst;
    k.gran = p->d_gran;
    k.ticket = p->d_ctl;
    k.status = p->d_ctl + 1;
    k.out_frames = p->out_frames;
    k.chunk_in = p->chunk_in;
    k.chunk_out = p->chunk_out;
    k.n_sources = p->n_sources;
    k.n_tiles = p->n_tiles;
    k.F = p->F;
    k.T = p->T;
    k.qF = p->F / p->T;
    k.rF = p->F % p->T;
    k.Tf = (float)p->T;
    k.rcpT = 1.0f / (float)p->T;
    k.epoch = p->epoch;
    k.J = p->J;
    k.stage_bytes = p->stage_bytes;
    k.ticket_base = p->ticket_base;
    k.u = p->uni;
    void *args[] = {&k};
    hipError_t e = hipLaunchKernel(p->kernel, dim3(p->n_tiles), dim3(p->threads), args, p->lds_bytes, s);
    if (e != hipSuccess) {
        rh::set_hip_error(e, "k_rlm_stereo launch");
        return RH_ERR_HIP;
    }
    p->ticket_base += p->n_tiles;  // every launch takes exactly n_tiles tickets
    return RH_OK;
}

rh_status rh_rlm_last_status(rh_rlm *p) {
    RH_REQUIRE_INIT();
    if (!p) return RH_ERR_INVALID;
    uint32_t ctl[2] = {0, 0};
    RH_HIP_TRY(hipMemcpy(ctl, p->d_ctl, 8, hipMemcpyDeviceToHost));  // synchronises with the device
    if (ctl[1]) {  // sticky until read
        RH_HIP_TRY(hipMemset(p->d_ctl + 1, 0, 4));
        return RH_ERR_TIMEOUT;
    }
    return RH_OK;
}

rh_status rh_rlm_geometry(rh_rlm *p, uint32_t *threads, uint32_t *frames_per_lane, uint32_t *lds_bytes, uint32_t *lookback_tiles) {
    if (!p) return RH_ERR_INVALID;
    if (threads) *threads = p->threads;
    if (frames_per_lane) *frames_per_lane = p->R;
    if (lds_bytes) *lds_bytes = p->lds_bytes;
    if (lookback_tiles) *lookback_tiles = p->J;
    return RH_OK;
}

// Time-parallel standalone biquad (rh_biquad mode 1): scheduled for the next round; the
// sequential mode-0 kernel is the one shipped for the standalone op.
rh_status rh_biquad_scan(float *, const float *, uint64_t, uint32_t, uint32_t, const float *, float *, rh_stream) {
    return RH_ERR_UNSUPPORTED;
}

}  // extern "C"

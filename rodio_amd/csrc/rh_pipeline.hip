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
constexpr int kHeaderBytes = 640;  // wagg[2][8][4] f64 (512 B) + cbuf[2][4] f64 (64 B) + misc (64 B)

struct SrcDesc {
    const float *data;
    uint64_t frames;      // N_s
    uint64_t out_frames;  // M_s
};

// Host-computed (f64 -> f32) powers of the companion matrix A = [[-a1,-a2],[1,0]], row major.
// Wave-uniform tables travel in the kernel argument block (scalar registers); only the
// per-lane tables live in memory.
struct Uniforms {
    // H(z) = b0 + (c1 z^-1 + c2 z^-2)/A(z), c1 = b1 - b0*a1, c2 = b2 - b0*a2: the recursive part
    // w = y - b0*x is the state that is scanned.  w is smooth whenever the poles sit near z = 1
    // (also for a high-pass, whose y is not), which keeps the zero-state run and the
    // homogeneous correction of the same magnitude as w instead of cancelling large terms.
    float b0, c1, c2, a1, a2;
    double scanM[6][4];           // A^(R*2^k)
    double waveM[4];              // A^(64R)
    float g1[kMaxR], g2[kMaxR];   // (A^(r+1))[0][0], [0][1]: homogeneous response inside a run
};
struct Tables {
    double laneM[64][4];           // A^(R*lane)
    double carryM[kMaxThreads][4]; // A^(R*tid)
    double lookM[kMaxLook][4];     // A^(L*j)
};

struct Params {
    const SrcDesc *srcs;
    const Tables *tabs;
    float *out;
    unsigned long long *gran;  // [S][tiles][4]
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

// y = M * x for a row-major 2x2
// The filter STATE algebra (scan, wave chain, tile carry) runs in f64: a rounding error in a
// state is later multiplied by ||A^n|| (up to ~1/(e(1-|pole|))), so f32 there costs 10-100x the
// reference's own error.  Samples, taps and the per-sample correction stay f32.
__device__ __forceinline__ void mat_acc(const double *M, double x1, double x2, double &y1, double &y2) {
    y1 = __builtin_fma(M[0], x1, __builtin_fma(M[1], x2, y1));
    y2 = __builtin_fma(M[2], x1, __builtin_fma(M[3], x2, y2));
}

template <int R, int KV, int D, bool FILT>
__global__ __launch_bounds__(kMaxThreads) void k_rlm_stereo(const Params p) {
    static_assert(R % 2 == 0 && R <= kMaxR, "R must be even");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *wagg = reinterpret_cast<double *>(smem);            // [2][8][4]
    double *cbuf = reinterpret_cast<double *>(smem + 512);      // [2][4]
    uint32_t *misc = reinterpret_cast<uint32_t *>(smem + 576);  // ticket
    unsigned char *inbuf = smem + kHeaderBytes;                 // [2][stage_bytes]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
        i_base &= ~1ull;  // 16-byte aligned float4 loads
        cursor_resolve(cursor_at(m_tile0 + L - 1, p), p, i_end, nn);
        i_end += 1;
    }
    uint32_t nvec = (uint32_t)((i_end - i_base + 2) / 2);
    if (nvec > (uint32_t)(KV * NT)) nvec = KV * NT;  // host sizes KV so this never bites

    // ---- per-lane tap table: LDS byte offset of frame i(m) and the lerp numerator ------------
    int offA[R + 2];
    float numf[R + 2];
    {
        Cursor c = cursor_at(first ? 0 : m0 - 2, p);
#pragma unroll
        for (int rr = 0; rr < R + 2; ++rr) {
            const bool dummy = first && rr < 2;
            uint64_t i;
            uint32_t num;
            cursor_resolve(c, p, i, num);
            offA[rr] = dummy ? 0 : (int)((i - i_base) * 8);
            numf[rr] = dummy ? 0.0f : (float)num;
            if (!dummy) cursor_next(c, p);
        }
    }

    const Tables *__restrict__ tb = p.tabs;
    double lM[4], cM[4];
    if (FILT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            lM[q] = tb->laneM[lane][q];
            cM[q] = tb->carryM[tid][q];
        }
    }
    const float b0 = p.u.b0, c1 = p.u.c1, c2 = p.u.c2, na1 = -p.u.a1, na2 = -p.u.a2;

    float2 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float2(0.0f, 0.0f);
    double Qr[D][4];  // start-of-run states (zero tile carry) of the D sources in flight
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int q = 0; q < 4; ++q) Qr[d][q] = 0.0;

    bool dead = false;  // a bounded wait expired: never spin again in this workgroup
    float4 pre[KV];
    auto issue_loads = [&](uint32_t s) {
        const SrcDesc sd = p.srcs[s];
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const uint32_t j = tid + k * NT;
            const uint64_t f = i_base + 2ull * j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < nvec && sd.frames) {
                if (f + 1 < sd.frames) {
                    v = *reinterpret_cast<const float4 *>(sd.data + f * 2);
                } else {  // past the end: replicate the last frame (verbatim rule, see header)
                    const uint64_t f0 = f < sd.frames ? f : sd.frames - 1;
                    const float2 a = *reinterpret_cast<const float2 *>(sd.data + f0 * 2);
                    const float2 b = *reinterpret_cast<const float2 *>(sd.data + (sd.frames - 1) * 2);
                    v = make_float4(a.x, a.y, b.x, b.y);
                }
            }
            pre[k] = v;
        }
    };
    auto commit_loads = [&](uint32_t s) {
        unsigned char *dstb = inbuf + (size_t)(s & 1) * p.stage_bytes;
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const uint32_t j = tid + k * NT;
            if (j < nvec) *reinterpret_cast<float4 *>(dstb + (size_t)j * 16) = pre[k];
        }
    };

    const uint32_t S = p.n_sources;
    if (S > 0) {
        issue_loads(0);
        commit_loads(0);
    }
    __syncthreads();

    const uint32_t n_iter = FILT ? S + D : S;
    for (uint32_t s = 0; s < n_iter; ++s) {
        double Qnew[4] = {0., 0., 0., 0.};
        double P[4] = {0., 0., 0., 0.};
        if (s < S) {
            if (s + 1 < S) issue_loads(s + 1);
            const uint64_t Ms = p.srcs[s].out_frames;
            const int nvalid = Ms > m0 ? (Ms - m0 >= (uint64_t)R ? R : (int)(Ms - m0)) : 0;
            const unsigned char *buf = inbuf + (size_t)(s & 1) * p.stage_bytes;
            auto tap = [&](int rr) -> float2 {
                const float2 a = *reinterpret_cast<const float2 *>(buf + offA[rr]);
                const float2 b = *reinterpret_cast<const float2 *>(buf + offA[rr] + 8);
                float2 x;  // math.rs:25: first + (second - first) * num / den
                x.x = a.x + div_T((b.x - a.x) * numf[rr], p.Tf, p.rcpT);
                x.y = a.y + div_T((b.y - a.y) * numf[rr], p.Tf, p.rcpT);
                return x;
            };
            if (FILT) {
                float2 x2 = first ? make_float2(0.f, 0.f) : tap(0);
                float2 x1 = first ? make_float2(0.f, 0.f) : tap(1);
                float2 y1 = make_float2(0.f, 0.f), y2 = make_float2(0.f, 0.f);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float2 x = tap(r + 2);
                    float2 y;  // zero-state step of the recursive part w; the w1 term goes last (shortest chain)
                    y.x = fma_(na1, y1.x, fma_(na2, y2.x, fma_(c2, x2.x, c1 * x1.x)));
                    y.y = fma_(na1, y1.y, fma_(na2, y2.y, fma_(c2, x2.y, c1 * x1.y)));
                    if (r < nvalid) {
                        acc[r].x += fma_(b0, x.x, y.x);
                        acc[r].y += fma_(b0, x.y, y.y);
                    }
                    y2 = y1;
                    y1 = y;
                    x2 = x1;
                    x1 = x;
                }
                // ---- wave64 inclusive scan of the run end states over A^(R*2^k) ----
                P[0] = y1.x;
                P[1] = y2.x;
                P[2] = y1.y;
                P[3] = y2.y;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int d = 1 << k;
                    const double q0 = __shfl_up(P[0], d), q1 = __shfl_up(P[1], d);
                    const double q2 = __shfl_up(P[2], d), q3 = __shfl_up(P[3], d);
                    if (lane >= d) {
                        const double *M = p.u.scanM[k];
                        mat_acc(M, q0, q1, P[0], P[1]);
                        mat_acc(M, q2, q3, P[2], P[3]);
                    }
                }
                if (lane == 63) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) wagg[((s & 1) * 8 + wave) * 4 + q] = P[q];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {  // exclusive: state at the start of this lane's run
                    const double up = __shfl_up(P[q], 1);
                    Qnew[q] = lane ? up : 0.0;
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float2 x = tap(r + 2);
                    if (r < nvalid) {
                        acc[r].x += x.x;
                        acc[r].y += x.y;
                    }
                }
            }
        }
        // ---- wave 0: gather the tile carry of source s-D from the J previous tiles -------------
        if (FILT && wave == 0) {
            double c[4] = {0., 0., 0., 0.};
            if (s >= (uint32_t)D && tile > 0) {
                const uint32_t sp = s - D;
                const bool need = (uint32_t)lane < p.J && (uint32_t)lane < tile;
                const unsigned long long *g = p.gran + ((uint64_t)sp * p.n_tiles + (tile - 1 - (need ? lane : 0))) * 4;
                unsigned long long v[4] = {0, 0, 0, 0};
                bool ok = !need;
                uint32_t spins = 0;
                while (!dead) {
                    if (!ok) {
                        bool all = true;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v[q] = __hip_atomic_load(g + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            all = all && ((uint32_t)(v[q] >> 32) == p.epoch);
                        }
                        ok = all;
                    }
                    if (__all(ok)) break;
                    if (++spins > kSpinLimit) {
                        if (lane == 0) atomicOr(p.status, 1u);
                        dead = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
                if (need && ok) {
                    const double *M = tb->lookM[lane];
                    mat_acc(M, (double)__uint_as_float((uint32_t)v[0]), (double)__uint_as_float((uint32_t)v[1]), c[0], c[1]);
                    mat_acc(M, (double)__uint_as_float((uint32_t)v[2]), (double)__uint_as_float((uint32_t)v[3]), c[2], c[3]);
                }
                if (p.J > 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int d = 32; d >= 1; d >>= 1) c[q] += __shfl_xor(c[q], d);
                    }
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) cbuf[(s & 1) * 4 + q] = c[q];
            }
        }
        if (s + 1 < S) commit_loads(s + 1);
        __syncthreads();
        if (FILT) {
            if (s < S) {
                // chain the waves: state at the start of this wave (zero tile carry)
                double Wst[4] = {0., 0., 0., 0.};
                const double *wa = wagg + (s & 1) * 32;
                for (int u = 0; u < wave; ++u) {
                    double n0 = wa[u * 4 + 0], n1 = wa[u * 4 + 1], n2 = wa[u * 4 + 2], n3 = wa[u * 4 + 3];
                    mat_acc(p.u.waveM, Wst[0], Wst[1], n0, n1);
                    mat_acc(p.u.waveM, Wst[2], Wst[3], n2, n3);
                    Wst[0] = n0; Wst[1] = n1; Wst[2] = n2; Wst[3] = n3;
                }
                if (wave == W - 1 && lane < 4) {  // publish the tile aggregate: 4 granules
                    double e0 = wa[wave * 4 + 0], e1 = wa[wave * 4 + 1], e2 = wa[wave * 4 + 2], e3 = wa[wave * 4 + 3];
                    mat_acc(p.u.waveM, Wst[0], Wst[1], e0, e1);
                    mat_acc(p.u.waveM, Wst[2], Wst[3], e2, e3);
                    // a published state is rounded to f32 once, like any output sample
                    const float ev = (float)(lane == 0 ? e0 : lane == 1 ? e1 : lane == 2 ? e2 : e3);
                    const unsigned long long word = ((unsigned long long)p.epoch << 32) | __float_as_uint(ev);
                    __hip_atomic_store(p.gran + ((uint64_t)s * p.n_tiles + tile) * 4 + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                mat_acc(lM, Wst[0], Wst[1], Qnew[0], Qnew[1]);
                mat_acc(lM, Wst[2], Wst[3], Qnew[2], Qnew[3]);
            }
            if (s >= (uint32_t)D) {  // finish source s-D: add the homogeneous response to its true start state
                const uint32_t sp = s - D;
                const uint64_t Ms = p.srcs[sp].out_frames;
                const int nvalid = Ms > m0 ? (Ms - m0 >= (uint64_t)R ? R : (int)(Ms - m0)) : 0;
                const double *cb = cbuf + (s & 1) * 4;
                double D0 = Qr[0][0], D1 = Qr[0][1], D2 = Qr[0][2], D3 = Qr[0][3];
                mat_acc(cM, cb[0], cb[1], D0, D1);
                mat_acc(cM, cb[2], cb[3], D2, D3);
                const float S0 = (float)D0, S1 = (float)D1, S2 = (float)D2, S3 = (float)D3;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (r < nvalid) {
                        const float h1 = p.u.g1[r], h2 = p.u.g2[r];
                        acc[r].x += fma_(h1, S0, h2 * S1);
                        acc[r].y += fma_(h1, S2, h2 * S3);
                    }
                }
            }
#pragma unroll
            for (int d = 0; d + 1 < D; ++d)
#pragma unroll
                for (int q = 0; q < 4; ++q) Qr[d][q] = Qr[d + 1][q];
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
            *reinterpret_cast<float2 *>(o + r * 2) = acc[r];
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
void put(double *dst, const M2 &m) {
    dst[0] = m.a;
    dst[1] = m.b;
    dst[2] = m.c;
    dst[3] = m.d;
}
double norm(const M2 &m) { return std::fabs(m.a) + std::fabs(m.b) + std::fabs(m.c) + std::fabs(m.d); }

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
    // ---- tables ------------------------------------------------------------------------------
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
    const uint64_t R = bestR, L = (uint64_t)bestR * bestT;
    for (int k = 0; k < 6; ++k) put(U.scanM[k], mpow(A, R << k));
    put(U.waveM, mpow(A, 64 * R));
    for (int r = 0; r < bestR; ++r) {
        const M2 m = mpow(A, r + 1);
        U.g1[r] = (float)m.a;
        U.g2[r] = (float)m.b;
    }
    for (int l = 0; l < 64; ++l) put(h->laneM[l], mpow(A, R * l));
    for (int t = 0; t < bestT; ++t) put(h->carryM[t], mpow(A, R * t));
    const M2 AL = mpow(A, L);
    uint32_t J = 0;
    if (p->filt) {
        M2 cur{1, 0, 0, 1};
        for (int j = 0; j < kMaxLook; ++j) {
            put(h->lookM[j], cur);
            J = j + 1;
            cur = mul(cur, AL);
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

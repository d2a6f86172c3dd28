// rh_biquad_scan.hip -- the stand-alone BltFilter (src/source/blt.rs:397-492, :502-544, :558-560) as a TIME-PARALLEL
// kernel for batches of streams: `rh_biquad` mode 1.
//
// The algebra is DESIGN.md 4.2 (the fused kernel's): H(z) = b0 + (c1 z^-1 + c2 z^-2)/A(z); the recursive part
// w = y - b0*x is what travels along time; a lane runs w over its R frames from a ZERO state, the true start states are
// linear in the run-end states of everything before -- powers of B = Tm A Tm^-1 (Tm: the basis in which those powers are
// benign in f32; all tables computed on the host in f64 and rounded once) -- and the correction g[r] * (start state) is
// added once.  What differs from the fused kernel is everything around it (this path has no mixer and as many bytes to
// write as to read), and that follows the limiter (rh_limit.hip):
//   * a workgroup of NW waves = one tile of LW = NW*64*R frames of ONE stream; tiles by atomic ticket, tile-major over the
//     streams, persistent grid;
//   * the next tile's samples arrive by LDS-DMA while the current one is worked on (plus the two frames in front of a wave's
//     share: x[n-1], x[n-2] of its first frame); lanes read their runs from swizzled LDS slots, results go back to the same
//     slots and leave as whole lines: 4 B in + 4 B out per sample;
//   * wave scan (DPP Kogge-Stone over 2x2 matrices) -> the waves' aggregates meet in LDS (ONE barrier) -> workgroup aggregate
//     published -> look-back over the J workgroup tiles in front whose weight B^(LW*j) is above 2^-40 (a stable filter forgets:
//     no chained prefix, a tile never waits for a predecessor's result, only for its zero-state aggregate) -> correction;
//   * any channel count the variants list, any block length, an optional carried state {x1,x2,y1,y2} per channel in the
//     layout of mode 0 (blt.rs:404-407) -- so a stream can be filtered block by block on this path too.
// <= 1e-5 abs against the reference-order recurrence (mode 0 stays the bit-exact path), and no further from the f64 truth.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <memory>
#include <mutex>
#include <vector>

#include "rh_common.h"

namespace {

#include "rh_scan_common.h"

constexpr int kMaxR = 16, kMaxNW = 8;
constexpr uint32_t kSpinLimit = 1u << 22;

struct BqTabs {  // per lane (device memory, cached per filter and geometry): powers of B, row-major 2x2
    float laneM[64][4];  // B^(R*l)
    float b15[64][4];    // B^(R*((l&15)+1))
    float b31[64][4];    // B^(R*((l&31)+1))
    float lookM[64][4];  // B^(LW*j): weight of the j-th workgroup tile in front (j = 0: the nearest)
};
struct BqArgs {
    float *dst;
    const float *src;
    float *gran;            // [S][tiles][2*C] hand-off words: the tiles' zero-state aggregates (scan basis)
    float *gran_other;      // the table of the NEXT launch on this stream (the two alternate): a tile that is done sets its record there back to
                            // "not yet", so that launch needs no kernel in front of it (nullptr: it initialises its own); rh_limit.hip has the same
    uint32_t ticket_base;   // value of ctl[0] when this launch starts (the counter is never reset)
    const float *state_in;  // [S][C][4] {x1,x2,y1,y2} snapshot, or nullptr (zero state)
    const BqTabs *tabs;
    uint32_t *ctl;          // [0] ticket
    uint32_t *status;          // the library's sticky failure word (rh_async_status)
    uint32_t spin;             // polls of one hand-off before the tile gives up (kSpinLimit; RH_SCAN_SPIN_LIMIT overrides)
    uint32_t dma_top;          // 1: the next tile's samples are requested at the top of a tile, not right in front of its poll
    uint64_t frames, stride;
    uint32_t n_streams, tiles, J;
    float b0, c1, c2, na1, na2;
    float Tm[4];
    float scanM[4][4];        // B^(R*2^k)
    float g[kMaxR][2];        // row 0 of A^(r+1) Tm^-1
    float waveM[kMaxNW][4];   // B^(L*k)
    float BL[4];              // B^L
};

template <class MP>  // (a pointer to four floats in any address space: the matrices of the argument block are read from the constant one)
__device__ __forceinline__ void mat_acc(MP M, float x1, float x2, float &y1, float &y2) {
    y1 = fma_(M[0], x1, fma_(M[1], x2, y1));
    y2 = fma_(M[2], x1, fma_(M[3], x2, y2));
}

// Inclusive wave64 scan of 2-vectors under P_l = sum_{k<=l} B^(R*(l-k)) p_k (the fused kernel's, rh_pipeline.hip).
template <class SM>
__device__ __forceinline__ void scan_mat(float &P0, float &P1, SM &sm, const float *b15, const float *b31) {
#define RH_STEP(K, N)                                                                          \
    {                                                                                          \
        const float q0 = dpp0<kRowShr + N, 0xf>(P0), q1 = dpp0<kRowShr + N, 0xf>(P1);          \
        mat_acc(sm[K], q0, q1, P0, P1);                                                        \
    }
    RH_STEP(0, 1)
    RH_STEP(1, 2)
    RH_STEP(2, 4)
    RH_STEP(3, 8)
#undef RH_STEP
    {
        const float q0 = dpp0<kBcast15, 0xa>(P0), q1 = dpp0<kBcast15, 0xa>(P1);
        mat_acc(b15, q0, q1, P0, P1);
    }
    {
        const float q0 = dpp0<kBcast31, 0xc>(P0), q1 = dpp0<kBcast31, 0xc>(P1);
        mat_acc(b31, q0, q1, P0, P1);
    }
}

// The kernel's argument block, read where it lies: in the constant address space (the kernarg segment; it is the kernel's only argument, so it
// sits at offset 0 of __builtin_amdgcn_kernarg_segment_ptr()).  Read through the by-value parameter, the compiler loads every scalar of it once, in
// front of the persistent loop, and keeps them all in SGPRs -- more than the file holds beside the loop's own state: 9 % of the kernel's vector
// instructions were v_readlane / v_writelane moving spilled scalars.  Read through a pointer that is made opaque per tile (and per phase), every
// phase loads the constants it uses where it uses them (scalar-cache hits) and nothing outlives it.  (rh_limit.hip: the same.)
typedef const __attribute__((address_space(4))) BqArgs *BqArgsC;

template <int C, int R, int NW, bool FULL>
__device__ __forceinline__ void bq_tile(BqArgsC kargs, v4f *lds, const v4f *halo, float (*xZ)[2 * C], const int lane_, const int wave, const uint32_t tile, const uint32_t stream,
                                        const float (*tab)[64], const uint32_t nf, const float *next_src, v4f *next_buf, v4f *next_halo, bool &dead, const uint32_t ticket_ahead,
                                        uint32_t *ticket_slot) {
    int lane = lane_;
    asm volatile("" : "+v"(lane));  // per-tile address arithmetic is recomputed, not hoisted into registers that live for the whole kernel
#define RH_ARGS_FRESH() asm volatile("" : "+s"(kargs))
    RH_ARGS_FRESH();
#define a (*kargs)
    constexpr int V = C * R / 4;
    constexpr int kPrefixUnroll = C <= 2 ? NW : 1;
    constexpr int HV = (2 * C + 3) / 4;  // vectors that hold the two frames in front of a run
    constexpr uint32_t L = 64u * R, LW = L * NW;
    constexpr uint32_t G = 2 * C;
    const uint64_t f0 = (uint64_t)tile * LW + (uint64_t)wave * L;
    const uint32_t nfl = FULL ? (uint32_t)R : (nf > (uint32_t)lane * R ? (nf - lane * R < (uint32_t)R ? nf - lane * R : R) : 0u);
    const float *src = a.src + stream * a.stride + f0 * C;
    float *dst = a.dst + stream * a.stride + f0 * C;
    const uint32_t nfloat = nf * C;
    const float *const gstream = a.gran + (uint64_t)stream * a.tiles * G;
    float *const rec = a.gran + ((uint64_t)stream * a.tiles + tile) * G;
    const float *const init = a.state_in ? a.state_in + (uint64_t)stream * C * 4 : nullptr;

    if (!FULL) {  // a short share (end of a stream): fetched here, guarded, into the same slots
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const uint32_t q = k * 64 + lane, o = 4u * q;
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if (o + 4 <= nfloat) v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(src + o));
            else if (o < nfloat) {
                v.x = src[o];
                if (o + 1 < nfloat) v.y = src[o + 1];
                if (o + 2 < nfloat) v.z = src[o + 2];
            }
            lds[slot_of<V>(q / V, q % V)] = v;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- x[n-1], x[n-2] of the run's first frame: the previous lane's last two frames; lane 0 takes them from the halo (the
    //      vectors in front of the share: DMA for whole shares, read here for a short one), the stream's first run from the
    //      carried state (blt.rs:404-407: x_n1, x_n2) or zeros
    float x1[C], x2[C];
    {
        float h[HV * 4];
        if (lane > 0) {
#pragma unroll
            for (int i = 0; i < HV; ++i) {
                const v4f v = lds[slot_of<V>(lane - 1, V - HV + i)];
                h[4 * i] = v.x, h[4 * i + 1] = v.y, h[4 * i + 2] = v.z, h[4 * i + 3] = v.w;
            }
        } else if (f0 > 0 && (FULL || nf > 0)) {  // (an EMPTY share past the end of the stream has no frames in front of it to read:
                                                  //  its address may lie beyond the last stream's buffer)
#pragma unroll
            for (int i = 0; i < HV; ++i) {
                const v4f v = FULL ? halo[i] : *reinterpret_cast<const v4f *>(src - 4 * (HV - i));
                h[4 * i] = v.x, h[4 * i + 1] = v.y, h[4 * i + 2] = v.z, h[4 * i + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < HV * 4; ++i) h[i] = 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                h[HV * 4 - C + c] = (init && f0 == 0) ? init[4 * c] : 0.0f;          // x[-1]
                h[HV * 4 - 2 * C + c] = (init && f0 == 0) ? init[4 * c + 1] : 0.0f;  // x[-2]
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            x1[c] = h[HV * 4 - C + c];
            x2[c] = h[HV * 4 - 2 * C + c];
        }
    }
    // ---- zero-state run: w[n] = c1 x[n-1] + c2 x[n-2] - a1 w[n-1] - a2 w[n-2]; y = b0 x + w kept per sample ----
    float yz[R][C], w1[C], w2[C];
#pragma unroll
    for (int c = 0; c < C; ++c) w1[c] = w2[c] = 0.0f;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const v4f v = lds[slot_of<V>(lane, j)];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (4 * j + i) / C, c = (4 * j + i) % C;
            const float x = e[i];
#ifdef RH_BQ_NO_ARITH  // diagnostics builds (tools/build_scan_variant.sh; wrong results, timing only): the per-sample recurrence, the scans and the correction deleted
            const float w = x1[c];
            yz[r][c] = x;
#else
            const float w = fma_(a.na1, w1[c], fma_(a.na2, w2[c], fma_(a.c2, x2[c], a.c1 * x1[c])));
            yz[r][c] = fma_(a.b0, x, w);
#endif
            if (FULL || (uint32_t)r < nfl) {
                w2[c] = w1[c];
                w1[c] = w;
                x2[c] = x1[c];
                x1[c] = x;
            }
        }
    }
    // ---- run-end states in the scan basis, wave scan, the waves' aggregates through LDS ----------------------------------
    float P[C][2], Q[C][2];
    {
        float b15[4], b31[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b15[q] = tab[4 + q][lane], b31[q] = tab[8 + q][lane];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            P[c][0] = P[c][1] = 0.0f;
            mat_acc(a.Tm, w1[c], w2[c], P[c][0], P[c][1]);
#ifndef RH_BQ_NO_ARITH
            scan_mat(P[c][0], P[c][1], a.scanM, b15, b31);
#endif
            Q[c][0] = dpp0<kWaveShr1, 0xf>(P[c][0]);  // exclusive: lane 0 gets 0
            Q[c][1] = dpp0<kWaveShr1, 0xf>(P[c][1]);
            if (lane == 63) xZ[wave][2 * c] = P[c][0], xZ[wave][2 * c + 1] = P[c][1];
        }
    }
    // the ticket taken at the top of the tile (for the tile after next) goes to LDS only here: its atomic has had the whole run to
    // return, instead of holding wave 0 -- and with it the workgroup at this barrier -- for a device-scope round trip
    if (threadIdx.x == 0) *ticket_slot = ticket_ahead;
    __syncthreads();  // the waves' aggregates are in LDS
    RH_ARGS_FRESH();
    float Wp[C][2], ZT[C][2];  // state at this wave's start from the waves in front (zero tile start); the tile's aggregate
#pragma unroll
    for (int c = 0; c < C; ++c) Wp[c][0] = Wp[c][1] = ZT[c][0] = ZT[c][1] = 0.0f;
#pragma unroll kPrefixUnroll  // few channels: all LDS reads of the loop leave together (one latency instead of NW)
    for (int k = 0; k < NW; ++k) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (k == wave) Wp[c][0] = ZT[c][0], Wp[c][1] = ZT[c][1];
            float n0 = xZ[k][2 * c], n1 = xZ[k][2 * c + 1];
            mat_acc(a.BL, ZT[c][0], ZT[c][1], n0, n1);
            ZT[c][0] = n0, ZT[c][1] = n1;
        }
    }
    if (wave == 0 && lane < 2 * C) {
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) v = lane == 2 * c ? ZT[c][0] : (lane == 2 * c + 1 ? ZT[c][1] : v);
        word_store(rec + lane, v);
    }
    // (a.dma_top = 0 only: the next tile's samples and the two frames in front of its share requested right before the poll --
    //  the poll then retires behind them, vmcnt being in order: 6 % slower than requesting them at the top of the tile)
    if (next_src) {
        dma_share<V>(next_src, next_buf, lane);
        if (lane < HV) glds16(next_src - 4 * HV, (uint32_t)lane * 16u, (uint32_t)(uintptr_t)(lds_u8 *)next_halo);
    }
    float lk[4], lM[4];  // per-lane matrices of the look-back and of the correction: read here, a poll ahead of their use
#pragma unroll
    for (int q = 0; q < 4; ++q) lk[q] = tab[12 + q][lane], lM[q] = tab[q][lane];
    // ---- look-back: T_in = sum_{j<J} B^(LW*j) ZT(t-1-j)  (+ B^(LW*t) z_state while the carried state still reaches) -------
    float Tin[C][2];
    {
        const int64_t idx = (int64_t)tile - 1 - lane;
        const bool reach = (uint32_t)lane < a.J;
        const bool real = reach && idx >= 0;
        float ag[2 * C];
#pragma unroll
        for (int c = 0; c < 2 * C; ++c) ag[c] = 0.0f;
        if (reach && idx == -1 && init) {  // the state the block starts from: w = y - b0*x (companion) -> scan basis
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float s1 = init[4 * c + 2] - a.b0 * init[4 * c], s2 = init[4 * c + 3] - a.b0 * init[4 * c + 1];
                mat_acc(a.Tm, s1, s2, ag[2 * c], ag[2 * c + 1]);
            }
        }
        const float *pr = gstream + (real ? (uint64_t)idx : 0) * G;
#ifdef RH_BQ_NO_LOOKBACK  // diagnostics builds: no tile looks at the tiles in front of it
        bool have = true;
#else
        bool have = !real;
#endif
        uint32_t spins = 0;
        while (true) {
            if (!have) {
                float gv[2 * C];
                load_words<2 * C, (2 * C) % 4 == 0 ? 4 : 2>(pr, gv);
                wait_loads(gv);
                bool ok = true;
#pragma unroll
                for (int c = 0; c < 2 * C; ++c) ok = ok && word_ok(gv[c]);
                if (ok) {
                    have = true;
#pragma unroll
                    for (int c = 0; c < 2 * C; ++c) ag[c] = gv[c];
                }
            }
            if (__all(have)) break;
            if (++spins > a.spin) {
                dead = true;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float t0 = 0.0f, t1 = 0.0f, tot;
            if (reach) mat_acc(lk, ag[2 * c], ag[2 * c + 1], t0, t1);
            (void)wave_excl_sum(t0, tot);
            Tin[c][0] = tot;
            (void)wave_excl_sum(t1, tot);
            Tin[c][1] = tot;
        }
        if (dead) {  // a hand-off never arrived: fail the call (status word) and poison the tile
            if (lane == 0) atomicOr(a.status, 1u);
#pragma unroll
            for (int c = 0; c < C; ++c) Tin[c][0] = Tin[c][1] = __builtin_nanf("");
        }
    }
    // ---- the homogeneous response to the lane's true start state, the result back into the LDS slots ----------------------
    {
        const auto *wM = a.waveM[wave];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float s0 = Wp[c][0], s1 = Wp[c][1];
            mat_acc(wM, Tin[c][0], Tin[c][1], s0, s1);          // the wave's start state
            mat_acc(lM, s0, s1, Q[c][0], Q[c][1]);              // the lane's
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < V; ++j) {
        v4f v;
        float e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (4 * j + i) / C, c = (4 * j + i) % C;
#ifdef RH_BQ_NO_ARITH
            e[i] = yz[r][c] + Q[c][0];
#else
            e[i] = fma_(a.g[r][0], Q[c][0], fma_(a.g[r][1], Q[c][1], yz[r][c]));
#endif
        }
        v.x = e[0], v.y = e[1], v.z = e[2], v.w = e[3];
        lds[slot_of<V>(lane, j)] = v;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const uint32_t q = k * 64 + lane, o = 4u * q;
        const v4f v = lds[slot_of<V>(q / V, q % V)];
        // (streaming stores: every output byte is written once; a 1:1 stream of reads and writes moves 3-8 % faster with them --
        // tools/ubench/write_bw.hip, this kernel 0.212 -> 0.204 ms)
        if (FULL || o + 4 <= nfloat) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(dst + o));
        else if (o < nfloat) {
            dst[o] = v.x;
            if (o + 1 < nfloat) dst[o + 1] = v.y;
            if (o + 2 < nfloat) dst[o + 2] = v.z;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // the NEXT launch's table: this tile's record back to "not yet" (see BqArgs::gran_other)
    if (a.gran_other && wave == 0 && (uint32_t)lane < G) a.gran_other[((uint64_t)stream * a.tiles + tile) * G + lane] = __uint_as_float(kNotYet);
#undef a
#undef RH_ARGS_FRESH
}

template <int C, int R, int NW>
__global__ __launch_bounds__(64 * NW, (NW >= 4 ? (C * R <= 16 ? 4 : 2) : 1)) void k_biquad_scan(const BqArgs a_by_value) {
    (void)a_by_value;
    BqArgsC kargs = (BqArgsC)__builtin_amdgcn_kernarg_segment_ptr();
#define a (*kargs)
    static_assert((C * R) % 4 == 0 && R <= kMaxR && NW <= kMaxNW && R >= 2, "a lane's run is whole 16-byte vectors");
    constexpr int V = C * R / 4;
    constexpr int HV = (2 * C + 3) / 4;
    constexpr uint32_t L = 64u * R, LW = L * NW;
    __shared__ __attribute__((aligned(1024))) v4f bufs[NW][2][64 * V];
    __shared__ __attribute__((aligned(16))) v4f halos[NW][2][4];  // HV <= 4 vectors in front of a share
    __shared__ float xZ[2][NW][2 * C];  // by tile parity: one barrier per tile separates a tile's writes from its reads, not from the next tile's writes
    __shared__ float tab[16][64];  // laneM, b15, b31, lookM per lane (BqTabs)
    __shared__ uint32_t s_ticket[3];
    static_assert(HV <= 4, "halo");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t total = a.n_streams * a.tiles;
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            tab[q][lane] = a.tabs->laneM[lane][q];
            tab[4 + q][lane] = a.tabs->b15[lane][q];
            tab[8 + q][lane] = a.tabs->b31[lane][q];
            tab[12 + q][lane] = a.tabs->lookM[lane][q];
        }
    }
    auto share = [&](uint32_t ticket, const float *&src, uint32_t &nf, bool &first) {
        const uint32_t tile = ticket / a.n_streams, stream = ticket - tile * a.n_streams;
        const uint64_t f0 = (uint64_t)tile * LW + (uint64_t)wave * L;
        nf = f0 >= a.frames ? 0u : (a.frames - f0 < L ? (uint32_t)(a.frames - f0) : L);
        src = a.src + stream * a.stride + f0 * C;
        first = f0 == 0;
    };
    auto fetch = [&](const float *src, bool first, v4f *buf, v4f *halo) {
        dma_share<V>(src, buf, lane);
        if (!first && lane < HV) glds16(src - 4 * HV, (uint32_t)lane * 16u, (uint32_t)(uintptr_t)(lds_u8 *)halo);
    };
    if (threadIdx.x == 0) {
        s_ticket[0] = atomicAdd(a.ctl, 1u) - a.ticket_base;
        s_ticket[1] = atomicAdd(a.ctl, 1u) - a.ticket_base;
    }
    __syncthreads();
    uint32_t cur = s_ticket[0], nxt = s_ticket[1], n = 0;
    bool prev_full = false, dead = false;
    if (cur < total) {
        const float *src;
        uint32_t nf;
        bool first;
        share(cur, src, nf, first);
        if (nf == L) fetch(src, first, bufs[wave][0], halos[wave][0]);
    }
    while (cur < total) {
        asm volatile("" : "+s"(kargs));  // (the arguments are re-read where they are used: see BqArgsC)
        uint32_t ticket_ahead = 0;
        if (threadIdx.x == 0) ticket_ahead = atomicAdd(a.ctl, 1u) - a.ticket_base;  // stored by bq_tile in front of its barrier
        uint32_t *const ticket_slot = &s_ticket[(n + 2) % 3];
        const uint32_t tile = cur / a.n_streams, stream = cur - tile * a.n_streams;
        const float *src;
        uint32_t nf;
        bool first;
        share(cur, src, nf, first);
        if (prev_full) wait_vm<V>();  // this tile's DMA is older than the V output stores of the previous tile
        else wait_vm<0>();
        const float *src2 = nullptr;
        bool first2 = false;
        if (nxt < total) {
            uint32_t nf2;
            share(nxt, src2, nf2, first2);
            if (nf2 != L) src2 = nullptr;
        }
        // (the halo of a stream's first share does not exist: bq_tile takes the carried state there, and its DMA must not run)
        const float *dma_src = src2;
        v4f *nb = bufs[wave][(n + 1) & 1], *nh = halos[wave][(n + 1) & 1];
        if (src2 && (first2 || a.dma_top)) {  // fetched here: a stream's first share has no halo; dma_top: see BqArgs
            fetch(src2, first2, nb, nh);
            dma_src = nullptr;
        }
        // FULL is the TILE's property, the same for every wave of the workgroup: all of them run one instantiation, its barrier included
        if (((uint64_t)tile + 1) * (uint64_t)(L * NW) <= a.frames) bq_tile<C, R, NW, true>(kargs, bufs[wave][n & 1], halos[wave][n & 1], xZ[n & 1], lane, wave, tile, stream, tab, nf, dma_src, nb, nh, dead, ticket_ahead, ticket_slot);
        else bq_tile<C, R, NW, false>(kargs, bufs[wave][n & 1], halos[wave][n & 1], xZ[n & 1], lane, wave, tile, stream, tab, nf, dma_src, nb, nh, dead, ticket_ahead, ticket_slot);
        prev_full = nf == L;
        cur = nxt;
        nxt = s_ticket[(n + 2) % 3];
        ++n;
    }
    wait_vm<0>();
#undef a
}

// ---- carried state: {x1,x2,y1,y2} per channel (blt.rs:404-407), the layout of mode 0 ----------------------------------------
// in front of the launch: snapshot of the state (the tiles read it while nothing has been written yet) and of the block's last
// two input frames (in place, dst == src, they are gone afterwards)
__global__ void k_bq_pre(uint32_t *ctl, uint32_t *words, uint64_t n_words, float *snap, float *xlast, const float *state, const float *src, uint64_t frames, uint64_t stride,
                         uint32_t C, uint32_t n) {
    const uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (uint64_t)gridDim.x * blockDim.x;
    // the scratch of the launch: control words zeroed, every hand-off word "not yet" -- by a kernel, in the stream's own buffer
    // (why not hipMemsetAsync on hipMallocAsync memory: rh_limit.hip, k_limit_init)
    if (i0 < 16) ctl[i0] = 0u;
    for (uint64_t i = i0; i < n_words; i += step) words[i] = 0xffffffffu;
    if (!state) return;
    for (uint64_t i = i0; i < n; i += step) {  // (stream, channel)
        const uint32_t s = (uint32_t)i / C, c = (uint32_t)i - s * C;
        for (int k = 0; k < 4; ++k) snap[4 * i + k] = state[4 * i + k];
        const float *x = src + (uint64_t)s * stride;
        xlast[2 * i] = frames >= 1 ? x[(frames - 1) * C + c] : state[4 * i];
        xlast[2 * i + 1] = frames >= 2 ? x[(frames - 2) * C + c] : (frames == 1 ? state[4 * i] : state[4 * i + 1]);
    }
}
// behind it: the new state from the block's last two inputs and outputs (y1 = the stored output: what the recurrence carries)
__global__ void k_bq_post(float *state, const float *snap, const float *xlast, const float *dst, uint64_t frames, uint64_t stride, uint32_t C, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = i / C, c = i - s * C;
    const float *y = dst + (uint64_t)s * stride;
    state[4 * i] = xlast[2 * i];
    state[4 * i + 1] = xlast[2 * i + 1];
    state[4 * i + 2] = frames >= 1 ? y[(frames - 1) * C + c] : snap[4 * i + 2];
    state[4 * i + 3] = frames >= 2 ? y[(frames - 2) * C + c] : (frames == 1 ? snap[4 * i + 2] : snap[4 * i + 3]);
}

// ---- host: the tables ----------------------------------------------------------------------------------------------------
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
    dst[0] = (float)m.a, dst[1] = (float)m.b, dst[2] = (float)m.c, dst[3] = (float)m.d;
}
double norm1(const M2 &m) { return std::fabs(m.a) + std::fabs(m.b) + std::fabs(m.c) + std::fabs(m.d); }
// scan basis for the companion matrix of z^2 + a1 z + a2 (rh_pipeline.hip)
void scan_basis(double a1, double a2, M2 &T, M2 &Tinv) {
    const double disc = a1 * a1 - 4.0 * a2, re = -0.5 * a1;
    double mu, nu;
    if (disc < 0.0 && std::sqrt(-disc) * 0.5 > 1e-3) {
        mu = re;
        nu = std::sqrt(-disc) * 0.5;
    } else {
        const double sq = disc > 0.0 ? std::sqrt(disc) * 0.5 : 0.0;
        const double l1 = re + sq, l2 = re - sq;
        mu = std::fabs(l1) < std::fabs(l2) ? l1 : l2;
        nu = 1.0;
    }
    T = {1.0, -mu, 0.0, nu};
    Tinv = {1.0, mu / nu, 0.0, 1.0 / nu};
}

struct BqPlan {
    float co[5];
    int R, NW;
    BqArgs proto;  // the uniform constants
    BqTabs *d_tabs = nullptr;
    BqPlan() = default;
    BqPlan(const BqPlan &) = delete;
    BqPlan &operator=(const BqPlan &) = delete;
    ~BqPlan() {
        if (d_tabs) (void)hipFree(d_tabs);  // (hipFree waits for the device: no queued launch still reads the table)
    }
};
// The cache: most recently used first, at most kMaxPlans filters (a swept cutoff does not grow it); a caller holds its plan
// through a shared_ptr, so a plan another thread evicts meanwhile lives until the launch that uses it has been enqueued.
constexpr size_t kMaxPlans = 32;
std::mutex g_mu;
std::list<std::shared_ptr<BqPlan>> g_plans;

// plan for (coefficients, R, NW); nullptr: the filter does not forget within 64 workgroup tiles (or is unstable)
std::shared_ptr<const BqPlan> get_plan(const float co[5], int R, int NW) {
    std::lock_guard<std::mutex> lock(g_mu);
    for (auto it = g_plans.begin(); it != g_plans.end(); ++it)
        if ((*it)->R == R && (*it)->NW == NW && std::memcmp((*it)->co, co, sizeof((*it)->co)) == 0) {
            g_plans.splice(g_plans.begin(), g_plans, it);
            return g_plans.front();
        }
    const double a1 = co[3], a2 = co[4];
    if (!(std::fabs(a2) < 1.0 && std::fabs(a1) < 1.0 + a2)) return nullptr;  // stability triangle
    const M2 A{-a1, -a2, 1.0, 0.0};
    M2 Tm, Ti;
    scan_basis(a1, a2, Tm, Ti);
    const M2 B = mul(mul(Tm, A), Ti);
    const uint64_t L = 64ull * R, LW = L * NW;
    uint32_t J = 0;
    {
        const M2 BLW = mpow(B, LW);
        M2 cur = BLW;
        for (uint32_t j = 1; j <= 64; ++j) {
            if (norm1(cur) < 0x1p-40) {
                J = j;
                break;
            }
            cur = mul(cur, BLW);
        }
    }
    if (J == 0) return nullptr;
    auto pp = std::make_shared<BqPlan>();
    BqPlan &p = *pp;
    std::memcpy(p.co, co, sizeof(p.co));
    p.R = R;
    p.NW = NW;
    BqArgs &u = p.proto;
    std::memset(&u, 0, sizeof(u));
    u.J = J;
    u.b0 = co[0];
    u.c1 = (float)((double)co[1] - (double)co[0] * a1);
    u.c2 = (float)((double)co[2] - (double)co[0] * a2);
    u.na1 = -co[3];
    u.na2 = -co[4];
    put(u.Tm, Tm);
    for (int k = 0; k < 4; ++k) put(u.scanM[k], mpow(B, (uint64_t)R << k));
    for (int r = 0; r < R; ++r) {
        const M2 m = mul(mpow(A, r + 1), Ti);
        u.g[r][0] = (float)m.a;
        u.g[r][1] = (float)m.b;
    }
    for (int k = 0; k < NW; ++k) put(u.waveM[k], mpow(B, L * k));
    put(u.BL, mpow(B, L));
    BqTabs *h = new BqTabs();
    for (int l = 0; l < 64; ++l) {
        put(h->laneM[l], mpow(B, (uint64_t)R * l));
        put(h->b15[l], mpow(B, (uint64_t)R * ((l & 15) + 1)));
        put(h->b31[l], mpow(B, (uint64_t)R * ((l & 31) + 1)));
        put(h->lookM[l], mpow(B, LW * l));
    }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p.d_tabs), sizeof(BqTabs));
    if (e == hipSuccess) e = hipMemcpy(p.d_tabs, h, sizeof(BqTabs), hipMemcpyHostToDevice);  // synchronous: done when this returns
    delete h;
    if (e != hipSuccess) {
        rh::set_hip_error(e, "rh_biquad mode 1 tables");
        return nullptr;
    }
    g_plans.push_front(pp);
    while (g_plans.size() > kMaxPlans) g_plans.pop_back();
    return pp;
}

using BqFn = void (*)(const BqArgs);
struct BqVariant {
    int C, R, NW;
    BqFn fn;
};
#define RH_BV(c, r, nw) BqVariant{c, r, nw, &k_biquad_scan<c, r, nw>}
const BqVariant kVariants[] = {
    RH_BV(1, 16, 8), RH_BV(1, 16, 1), RH_BV(2, 8, 8), RH_BV(2, 16, 8), RH_BV(2, 8, 4), RH_BV(2, 8, 1), RH_BV(3, 4, 8), RH_BV(3, 4, 1),
    RH_BV(4, 4, 8),  RH_BV(4, 4, 1),  RH_BV(5, 4, 8), RH_BV(5, 4, 1),  RH_BV(6, 4, 8), RH_BV(6, 4, 1), RH_BV(7, 4, 8), RH_BV(7, 4, 1),
    RH_BV(8, 4, 8),  RH_BV(8, 4, 1),
};
#undef RH_BV

}  // namespace

namespace rh {
// rh_biquad mode 1.  RH_ERR_UNSUPPORTED when this kernel does not take the call (channel count without a variant, rows that
// are not 16-byte aligned, a filter that does not forget within 64 tiles): the caller decides what then.
rh_status biquad_scan_launch(float *dst, const float *src, uint64_t frames, uint32_t channels, uint32_t n_streams, const float co[5], float *state, hipStream_t s) {
    const uint64_t stride = frames * channels;
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0 && (n_streams == 1 || stride % 4 == 0);
    if (!aligned) return RH_ERR_UNSUPPORTED;
    // Geometry: the longest tile a stream fills at least half of, more frames per lane among equals (rh_limit.hip has the
    // measurements: 8192-frame tiles from 64 x 1 Mi frames down to 256 x 8192, 0.240 ms with R = 16 against 0.261 ms with R = 8)
    const BqVariant *v = nullptr;
    if (rh::knob(rh::K_BIQUAD_R) || rh::knob(rh::K_BIQUAD_NW)) {  // tuning aids: the variant closest to the request
        const int want_R = rh::knob(rh::K_BIQUAD_R) ? atoi(rh::knob(rh::K_BIQUAD_R)) : 16, want_NW = rh::knob(rh::K_BIQUAD_NW) ? atoi(rh::knob(rh::K_BIQUAD_NW)) : 8;
        for (const BqVariant &c : kVariants) {
            if (c.C != (int)channels) continue;
            auto score = [&](const BqVariant &x) { return 10 * std::abs(x.NW - want_NW) + std::abs(x.R - want_R); };
            if (!v || score(c) < score(*v)) v = &c;
        }
    } else {
        auto tile_of = [](const BqVariant &x) { return (uint64_t)64 * x.R * x.NW; };
        for (const BqVariant &c : kVariants) {
            if (c.C != (int)channels) continue;
            if (!v) {
                v = &c;
                continue;
            }
            const bool fits_c = tile_of(c) <= 2 * frames, fits_v = tile_of(*v) <= 2 * frames;
            const bool better = fits_c != fits_v ? fits_c
                                : (fits_c ? (tile_of(c) > tile_of(*v) || (tile_of(c) == tile_of(*v) && c.R > v->R)) : tile_of(c) < tile_of(*v));
            if (better) v = &c;
        }
    }
    if (!v) return RH_ERR_UNSUPPORTED;
    const std::shared_ptr<const BqPlan> pl = get_plan(co, v->R, v->NW);
    if (!pl) return RH_ERR_UNSUPPORTED;
    const uint32_t R = (uint32_t)v->R, NW = (uint32_t)v->NW, LW = 64u * R * NW;
    const uint64_t tiles64 = (frames + LW - 1) / LW;
    if (tiles64 > 0x7fffffffull || tiles64 * n_streams >= 0xfff00000ull) return RH_ERR_UNSUPPORTED;

    BqArgs a = pl->proto;
    a.dst = dst;
    a.src = src;
    a.tabs = pl->d_tabs;
    a.frames = frames;
    a.stride = stride;
    a.n_streams = n_streams;
    a.tiles = (uint32_t)tiles64;
    const size_t n_sc = (size_t)n_streams * channels;
    // two hand-off tables in rotation, as in rh_limit.hip: a launch without a carried state that follows one of its own shape on this
    // stream finds its table cleared by that launch and needs no k_bq_pre in front of it
    const size_t gran_bytes = (((size_t)n_streams * tiles64 * 2 * channels * sizeof(float)) + 63) & ~size_t(63);
    const size_t head = 64 + ((n_sc * 6 * 4 + 63) & ~size_t(63));  // control words, state snapshot [n][4], last inputs [n][2]
    unsigned char *scratch = nullptr;
    std::unique_lock<std::mutex> scratch_hold;
    rh::ScratchAux *aux = nullptr;
    RH_HIP_TRY(rh::stream_scratch(s, head + 2 * gran_bytes, reinterpret_cast<void **>(&scratch), scratch_hold, &aux));
    uint64_t tag = 0x4251554144ull;  // "BQUAD", then the shape (FNV-1a)
    for (uint64_t v_ : {(uint64_t)n_streams, tiles64, (uint64_t)channels, (uint64_t)head, (uint64_t)gran_bytes, (uint64_t)reinterpret_cast<uintptr_t>(scratch)}) tag = (tag ^ v_) * 0x100000001b3ull;
    tag |= 1;
    const char *init_knob = rh::knob(rh::K_LIMIT_INIT);  // RH_LIMIT_INIT=1: both scan kernels initialise their tables in front of every launch
    const bool clean = !state && aux->tag == tag && !(init_knob && init_knob[0] == '1');
    a.ctl = reinterpret_cast<uint32_t *>(scratch);
    a.status = rh::g_async_status;
    a.dma_top = rh::knob(rh::K_SCAN_DMA_TOP) ? (uint32_t)atoi(rh::knob(rh::K_SCAN_DMA_TOP)) : 1u;  // measured: 0.312 -> 0.286 ms (limiter), 0.234 -> 0.221 ms (biquad), 64 x 1 Mi frames
    a.spin = rh::knob(rh::K_SCAN_SPIN_LIMIT) ? (uint32_t)strtoul(rh::knob(rh::K_SCAN_SPIN_LIMIT), nullptr, 10) : kSpinLimit;
    float *snap = reinterpret_cast<float *>(scratch + 64), *xlast = snap + n_sc * 4;
    hipError_t e = hipSuccess;
    if (!clean) {
        const uint64_t n_words = 2 * gran_bytes / 4;
        const unsigned pre_wgs = (unsigned)std::min<uint64_t>(1024, (std::max<uint64_t>(n_words, n_sc) + 255) / 256);
        hipLaunchKernelGGL(k_bq_pre, dim3(pre_wgs), dim3(256), 0, s, a.ctl, reinterpret_cast<uint32_t *>(scratch + head), n_words, snap, xlast, state, src, frames, stride, channels, (uint32_t)n_sc);
        e = hipGetLastError();
        aux->tag = state ? 0 : tag;
        aux->ticket_base = 0;
        aux->parity = 0;
    }
    a.gran = reinterpret_cast<float *>(scratch + head + (state ? 0 : aux->parity) * gran_bytes);
    a.gran_other = state ? nullptr : reinterpret_cast<float *>(scratch + head + (aux->parity ^ 1u) * gran_bytes);
    a.ticket_base = aux->ticket_base;
    if (state) a.state_in = snap;
    if (e == hipSuccess) {
        static int occupancy[sizeof(kVariants) / sizeof(kVariants[0])];  // asked once per variant
        int &per_cu_cached = occupancy[v - kVariants];
        if (per_cu_cached == 0) {
            int q = 0;
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, reinterpret_cast<const void *>(v->fn), 64 * (int)NW, 0);
            per_cu_cached = q < 1 ? 1 : q;
        }
        int per_cu = per_cu_cached;
        if (per_cu * (int)NW > 16) per_cu = 16 / (int)NW > 0 ? 16 / (int)NW : 1;
        if (const char *w = rh::knob(rh::K_BIQUAD_WGS)) per_cu = atoi(w) > 0 ? atoi(w) : per_cu;
        uint64_t grid = (uint64_t)rh::g_num_cus * (uint64_t)per_cu;
        const uint64_t total = tiles64 * n_streams;
        if (grid > total) grid = total;
        if (e == hipSuccess) {
            void *args[] = {&a};
            e = hipLaunchKernel(reinterpret_cast<const void *>(v->fn), dim3((uint32_t)grid), dim3(64 * NW), args, 0, s);
            if (e == hipSuccess && !state) {  // every workgroup takes two tickets ahead and one per tile it works on
                aux->ticket_base += (uint32_t)(total + 2 * grid);
                aux->parity ^= 1u;
            }
        }
    }
    if (e != hipSuccess) aux->tag = 0;
    if (e == hipSuccess && state) {
        hipLaunchKernelGGL(k_bq_post, dim3((unsigned)((n_sc + 255) / 256)), dim3(256), 0, s, state, snap, xlast, dst, frames, stride, channels, (uint32_t)n_sc);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        rh::set_hip_error(e, "rh_biquad mode 1 launch");
        return RH_ERR_HIP;
    }
    return RH_OK;
}
}  // namespace rh

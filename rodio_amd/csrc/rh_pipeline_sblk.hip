// rh_pipeline_sblk.hip -- a block of a stream on the summed state in ONE launch (k_rlm_sblk).
//
// What it replaces (mixer.rs:120-136 pulled block-wise: `GpuMixer`, rh_rlm_stream_block / rh_rlm_stream_block_v while the sources run
// together): until round 5 a block was two dependent launches -- k_mix_rows summed the sources' rows at the input rate into one mixed row
// (29 us for a 64 Ki-frame block of 256 sources, 4.7 TB/s: a short kernel over 128 MiB), k_rlm_fast converted and filtered that row (9 us, a
// latency chain over 62 tiles) -- 41 us per block against 17 us of bytes.  Here a block is one kernel:
//
//   * a TILE is a window of KV KiB of every source's row -- 128 * KV stereo frames, windows overlapping by H = 4 frames, so that a tile owns
//     the output frames whose SECOND tap lies in its stride and finds the first tap of its first frame and the taps of the two frames the filter
//     looks back at inside its own window (k_rlm_chunk gets those 4 mixed frames from the tile in front, a memory round trip after the source
//     loop; here they cost 32 bytes per source and tile read twice: 1.0 to 3 % of the bytes);
//   * a tile is a WORKGROUP OF 8 WAVES that share the source list: every wave pulls its eighth of the sources through an LDS-DMA ring of its
//     own (NS stages of KV KiB: 8 x NS x KV KiB in flight per CU, which is what a CU needs to hold its share of the chip's bandwidth while a block
//     has fewer tiles than the chip has CUs) and sums them in registers; the eight partial sums meet in the LDS and wave 0 does what
//     k_rlm_chunk's lone wave does behind its loop: the lerp, the zero-state biquad over runs of R frames per lane, the wave scan of the
//     run-end states, ONE published aggregate per tile, the look-back over the J tiles in front (decoupled: zero-state aggregates, no chain),
//     the homogeneous correction, whole-line stores through the LDS;
//   * nothing comes from the host per block but kernel arguments: the tile boundaries are the converter's own index function evaluated on the
//     device (m_lo(t) = ceil((g0 + t P + H - 1) T / F), continuous conversion), the look-back weights B^d come from ONE table of powers of the
//     filter's state matrix (d = 0 .. the distance after which the filter has forgotten to 2^-40), built when the stream's first block runs;
//   * the stream's state crosses the block boundary as it always did: the summed filter state at the block's first output frame (4 floats,
//     scan basis: StreamArgs::win) enters the tiles it still reaches with the weight B^(m_lo(t) - m0); the last tile writes the state at the
//     block's end (StreamArgs::wout).  The converter needs no state: the caller passes rows that start at the first tap of frame m0 - 2.
//
// Blocks the kernel does not take (spans that restart the converter, ratios that put more than 64 R frames into a window, more tiles than
// fit the chip at once under rh_rlm_set_exclusive(0), a filter that forgets too slowly) run as before.  RH_NO_SBLK=1 is the A/B.
#include <hip/hip_ext.h>
#include "rh_pipeline_dev.h"

namespace {

struct SblkArgs {
    const float *uni;               // the plan's Uniforms as floats in device memory
    const float *powD;              // [Dmax + 1][4]: B^d in the scan basis
    unsigned long long *gran;       // [tiles][4] {epoch, f32 bits}: the tiles' zero-state aggregates
    uint64_t src_off;               // bytes added to every source pointer of the table (a stream whose rows moved on together)
    uint32_t Dmax;
    uint32_t P;                     // stride of the windows in frames (window - P >= H frames of overlap; P * frame bytes a multiple of 16)
    // The converter's position, relative to the block (everything a tile computes fits 32 bits then: the host checks): output frames are counted
    // from frame mb = m0 - mb_off (mb_off = 2: the two frames the filter looks back at; less at the very start of a stream), whose first tap is
    // frame ib of the rows with numerator rb:  frame mb + u reads row frame ib + (rb + u F) / T, numerator (rb + u F) mod T.
    uint32_t ib, rb, mb_off;
    // rh_rlm_stream_overlap: the stream's state between two blocks that are in flight together, as tagged words {tag, f32 bits} beside the plain
    // ones.  hand_in != nullptr: this block was launched WITHOUT a barrier behind the block in front (hipExtAnyOrderLaunch, the same queue):
    // its tiles that the state still reaches wait for the words tagged hand_tag.  hand_out != nullptr: the last tile leaves them, tagged p.epoch.
    const unsigned long long *hand_in;
    unsigned long long *hand_out;
    uint32_t hand_tag;
};

template <int N>
__device__ __forceinline__ void wait_vm_le(uint32_t after) {  // at most `after` groups of N instructions outstanding (after < 16, uniform)
    switch (after) {
    case 0: wait_vm<0>(); break;
    case 1: wait_vm<(N * 1 < 63 ? N * 1 : 63)>(); break;
    case 2: wait_vm<(N * 2 < 63 ? N * 2 : 63)>(); break;
    case 3: wait_vm<(N * 3 < 63 ? N * 3 : 63)>(); break;
    case 4: wait_vm<(N * 4 < 63 ? N * 4 : 63)>(); break;
    case 5: wait_vm<(N * 5 < 63 ? N * 5 : 63)>(); break;
    case 6: wait_vm<(N * 6 < 63 ? N * 6 : 63)>(); break;
    case 7: wait_vm<(N * 7 < 63 ? N * 7 : 63)>(); break;
    case 8: wait_vm<(N * 8 < 63 ? N * 8 : 63)>(); break;
    case 9: wait_vm<(N * 9 < 63 ? N * 9 : 63)>(); break;
    case 10: wait_vm<(N * 10 < 63 ? N * 10 : 63)>(); break;
    default: wait_vm<(N * 11 < 63 ? N * 11 : 63)>(); break;
    }
}

constexpr int kSblkWaves = 8, kSblkH = 4;  // loader waves of a tile (one more wave works out the tile's bounds and taps meanwhile, and runs what follows the sum)

// R: frames per lane of wave 0's runs; C: channels; KV: KiB of a window; NS: ring stages per wave.
template <int R, int C, int KV, int NS>
__global__ __launch_bounds__(64 * (kSblkWaves + 1), (NS * KV <= 8 ? 5 : 3)) void k_rlm_sblk(const Params p, const SblkArgs q) {
    typedef Chan<C> CH;
    typedef typename CH::V V;
    constexpr int W = kSblkWaves, H = kSblkH;
    constexpr uint32_t FB = CH::kFB;
    constexpr uint32_t kStage = KV * 1024, Wd = kStage / FB;  // bytes / frames of a window
    constexpr uint32_t kRing = NS * kStage;                   // a wave's ring
    constexpr uint32_t kImg = W * kRing;                      // the mixed window: kStage bytes + 16 zero bytes behind it (a dummy tap; the second tap of a verbatim last frame)
    constexpr uint32_t kTile = kImg + kStage + 32;            // the tile a workgroup took by ticket
    constexpr uint32_t kRow = (R + 1) * FB;                   // wave 0's output rows, in wave 1's ring (idle by then)
    static_assert(NS <= 12 && 64 * kRow <= kRing && NS >= 2, "geometry");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kImg + kStage + 64];
    lds_u8 *const lds = (lds_u8 *)smem;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
    const int lane = threadIdx.x & 63;
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane((uint32_t)threadIdx.x >> 6);
    const bool tail_wave = wave_id == 0;                // wave 0: the tile's bounds, taps, and everything behind the sum
    const uint32_t wave = tail_wave ? 0u : wave_id - 1u;  // loaders 0 .. W - 1
    // Every tile resident at once and nothing else on the chip (the host knows): tile = workgroup.  Otherwise tiles by ticket from eight
    // counters, one per XCD, as in k_rlm_chunk: a tile then only ever waits for tiles that hold a slot or are done.
    uint32_t tile = blockIdx.x;
    if (!p.direct) {
        if (threadIdx.x == 0) {
            const uint32_t x = blockIdx.x & 7u;
            const uint32_t k = atomicAdd(p.ticket + 32u * (1u + x), 1u) - p.shard_base;
            *(RH_LDS uint32_t *)(lds + kTile) = x + 8u * k;
        }
        __syncthreads();
        tile = __builtin_amdgcn_readfirstlane(*(RH_LDS uint32_t *)(lds + kTile));
    }
    if (tile >= p.n_tiles) return;
    const uint32_t Ns = p.eq_frames, S = p.n_sources, P = q.P;
    const uint32_t nvec = Ns * C / 4;  // 16-byte vectors of a row (host: whole vectors)

    // ---- this wave's share of the source list: pointers and gains in vector registers (lane l: source s_begin + l), read back with
    // v_readlane inside the loop -- no scalar memory traffic next to the ds_reads' lgkmcnt ----
    const uint32_t Sw = (S + W - 1) / W;
    const uint32_t s_begin = wave * Sw < S ? wave * Sw : S;
#if defined(RH_SBLK_DIAG) && RH_SBLK_DIAG == 2  // diagnostics builds (wrong results): the kernel without its loads
    const uint32_t n_mine = 0u;
#else
    const uint32_t n_mine = tail_wave ? 0u : (S - s_begin < Sw ? S - s_begin : Sw);
#endif
    const uint32_t v0 = tile * (P * FB / 16u);
    uint32_t goff[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        uint32_t j = v0 + (uint32_t)k * 64 + lane;
        j = j < nvec ? j : nvec - 1;  // past the end of the row: its last vector again (finite, never a tap of a stored frame)
        goff[k] = j * 16;
    }
    const bool lin = v0 + (uint32_t)(KV * 64) <= nvec;
    const uint32_t ring0 = lds0 + wave * kRing;
    v4f acc[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) acc[k] = v4f{0.f, 0.f, 0.f, 0.f};

    // ---- wave 0: the tile's bounds, the lanes' taps and the look-back weights, worked out while the windows stream ----
    const uint32_t u_end = q.mb_off + p.st_active;  // (output frames relative to frame mb, see SblkArgs)
    auto ulo_of = [&](uint32_t t) -> uint32_t {  // the first output frame whose second tap lies in the stride of tile t or behind it
        if (t == 0) return q.mb_off;
        const uint32_t a = t * P + (uint32_t)(H - 1);          // its first tap is row frame a or behind it
        const uint32_t need = a > q.ib ? (a - q.ib) * p.T : 0u;  // (rb + u F) >= need
        uint32_t u = need > q.rb ? (need - q.rb + p.F - 1u) / p.F : 0u;
        u = u < q.mb_off ? q.mb_off : u;
        return u > u_end ? u_end : u;
    };
    uint32_t n_t = 0;       // the tile's output frames
    uint32_t u_lo = 0;
    int nfl = 0;
    int offA[R + 2];
    float wgt[R + 2];
    float lM[4], b15[4], b31[4], kM[4], pwv[4], wM[4], eM[4], win[2 * C];
    uint32_t Jc = 0;
    bool win_on = false, last_tile = false;
    float U = 0.f;
    uint64_t a_out = 0;
#pragma unroll
    for (int k = 0; k < 2 * C; ++k) win[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) lM[k] = b15[k] = b31[k] = kM[k] = pwv[k] = wM[k] = eM[k] = 0.f;
#pragma unroll
    for (int rr = 0; rr < R + 2; ++rr) offA[rr] = (int)(kImg + kStage), wgt[rr] = 0.f;

    // ---- the ring: prime it ----
    uint32_t plo = 0, phi = 0;
    float gv = 0.f;
    if ((uint32_t)lane < n_mine) {  // lane l: source s_begin + l
        const SrcDesc *d = p.srcs + (s_begin + (uint32_t)lane);
        const uint64_t a = (uint64_t)(uintptr_t)d->data + q.src_off;
        plo = (uint32_t)a;
        phi = (uint32_t)(a >> 32);
        gv = d->gain;
    }
    // The table has landed HERE: the compiler waits for its own loads where their values are first used, and it does not see the DMA below -- a
    // gain first used inside the loop would put an `s_waitcnt vmcnt(0)` there, draining the ring in every iteration (measured: 0.75 us per
    // source, one memory round trip each).
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(plo), "+v"(phi), "+v"(gv) : : "memory");
    auto stage_source = [&](uint32_t i, uint32_t st) {
        const uint64_t a = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)phi, (int)i) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)plo, (int)i);
        const void *data = (const void *)(uintptr_t)a;
        if (lin) {
            glds16_run<KV>(data, goff[0], ring0 + st * kStage);
        } else {
#pragma unroll
            for (int k = 0; k < KV; ++k) glds16(data, goff[k], ring0 + st * kStage + k * 1024);
        }
    };
    uint32_t issued = 0;
    for (; issued < n_mine && issued < (uint32_t)NS; ++issued) stage_source(issued, issued);

    if (tail_wave) {
        u_lo = ulo_of(tile);
        last_tile = tile + 1u == p.n_tiles;
        // (the last tile takes what is left: the verbatim last frame of a stream, whose second tap lies behind the row, belongs to no stride)
        const uint32_t u_hi = last_tile ? u_end : ulo_of(tile + 1u);
        n_t = u_hi - u_lo;  // <= 64 * R (host)
        nfl = (int)n_t - lane * R < 0 ? 0 : ((int)n_t - lane * R > R ? R : (int)n_t - lane * R);
        const uint32_t nl0 = (n_t + R - 1) / R;
        const uint32_t v = nl0 ? n_t - (nl0 - 1) * R : 0;  // frames of the last lane's run, 1 .. R
        const Tables *__restrict__ tb = p.tabs;
        Jc = p.J < tile ? p.J : tile;
        // lane j < Jc weighs the aggregate of tile - 1 - j with B^(m_lo(tile) - m_lo(tile - j))
        uint32_t dj = 0;
        if ((uint32_t)lane < Jc) dj = u_lo - ulo_of(tile - (uint32_t)lane);
        const bool far = dj > q.Dmax;  // (forgotten: weight 0)
        const float *kp = q.powD + 4 * (uint64_t)(far ? 0u : dj);
        const uint32_t dW = u_lo - q.mb_off;
        win_on = p.st_win != nullptr && dW <= q.Dmax;
        const float *wp = q.powD + 4 * (uint64_t)(win_on ? dW : 0u);
        const float *ep = q.powD + 4 * (uint64_t)(n_t <= q.Dmax ? n_t : q.Dmax);
        const float *pw = q.powD + 4 * v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lM[k] = tb->laneM[lane][k];
            b15[k] = tb->bc15M[lane][k];
            b31[k] = tb->bc31M[lane][k];
            kM[k] = ((uint32_t)lane < Jc && !far) ? kp[k] : 0.f;
            wM[k] = wp[k];
            eM[k] = n_t <= q.Dmax ? ep[k] : 0.f;
            pwv[k] = pw[k];
        }
        if (win_on && !q.hand_in) {
#pragma unroll
            for (int k = 0; k < 2 * C; ++k) win[k] = p.st_win[k];
        }
        U = q.uni[lane < 61 ? lane : 60];
        a_out = (uint64_t)(uintptr_t)(p.out + ((uint64_t)(u_lo - q.mb_off) + (uint32_t)lane) * C);
        // taps and weights of the lane's R + 2 frames (frames u0 - 2 .. u0 + R - 1 behind frame mb): offsets into the mixed window
        const uint32_t u0 = u_lo + (uint32_t)lane * R;
        const bool early = u0 < 2u;                   // the stream's start: x'[-1] = x'[-2] = 0 (mb_off < 2 only there)
        const uint32_t uf = early ? 0u : u0 - 2u;
        const uint32_t x = q.rb + uf * p.F;
        uint32_t il = q.ib + x / p.T, num = x % p.T;  // row frame and numerator of frame mb + uf
        const int wbase = (int)(tile * P);            // row frame of the window's first frame
#pragma unroll
        for (int rr = 0; rr < R + 2; ++rr) {
            const bool dummy = early && (uint32_t)rr + u0 < 2u;
            int li = (int)il;
            uint32_t nm = num;
            if (li + 1 >= (int)Ns) {  // the last frame is emitted verbatim (sample_rate.rs:193-200); frames past it are never stored
                li = (int)Ns - 1;
                nm = 0;
            }
            int f = li - wbase;       // frame of the window
            f = f < 0 ? 0 : (f > (int)Wd - 1 ? (int)Wd - 1 : f);  // (only frames that are not stored leave the range)
            offA[rr] = dummy ? (int)(kImg + kStage) : (int)kImg + f * (int)FB;
            wgt[rr] = dummy ? 0.0f : (float)nm / p.Tf;
            if (!dummy) {  // the next frame: + F / T
                il += p.qF;
                num += p.rF;
                if (num >= p.T) {
                    num -= p.T;
                    il += 1;
                }
            }
        }
    }

    // ---- this wave's sources (at most 64: the host keeps larger batches on the two-launch path): windows through the ring, summed in registers ----
    for (uint32_t i = 0; i < n_mine; ++i) {
        const float g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gv), (int)i));
        wait_vm_le<KV>(issued - i - 1u);
        const lds_u8 *buf = lds + wave * kRing + (i % (uint32_t)NS) * kStage;
        v4f v[KV];
#pragma unroll
        for (int k = 0; k < KV; ++k) v[k] = *(const lds_f4 *)(buf + k * 1024 + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the window is in registers: its stage is free ...
        if (issued < n_mine) {                              // ... for the source NS ahead
            stage_source(issued, i % (uint32_t)NS);
            ++issued;
        }
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            acc[k].x = fma_(g, v[k].x, acc[k].x);
            acc[k].y = fma_(g, v[k].y, acc[k].y);
            acc[k].z = fma_(g, v[k].z, acc[k].z);
            acc[k].w = fma_(g, v[k].w, acc[k].w);
        }
    }
    // ---- the eight partial sums meet in the LDS ----
    if (!tail_wave) {
#pragma unroll
        for (int k = 0; k < KV; ++k) *(lds_f4 *)(lds + wave * kRing + k * 1024 + lane * 16) = acc[k];
    }
    __syncthreads();
    if (!tail_wave) return;
#if defined(RH_SBLK_DIAG) && RH_SBLK_DIAG == 1  // diagnostics builds (wrong results): the kernel without what follows the sum
    if (lane == 0) p.out[tile] = (float)offA[0] + wgt[1] + lM[0] + kM[1] + (float)nfl + U + pwv[0] + wM[0] + eM[0] + win[0] + (float)a_out + (last_tile ? 1.f : 0.f) + (win_on ? 1.f : 0.f);
    return;
#endif
#pragma unroll
    for (int w = 0; w < W; ++w) {
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const v4f t = *(const lds_f4 *)(lds + (uint32_t)w * kRing + k * 1024 + lane * 16);
            acc[k].x += t.x, acc[k].y += t.y, acc[k].z += t.z, acc[k].w += t.w;
        }
    }
#pragma unroll
    for (int k = 0; k < KV; ++k) *(lds_f4 *)(lds + kImg + (uint32_t)(k * 64 + lane) * 16) = acc[k];
    if (lane == 0) *(lds_f4 *)(lds + kImg + kStage) = v4f{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    // ---- the lane's run of the mixed stream: lerp, zero-state biquad; the run-end state after nfl frames (as k_rlm_chunk) ----
    const float b0 = readlane_f(U, 0), c1 = readlane_f(U, 1), c2 = readlane_f(U, 2), na1 = -readlane_f(U, 3), na2 = -readlane_f(U, 4);
    V out[R];
    V E1 = CH::zero(), E2 = CH::zero();
    {
        V ta[R + 2], tb2[R + 2];
#pragma unroll
        for (int rr = 0; rr < R + 2; ++rr) {
            ta[rr] = CH::ld_lds(lds + offA[rr]);
            tb2[rr] = CH::ld_lds(lds + offA[rr] + FB);
        }
        auto tap = [&](int rr) -> V {
            V x;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) CH::set(x, ch, fma_(CH::get(tb2[rr], ch) - CH::get(ta[rr], ch), wgt[rr], CH::get(ta[rr], ch)));
            return x;
        };
        V x2 = tap(0);  // (dummy taps read zeros with weight 0)
        V x1 = tap(1);
        V w1 = CH::zero(), w2 = CH::zero();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const V x = tap(r + 2);
            V w;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                const float wc = fma_(na1, CH::get(w1, ch), fma_(na2, CH::get(w2, ch), fma_(c2, CH::get(x2, ch), c1 * CH::get(x1, ch))));
                CH::set(w, ch, wc);
                CH::set(out[r], ch, fma_(b0, CH::get(x, ch), wc));
            }
            w2 = w1;
            w1 = w;
            x2 = x1;
            x1 = x;
            E1 = vsel(r + 1 == nfl, w1, E1);
            E2 = vsel(r + 1 == nfl, w2, E2);
        }
    }
    // ---- scan of the run-end states (scan basis) ----
    float Pq[2 * C];
#pragma unroll
    for (int k = 0; k < 2 * C; ++k) Pq[k] = 0.f;
    {
        const float Tm[4] = {readlane_f(U, 5), readlane_f(U, 6), readlane_f(U, 7), readlane_f(U, 8)};
#pragma unroll
        for (int ch = 0; ch < C; ++ch) mat_acc(Tm, CH::get(E1, ch), CH::get(E2, ch), Pq[2 * ch], Pq[2 * ch + 1]);
    }
    float own[2 * C];
#pragma unroll
    for (int k = 0; k < 2 * C; ++k) own[k] = Pq[k];
#define RH_SSCAN(K, N)                                                                             \
    {                                                                                              \
        float sq[2 * C];                                                                           \
        _Pragma("unroll") for (int k = 0; k < 2 * C; ++k) sq[k] = dpp0<kDppRowShr + N, 0xf>(Pq[k]); \
        const float sM[4] = {readlane_f(U, 9 + 4 * K), readlane_f(U, 10 + 4 * K), readlane_f(U, 11 + 4 * K), readlane_f(U, 12 + 4 * K)}; \
        _Pragma("unroll") for (int ch = 0; ch < C; ++ch) mat_acc(sM, sq[2 * ch], sq[2 * ch + 1], Pq[2 * ch], Pq[2 * ch + 1]); \
    }
    RH_SSCAN(0, 1)
    RH_SSCAN(1, 2)
    RH_SSCAN(2, 4)
    RH_SSCAN(3, 8)
#undef RH_SSCAN
    {
        float sq[2 * C];
#pragma unroll
        for (int k = 0; k < 2 * C; ++k) sq[k] = dpp0<kDppBcast15, 0xa>(Pq[k]);
#pragma unroll
        for (int ch = 0; ch < C; ++ch) mat_acc(b15, sq[2 * ch], sq[2 * ch + 1], Pq[2 * ch], Pq[2 * ch + 1]);
    }
    {
        float sq[2 * C];
#pragma unroll
        for (int k = 0; k < 2 * C; ++k) sq[k] = dpp0<kDppBcast31, 0xc>(Pq[k]);
#pragma unroll
        for (int ch = 0; ch < C; ++ch) mat_acc(b31, sq[2 * ch], sq[2 * ch + 1], Pq[2 * ch], Pq[2 * ch + 1]);
    }
    float A[2 * C];  // the tile aggregate: the short last run on top of the inclusive prefix of the lane before it
    {
        const int nl = (int)((n_t + R - 1) / R);  // lanes with frames (uniform)
#pragma unroll
        for (int k = 0; k < 2 * C; ++k) A[k] = 0.f;
        if (nl >= 1) {
#pragma unroll
            for (int k = 0; k < 2 * C; ++k) A[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(own[k]), nl - 1));
        }
        if (nl >= 2) {
            const float M[4] = {readfirstlane_f(pwv[0]), readfirstlane_f(pwv[1]), readfirstlane_f(pwv[2]), readfirstlane_f(pwv[3])};  // B^v
            float xp[2 * C];
#pragma unroll
            for (int k = 0; k < 2 * C; ++k) xp[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Pq[k]), nl - 2));
#pragma unroll
            for (int ch = 0; ch < C; ++ch) mat_acc(M, xp[2 * ch], xp[2 * ch + 1], A[2 * ch], A[2 * ch + 1]);
        }
        if (lane < 2 * C) {
            float ev = A[0];
#pragma unroll
            for (int k = 1; k < 2 * C; ++k) ev = lane == k ? A[k] : ev;
            __hip_atomic_store(q.gran + (uint64_t)tile * 4 + (uint32_t)lane, ((unsigned long long)p.epoch << 32) | __float_as_uint(ev), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    float Q[2 * C];
#pragma unroll
    for (int k = 0; k < 2 * C; ++k) Q[k] = dpp0<kDppWaveShr1, 0xf>(Pq[k]);  // exclusive: the prefix of the lanes before (all of them whole runs)
    // ---- the tile carry: lane j < Jc polls tile - 1 - j; the stream's state at the block start where it still reaches ----
    float c[2 * C];
#pragma unroll
    for (int k = 0; k < 2 * C; ++k) c[k] = 0.f;
    bool dead = false;
    if (Jc) {
        const bool want = (uint32_t)lane < Jc;
        const unsigned long long *gp = q.gran + (uint64_t)(tile - 1u - (want ? (uint32_t)lane : 0u)) * 4;
        unsigned long long gvw[2 * C];
#pragma unroll
        for (int k = 0; k < 2 * C; ++k) gvw[k] = 0;
        bool ok = false;
        uint32_t spins = 0;
        while (true) {
            if (want && !ok) {
                bool all = true;
#pragma unroll
                for (int k = 0; k < 2 * C; ++k) {
                    gvw[k] = __hip_atomic_load(gp + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    all = all && ((uint32_t)(gvw[k] >> 32) == p.epoch);
                }
                ok = all;
            }
            if (__all(ok || !want)) break;
            if (++spins > kSpinLimit) {
                if (lane == 0) atomicOr(p.status, 1u);
                dead = true;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (want && ok && !dead) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) mat_acc(kM, __uint_as_float((uint32_t)gvw[2 * ch]), __uint_as_float((uint32_t)gvw[2 * ch + 1]), c[2 * ch], c[2 * ch + 1]);
        }
#pragma unroll
        for (int k = 0; k < 2 * C; ++k) {  // sum over lanes 0..31 -> uniform
            c[k] += dpp0<kDppRowShr + 1, 0xf>(c[k]);
            c[k] += dpp0<kDppRowShr + 2, 0xf>(c[k]);
            c[k] += dpp0<kDppRowShr + 4, 0xf>(c[k]);
            c[k] += dpp0<kDppRowShr + 8, 0xf>(c[k]);
            c[k] = readlane_f(c[k], 15) + readlane_f(c[k], 31);
        }
    }
    if (win_on && q.hand_in) {  // the block in front still runs: its last tile leaves the state as tagged words (it holds a slot or is done: see sblk_try)
        unsigned long long hv[2 * C];
        uint32_t spins = 0;
        while (true) {
            bool all = true;
#pragma unroll
            for (int k = 0; k < 2 * C; ++k) {
                hv[k] = __hip_atomic_load(q.hand_in + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                all = all && ((uint32_t)(hv[k] >> 32) == q.hand_tag);
            }
            if (all) break;
            if (++spins > kSpinLimit) {
                if (lane == 0) atomicOr(p.status, 1u);
                dead = true;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
#pragma unroll
        for (int k = 0; k < 2 * C; ++k) win[k] = __uint_as_float((uint32_t)hv[k]);
    }
    if (win_on) {  // + B^(m_lo - m0) * (the stream's state at m0)
        const float M[4] = {readfirstlane_f(wM[0]), readfirstlane_f(wM[1]), readfirstlane_f(wM[2]), readfirstlane_f(wM[3])};
#pragma unroll
        for (int ch = 0; ch < C; ++ch) mat_acc(M, readfirstlane_f(win[2 * ch]), readfirstlane_f(win[2 * ch + 1]), c[2 * ch], c[2 * ch + 1]);
    }
    if (dead) {  // a hand-off that never arrived: the status word fails the call, the tile is poisoned
#pragma unroll
        for (int k = 0; k < 2 * C; ++k) c[k] = __builtin_nanf("");
    }
    if (last_tile && p.st_mode == 1 && p.st_wout && lane < 2 * C) {  // the stream's state at the block's end: A + B^(n_t) * carry
        const float M[4] = {readfirstlane_f(eM[0]), readfirstlane_f(eM[1]), readfirstlane_f(eM[2]), readfirstlane_f(eM[3])};
        float e[2 * C];
#pragma unroll
        for (int k = 0; k < 2 * C; ++k) e[k] = A[k];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) mat_acc(M, c[2 * ch], c[2 * ch + 1], e[2 * ch], e[2 * ch + 1]);
        float ev = e[0];
#pragma unroll
        for (int k = 1; k < 2 * C; ++k) ev = lane == k ? e[k] : ev;
        p.st_wout[lane] = ev;
        if (q.hand_out) __hip_atomic_store(q.hand_out + lane, ((unsigned long long)p.epoch << 32) | __float_as_uint(ev), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int ch = 0; ch < C; ++ch) mat_acc(lM, c[2 * ch], c[2 * ch + 1], Q[2 * ch], Q[2 * ch + 1]);  // start state of the lane's run = Q + B^(R*lane) * carry
    {
        lds_u8 *row = lds + kRing + (uint32_t)lane * kRow;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            V y;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) CH::set(y, ch, fma_(readlane_f(U, 25 + 2 * r), Q[2 * ch], fma_(readlane_f(U, 26 + 2 * r), Q[2 * ch + 1], CH::get(out[r], ch))));
            if (C == 2) *(lds_f2 *)(row + r * FB) = v2f{CH::get(y, 0), CH::get(y, C - 1)};
            else *(RH_LDS float *)(row + r * FB) = CH::get(y, 0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    {
        float *ot = (float *)(uintptr_t)a_out;  // this lane's frame of every group of 64
        for (uint32_t f0 = 0; f0 < n_t; f0 += 64) {
            const uint32_t f = f0 + (uint32_t)lane;
            if (f < n_t) {
                const lds_u8 *src2 = lds + kRing + (f / R) * kRow + (f % R) * FB;
                if (C == 2) {
                    const v2f a = *(const lds_f2 *)src2;
                    *reinterpret_cast<float2 *>(ot + (uint64_t)f0 * 2) = make_float2(a.x, a.y);
                } else {
                    ot[f0] = *(const RH_LDS float *)src2;
                }
            }
        }
    }
}

struct Inst {
    int R, C, KV, NS;
    const void *fn;
};
#define RH_SBLK(R, C, KV, NS) \
    Inst { R, C, KV, NS, reinterpret_cast<const void *>(&k_rlm_sblk<R, C, KV, NS>) }
const Inst kInst[] = {
    RH_SBLK(3, 2, 1, 12), RH_SBLK(5, 2, 2, 6), RH_SBLK(7, 2, 3, 4), RH_SBLK(9, 2, 4, 3),
    RH_SBLK(5, 1, 1, 12), RH_SBLK(9, 1, 2, 6),
};
#undef RH_SBLK

}  // namespace

// ---- host side ----------------------------------------------------------------------------------------------------------------------
namespace rhp {

struct SblkPlan {
    int R = 0;                 // the R the filter tables were built for (0: none yet)
    Uniforms uni;
    Tables *d_tabs = nullptr;
    float *d_uni = nullptr, *d_pow = nullptr;
    uint32_t Dmax = 0;
    unsigned long long *d_gran = nullptr;  // [3][cap_tiles][4]: the tiles' aggregates, one set per block in flight and one spare
    size_t cap_tiles = 0;
    bool unusable = false;     // the filter forgets too slowly for a table of powers
    // rh_rlm_stream_overlap: the stream's state as tagged words, one set per block in flight (and one spare)
    unsigned long long *d_hand = nullptr;  // [3][4]
    uint32_t last_tag = 0;                 // the tag the block in front leaves there (its epoch), in set last_set
    int last_set = 0;
    bool last_handed = false;              // ... if it leaves one (a block of a running stream under rh_rlm_stream_overlap)
    hipStream_t last_stream = nullptr;     // the stream the block in front was launched on
    uint32_t n_chained = 0;                // blocks launched without a barrier behind the block in front (diagnostics)
    uint32_t blk = 0;                      // blocks of the current stream that this kernel ran
    bool prev_sblk = false;                // ... and the block before this one was one of them
    uint64_t seen_version = ~0ull;         // the source table the block in front read (a block behind a new upload starts behind a barrier)
};

void sblk_other_block(rh_rlm *p, bool stream_begins) {  // a block of the stream ran elsewhere (or the stream begins): the state lives in the plain words again
    SblkPlan *s = static_cast<SblkPlan *>(p->sblk);
    if (!s) return;
    s->prev_sblk = false;
    if (stream_begins) s->n_chained = 0;
}

uint32_t sblk_chained_blocks(const rh_rlm *p) {
    const SblkPlan *s = static_cast<const SblkPlan *>(p->sblk);
    return s ? s->n_chained : 0u;
}

void sblk_free(rh_rlm *p) {
    SblkPlan *s = static_cast<SblkPlan *>(p->sblk);
    if (!s) return;
    if (s->d_tabs) (void)hipFree(s->d_tabs);
    if (s->d_uni) (void)hipFree(s->d_uni);
    if (s->d_pow) (void)hipFree(s->d_pow);
    if (s->d_gran) (void)hipFree(s->d_gran);
    if (s->d_hand) (void)hipFree(s->d_hand);
    delete s;
    p->sblk = nullptr;
}

static rh_status sblk_tables(rh_rlm *p, SblkPlan &s, int R) {
    if (s.R == R) return RH_OK;
    const rh_status w = wait_idle(p);  // an earlier block may still read the tables
    if (w != RH_OK) return w;
    if (s.d_tabs) RH_HIP_TRY(hipFree(s.d_tabs));
    if (s.d_uni) RH_HIP_TRY(hipFree(s.d_uni));
    s.d_tabs = nullptr, s.d_uni = nullptr, s.R = 0;
    const M2 A{-(double)p->coeffs[3], -(double)p->coeffs[4], 1.0, 0.0};
    M2 Tm, Ti;
    scan_basis((double)p->coeffs[3], (double)p->coeffs[4], Tm, Ti);
    const M2 B = mul(mul(Tm, A), Ti);
    if (!s.d_pow) {  // B^d until the filter has forgotten (||B^d|| < 2^-40), once per handle
        std::vector<float> pw;
        M2 cur{1, 0, 0, 1};
        uint32_t d = 0;
        for (; d <= (1u << 16); ++d) {
            pw.resize((size_t)(d + 1) * 4);
            put(&pw[(size_t)d * 4], cur);
            if (d > 0 && norm(cur) < 0x1p-40) break;
            cur = mul(cur, B);
        }
        if (d > (1u << 16)) {
            s.unusable = true;
            return RH_OK;
        }
        RH_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s.d_pow), pw.size() * 4));
        RH_HIP_TRY(hipMemcpy(s.d_pow, pw.data(), pw.size() * 4, hipMemcpyHostToDevice));
        s.Dmax = d;
    }
    Tables *h = new Tables();
    std::memset(h, 0, sizeof(Tables));
    Uniforms &U = s.uni;
    std::memset(&U, 0, sizeof(U));
    U.b0 = p->coeffs[0];
    U.c1 = (float)((double)p->coeffs[1] - (double)p->coeffs[0] * (double)p->coeffs[3]);
    U.c2 = (float)((double)p->coeffs[2] - (double)p->coeffs[0] * (double)p->coeffs[4]);
    U.a1 = p->coeffs[3];
    U.a2 = p->coeffs[4];
    put(U.Tm, Tm);
    for (int k = 0; k < 4; ++k) put(U.scanM[k], mpow(B, (uint64_t)R << k));
    for (int r = 0; r < R; ++r) {
        const M2 m = mul(mpow(A, r + 1), Ti);
        U.g[r][0] = (float)m.a;
        U.g[r][1] = (float)m.b;
    }
    for (int l = 0; l < 64; ++l) {
        put(h->laneM[l], mpow(B, (uint64_t)R * l));
        put(h->bc15M[l], mpow(B, (uint64_t)R * ((l & 15) + 1)));
        put(h->bc31M[l], mpow(B, (uint64_t)R * ((l & 31) + 1)));
    }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&s.d_tabs), sizeof(Tables));
    if (e == hipSuccess) e = hipMemcpy(s.d_tabs, h, sizeof(Tables), hipMemcpyHostToDevice);
    static_assert(offsetof(Uniforms, Tm) == 20 && offsetof(Uniforms, scanM) == 36 && offsetof(Uniforms, g) == 100, "the kernel reads the Uniforms by float index");
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s.d_uni), sizeof(Uniforms));
    if (e == hipSuccess) e = hipMemcpy(s.d_uni, &U, sizeof(Uniforms), hipMemcpyHostToDevice);
    delete h;
    if (e != hipSuccess) {
        rh::set_hip_error(e, "k_rlm_sblk tables");
        return e == hipErrorOutOfMemory ? RH_ERR_NOMEM : RH_ERR_HIP;
    }
    s.R = R;
    return RH_OK;
}

// One block of a stream on the summed state (stream_block_summed): `out` output frames from global frame sa.m0 of rows that start at global
// input frame sa.g0 and hold `avail` frames each; the source table is on the device.  *taken = false: the block is not this kernel's.
rh_status sblk_try(rh_rlm *p, uint32_t n_sources, uint64_t avail, uint64_t out, float *dst, const StreamArgs &sa, hipStream_t hs, bool *taken) {
    *taken = false;
    const uint32_t C = p->cfg.channels;
    if (!p->filt || (C != 1 && C != 2) || p->st_chunk_in || n_sources < 2 || n_sources > 64u * kSblkWaves || out == 0 || rh::knob(rh::K_NO_SBLK) || rh::knob(rh::K_NO_MIX_FIRST)) return RH_OK;
    if ((avail * C) % 4 != 0 || avail < 8 || avail >= (1ull << 29)) return RH_OK;
    const uint64_t F = p->F, T = p->T, H = kSblkH, FB = 4ull * C;
    if (2 * F > 3 * T) return RH_OK;  // the two frames the filter looks back at start at most 3 input frames in front of a frame's first tap
    if (sa.m0 + out >= (1ull << 44) || sa.g0 >= (1ull << 40)) return RH_OK;  // (m * F stays inside 64 bits)
    if ((out + 8) * F + T >= (1ull << 31) || (avail + 8) * T >= (1ull << 31)) return RH_OK;  // (the tiles count in 32 bits, relative to the block)
    // the input frames the block's output reaches: up to the second tap of its last frame
    const uint64_t i_last = (uint64_t)(((unsigned __int128)(sa.m0 + out - 1) * F) / T);
    uint64_t reach = i_last + 2 > sa.g0 ? i_last + 2 - sa.g0 : 1;
    reach = reach < avail ? reach : avail;
    if (!p->sblk) p->sblk = new SblkPlan();
    SblkPlan &s = *static_cast<SblkPlan *>(p->sblk);
    if (s.unusable) return RH_OK;
    // the instance: the smallest window whose tiles fit the chip one per CU (more, smaller tiles would queue behind each other; fewer, larger
    // ones leave CUs idle); RH_SBLK_KV pins it
    const bool ovl = p->st_overlap && p->exclusive && !rh::knob(rh::K_SBLK_NO_OVERLAP);  // rh_rlm_stream_overlap (rows resident: the caller's promise)
    const Inst *pick = nullptr;
    uint64_t tiles = 0, P = 0;
    const char *pin = rh::knob(rh::K_SBLK_KV);
    const Inst *const tab = kInst;
    const size_t n_tab = sizeof(kInst) / sizeof(kInst[0]);
    for (size_t ii = 0; ii < n_tab; ++ii) {
        const Inst &in = tab[ii];
        if ((uint32_t)in.C != C) continue;
        if (pin && atoi(pin) != in.KV) continue;
        const uint64_t Wd = (uint64_t)in.KV * 1024 / FB, Pmax = Wd - H;
        uint64_t t = (reach > H ? reach - H + Pmax - 1 : Pmax) / Pmax;  // windows of stride Pmax that cover `reach` frames
        if (t == 0) t = 1;
        // ... at an even stride (whole 16-byte vectors), so that the tiles are of one size
        const uint64_t vf = 16 / FB;
        uint64_t Pe = ((reach > H ? reach - H : 1) + t - 1) / t;
        Pe = (Pe + vf - 1) / vf * vf;
        if (Pe > Pmax) Pe = Pmax / vf * vf;
        if ((Wd * T + F - 1) / F + 3 > 64ull * in.R) continue;  // more output frames in a window than 64 runs hold
        pick = &in;
        tiles = t;
        P = Pe;
        if (t <= (uint64_t)rh::g_num_cus) break;
    }
    if (!pick || tiles == 0 || tiles > 0x3fffffull) return RH_OK;
    {
        const rh_status st = sblk_tables(p, s, pick->R);
        if (st != RH_OK) return st;
        if (s.unusable || !s.d_tabs) return RH_OK;
    }
    // J: tiles in front of a tile that the filter has not forgotten (a tile owns at least (P - 1) T / F - 1 frames)
    const uint64_t n_min = (P - 1) * T / F >= 2 ? (P - 1) * T / F - 1 : 1;
    const uint64_t J = ((uint64_t)s.Dmax + n_min - 1) / n_min;
    if (J == 0 || J > 32) return RH_OK;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pick->fn, 64 * (kSblkWaves + 1), 0) != hipSuccess || per_cu < 1) return RH_OK;
    const bool direct = p->exclusive && tiles <= (uint64_t)rh::g_num_cus * (uint64_t)per_cu;
    if (!direct && tiles > 8ull * (uint64_t)rh::g_num_cus * (uint64_t)per_cu) return RH_OK;  // (long blocks: the two-launch form reaches the chip's rate there)
    if ((size_t)tiles > s.cap_tiles) {
        const rh_status w = wait_idle(p);
        if (w != RH_OK) return w;
        if (s.d_gran) RH_HIP_TRY(hipFree(s.d_gran));
        s.d_gran = nullptr, s.cap_tiles = 0;
        const size_t cap = (size_t)tiles + 64;
        RH_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s.d_gran), 3 * cap * 32));
        RH_HIP_TRY(rh::fill_now(s.d_gran, 0, 3 * cap * 32));  // tag 0 = never written (launch tags start at 1)
        s.cap_tiles = cap;
    }
    // rh_rlm_stream_overlap -- a block that starts while the block in front still runs.  Two streams are two hardware queues, and a kernel that
    // waits for one that sits undispatched in ANOTHER queue waits for the scheduler's time slice (measured: 28-65 ms a block), so both blocks go
    // to the caller's stream, this one WITHOUT the barrier a launch normally carries (hipExtAnyOrderLaunch).  What that does on this part
    // (tools/ubench/any_order.hip, profiles/r06_any_order_ubench.txt): every XCD takes the queue's kernels in order on its own -- this block's
    // workgroups start on an XCD as soon as that XCD has finished its share of the block in front, while the other XCDs still work on theirs.
    // So the tail of one block and the ramp of the next overlap across XCDs, the block in front has all its workgroups on the chip when this
    // one's first workgroup starts (they fit it at once: `direct`), and the only thing this block needs of it -- the stream's state -- is
    // waited for inside the kernel by the few tiles it still reaches (tagged words, SblkArgs::hand_in).  The next launch WITH a barrier --
    // anything else on the stream -- waits for both.  Conditions: the block in front was one of these, on this stream, reading the same source
    // table (nothing of ours was queued in between), and left the tagged state.  Three sets of hand-off tables in rotation: block k + 2 starts
    // on an XCD only when block k + 1 is done there, and block k + 1's tiles there looked back at tiles of the other XCDs, which started behind
    // block k's (tiles >= 16: every XCD holds a tile that looks back across all eight) -- the third set is slack on top of that argument.
    const int set = (int)(s.blk % 3u);
    if (ovl && !s.d_hand) {
        RH_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s.d_hand), 3 * 4 * sizeof(unsigned long long)));
        RH_HIP_TRY(rh::fill_now(s.d_hand, 0, 3 * 4 * sizeof(unsigned long long)));  // tag 0 = never written
    }
    bool chained = ovl && s.prev_sblk && s.last_handed && s.last_stream == hs && s.seen_version == p->srcs_version && direct && tiles >= 16 && sa.win != nullptr;
    {
        const rh_status w = pre_launch(p, hs);
        if (w != RH_OK) return w;
    }
    p->epoch += 1;
    if (p->epoch == 0) {  // tag wrap: start over from clean tables
        if (p->d_gran) RH_HIP_TRY(hipMemsetAsync(p->d_gran, 0, p->gran_words * 8, hs));
        RH_HIP_TRY(hipMemsetAsync(s.d_gran, 0, 3 * s.cap_tiles * 32, hs));
        if (s.d_hand) RH_HIP_TRY(hipMemsetAsync(s.d_hand, 0, 3 * 4 * sizeof(unsigned long long), hs));
        chained = false;  // (behind the fills, and no tag of the old count is waited for)
        p->epoch = 1;
    }
    Params k;
    std::memset(&k, 0, sizeof k);
    k.srcs = p->d_srcs;
    k.tabs = s.d_tabs;
    k.out = dst;
    k.gran = nullptr;
    k.ticket = p->d_ctl;
    k.status = p->d_ctl + 1;
    k.out_frames = out;
    k.chunk_in = k.chunk_out = 0;
    k.n_sources = n_sources;
    k.n_tiles = (uint32_t)tiles;
    k.F = p->F;
    k.T = p->T;
    k.qF = p->F / p->T;
    k.rF = p->F % p->T;
    k.Tf = (float)p->T;
    k.rcpT = rh::lerp_rcp(p->T);  // 0: this T did not pass the exhaustive check of the short division (rh_common.h)
    k.epoch = p->epoch;
    k.J = (uint32_t)J;
    k.direct = direct ? 1u : 0u;
    k.shard_base = p->shard_base;
    k.eq_frames = (uint32_t)avail;
    k.st_mode = sa.mode;
    k.st_active = (uint32_t)out;
    k.st_m0 = sa.m0;
    k.st_g0 = sa.g0;
    k.st_win = sa.win;
    k.st_wout = sa.wout;
    k.u = s.uni;
    SblkArgs q;
    {
        const uint64_t mb = sa.m0 >= 2 ? sa.m0 - 2 : 0;
        const unsigned __int128 pp = (unsigned __int128)mb * F;
        const uint64_t ib_g = (uint64_t)(pp / T);
        if (ib_g < sa.g0) return RH_OK;  // (the rows start behind the first tap of frame m0 - 2: not a block of this stream's own making)
        q.ib = (uint32_t)(ib_g - sa.g0);
        q.rb = (uint32_t)(pp % T);
        q.mb_off = (uint32_t)(sa.m0 - mb);
    }
    q.uni = s.d_uni;
    q.powD = s.d_pow;
    q.gran = s.d_gran + (size_t)set * s.cap_tiles * 4;
    q.src_off = sa.src_off;
    q.Dmax = s.Dmax;
    q.P = (uint32_t)P;
    const bool hands = ovl && sa.mode == 1 && sa.wout != nullptr && direct;
    q.hand_in = chained ? s.d_hand + (size_t)s.last_set * 4 : nullptr;
    q.hand_tag = s.last_tag;
    q.hand_out = hands ? s.d_hand + (size_t)set * 4 : nullptr;
    void *args[] = {&k, &q};
    const uint32_t grid = direct ? (uint32_t)tiles : (((uint32_t)tiles + 7u) & ~7u);
    const hipError_t e = chained ? hipExtLaunchKernel(pick->fn, dim3(grid), dim3(64 * (kSblkWaves + 1)), args, 0, hs, nullptr, nullptr, hipExtAnyOrderLaunch)
                                 : hipLaunchKernel(pick->fn, dim3(grid), dim3(64 * (kSblkWaves + 1)), args, 0, hs);
    if (e != hipSuccess) {
        rh::set_hip_error(e, "k_rlm_sblk");
        return RH_ERR_HIP;
    }
    s.last_handed = hands;
    s.last_tag = p->epoch;
    s.last_set = set;
    s.last_stream = hs;
    s.seen_version = p->srcs_version;
    if (chained) s.n_chained += 1;
    s.prev_sblk = true;
    s.blk += 1;
    if (!direct) p->shard_base += grid / 8u;
    p->n_tiles = (uint32_t)tiles;
    *taken = true;
    return mark_launch(p, hs);
}

}  // namespace rhp

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
// Two kernels share the decomposition:
//   k_rlm_fast  equal-length batches (the benchmark): everything that couples lanes and tiles is done
//               once on the SUM over the sources; tiles never wait for each other inside the source loop
//   k_rlm_wave  ragged batches: per-source scan and per-(source, tile) carries, exchanged in groups of 8
//
// Work decomposition (wave64, no MFMA -- there is no contraction here):
//   * one WAVE (= one 64-lane workgroup) owns a tile of L = 64*R consecutive OUTPUT frames for
//     ALL sources; a lane owns a run of R consecutive frames.  The wave walks the sources in
//     insertion order and keeps the mix accumulators (R stereo frames per lane) in registers,
//     so the mixer sum costs no memory traffic.  Single-wave workgroups need no barrier at all.
//   * the input frames of (source, tile) are contiguous in HBM.  They are fetched with
//     global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip) into a ring of NS LDS stages and retired
//     with a counted s_waitcnt vmcnt; the lerp taps are ds_read2_b64, all issued before the
//     arithmetic.  HBM traffic = input once + mixed output once (+ one 128-byte line per chunk).
//     The DMA is issued from inline asm so that hipcc neither drains it at the next load
//     (cdna_hip_programming.md "Pipelining across barriers") nor counts it: every wait for it
//     is placed by hand below.
//   * the biquad is a linear recurrence along time.  Each lane runs it over its run from a zero
//     state; the run-end states are combined with a wave64 Kogge-Stone scan (DPP, no LDS) over
//     the 2x2 state-matrix powers B^(R*2^k) (host-computed in f64), and tiles are chained
//     through HBM "granules" ({epoch,value} 8-byte words, one agent-scope relaxed store each;
//     cdna_hip_programming.md G16 form R2).  A tile needs only the zero-state aggregates of its
//     J predecessors, where J is the number of tiles after which ||B^(L*J)|| < 2^-40 (the filter
//     is stable, so older history is below f32 resolution): no chained inclusive prefix, hence
//     no serial dependency along the tiles.  The homogeneous response g1[r]*S1 + g2[r]*S2 to the
//     true start state S of a lane is added once, after the last source, from summed states.
//   * tiles are numbered by an atomic ticket, so a tile only ever waits for tiles that already
//     hold a wave slot: progress does not depend on dispatch order or residency.
//
// This unit: the kernels, their instances and the launch (rlm_launch).  The host side around it -- handles, plans and
// tables, filter classes, the one-shot entry points (rh_pipeline_plan.hip) and block streaming (rh_pipeline_stream.hip) --
// shares rh_pipeline_internal.h with it.
#include "rh_pipeline_dev.h"

namespace {


// (the pairs of a ragged batch in which a source is about to end: defined below, next to k_rlm_resid)
__device__ __forceinline__ uint32_t rag_find_pairs(const Params &p, const uint32_t m_tile0, const uint32_t m_stable, const int lane, lds_u8 *lds, const uint32_t list_off);
template <int R, int KV, int C>
__device__ __forceinline__ void rag_run_pairs(const Params &p, unsigned long long *const rows, const uint32_t tile, const uint32_t n_pairs, const int lane, lds_u8 *lds, const uint32_t lds0,
                                              const uint32_t list_off, const uint32_t i_base, const uint32_t nvec, const uint32_t (&goff)[KV], const int (&offA)[R + 2], const float (&wgt)[R + 2],
                                              const float (&lM)[4], const float (&b15)[4], const float (&b31)[4], const float (&kM)[4], typename Chan<C>::V (&acc)[R]);

// SUMF (with RAG): sum first -- the stable sources of a tile share the tile's span, taps and weights, so their spans are SUMMED as
// they land (one FMA per float) and the lerp, the zero-state filter and everything behind it run once, on the sum (§4.6).
template <int R, int KV, int NS, bool FILT, bool RAG = false, int C = 2, bool SUMF = false>
__global__ __launch_bounds__(64, (R <= 4 ? (KV <= 5 ? 5 : 4) : R <= 6 ? 3 : R <= 12 ? 2 : 1)) void k_rlm_fast(const Params p) {  // (no spills: tests/test_code_objects.py)
    typedef Chan<C> CH;
    typedef typename CH::V V;
    constexpr uint32_t FB = CH::kFB, VF = CH::kVF;
    static_assert(R <= kMaxR, "frames per lane");
    static_assert(NS >= 2 && NS <= 4, "ring depth");
    static_assert(KV * (NS - 1) < 64, "vmcnt range");
    constexpr uint32_t kStage = KV * 1024;  // bytes of one LDS input stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_u8 *const lds = (lds_u8 *)smem;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;

    const int lane = threadIdx.x;
    // batch mode (no mixer: every source keeps its own output row): tickets run tile-major over the
    // sources, so the predecessor tiles of a stream always hold earlier tickets.  One device-scope counter hands out ~85
    // tickets per microsecond (MI355X_MICROARCH.md "dequeue") -- 64 streams x 820 tiles would spend 0.6 ms on that alone --
    // so the streams are dealt over `shards` counters on separate cache lines, one per XCD (workgroup b runs on XCD b % 8):
    // a stream's tiles all come from one counter, which keeps the order that matters.
    uint32_t ticket, stream, tile;
    if (p.direct) {
        ticket = blockIdx.x;
        stream = 0;
        tile = ticket;
    } else if (p.shards > 1 && p.batch_streams == 0) {
        // one stream, tiles by ticket, eight counters (see k_rlm_chunk): workgroup b takes ticket k of counter b % 8 and works on tile
        // b % 8 + 8 k; the grid is whole rounds of eight, workgroups past the last tile leave
        const uint32_t x = blockIdx.x & 7u;
        const uint32_t k_ = __builtin_amdgcn_readfirstlane(lane == 0 ? atomicAdd(p.ticket + 32u * (1u + x), 1u) - p.shard_base : 0u);
        tile = x + 8u * k_;
        ticket = tile;
        stream = 0;
        if (tile >= p.n_tiles) return;
    } else if (p.shards > 1) {
        const uint32_t x = blockIdx.x % p.shards, per = p.batch_streams / p.shards;
        ticket = __builtin_amdgcn_readfirstlane(lane == 0 ? atomicAdd(p.ticket + 32u * (1u + x), 1u) - p.shard_base : 0u);
        stream = x + p.shards * (ticket % per);
        tile = ticket / per;
    } else {
        ticket = __builtin_amdgcn_readfirstlane(lane == 0 ? atomicAdd(p.ticket, 1u) - p.ticket_base : 0u);
        stream = p.batch_streams ? ticket % p.batch_streams : 0u;
        tile = p.batch_streams ? ticket / p.batch_streams : ticket;
    }
    ticket = __builtin_amdgcn_readfirstlane(ticket);  // wave-uniform by construction: say so (SGPRs, not VGPRs)
    stream = __builtin_amdgcn_readfirstlane(stream);
    tile = __builtin_amdgcn_readfirstlane(tile);
    constexpr uint32_t L = 64u * R;
    const uint32_t m_tile0 = tile * L;
    const uint32_t m0 = m_tile0 + (uint32_t)lane * R;
    const uint64_t mg0 = p.st_mode ? p.st_m0 : 0;  // block streaming: this launch starts at global output frame st_m0
    const bool first = (mg0 + m0 == 0);  // stream start: x'[-1] = x'[-2] = 0
    const uint32_t Mout = (uint32_t)p.out_frames;
    const uint32_t Ns = p.eq_frames;  // every source has Ns frames (and Mout output frames)
    const uint64_t g0 = p.st_mode ? p.st_g0 : 0;  // ... and the buffers start at global input frame st_g0
    // mode 1: lanes at or past st_active belong to the next block (their input has not arrived yet)
    const bool lane_on = p.st_mode != 1 || m0 < p.st_active;
    const bool state_tile = p.st_mode == 1 && tile == p.st_active / L;  // holds the lane whose start state is the block's end state

    uint32_t i_base, nvec;
    {
        uint64_t ib, ie;
        uint32_t nn;
        cursor_resolve(cursor_at(mg0 + m_tile0 >= 2 ? mg0 + m_tile0 - 2 : 0, p), p, ib, nn);
        ib = ib > g0 ? ib - g0 : 0;  // index inside the buffers
        ib &= ~(uint64_t)(128u / FB - 1u);  // the staged span starts on a 128-byte line: every DMA instruction covers whole lines
        cursor_resolve(cursor_at(mg0 + m_tile0 + L - 1, p), p, ie, nn);
        ie = ie > g0 ? ie - g0 : 0;
        ie += 1;
        uint32_t nv = (uint32_t)((ie - ib + VF) / VF);
        if (nv > (uint32_t)(KV * 64)) nv = KV * 64;  // host sizes KV so this never bites
        i_base = __builtin_amdgcn_readfirstlane((uint32_t)ib);
        nvec = __builtin_amdgcn_readfirstlane(nv);
    }
    // The tile reaches past the end of the sources: lanes beyond it re-fetch the last 16-byte vector
    // (finite data no valid output reads), and the last frame's second tap is replaced by the first
    // (sample_rate.rs:193-200: the last frame is emitted verbatim).  Same for every source here.
    const bool edge = i_base + VF * nvec > Ns;
    const uint32_t lastoff = ((Ns - 1) & ~(VF - 1u)) * FB;
    uint32_t goff[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        uint32_t j = lane + k * 64;
        j = j < nvec ? j : nvec - 1;
        goff[k] = (i_base + VF * j) * FB;
        if (edge && goff[k] > lastoff) goff[k] = lastoff;
    }
    int offA[R + 2];
    float wgt[R + 2];
    {
        const uint32_t dthr = Ns - 1 - i_base;
        const int thr = (int)((dthr < (1u << 27) ? dthr : (1u << 27)) * FB);
        Cursor c = cursor_at(first ? 0 : mg0 + m0 - 2, p);
#pragma unroll
        for (int rr = 0; rr < R + 2; ++rr) {
            const bool dummy = first && rr < 2;
            uint64_t i;
            uint32_t num;
            cursor_resolve(c, p, i, num);
            offA[rr] = dummy ? 0 : (int)(((uint32_t)(i - g0) - i_base) * FB);
            if (edge && offA[rr] >= thr) num = 0;  // verbatim last frame (and frames past it, never stored)
            if (edge && offA[rr] > thr) offA[rr] = thr;
            wgt[rr] = dummy ? 0.0f : (FILT ? (float)num / p.Tf : (float)num);
            if (!dummy) cursor_next(c, p);
        }
    }
    const float b0 = p.u.b0, c1 = p.u.c1, c2 = p.u.c2, na1 = -p.u.a1, na2 = -p.u.a2;

    V acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = CH::zero();
    V E1 = CH::zero(), E2 = CH::zero();  // sum over the sources of the run-end states (w[-1], w[-2])

    const uint32_t S = p.batch_streams ? 1u : p.n_sources;
    typedef __attribute__((address_space(4))) const uint64_t cu64;
    cu64 *const desc = (cu64 *)(uintptr_t)(p.srcs + stream);  // 32-byte descriptors: the pointer is the first qword ...
    typedef __attribute__((address_space(4))) const float cf32;
    cf32 *const dgain = (cf32 *)(uintptr_t)(p.srcs + stream);  // ... the gain is float 4
    auto stage_source = [&](const void *data, uint32_t stage_off) {
#pragma unroll
        for (int k = 0; k < KV; ++k) glds16(data, goff[k], lds0 + stage_off + k * 1024);
    };
    const bool live = Mout > m_tile0 && Ns > 0;  // false only for the state tile of a block that ends on a tile boundary
    // The ring: NS stages, and -- because a source's taps are pulled into registers in one go -- NS
    // sources in flight: the stage of source s is re-targeted by the DMA of source s+NS as soon as the
    // taps of s have returned, BEFORE the arithmetic of s.
    uint32_t st_cur = 0;
    uint64_t ptr_pref = 0;
    float g_next = 1.0f;
    if (!RAG) {
#pragma unroll
        for (int d = 0; d < NS; ++d)
            if ((uint32_t)d < S && live) stage_source((const void *)(uintptr_t)desc[4 * d], d * kStage);
        ptr_pref = NS < S ? desc[4 * NS] : 0;  // fetched one iteration ahead of its use
        g_next = S ? dgain[4] : 1.0f;          // gain of source 0
    }
    RH_PH_DECL

    // Software pipeline over the sources: while source s is being computed, the taps of source s+1 are already on their
    // way from the LDS (two register sets, the loop body is written out twice so that they swap without moves).
    // Per iteration: taps(s) complete -> stage(s) is free -> DMA of source s+NS into it -> stage(s+1) landed ->
    // tap reads of s+1 issued -> arithmetic of s.
    auto read_taps = [&](uint32_t stage_off, V (&qa)[R + 2], V (&qb)[R + 2]) {
        const lds_u8 *buf = lds + stage_off;
#pragma unroll
        for (int rr = 0; rr < R + 2; ++rr) {
            qa[rr] = CH::ld_lds(buf + offA[rr]);
            qb[rr] = CH::ld_lds(buf + offA[rr] + FB);
        }
    };
    auto compute = [&](const V (&ta)[R + 2], const V (&tb2)[R + 2], const float g) {
        auto tap = [&](int rr) -> V {
            const V a = ta[rr], b = tb2[rr];
            V x;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float ac = CH::get(a, c), bc = CH::get(b, c);
                if (FILT) {
                    CH::set(x, c, fma_(bc - ac, wgt[rr], ac));
                } else {  // amplify.rs:64 then math.rs:25: first + (second - first) * num / den, exactly
                    const float ag = ac * g, bg = bc * g;
                    CH::set(x, c, ag + div_T((bg - ag) * wgt[rr], p.Tf, p.rcpT));
                }
            }
            return x;
        };
        if (FILT) {
            V x2 = first ? CH::zero() : tap(0);
            V x1 = first ? CH::zero() : tap(1);
            V w1 = CH::zero(), w2 = CH::zero();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const V x = tap(r + 2);
                V w;  // zero-state step of the recursive part; the w1 term goes last (shortest chain)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float wc = fma_(na1, CH::get(w1, c), fma_(na2, CH::get(w2, c), fma_(c2, CH::get(x2, c), c1 * CH::get(x1, c))));
                    CH::set(w, c, wc);
                    CH::set(acc[r], c, fma_(g, fma_(b0, CH::get(x, c), wc), CH::get(acc[r], c)));  // the source's gain rides on the mix (linear chain)
                }
                w2 = w1;
                w1 = w;
                x2 = x1;
                x1 = x;
            }
            E1 = vfma_s(g, w1, E1);
            E2 = vfma_s(g, w2, E2);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const V x = tap(r + 2);
                acc[r] += x;
            }
        }
    };
    auto iteration = [&](const uint32_t s, const V (&ca)[R + 2], const V (&cb)[R + 2], V (&na)[R + 2], V (&nb)[R + 2]) {
        // the taps of source s are in registers: its stage is free for source s+NS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RH_PH(2)
        if (s + NS < S) stage_source((const void *)(uintptr_t)ptr_pref, st_cur);
        ptr_pref = s + NS + 1 < S ? desc[4 * (uint64_t)(s + NS + 1)] : 0;
        const float g = g_next;  // Amplify factor of source s
        g_next = s + 1 < S ? dgain[8 * (uint64_t)(s + 1) + 4] : 1.0f;
        st_cur += kStage;
        if (st_cur >= NS * kStage) st_cur = 0;
        if (s + 1 < S) {  // the stage of source s+1 has landed when at most the groups issued after it are outstanding
            const uint32_t left = S - 2 - s;
            wait_groups<KV, NS>((int)(left < (uint32_t)(NS - 1) ? left : (uint32_t)(NS - 1)));
            read_taps(st_cur, na, nb);
        }
        RH_PH(1)
        compute(ca, cb, g);
        RH_PH(4)
    };
    V tA[R + 2], tB[R + 2], uA[R + 2], uB[R + 2];
    if (!RAG) {
        if (live && S) {
            const uint32_t left = S - 1;
            wait_groups<KV, NS>((int)(left < (uint32_t)(NS - 1) ? left : (uint32_t)(NS - 1)));
            read_taps(0, tA, tB);
        }
        for (uint32_t s = 0; s < S && live; s += 2) {
            iteration(s, tA, tB, uA, uB);
            if (s + 1 < S) iteration(s + 1, uA, uB, tA, tB);
        }
    } else if (live) {
        // The same pipeline over the SUBSEQUENCE of stable sources.  q[0] is the source being computed, q[1..NS-1] are
        // in flight, q[NS] is the next one to fetch (S: none left).  `issued` / `waited` count DMA groups.
        typedef __attribute__((address_space(4))) const uint32_t cu32;
        cu32 *const dwords = (cu32 *)(uintptr_t)p.srcs;
        // stable: whole for this tile and the J after it -- or until the mix itself ends (such sources share one length,
        // eq_frames: the host checks it; the end-of-source handling above is theirs)
        const uint32_t m_far = m_tile0 + (p.J + 1u) * L;
        const uint32_t m_stable = m_far < Mout ? m_far : Mout;
        auto next_stable = [&](uint32_t from) {
            uint32_t k = from;
            while (k < S && dwords[8 * (uint64_t)k + 3] < m_stable) ++k;
            return k;
        };
        // Inside a source (no tile edge, the span fills all but the last instruction) the DMA instructions of a stage lie 1 KiB apart
        // in memory as in the LDS: they go out as runs that share one M0 and one offset register (glds16_run) -- a tile in the middle
        // of the batch, i.e. nearly every one.
        constexpr int KF = KV - 1;
        const bool lin = SUMF && !edge && nvec >= (uint32_t)(KF * 64);
        auto stage_rag = [&](const void *data, uint32_t stage_off) {
            if (lin) {
                glds16_run<KF>(data, goff[0], lds0 + stage_off);
                glds16(data, goff[KV - 1], lds0 + stage_off + (KV - 1) * 1024);
            } else {
                stage_source(data, stage_off);
            }
        };
        uint32_t q[NS + 1];
        uint32_t issued = 0, waited = 0;
        {
            uint32_t nxt = next_stable(0);
#pragma unroll
            for (int d = 0; d < NS; ++d) {
                q[d] = nxt;
                if (nxt < S) {
                    stage_rag((const void *)(uintptr_t)desc[4 * (uint64_t)nxt], d * kStage);
                    ++issued;
                    nxt = next_stable(nxt + 1);
                }
            }
            q[NS] = nxt;
        }
        auto rag_iteration = [&](const V (&ca)[R + 2], const V (&cb)[R + 2], V (&na)[R + 2], V (&nb)[R + 2]) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the taps of q[0] are in registers: its stage is free
            if (q[NS] < S) {
                stage_source((const void *)(uintptr_t)desc[4 * (uint64_t)q[NS]], st_cur);
                ++issued;
            }
            const float g = dgain[8 * (uint64_t)q[0] + 4];
            st_cur += kStage;
            if (st_cur >= NS * kStage) st_cur = 0;
            if (q[1] < S) {  // the stage of q[1] has landed when only the groups issued after it are outstanding
                ++waited;
                wait_groups<KV, NS>((int)(issued - waited));
                read_taps(st_cur, na, nb);
            }
            compute(ca, cb, g);
#pragma unroll
            for (int d = 0; d < NS; ++d) q[d] = q[d + 1];
            q[NS] = q[NS] < S ? next_stable(q[NS] + 1) : S;
        };
        if constexpr (SUMF) {
            v4f accv[KV];
#pragma unroll
            for (int k = 0; k < KV; ++k) accv[k] = v4f{0.f, 0.f, 0.f, 0.f};
            // Descriptor words are scalar loads, and a scalar load that misses its cache is a round trip to L2 the wave sits out (the
            // phase profile of the first version: a third of a heavy tile's time between "the stage has landed" and "the next DMA is out").
            // So every word is asked for an iteration ahead of its use: the row of q[NS], the gain of q[1], the end of the source behind q[NS].
            uint64_t ptr_next = q[NS] < S ? desc[4 * (uint64_t)q[NS]] : 0;
            float g_cur = q[0] < S ? dgain[8 * (uint64_t)q[0] + 4] : 0.f;
            while (q[0] < S) {
                const uint32_t cand = q[NS] + 1;  // almost always stable too
                const uint32_t cand_end = cand < S ? dwords[8 * (uint64_t)cand + 3] : 0u;
                const float g_nxt = q[1] < S ? dgain[8 * (uint64_t)q[1] + 4] : 0.f;
                ++waited;
                wait_groups<KV, NS>((int)(issued - waited));  // the stage of q[0] has landed
                RH_PH(2)
                const lds_u8 *buf = lds + st_cur;
                v4f v[KV];
#pragma unroll
                for (int k = 0; k < KV; ++k) v[k] = *(const lds_f4 *)(buf + k * 1024 + lane * 16);
                const float g = g_cur;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the span is in registers: its stage is free
                if (q[NS] < S) {
                    stage_rag((const void *)(uintptr_t)ptr_next, st_cur);
                    ++issued;
                }
                RH_PH(1)
#pragma unroll
                for (int k = 0; k < KV; ++k) {
                    accv[k].x = fma_(g, v[k].x, accv[k].x);
                    accv[k].y = fma_(g, v[k].y, accv[k].y);
                    accv[k].z = fma_(g, v[k].z, accv[k].z);
                    accv[k].w = fma_(g, v[k].w, accv[k].w);
                }
                st_cur += kStage;
                if (st_cur >= NS * kStage) st_cur = 0;
#pragma unroll
                for (int d = 0; d < NS; ++d) q[d] = q[d + 1];
                q[NS] = cand >= S ? S : (cand_end >= m_stable ? cand : next_stable(cand + 1));
                ptr_next = q[NS] < S ? desc[4 * (uint64_t)q[NS]] : 0;
                g_cur = g_nxt;
                RH_PH(4)
            }
            wait_vm<0>();
            // the summed span takes the place of a source's in stage 0: one lerp, one zero-state run
#pragma unroll
            for (int k = 0; k < KV; ++k) *(lds_f4 *)(lds + k * 1024 + lane * 16) = accv[k];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            read_taps(0, tA, tB);
            compute(tA, tB, 1.0f);
        } else {
            if (q[0] < S) {
                ++waited;
                wait_groups<KV, NS>((int)(issued - waited));
                read_taps(0, tA, tB);
            }
            while (q[0] < S) {
                rag_iteration(tA, tB, uA, uB);
                if (q[0] < S) rag_iteration(uA, uB, tA, tB);
            }
        }
    }
    wait_vm<0>();  // nothing of this wave may still be in flight towards its LDS

    if (FILT && (live || state_tile)) {
        if (!lane_on) E1 = E2 = CH::zero();
        const Tables *__restrict__ tb = p.tabs;
        float lM[4], b15[4], b31[4], kM[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            lM[q] = tb->laneM[lane][q];
            b15[q] = tb->bc15M[lane][q];
            b31[q] = tb->bc31M[lane][q];
            kM[q] = tb->lookM[lane & (kMaxLook - 1)][q];
        }
        // ---- summed run-end states in the scan basis, then the wave64 inclusive scan (P[2c], P[2c+1]: channel c) ----
        float P[2 * C];
#pragma unroll
        for (int q = 0; q < 2 * C; ++q) P[q] = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) mat_acc(p.u.Tm, CH::get(E1, c), CH::get(E2, c), P[2 * c], P[2 * c + 1]);
#define RH_SCAN_STEP(K, N)                                                                          \
    {                                                                                               \
        float sq[2 * C];                                                                            \
        _Pragma("unroll") for (int q = 0; q < 2 * C; ++q) sq[q] = dpp0<kDppRowShr + N, 0xf>(P[q]);   \
        _Pragma("unroll") for (int c = 0; c < C; ++c) mat_acc(p.u.scanM[K], sq[2 * c], sq[2 * c + 1], P[2 * c], P[2 * c + 1]); \
    }
        RH_SCAN_STEP(0, 1)
        RH_SCAN_STEP(1, 2)
        RH_SCAN_STEP(2, 4)
        RH_SCAN_STEP(3, 8)
#undef RH_SCAN_STEP
        {  // rows 1 and 3 take the inclusive prefix of the row before them
            float sq[2 * C];
#pragma unroll
            for (int q = 0; q < 2 * C; ++q) sq[q] = dpp0<kDppBcast15, 0xa>(P[q]);
#pragma unroll
            for (int c = 0; c < C; ++c) mat_acc(b15, sq[2 * c], sq[2 * c + 1], P[2 * c], P[2 * c + 1]);
        }
        {  // rows 2 and 3 take the inclusive prefix of lanes 0..31
            float sq[2 * C];
#pragma unroll
            for (int q = 0; q < 2 * C; ++q) sq[q] = dpp0<kDppBcast31, 0xc>(P[q]);
#pragma unroll
            for (int c = 0; c < C; ++c) mat_acc(b31, sq[2 * c], sq[2 * c + 1], P[2 * c], P[2 * c + 1]);
        }
        {  // publish the tile aggregate (lane 63's inclusive prefix): 2C granules of the tile's 4-word slot, one store
            float e[2 * C];
#pragma unroll
            for (int q = 0; q < 2 * C; ++q) e[q] = readlane_f(P[q], 63);
            if (lane < 2 * C) {
                float ev = e[0];
#pragma unroll
                for (int q = 1; q < 2 * C; ++q) ev = lane == q ? e[q] : ev;
                const unsigned long long word = ((unsigned long long)p.epoch << 32) | __float_as_uint(ev);
                __hip_atomic_store(p.gran + ((uint64_t)stream * p.n_tiles + tile) * 4 + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        float Q[2 * C];
#pragma unroll
        for (int q = 0; q < 2 * C; ++q) Q[q] = dpp0<kDppWaveShr1, 0xf>(P[q]);  // exclusive: lane 0 gets 0
        // ---- the tile carry: lane j < J polls predecessor tile-1-j (they finish about now) ----
        const uint32_t Jc = p.J < tile ? p.J : tile;
        float c[2 * C];
#pragma unroll
        for (int q = 0; q < 2 * C; ++q) c[q] = 0.f;
        if (Jc > 0) {
            const bool want = (uint32_t)lane < Jc;
            const unsigned long long *gp = p.gran + ((uint64_t)stream * p.n_tiles + (tile - 1 - (want ? lane : 0))) * 4;
            unsigned long long gv[2 * C];
#pragma unroll
            for (int q = 0; q < 2 * C; ++q) gv[q] = 0;
            bool ok = false, dead = false;
            uint32_t spins = 0;
            while (true) {
                if (want && !ok) {
                    bool all = true;
#pragma unroll
                    for (int q = 0; q < 2 * C; ++q) {
                        gv[q] = __hip_atomic_load(gp + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        all = all && ((uint32_t)(gv[q] >> 32) == p.epoch);
                    }
                    ok = all;
                }
                if (__all(ok || !want)) break;
                if (++spins > kSpinLimit) {
                    if (lane == 0) atomicOr(p.status, 1u);
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            if (lane == 0 && spins) atomicAdd(p.status + 1, spins);  // statistics: polls that found nothing yet
            if (want && ok && !dead) {
#pragma unroll
                for (int ch = 0; ch < C; ++ch) mat_acc(kM, __uint_as_float((uint32_t)gv[2 * ch]), __uint_as_float((uint32_t)gv[2 * ch + 1]), c[2 * ch], c[2 * ch + 1]);
            }
            // A carry that never arrived: the status word fails the call (rh_rlm_last_status), and the tile is poisoned so
            // that a block served without that check can never pass for audio.
            if (dead) {
#pragma unroll
                for (int q = 0; q < 2 * C; ++q) c[q] = __builtin_nanf("");
            }
#pragma unroll
            for (int q = 0; q < 2 * C; ++q) {  // sum over lanes 0..31 -> uniform
                c[q] += dpp0<kDppRowShr + 1, 0xf>(c[q]);
                c[q] += dpp0<kDppRowShr + 2, 0xf>(c[q]);
                c[q] += dpp0<kDppRowShr + 4, 0xf>(c[q]);
                c[q] += dpp0<kDppRowShr + 8, 0xf>(c[q]);
                c[q] = readlane_f(c[q], 15) + readlane_f(c[q], 31);
            }
        }
        if (p.st_mode && tile < p.J) {  // the stream's state at the block start still reaches this tile: + B^(L*tile) * W_in
            const float *M = tb->lookM[tile];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) mat_acc(M, p.st_win[2 * ch], p.st_win[2 * ch + 1], c[2 * ch], c[2 * ch + 1]);
        }
        // the merged homogeneous response: start state = Q + B^(R*lane) * carry
#pragma unroll
        for (int ch = 0; ch < C; ++ch) mat_acc(lM, c[2 * ch], c[2 * ch + 1], Q[2 * ch], Q[2 * ch + 1]);
        if (state_tile && (uint32_t)lane == (p.st_active % L) / R) {  // the first lane of the next block: its start state is the block's end state
#pragma unroll
            for (int q = 0; q < 2 * C; ++q) p.st_wout[q] = Q[q];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) CH::set(acc[r], ch, fma_(p.u.g[r][0], Q[2 * ch], fma_(p.u.g[r][1], Q[2 * ch + 1], CH::get(acc[r], ch))));
        }
        if constexpr (RAG && SUMF) {
            // The tile's own pairs in which a source is about to end, onto the mix in registers (the ring's first two stages and
            // 128 bytes behind the ring are theirs now).  Few tiles have any, and those are the lighter ones.
            const uint32_t m_far = m_tile0 + (p.J + 1u) * L;
            const uint32_t m_stable = m_far < Mout ? m_far : Mout;
            if (p.rag_merge && live && m_stable > p.rag_pairs_from && m_tile0 < p.rag_pairs_to) {
                RH_PH(5)
                const uint32_t n_pairs = rag_find_pairs(p, m_tile0, m_stable, lane, lds, NS * kStage);
                if (n_pairs)
                    rag_run_pairs<R, KV, C>(p, p.gran - (uint64_t)p.n_sources * p.n_tiles * 4, tile, n_pairs, lane, lds, lds0, NS * kStage, i_base, nvec, goff, offA, wgt, lM, b15, b31, kM, acc);
                RH_PH(3)
            }
        }
    }
#ifdef RH_PHASE_PROFILE
    RH_PH(5)
    if (p.prof && lane == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        ph_t[6] = ((unsigned long long)xcc << 32) | hwid;
        ph_t[7] = ph_start;
        for (int i = 0; i < 8; ++i) p.prof[(uint64_t)ticket * 8 + i] = ph_t[i];
    }
#endif

    // ---- mixed output: R frames per lane -----------------------------------------------------
    float *o = p.out + (uint64_t)stream * p.out_stride + (uint64_t)m0 * C;
    if (C == 1) {  // mono: a lane's run is R floats; 16-byte stores where the run is whole vectors inside the mix (R % 4 == 0: m0 * 4 is 16-byte aligned)
        float *of = reinterpret_cast<float *>(o);
        if (R % 4 == 0) {
#pragma unroll
            for (int r = 0; r + 3 < R; r += 4) {
                const uint32_t m = m0 + r;
                if (m + 3 < Mout) *reinterpret_cast<float4 *>(of + r) = make_float4(CH::get(acc[r], 0), CH::get(acc[r + 1], 0), CH::get(acc[r + 2], 0), CH::get(acc[r + 3], 0));
                else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (m + k < Mout) of[r + k] = CH::get(acc[r + k], 0);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (m0 + r < Mout) of[r] = CH::get(acc[r], 0);
        }
        return;
    }
    if (R % 2 == 0 && !RAG && FILT && p.batch_streams) {
        // Batch mode writes as many bytes as it reads.  A lane's run is R*8 contiguous bytes, so a wave's float4 store hits
        // 64 different 128-byte lines with 16 bytes each: ten partial-line writes where one full line would do.  The runs
        // therefore go through the (now idle) LDS stage -- rows padded by 8 bytes -- and leave as whole lines: lane l stores
        // frames 2q, 2q+1 of the tile, q = k*64 + l.
        constexpr uint32_t kRow = R * 8 + 8;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lds_u8 *row = lds + (uint32_t)lane * kRow;
#pragma unroll
        for (int r = 0; r < R; ++r) *(lds_f2 *)(row + r * 8) = v2f{CH::get(acc[r], 0), CH::get(acc[r], C - 1)};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float *ot = p.out + (uint64_t)stream * p.out_stride + (uint64_t)m_tile0 * 2;
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
            const uint32_t f = 2u * (k * 64u + (uint32_t)lane);
            const lds_u8 *src2 = lds + (f / R) * kRow + (f % R) * 8u;
            const v2f a = *(const lds_f2 *)src2, b = *(const lds_f2 *)(src2 + 8);
            const uint32_t m = m_tile0 + f;
            if (m + 1 < Mout) *reinterpret_cast<float4 *>(ot + f * 2) = make_float4(a.x, a.y, b.x, b.y);
            else if (m < Mout) *reinterpret_cast<float2 *>(ot + f * 2) = make_float2(a.x, a.y);
        }
    } else if (R % 2 == 0) {
#pragma unroll
        for (int r = 0; r + 1 < R; r += 2) {
            const uint32_t m = m0 + r;
            if (m + 1 < Mout) {
                *reinterpret_cast<float4 *>(o + r * 2) = make_float4(CH::get(acc[r], 0), CH::get(acc[r], C - 1), CH::get(acc[r + 1], 0), CH::get(acc[r + 1], C - 1));
            } else if (m < Mout) {
                *reinterpret_cast<float2 *>(o + r * 2) = make_float2(CH::get(acc[r], 0), CH::get(acc[r], C - 1));
            }
        }
    } else {  // odd R: a lane's run starts on an 8-byte boundary only
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (m0 + r < Mout) *reinterpret_cast<float2 *>(o + r * 2) = make_float2(CH::get(acc[r], 0), CH::get(acc[r], C - 1));
    }
}

// (C = 1: the frames are mono; the per-source aggregates keep their 4-word slots, the second channel's words travel as zeros)
template <int R, int KV, int NS, bool FILT, int C = 2>
__global__ __launch_bounds__(64, (R <= 8 && KV <= 5 ? 3 : 2)) void k_rlm_wave(const Params p) {  // (no spills: tests/test_code_objects.py)
    typedef Chan<C> CH;
    typedef typename CH::V V;
    constexpr uint32_t FB = CH::kFB, VF = CH::kVF;
    static_assert(R <= kMaxR, "frames per lane");
    static_assert(NS >= 2 && NS <= 4, "ring depth");
    static_assert(KV * (NS - 1) < 64, "vmcnt range");
    constexpr uint32_t kStage = KV * 1024;        // bytes of one LDS input stage
    constexpr uint32_t kPubBase = NS * kStage;    // 8 x 16 B: this tile's aggregates of the current source group
    constexpr uint32_t kCntBase = kPubBase + 128;              // 32 B: per recent source, how many predecessor tiles hold its own aggregate
    constexpr uint32_t kLookBase = kCntBase + 32;              // kMaxLook x 16 B: B^(L*j)
    constexpr uint32_t kGranBase = kLookBase + kMaxLook * 16;  // NI x 1 KiB: predecessor aggregates of one source group
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // explicit LDS address space: a generic pointer here turns every tap into a flat_load
    lds_u8 *const lds = (lds_u8 *)smem;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;  // LDS byte address of the dynamic region

    const int lane = threadIdx.x;
    const uint32_t tile = __builtin_amdgcn_readfirstlane(lane == 0 ? atomicAdd(p.ticket, 1u) - p.ticket_base : 0u);
    constexpr uint32_t L = 64u * R;
    const uint32_t m_tile0 = tile * L;  // host: out_frames < 2^31
    const uint32_t m0 = m_tile0 + (uint32_t)lane * R;
    const uint64_t mg0 = p.st_mode ? p.st_m0 : 0;  // block streaming: this launch starts at global output frame st_m0 ...
    const uint64_t g0 = p.st_mode ? p.st_g0 : 0;   // ... and the buffers start at global input frame st_g0
    const bool first = (mg0 + m0 == 0);  // stream start: x'[-1] = x'[-2] = 0
    const uint32_t Mout = (uint32_t)p.out_frames;
    const uint32_t col = tile + p.col0;  // this tile's column in the per-source aggregate rows
    const uint32_t ncol = p.gran_cols;

    // ---- input span of this tile (identical for every source) -------------------------------
    uint32_t i_base, nvec;
    {
        uint64_t ib, ie;
        uint32_t nn;
        cursor_resolve(cursor_at(mg0 + m_tile0 >= 2 ? mg0 + m_tile0 - 2 : 0, p), p, ib, nn);
        ib = ib > g0 ? ib - g0 : 0;  // index inside the buffers
        ib &= ~(uint64_t)(128u / FB - 1u);  // the staged span starts on a 128-byte line: every DMA instruction covers whole lines
        cursor_resolve(cursor_at(mg0 + m_tile0 + L - 1, p), p, ie, nn);
        ie = ie > g0 ? ie - g0 : 0;
        ie += 1;
        uint32_t nv = (uint32_t)((ie - ib + VF) / VF);
        if (nv > (uint32_t)(KV * 64)) nv = KV * 64;  // host sizes KV so this never bites
        i_base = __builtin_amdgcn_readfirstlane((uint32_t)ib);  // host: in_frames < 2^29
        nvec = __builtin_amdgcn_readfirstlane(nv);
    }

    // per-lane byte offsets of the KV staging vectors inside a source (surplus lanes re-fetch the
    // last vector into unused slots); 8*frames < 2^32
    uint32_t goff[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        uint32_t j = lane + k * 64;
        j = j < nvec ? j : nvec - 1;
        goff[k] = (i_base + VF * j) * FB;
    }

    // ---- per-lane tap table: LDS byte offset of frame i(m) and the lerp weight ----------------
    // FILT: weight = num/T (the lerp becomes one FMA, <= 1.5 ulp from math.rs:25 -- far inside
    // the 1e-5 budget of the filtered pipeline); !FILT: weight = num, divided exactly per sample
    // so that resample+mix alone stays bit-identical to the reference.
    int offA[R + 2];
    float wgt[R + 2];
    {
        Cursor c = cursor_at(first ? 0 : mg0 + m0 - 2, p);
#pragma unroll
        for (int rr = 0; rr < R + 2; ++rr) {
            const bool dummy = first && rr < 2;
            uint64_t i;
            uint32_t num;
            cursor_resolve(c, p, i, num);
            offA[rr] = dummy ? 0 : (int)(((uint32_t)(i - g0) - i_base) * FB);
            wgt[rr] = dummy ? 0.0f : (FILT ? (float)num / p.Tf : (float)num);
            if (!dummy) cursor_next(c, p);
        }
    }

    // ---- carry look-back geometry ----------------------------------------------------------------
    // Aggregates travel in groups of 8 sources.  One LDS-DMA instruction fetches, for the 8 sources
    // of a group, the aggregates of 4 predecessor tiles: lane = src*8 + pred*2 + half (16 B each);
    // NI = ceil(J/4) instructions cover all J predecessors.  Lane l < 32 then owns the 32-byte set
    // (src = l>>2, pred = l&3) of every instruction.
    const uint32_t Jc = p.J < col ? p.J : col;  // uniform: predecessors that exist (streaming: + the block-start state)
    const uint32_t NI = (Jc + 3) >> 2;
    const uint32_t gsrc = lane >> 3, gpred = (lane >> 1) & 3;
    const uint32_t gr_off0 = gsrc * ncol * 32u + (Jc - 1 - gpred) * 32u + (lane & 1) * 16u;  // instruction 0; -128 per further one
    const uint32_t set_src = lane >> 2, set_pred = lane & 3;  // lanes < 32

    const Tables *__restrict__ tb = p.tabs;
    float lM[4], b15[4], b31[4];
    if (FILT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            lM[q] = tb->laneM[lane][q];
            b15[q] = tb->bc15M[lane][q];
            b31[q] = tb->bc31M[lane][q];
        }
        if (lane < kMaxLook) *(lds_f4 *)(lds + kLookBase + lane * 16) = v4f{tb->lookM[lane][0], tb->lookM[lane][1], tb->lookM[lane][2], tb->lookM[lane][3]};
    }
    const float b0 = p.u.b0, c1 = p.u.c1, c2 = p.u.c2, na1 = -p.u.a1, na2 = -p.u.a2;

    V acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = CH::zero();
    // The homogeneous corrections of all sources share g[] and laneM, so they are summed as STATES
    // and applied once after the last source: Qacc = sum of the lanes' zero-carry start states,
    // Cacc = this lane's share (its sets) of the summed tile carries.
    float Qacc[4] = {0.f, 0.f, 0.f, 0.f}, Cacc[4] = {0.f, 0.f, 0.f, 0.f};
    // A source that stays whole for this tile and the next J ("stable") never needs an aggregate of its own: every
    // tile that consumes it applies its correction merged.  Such sources only add their run-end state to a sum that
    // is scanned and published ONCE per tile (row n_sources of the table), as in k_rlm_fast; the per-source scan and
    // exchange below is left to the few sources that end within J tiles.  Streaming keeps every source on its own
    // (the block-end states are per source).
    const bool indiv_all = p.col0 != 0;
    const uint32_t m_stable = m_tile0 + (p.J + 1u) * L;
    V E1s = CH::zero(), E2s = CH::zero();
    uint64_t mrg = 0;      // bit k: source s-1-k was merged into Qacc
    uint32_t pubmask = 0;  // bit (s&7): source s of the current group has an aggregate in the pub area
    uint32_t grp_mask = 0; // merged flags (bit 7-src) of the group whose aggregates are in flight
    uint32_t grp_first = 0;
    uint32_t iss = 0;      // bit d: the stage of source s+d was filled by LDS-DMA (KV operations)
    bool dead = false;     // a bounded wait expired: never spin again in this wave

    const uint32_t S = p.n_sources;

    // ---- staging: source frames [i_base, i_base + 2*nvec) -> LDS stage, 16 bytes per lane -------
    // Always KV LDS-DMA instructions.  If the tile reaches past the end of the source, the lanes
    // beyond it re-fetch the source's last 16-byte vector instead (an aligned 16-byte load that
    // starts inside the source cannot leave its page): their slots then hold finite data that no
    // valid output reads, except as the second tap of the source's last frame, which
    // sample_rate.rs:193-200 emits verbatim -- the edge variant of the run selects the first tap there.
    auto stage_source = [&](const void *data, uint32_t frames, uint32_t stage_off) {
        if (i_base + VF * nvec <= frames) {
#pragma unroll
            for (int k = 0; k < KV; ++k) glds16(data, goff[k], lds0 + stage_off + k * 1024);
        } else {
            const uint32_t lastoff = ((frames - 1) & ~(VF - 1u)) * FB;
#pragma unroll
            for (int k = 0; k < KV; ++k) glds16(data, goff[k] < lastoff ? goff[k] : lastoff, lds0 + stage_off + k * 1024);
        }
    };
    // The descriptor table is constant for the launch: read it through the constant address space so
    // that it is an s_load (lgkmcnt), not a vector load whose wait would drain the DMA ring.
    typedef __attribute__((address_space(4))) const uint32_t cu32;
    cu32 *const desc = (cu32 *)(uintptr_t)p.srcs;
    struct Desc {
        const void *data;
        uint32_t frames, out_frames;
        float gain;
    };
    auto load_desc = [&](uint32_t s) -> Desc {
        Desc d{nullptr, 0, 0, 1.0f};
        if (s < S) {
            const uint64_t lo = desc[8 * (uint64_t)s], hi = desc[8 * (uint64_t)s + 1];
            d.data = (const void *)(uintptr_t)(lo | (hi << 32));
            d.frames = desc[8 * (uint64_t)s + 2];
            d.out_frames = desc[8 * (uint64_t)s + 3];
            d.gain = __uint_as_float(desc[8 * (uint64_t)s + 4]);
        }
        return d;
    };

    // Sum of the lanes' carry shares (lanes < 32 hold them), as a wave-uniform value.
    auto reduce_carry = [&](float (&c)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c[q] += dpp0<kDppRowShr + 1, 0xf>(c[q]);
            c[q] += dpp0<kDppRowShr + 2, 0xf>(c[q]);
            c[q] += dpp0<kDppRowShr + 4, 0xf>(c[q]);
            c[q] += dpp0<kDppRowShr + 8, 0xf>(c[q]);
            c[q] = readlane_f(c[q], 15) + readlane_f(c[q], 31);
        }
    };
    // Poll one 32-byte aggregate per lane until it carries this launch's epoch (ordinary agent-scope
    // loads: hipcc waits for them, which also drains the DMA ring -- this is the slow path).
    auto poll_sets = [&](const unsigned long long *gp, bool want, unsigned long long (&gv)[4], bool &ok) {
        uint32_t spins = 0;
        if (lane == 0) atomicAdd(p.status + 1, 1u);  // statistics: carries that were not ready in time
        while (!dead) {
            if (want && !ok) {
                bool all = true;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    gv[q] = __hip_atomic_load(gp + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    all = all && ((uint32_t)(gv[q] >> 32) == p.epoch);
                }
                ok = all;
            }
            if (__all(ok || !want)) break;
#ifdef RH_PHASE_PROFILE
            if (lane == 0) atomicAdd(p.status + 2, 1u);  // statistics: polls that found nothing yet
#endif
            if (++spins > kSpinLimit) {
                if (lane == 0) atomicOr(p.status, 1u);
                dead = true;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    };

    // ---- prologue: sources 0..NS-2 into stages 0..NS-2 -------------------------------------------
    uint32_t Ms_ring[NS], Ns_ring[NS];  // [d]: out_frames / frames of source s+d
    float g_ring[NS];                  // ... and its Amplify factor
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        Ms_ring[d] = Ns_ring[d] = 0;
        g_ring[d] = 1.0f;
    }
    uint32_t st_cur = 0;  // LDS byte offset of the stage of source s
#pragma unroll
    for (int d = 0; d < NS - 1; ++d) {
        const Desc sd = load_desc(d);
        Ms_ring[d] = sd.out_frames;
        Ns_ring[d] = sd.frames;
        g_ring[d] = sd.gain;
        if (sd.out_frames > m_tile0) {
            stage_source(sd.data, sd.frames, d * kStage);
            iss |= 1u << d;
        }
    }
    Desc sd_pref = load_desc(NS - 1);  // fetched one iteration ahead of its use

    const uint32_t g_last = S ? (S - 1) >> 3 : 0;
    const uint32_t n_iter = FILT ? (S ? 8 * (g_last + kGroupLag) + 3 : 0) : S;
    RH_PH_DECL

    for (uint32_t s = 0; s < n_iter; ++s) {
        // (1) every 8th source: fetch the predecessor tiles' aggregates of the source group that was
        //     completed 8*(kGroupLag-1)+1 sources ago (consumed 2 iterations from now)
        if (FILT && (s & 7) == 0 && s >= 8 * kGroupLag) {
            grp_first = s - 8 * kGroupLag;
            grp_mask = (uint32_t)(mrg >> (8 * kGroupLag - 8)) & 0xffu;  // bit 7-src
            if (Jc > 0 && grp_mask && !dead) {
                const unsigned long long *gb = p.gran + ((uint64_t)grp_first * ncol + (col - Jc)) * 4;
                const bool src_on = (grp_mask >> (7 - gsrc)) & 1u;
                for (uint32_t i = 0; i < NI; ++i)
                    if (src_on && gpred + 4 * i < Jc) glds16_sc1(gb, gr_off0 - 128u * i, lds0 + kGranBase + 1024u * i);
            }
        }
        RH_PH(0)
        // (2) stage source s+NS-1 into the slot source s-1 has just left
        {
            uint32_t st_new = st_cur + (NS - 1) * kStage;
            if (st_new >= NS * kStage) st_new -= NS * kStage;
            Ms_ring[NS - 1] = sd_pref.out_frames;
            Ns_ring[NS - 1] = sd_pref.frames;
            g_ring[NS - 1] = sd_pref.gain;
            if (s + NS - 1 < S && sd_pref.out_frames > m_tile0) {
                stage_source(sd_pref.data, sd_pref.frames, st_new);
                iss |= 1u << (NS - 1);
            }
            sd_pref = load_desc(s + NS);
        }
        RH_PH(1)
        // (3) everything older than the newest (groups issued after source s) has landed: the stage
        //     of source s, and any aggregate fetch issued 2 or more iterations ago
        wait_groups<KV, NS>(__builtin_popcount(iss >> 1));
        RH_PH(2)
        // (3b) all lerp taps of source s leave for the LDS now, so that their latency overlaps the carry step
        const uint32_t Ms = Ms_ring[0];
        const bool active = s < S && Ms > m_tile0;  // this tile still holds frames of source s
        V ta[R + 2], tb2[R + 2];
        if (active) {
            const lds_u8 *buf = lds + st_cur;
#pragma unroll
            for (int rr = 0; rr < R + 2; ++rr) {
                ta[rr] = CH::ld_lds(buf + offA[rr]);
                tb2[rr] = CH::ld_lds(buf + offA[rr] + FB);
            }
            asm volatile("" ::: "memory");  // keep the reads above the carry step
        }
        // (4) every 8th source: the group's tile carries join Cacc (lane l < 32: source l>>2, predecessor l&3 + 4i)
        if (FILT && (s & 7) == 2 && s >= 8 * kGroupLag && Jc > 0 && grp_mask) {
            const bool src_on = lane < 32 && ((grp_mask >> (7 - set_src)) & 1u);
            for (uint32_t i = 0; i < NI; ++i) {
                // only the predecessors in which the source already ran on its own hold an aggregate of it
                const uint32_t have = src_on ? *(const RH_LDS unsigned char *)(lds + kCntBase + ((grp_first + set_src) & 31u)) : 0u;
                const bool want = src_on && set_pred + 4 * i < have;
                unsigned long long gv[4] = {0, 0, 0, 0};
                bool ok = false;
                if (want && !dead) {
                    const lds_u8 *gsl = lds + kGranBase + 1024u * i + lane * 32;
                    const v2u64 lo = *(const lds_u64x2 *)gsl, hi = *(const lds_u64x2 *)(gsl + 16);
                    gv[0] = lo.x; gv[1] = lo.y; gv[2] = hi.x; gv[3] = hi.y;
                    ok = true;
#pragma unroll
                    for (int q = 0; q < 4; ++q) ok = ok && ((uint32_t)(gv[q] >> 32) == p.epoch);
                }
#ifndef RH_DIAG_NO_CARRY_WAIT  // diagnostic build: free-running tiles (wrong results), to price the lock-step
                if (!dead && !__all(ok || !want)) {  // a neighbour is more than 8*(kGroupLag-1) sources behind
                    const unsigned long long *gp = p.gran + ((uint64_t)(grp_first + (want ? set_src : 0)) * ncol + (col - 1 - (want ? set_pred + 4 * i : 0))) * 4;
                    poll_sets(gp, want, gv, ok);
                }
#endif
                if (want && ok && !dead) {
                    const v4f k4 = *(const lds_f4 *)(lds + kLookBase + (set_pred + 4 * i) * 16);
                    const float kM[4] = {k4.x, k4.y, k4.z, k4.w};
                    mat_acc(kM, __uint_as_float((uint32_t)gv[0]), __uint_as_float((uint32_t)gv[1]), Cacc[0], Cacc[1]);
                    mat_acc(kM, __uint_as_float((uint32_t)gv[2]), __uint_as_float((uint32_t)gv[3]), Cacc[2], Cacc[3]);
                }
            }
        }
        RH_PH(3)

        // (5) source s: lerp, zero-state biquad run, ordered mix
        // frames past the end of the mix are never stored, so a source that ends with the mix needs no masking
        const bool full = Ms >= m_tile0 + L || Ms >= Mout;
        uint64_t merged = 0;
        if (active) {
            const uint32_t Ns = Ns_ring[0];
            const float g = g_ring[0];  // Amplify factor of source s: rides on the mix and on the run-end state
            // edge: some lane needs masking, or the staged span reaches past the source (verbatim last frame)
            const bool edge = !full || i_base + VF * nvec > Ns;
            const uint32_t dthr = Ns - 1 - i_base;  // active => i_base <= Ns-1
            const int thr = (int)((dthr < (1u << 27) ? dthr : (1u << 27)) * FB);
            auto tap = [&](int rr, auto edge_tag) -> V {
                const V a = ta[rr], b = tb2[rr];
                V x;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float ac = CH::get(a, c), bc = CH::get(b, c);
                    float xc;
                    if (FILT) {
                        xc = fma_(bc - ac, wgt[rr], ac);
                    } else {  // amplify.rs:64 then math.rs:25: first + (second - first) * num / den, exactly
                        const float ag = ac * g, bg = bc * g;
                        xc = ag + div_T((bg - ag) * wgt[rr], p.Tf, p.rcpT);
                    }
                    if (decltype(edge_tag)::value) {  // the source's last frame is emitted verbatim
                        const bool last = offA[rr] >= thr;
                        xc = last ? (FILT ? ac : ac * g) : xc;
                    }
                    CH::set(x, c, xc);
                }
                return x;
            };
            // lanes past the end of the source contribute nothing (the reference's iterator ended)
            const int nvalid = Ms >= m0 + R ? R : (Ms > m0 ? (int)(Ms - m0) : 0);
            if (FILT) {
                V w1 = CH::zero(), w2 = CH::zero();
                auto run = [&](auto masked) {
                    V x2 = first ? CH::zero() : tap(0, masked);
                    V x1 = first ? CH::zero() : tap(1, masked);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const V x = tap(r + 2, masked);
                        V w;  // zero-state step of the recursive part; the w1 term goes last (shortest chain)
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            const float wc = fma_(na1, CH::get(w1, c), fma_(na2, CH::get(w2, c), fma_(c2, CH::get(x2, c), c1 * CH::get(x1, c))));
                            CH::set(w, c, wc);
                            float yc = fma_(b0, CH::get(x, c), wc);
                            if (decltype(masked)::value) yc = r < nvalid ? yc : 0.0f;
                            CH::set(acc[r], c, fma_(g, yc, CH::get(acc[r], c)));
                        }
                        w2 = w1;
                        w1 = w;
                        x2 = x1;
                        x1 = x;
                    }
                };
                if (!edge) run(std::false_type{});
                else run(std::true_type{});
                RH_PH(4)
                if (!indiv_all && Ms >= m_stable) {  // stable: joins the summed state, nothing else
                    E1s = vfma_s(g, w1, E1s);
                    E2s = vfma_s(g, w2, E2s);
                } else {
                {  // how many predecessor tiles saw this source as not stable (and published it on its own)
                    const uint32_t tM = Ms / L;
                    uint32_t have = indiv_all ? Jc : (tile + p.J > tM ? tile + p.J - tM : 0u);
                    have = have < Jc ? have : Jc;
                    if (lane == 0) *(RH_LDS unsigned char *)(lds + kCntBase + (s & 31u)) = (unsigned char)have;
                }
                // ---- run end state in the scan basis, then the wave64 inclusive scan ----
                float P[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < C; ++c) mat_acc(p.u.Tm, CH::get(w1, c) * g, CH::get(w2, c) * g, P[2 * c], P[2 * c + 1]);
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
                // the tile aggregate (lane 63's inclusive prefix) waits in LDS for the group's publication
                if (lane == 63) *(lds_f4 *)(lds + kPubBase + (s & 7) * 16) = v4f{P[0], P[1], P[2], P[3]};
                pubmask |= 1u << (s & 7);
                float Qnew[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) Qnew[q] = dpp0<kDppWaveShr1, 0xf>(P[q]);  // exclusive: lane 0 gets 0
                if (full) {
                    merged = 1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) Qacc[q] += Qnew[q];
                } else {
                    // The source ends inside this tile while the mix goes on (at most one tile per
                    // source): its correction is masked per frame, so it cannot join the merged
                    // states.  Fetch its carry right now (lane j: predecessor j) and apply it exactly.
                    float c[4] = {0.f, 0.f, 0.f, 0.f};
                    if (Jc > 0) {
                        const bool want = (uint32_t)lane < Jc;
                        unsigned long long gv[4] = {0, 0, 0, 0};
                        bool ok = false;
                        poll_sets(p.gran + ((uint64_t)s * ncol + (col - 1 - (want ? lane : 0))) * 4, want, gv, ok);
                        if (want && ok && !dead) {
                            const v4f k4 = *(const lds_f4 *)(lds + kLookBase + lane * 16);
                            const float kM[4] = {k4.x, k4.y, k4.z, k4.w};
                            mat_acc(kM, __uint_as_float((uint32_t)gv[0]), __uint_as_float((uint32_t)gv[1]), c[0], c[1]);
                            mat_acc(kM, __uint_as_float((uint32_t)gv[2]), __uint_as_float((uint32_t)gv[3]), c[2], c[3]);
                        }
                        reduce_carry(c);
                    }
                    mat_acc(lM, c[0], c[1], Qnew[0], Qnew[1]);
                    mat_acc(lM, c[2], c[3], Qnew[2], Qnew[3]);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const bool v = r < nvalid;
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            const float h = fma_(p.u.g[r][0], Qnew[2 * c], p.u.g[r][1] * Qnew[2 * c + 1]);
                            CH::set(acc[r], c, CH::get(acc[r], c) + (v ? h : 0.0f));
                        }
                    }
                }
                }  // individual source
            } else {
                auto run = [&](auto masked) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        V x = tap(r + 2, masked);
                        if (decltype(masked)::value && !(r < nvalid)) continue;  // an ended source adds nothing, not even +0.0
                        acc[r] += x;
                    }
                };
                if (!edge) run(std::false_type{});
                else run(std::true_type{});
            }
        }
        // (6) the group is complete: publish this tile's aggregates, 8 sources x 4 granules in one store
        if (FILT && s < S && ((s & 7) == 7 || s == S - 1)) {
            if (lane < 32 && ((pubmask >> (lane >> 2)) & 1u)) {
                const float ev = *(const RH_LDS float *)(lds + kPubBase + lane * 4);
                const unsigned long long word = ((unsigned long long)p.epoch << 32) | __float_as_uint(ev);
                __hip_atomic_store(p.gran + ((uint64_t)((s & ~7u) + (lane >> 2)) * ncol + col) * 4 + (lane & 3), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            pubmask = 0;
        }
        // rotate the rings
        mrg = (mrg << 1) | merged;
        iss >>= 1;
#pragma unroll
        for (int d = 0; d + 1 < NS; ++d) {
            Ms_ring[d] = Ms_ring[d + 1];
            Ns_ring[d] = Ns_ring[d + 1];
            g_ring[d] = g_ring[d + 1];
        }
        st_cur += kStage;
        if (st_cur >= NS * kStage) st_cur = 0;
        RH_PH(5)
    }
    wait_vm<0>();  // nothing of this wave may still be in flight towards its LDS
#ifdef RH_PHASE_PROFILE
    if (p.prof && lane == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        ph_t[6] = ((unsigned long long)xcc << 32) | hwid;
        ph_t[7] = ph_start;
        for (int i = 0; i < 8; ++i) p.prof[(uint64_t)tile * 8 + i] = ph_t[i];
    }
#endif

    if (FILT && !indiv_all) {  // the stable sources' summed state: one scan, one published aggregate, one look-back
        float P[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) mat_acc(p.u.Tm, CH::get(E1s, c), CH::get(E2s, c), P[2 * c], P[2 * c + 1]);
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
        {
            const float q0 = dpp0<kDppBcast15, 0xa>(P[0]), q1 = dpp0<kDppBcast15, 0xa>(P[1]);
            const float q2 = dpp0<kDppBcast15, 0xa>(P[2]), q3 = dpp0<kDppBcast15, 0xa>(P[3]);
            mat_acc(b15, q0, q1, P[0], P[1]);
            mat_acc(b15, q2, q3, P[2], P[3]);
        }
        {
            const float q0 = dpp0<kDppBcast31, 0xc>(P[0]), q1 = dpp0<kDppBcast31, 0xc>(P[1]);
            const float q2 = dpp0<kDppBcast31, 0xc>(P[2]), q3 = dpp0<kDppBcast31, 0xc>(P[3]);
            mat_acc(b31, q0, q1, P[0], P[1]);
            mat_acc(b31, q2, q3, P[2], P[3]);
        }
        unsigned long long *const sum_row = p.gran + (uint64_t)S * ncol * 4;
        {
            const float e0 = readlane_f(P[0], 63), e1 = readlane_f(P[1], 63);
            const float e2 = readlane_f(P[2], 63), e3 = readlane_f(P[3], 63);
            if (lane < 4) {
                const float ev = lane == 0 ? e0 : lane == 1 ? e1 : lane == 2 ? e2 : e3;
                const unsigned long long word = ((unsigned long long)p.epoch << 32) | __float_as_uint(ev);
                __hip_atomic_store(sum_row + (uint64_t)col * 4 + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) Qacc[q] += dpp0<kDppWaveShr1, 0xf>(P[q]);  // exclusive: lane 0 gets 0
        if (Jc > 0) {  // lane j < Jc: predecessor j's summed aggregate (they finish about now)
            const bool want = (uint32_t)lane < Jc;
            unsigned long long gv[4] = {0, 0, 0, 0};
            bool ok = false;
            poll_sets(sum_row + (uint64_t)(col - 1 - (want ? lane : 0)) * 4, want, gv, ok);
            if (want && ok && !dead) {
                const v4f k4 = *(const lds_f4 *)(lds + kLookBase + lane * 16);
                const float kM[4] = {k4.x, k4.y, k4.z, k4.w};
                mat_acc(kM, __uint_as_float((uint32_t)gv[0]), __uint_as_float((uint32_t)gv[1]), Cacc[0], Cacc[1]);
                mat_acc(kM, __uint_as_float((uint32_t)gv[2]), __uint_as_float((uint32_t)gv[3]), Cacc[2], Cacc[3]);
            }
        }
    }
    if (FILT) {  // the merged homogeneous response: start state = Qacc + B^(R*lane) * (summed tile carries)
        if (dead) Cacc[0] = Cacc[1] = Cacc[2] = Cacc[3] = __builtin_nanf("");  // a carry never arrived: poison the tile (the status word fails the call)
        reduce_carry(Cacc);
        mat_acc(lM, Cacc[0], Cacc[1], Qacc[0], Qacc[1]);
        mat_acc(lM, Cacc[2], Cacc[3], Qacc[2], Qacc[3]);
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int c = 0; c < C; ++c) CH::set(acc[r], c, fma_(p.u.g[r][0], Qacc[2 * c], fma_(p.u.g[r][1], Qacc[2 * c + 1], CH::get(acc[r], c))));
        }
    }

    // ---- mixed output: R frames per lane -----------------------------------------------------
    float *o = p.out + (uint64_t)m0 * C;
    if (C == 1) {
        if (R % 4 == 0) {
#pragma unroll
            for (int r = 0; r + 3 < R; r += 4) {
                const uint32_t m = m0 + r;
                if (m + 3 < Mout) *reinterpret_cast<float4 *>(o + r) = make_float4(CH::get(acc[r], 0), CH::get(acc[r + 1], 0), CH::get(acc[r + 2], 0), CH::get(acc[r + 3], 0));
                else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (m + k < Mout) o[r + k] = CH::get(acc[r + k], 0);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (m0 + r < Mout) o[r] = CH::get(acc[r], 0);
        }
    } else if (R % 2 == 0) {
#pragma unroll
        for (int r = 0; r + 1 < R; r += 2) {
            const uint32_t m = m0 + r;
            if (m + 1 < Mout) {
                *reinterpret_cast<float4 *>(o + r * 2) = make_float4(CH::get(acc[r], 0), CH::get(acc[r], C - 1), CH::get(acc[r + 1], 0), CH::get(acc[r + 1], C - 1));
            } else if (m < Mout) {
                *reinterpret_cast<float2 *>(o + r * 2) = make_float2(CH::get(acc[r], 0), CH::get(acc[r], C - 1));
            }
        }
    } else {  // odd R: a lane's run starts on an 8-byte boundary only
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (m0 + r < Mout) *reinterpret_cast<float2 *>(o + r * 2) = make_float2(CH::get(acc[r], 0), CH::get(acc[r], C - 1));
    }
}

// =================================================================================================
// k_mix_rows -- "mix first".  The converter and the filter are linear and, for a batch of equal-length sources of one format,
// the same for every source (same taps, same weights, same span seams, same end):
//     sum_s filter(resample(g_s * x_s)) = filter(resample(sum_s g_s * x_s)).
// So a filtered equal-length batch (the benchmark) is mixed at the INPUT rate first -- this kernel: y[n] = sum_s g_s * x_s[n],
// sources in insertion order, one pass over every input byte as whole aligned 16-byte vectors, nothing else to do per byte --
// and the fused kernel then converts and filters ONE stream (y: 1/S of the input).  The streaming pass is what the roofline
// prices: S * N * C * 4 bytes in, N * C * 4 out.  (Without a filter the batch stays on the per-source path: that one is
// bit-exact with rodio's ordered sum of converted samples; the filtered path is compared at 1e-5 either way.)
// =================================================================================================
// Short rows (a stream's block: 64 Ki frames are 128 workgroups at one vector per lane, half the chip) are cut the other way as well: gridDim.y
// GROUPS of sources, group g summing its share of the source list into partial row g (y + g * group_stride, descriptor ydesc[g]); the fused launch
// behind takes the partial rows as its sources and adds them in order.  One group: the whole list into one row, as before.
template <int U>
__global__ __launch_bounds__(256) void k_mix_rows(const SrcDesc *__restrict__ srcs_all, const uint32_t n_sources_all, float *__restrict__ y_all, const uint64_t n_floats, SrcDesc *__restrict__ ydesc_all,
                                                  const uint32_t frames, const uint32_t out_frames, const float *desc_row_all, const uint64_t group_stride, const uint64_t src_off) {
    // (src_off: bytes added to every source pointer of the table -- a stream whose sources all moved on by the same amount since the table was
    // uploaded passes the distance instead of uploading it again: rh_pipeline_stream.hip)
    typedef __attribute__((address_space(4))) const uint64_t cu64;
    typedef __attribute__((address_space(4))) const float cf32;
    typedef RH_GLB const v4f glb_cf4;
    const uint32_t per_group = (n_sources_all + gridDim.y - 1) / gridDim.y, s_first = blockIdx.y * per_group;
    const uint32_t n_sources = s_first < n_sources_all ? (n_sources_all - s_first < per_group ? n_sources_all - s_first : per_group) : 0u;
    const SrcDesc *const srcs = srcs_all + s_first;
    float *const y = y_all + (uint64_t)blockIdx.y * group_stride;
    SrcDesc *const ydesc = ydesc_all + blockIdx.y;
    const float *const desc_row = desc_row_all + (uint64_t)blockIdx.y * group_stride;
    cu64 *const desc = (cu64 *)(uintptr_t)srcs;
    cf32 *const dgain = (cf32 *)(uintptr_t)srcs;
    const uint64_t nvec = n_floats / 4;
    const uint64_t base = (uint64_t)blockIdx.x * (256u * U) + threadIdx.x;
    v4f acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = v4f{0.f, 0.f, 0.f, 0.f};
    constexpr int SB = 8 / U > 2 ? 8 / U : 2;  // sources per step: 8 (or 2*U) vector loads per lane
    if (base + (uint64_t)(U - 1) * 256u < nvec) {  // every vector of this lane is whole and inside
        // Two register sets: while the vectors of one step are summed, the loads of the next are in flight and the pointers
        // and gains of the step after that are on their way through the scalar cache (the body is written out twice so that
        // the sets swap without moves).
        const uint32_t steps = n_sources / SB;
        uint64_t pN[SB];
        float gN[SB];
        auto fetch_desc = [&](uint32_t j) {
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                pN[k] = desc[4 * (uint64_t)(j * SB + k)] + src_off;
                gN[k] = dgain[8 * (uint64_t)(j * SB + k) + 4];
            }
        };
        auto issue = [&](v4f (&v)[SB][U], float (&g)[SB]) {
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                glb_cf4 *const ptr = (glb_cf4 *)(uintptr_t)pN[k];
                g[k] = gN[k];
#pragma unroll
                for (int u = 0; u < U; ++u) v[k][u] = __builtin_nontemporal_load(ptr + base + (uint64_t)u * 256u);
            }
        };
        auto consume = [&](const v4f (&v)[SB][U], const float (&g)[SB]) {
#pragma unroll
            for (int k = 0; k < SB; ++k)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    acc[u].x = fma_(g[k], v[k][u].x, acc[u].x);
                    acc[u].y = fma_(g[k], v[k][u].y, acc[u].y);
                    acc[u].z = fma_(g[k], v[k][u].z, acc[u].z);
                    acc[u].w = fma_(g[k], v[k][u].w, acc[u].w);
                }
        };
        v4f vA[SB][U], vB[SB][U];
        float gA[SB], gB[SB];
        // (the loads of the next step are issued unconditionally inside the loop: a conditional issue makes the compiler wait
        // for the newest load of either path, which drains the pipeline every step)
        if (steps) {
            fetch_desc(0);
            issue(vA, gA);
            if (steps > 1) fetch_desc(1);
            uint32_t j = 0;
            for (; j + 2 < steps; j += 2) {
                issue(vB, gB);         // step j+1
                fetch_desc(j + 2);
                consume(vA, gA);       // step j
                issue(vA, gA);         // step j+2
                fetch_desc(j + 3 < steps ? j + 3 : steps - 1);
                consume(vB, gB);       // step j+1
            }
            if (j + 1 < steps) {  // two steps left: A in flight, B's descriptors fetched
                issue(vB, gB);
                consume(vA, gA);
                consume(vB, gB);
            } else {
                consume(vA, gA);
            }
        }
        for (uint32_t s = steps * SB; s < n_sources; ++s) {
            glb_cf4 *const ptr = (glb_cf4 *)(uintptr_t)(desc[4 * (uint64_t)s] + src_off);
            const float g = dgain[8 * (uint64_t)s + 4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const v4f v = __builtin_nontemporal_load(ptr + base + (uint64_t)u * 256u);
                acc[u].x = fma_(g, v.x, acc[u].x);
                acc[u].y = fma_(g, v.y, acc[u].y);
                acc[u].z = fma_(g, v.z, acc[u].z);
                acc[u].w = fma_(g, v.w, acc[u].w);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) *reinterpret_cast<v4f *>(y + 4 * (base + (uint64_t)u * 256u)) = acc[u];
    } else {  // the last workgroup: vector by vector, guarded
        for (int u = 0; u < U; ++u) {
            const uint64_t i = base + (uint64_t)u * 256u;
            if (i >= nvec) break;
            v4f a = v4f{0.f, 0.f, 0.f, 0.f};
            for (uint32_t s = 0; s < n_sources; ++s) {
                glb_cf4 *const ptr = (glb_cf4 *)(uintptr_t)(desc[4 * (uint64_t)s] + src_off);
                const float g = dgain[8 * (uint64_t)s + 4];
                const v4f v = ptr[i];
                a.x = fma_(g, v.x, a.x), a.y = fma_(g, v.y, a.y), a.z = fma_(g, v.z, a.z), a.w = fma_(g, v.w, a.w);
            }
            *reinterpret_cast<v4f *>(y + 4 * i) = a;
        }
    }
    if (blockIdx.x == 0) {
        const uint64_t t = nvec * 4 + threadIdx.x;  // the floats behind the last whole vector (a mono batch of a length not divisible by 4)
        if (t < n_floats) {
            float a = 0.f;
            for (uint32_t s = 0; s < n_sources; ++s) a = fma_(dgain[8 * (uint64_t)s + 4], ((glb_cf32 *)(uintptr_t)(desc[4 * (uint64_t)s] + src_off))[t], a);
            y[t] = a;
        }
        if (threadIdx.x == 0) {  // the one-entry descriptor table the fused launch behind this one reads
            ydesc->data = desc_row;  // the row the launch behind this one reads: the mix, or what a filter makes of it
            ydesc->frames = frames;
            ydesc->out_frames = out_frames;
            ydesc->gain = 1.0f;
            ydesc->pad[0] = ydesc->pad[1] = ydesc->pad[2] = 0;
        }
    }
}

// The same sum for long rows, in the shape that reaches the read ceiling of this part (tools/ubench/stream_ring: 7.19 TB/s):
// one wave owns chunk t -- 8 KiB, aligned -- of EVERY source; it pulls the chunks through a ring of NS LDS stages with LDS-DMA
// (8 instructions of 1 KiB per source, no registers held by data in flight), reads its own 8 vectors of a landed stage back
// and accumulates.  A row of n_floats gives ceil(n_floats / 2048) waves: for rows that fill the chip (the host decides).
template <int NS>
__global__ __launch_bounds__(64) void k_mix_ring(const SrcDesc *__restrict__ srcs, const uint32_t n_sources, float *__restrict__ y, const uint64_t n_floats, SrcDesc *__restrict__ ydesc,
                                                 const uint32_t frames, const uint32_t out_frames, const float *desc_row, const uint64_t src_off) {
    constexpr int KV = 8;
    constexpr uint32_t kStage = KV * 1024;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * kStage];
    lds_u8 *const lds = (lds_u8 *)smem;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
    typedef __attribute__((address_space(4))) const uint64_t cu64;
    typedef __attribute__((address_space(4))) const float cf32;
    cu64 *const desc = (cu64 *)(uintptr_t)srcs;
    cf32 *const dgain = (cf32 *)(uintptr_t)srcs;
    const int lane = threadIdx.x;
    const uint64_t nvec = n_floats / 4;
    const uint64_t v0 = (uint64_t)blockIdx.x * (KV * 64);  // first vector of this wave's chunk
    // byte offsets of this lane's KV vectors inside a row; vectors past the end of the row re-fetch its last one (never stored)
    uint32_t goff[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        uint64_t j = v0 + (uint64_t)k * 64 + lane;
        j = j < nvec ? j : nvec - 1;
        goff[k] = (uint32_t)(j * 16);  // rows are < 2^32 bytes (frames < 2^29)
    }
    auto stage_source = [&](uint32_t s_, uint32_t stage) {
        const void *data = (const void *)(uintptr_t)(desc[4 * (uint64_t)s_] + src_off);
#pragma unroll
        for (int k = 0; k < KV; ++k) glds16(data, goff[k], lds0 + stage * kStage + k * 1024);
    };
    v4f acc[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) acc[k] = v4f{0.f, 0.f, 0.f, 0.f};
    if (v0 < nvec) {
#pragma unroll
        for (int d = 0; d < NS - 1; ++d)
            if ((uint32_t)d < n_sources) stage_source(d, d);
        uint32_t st = 0;
        float g_next = n_sources ? dgain[4] : 0.f;
        for (uint32_t s_ = 0; s_ < n_sources; ++s_) {
            const float g = g_next;
            g_next = s_ + 1 < n_sources ? dgain[8 * (uint64_t)(s_ + 1) + 4] : 0.f;
            // the stage that source s_-1 was read from is free (its reads were waited for): source s_+NS-1 goes there
            if (s_ + NS - 1 < n_sources) {
                uint32_t into = st + NS - 1;
                into = into >= (uint32_t)NS ? into - NS : into;
                stage_source(s_ + NS - 1, into);
                wait_vm<KV *(NS - 1)>();
            } else {
                const uint32_t left = n_sources - 1 - s_;  // groups issued after this source's
                wait_groups<KV, NS>((int)left);
            }
            const lds_u8 *buf = lds + st * kStage;
            v4f v[KV];
#pragma unroll
            for (int k = 0; k < KV; ++k) v[k] = *(const lds_f4 *)(buf + k * 1024 + lane * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the stage may be re-targeted by the next DMA
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                acc[k].x = fma_(g, v[k].x, acc[k].x);
                acc[k].y = fma_(g, v[k].y, acc[k].y);
                acc[k].z = fma_(g, v[k].z, acc[k].z);
                acc[k].w = fma_(g, v[k].w, acc[k].w);
            }
            st = st + 1 == (uint32_t)NS ? 0 : st + 1;
        }
        wait_vm<0>();
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const uint64_t j = v0 + (uint64_t)k * 64 + lane;
            if (j < nvec) *reinterpret_cast<v4f *>(y + 4 * j) = acc[k];
        }
    }
    if (blockIdx.x == 0) {
        const uint64_t t = nvec * 4 + lane;  // the floats behind the last whole vector
        if (t < n_floats) {
            float a = 0.f;
            for (uint32_t s_ = 0; s_ < n_sources; ++s_) a = fma_(dgain[8 * (uint64_t)s_ + 4], ((glb_cf32 *)(uintptr_t)(desc[4 * (uint64_t)s_] + src_off))[t], a);
            y[t] = a;
        }
        if (lane == 0) {
            ydesc->data = desc_row;  // the row the launch behind this one reads: the mix, or what a filter makes of it
            ydesc->frames = frames;
            ydesc->out_frames = out_frames;
            ydesc->gain = 1.0f;
            ydesc->pad[0] = ydesc->pad[1] = ydesc->pad[2] = 0;
        }
    }
}

// =================================================================================================
// k_rlm_chunk -- mix first in ONE kernel, for rows long enough to fill the chip (stereo).  Tile t owns the ALIGNED 8 KiB chunk t
// of every source: it sums the chunks through the LDS-DMA ring exactly as k_mix_ring does (the pass that reaches the read
// ceiling, no re-fetched line), and then converts and filters ITS part of the one mixed stream itself -- no mixed row in memory,
// no second launch.  What makes that possible:
//   * a tile's output frames are those whose SECOND tap lies in its chunk (frames m_lo[t] .. m_lo[t+1]-1, a table of the host's:
//     1114 or 1115 of them at 44.1 -> 48 kHz): the only input a tile lacks is the end of the chunk before it -- the first tap of
//     its first frame, and the taps of the two frames x'[m_lo-1], x'[m_lo-2] the filter looks back at.  Those are MIXED frames:
//     the tile before publishes its last 4 of them (tagged words, like the aggregates) as soon as its source loop ends;
//   * lanes take runs of R frames as in k_rlm_fast; the tile's last lane takes what is left (v <= R frames) and the lanes behind
//     it idle.  The scan is the uniform one; only the tile aggregate needs the short run: A = B^v * (prefix of the lane before)
//     + (own run), one 2x2 product with a table of B^v;
//   * tiles have different lengths, so the look-back weights B^(m_lo[t] - m_lo[t-j]) come from a table per tile (host, f64).
// Waits only ever go to EARLIER tiles, and the host launches this kernel only when every tile is resident at once.
// =================================================================================================
struct ChunkArgs {
    const uint32_t *m_lo;      // [n_tiles + 1]: first output frame of every tile; m_lo[n_tiles] = out_frames
    unsigned long long *halo;  // [n_tiles][8] {epoch, f32 bits}: the last 4 mixed frames of the tile's chunk
    const float *lookT;        // [n_tiles][J][4]: B^(m_lo[t] - m_lo[t-j]), j = 0 .. J-1 (the weight of tile t-1-j's aggregate)
    const float *powM;         // [R + 1][4]: B^v
    const float *uni;          // the plan's Uniforms as an array of floats in device memory (Params::u holds the same values)
};
constexpr uint32_t kChunkMultiMax = 7;  // classes in one launch (k_rlm_chunk_multi): what fits the 4 KiB kernarg segment
struct ChunkMulti {
    struct Entry {
        Params p;
        ChunkArgs q;
    };
    uint32_t n;                          // classes in this launch
    uint32_t first[kChunkMultiMax + 1];  // first workgroup of every class (multiples of 8: a workgroup's XCD is its class-relative index's too); [n] = the grid
    float *sum_out;                      // k_rlm_chunk_classes: where the SUM of the classes' mixes goes (the classes' own rows stay unwritten)
    Entry e[kChunkMultiMax];
};
static_assert(sizeof(ChunkMulti) <= 4096, "the kernarg segment");
#if defined(RH_CHUNK_DIAG) && RH_CHUNK_DIAG == 3  // diagnostics builds: shader cycles per phase behind the source loop, summed over the tiles into ctl[8..15]
#define RH_CPH(i) { const unsigned long long cph_now = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(p.ticket + 8 + (i), (uint32_t)(cph_now - cph_last)); cph_last = cph_now; }
#define RH_CPH_DECL unsigned long long cph_last = __builtin_readcyclecounter();
#else
#define RH_CPH(i)
#define RH_CPH_DECL
#endif
// C: channels of a frame; KV: KiB of a chunk (8 for stereo: 1024 frames; 4 for mono: 1024 frames too -- the tile stays at 64 runs of 18).
template <int R, int C, int KV>
__device__ __forceinline__ void rlm_chunk_tile(const Params &p, const ChunkArgs &q, const uint32_t block) {
    typedef Chan<C> CH;
    typedef typename CH::V V;
    constexpr int NS = 2, H = 4;
    constexpr uint32_t FB = CH::kFB;
    constexpr uint32_t kStage = KV * 1024, P = kStage / FB;   // bytes of a ring stage = one chunk; frames per chunk
    constexpr uint32_t MB = NS * kStage + 64;                 // the mixed chunk: halo frames at MB - H * FB .. MB, 16 spare bytes behind it
    __shared__ __attribute__((aligned(1024))) unsigned char smem[MB + kStage + 64];
    lds_u8 *const lds = (lds_u8 *)smem;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
    typedef __attribute__((address_space(4))) const uint64_t cu64;
    typedef __attribute__((address_space(4))) const float cf32;
    typedef __attribute__((address_space(4))) const uint32_t cu32;
    cu64 *const desc = (cu64 *)(uintptr_t)p.srcs;
    cf32 *const dgain = (cf32 *)(uintptr_t)p.srcs;
    const int lane = threadIdx.x;
    // Every tile resident at once (the host knows): tile = workgroup.  More tiles than slots: tiles are handed out by a ticket, so
    // that a tile only ever waits for tiles that already run or have finished (its waits go to earlier tiles only).
    uint32_t tile = block;
    if (!p.direct) {
        // One device-scope counter hands out ~85 tickets per microsecond: 12 us for the 1024 tiles of the benchmark batch, 4 % of
        // the launch.  So the tickets come from EIGHT counters on separate cache lines, one per XCD: workgroup b takes ticket k of
        // counter b % 8 and works on tile b % 8 + 8 k.  Every XCD dispatches its own workgroups (those with its b % 8) in order, so a
        // workgroup's tile is its own index unless its neighbours of the same XCD overtake it -- and a tile still only ever waits
        // for tiles that hold a slot or are done: the lowest unfinished tile of a counter is held by a workgroup that runs, or all
        // earlier workgroups of that XCD have finished and the next one starts.  (The grid is rounded up to whole rounds of eight, so
        // that every counter advances by the same amount per launch: workgroups past the last tile leave.)
        const uint32_t x = block & 7u;
        const uint32_t k = __builtin_amdgcn_readfirstlane(lane == 0 ? atomicAdd(p.ticket + 32u * (1u + x), 1u) - p.shard_base : 0u);
        tile = x + 8u * k;
        if (tile >= p.n_tiles) return;
    }
#ifdef RH_CHUNK_RAGSIM  // diagnostics builds: the load profile of a ragged batch (lengths uniform in [N/2, N]) on the equal batch's data -- timing only
    const uint32_t Ns = p.eq_frames;
    const uint32_t S = 2u * tile < p.n_tiles ? p.n_sources : (uint32_t)(((uint64_t)p.n_sources * 2u * (p.n_tiles - tile)) / p.n_tiles);
#else
    const uint32_t Ns = p.eq_frames, S = p.n_sources;
#endif
    const uint32_t m_lo = ((cu32 *)(uintptr_t)q.m_lo)[tile], m_hi = ((cu32 *)(uintptr_t)q.m_lo)[tile + 1];
    const uint32_t m0 = m_lo + (uint32_t)lane * R;
    int nfl;  // frames of this lane's run
    // what the part behind the source loop needs of the tile's bounds lives in VECTOR registers across the loop (scalar registers
    // are short there, and a scalar load behind the loop is a memory round trip nothing hides): the frame count and B^v
    uint32_t n_t_v;
    float pwv[4];
    {
        const uint32_t n_t = m_hi - m_lo;  // <= 64 * R (host)
        nfl = (int)n_t - lane * R < 0 ? 0 : ((int)n_t - lane * R > R ? R : (int)n_t - lane * R);
        const uint32_t nl0 = (n_t + R - 1) / R;
        const uint32_t v = nl0 ? n_t - (nl0 - 1) * R : 0;  // frames of the last lane's run, 1 .. R
        n_t_v = n_t;
        asm volatile("v_mov_b32 %0, %1" : "=v"(n_t_v) : "s"(n_t));
        const float *pw = q.powM + 4 * v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pwv[k] = pw[k];
            asm volatile("" : "+v"(pwv[k]));  // (a vector load of a uniform address: the values stay in vector registers)
        }
    }
    const bool first = (m0 == 0);                              // stream start: x'[-1] = x'[-2] = 0

    // ---- the sum of chunk `tile` of every source (k_mix_ring) ----
    const uint32_t nvec = Ns * C / 4;  // 16-byte vectors of a row (host: whole vectors)
    const uint32_t v0 = tile * (KV * 64);
    uint32_t goff[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        uint32_t j = v0 + (uint32_t)k * 64 + lane;
        j = j < nvec ? j : nvec - 1;  // past the end of the row: its last vector again (finite, never a tap of a stored frame)
        goff[k] = j * 16;
    }
    // every chunk but a row's last lies whole inside the row: its DMA instructions are 1 KiB apart in memory as in the LDS and go
    // out as one run (one M0, one offset register: glds16_run)
    const bool lin = v0 + (uint32_t)(KV * 64) <= nvec;
    auto stage_source = [&](uint32_t s_, uint32_t stage) {
        const void *data = (const void *)(uintptr_t)desc[4 * (uint64_t)s_];
        if (lin) {
            glds16_run<KV>(data, goff[0], lds0 + stage * kStage);
        } else {
#pragma unroll
            for (int k = 0; k < KV; ++k) glds16(data, goff[k], lds0 + stage * kStage + k * 1024);
        }
    };
    if (S) stage_source(0, 0);  // the first two chunks are on their way while the lane works out its taps
    if (S > 1) stage_source(1, 1);
    // the lane's rows of the filter tables and its look-back weight: fetched here, a source loop away from their use
    // Everything the part behind the source loop needs from the kernel arguments is worked out HERE and parked in vector
    // registers: a scalar load of a kernel argument behind the loop is a memory round trip that nothing hides (the compiler
    // re-loads arguments rather than keep them: twelve such trips, one after the other, before this was done).
    const Tables *__restrict__ tb = p.tabs;
    float lM[4], b15[4], b31[4], kM[4];
    constexpr int NHV = H * FB / 16;  // vectors that hold the chunk's last 4 frames: the last lanes' last vector
    const uint32_t Jc = p.J < tile ? p.J : tile;
    int want_look = (uint32_t)lane < Jc ? 1 : 0;
    uint64_t a_halo_pub = (uint64_t)(uintptr_t)(q.halo + (uint64_t)tile * 8 + (uint32_t)(lane >= 64 - NHV ? lane - (64 - NHV) : 0) * 4);
    uint64_t a_halo_poll = (uint64_t)(uintptr_t)(q.halo + (uint64_t)(tile ? tile - 1 : 0) * 8 + (lane < H * C ? lane : 0));
    uint64_t a_gran_pub = (uint64_t)(uintptr_t)(p.gran + (uint64_t)tile * 4 + (lane < 2 * C ? lane : 0));
    uint64_t a_gran_poll = (uint64_t)(uintptr_t)(p.gran + (uint64_t)(tile ? tile - 1 - (want_look ? lane : 0) : 0) * 4);
    uint64_t a_out = (uint64_t)(uintptr_t)(p.out + ((uint64_t)m_lo + (uint32_t)lane) * C);  // frame `lane` of the tile
    uint32_t epoch_v = p.epoch;
    float U = q.uni[lane < 61 ? lane : 60];  // lane l holds float l of the Uniforms: {b0, c1, c2, a1, a2, Tm[4], scanM[4][4], g[r][2] ...}
    asm volatile("" : "+v"(want_look), "+v"(a_halo_pub), "+v"(a_halo_poll), "+v"(a_gran_pub), "+v"(a_gran_poll), "+v"(a_out), "+v"(epoch_v), "+v"(U));
    {
        const uint32_t Jc0 = Jc;
        const float *kp = q.lookT + ((uint64_t)tile * p.J + ((uint32_t)lane < Jc0 ? lane : 0)) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lM[k] = tb->laneM[lane][k];
            b15[k] = tb->bc15M[lane][k];
            b31[k] = tb->bc31M[lane][k];
            kM[k] = kp[k];
        }
    }
    // ---- taps and weights of the lane's R + 2 frames: LDS offsets into [halo | chunk] ----
    int offA[R + 2];
    float wgt[R + 2];
    {
        const int64_t fbase = (int64_t)tile * P;
        Cursor c = cursor_at(first ? 0 : m0 - 2, p);
#pragma unroll
        for (int rr = 0; rr < R + 2; ++rr) {
#if defined(RH_CHUNK_DIAG) && RH_CHUNK_DIAG == 2
            offA[rr] = (int)MB + rr * 8;
            wgt[rr] = 0.5f;
            continue;
#endif
            const bool dummy = first && rr < 2;
            uint64_t i;
            uint32_t num;
            cursor_resolve(c, p, i, num);
            if (i + 1 >= Ns) {  // the last frame is emitted verbatim (sample_rate.rs:193-200); frames past it are never stored
                i = Ns - 1;
                num = 0;
            }
            int64_t f = (int64_t)i - fbase;
            f = f < -H ? -H : (f > (int64_t)P - 1 ? (int64_t)P - 1 : f);  // (only frames that are not stored leave the range)
            offA[rr] = dummy ? (int)MB : (int)MB + (int)f * (int)FB;
            wgt[rr] = dummy ? 0.0f : (float)num / p.Tf;
            if (!dummy) cursor_next(c, p);
        }
    }
    v4f acc[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) acc[k] = v4f{0.f, 0.f, 0.f, 0.f};
    {
        uint32_t st = 0;
        float g_next = S ? dgain[4] : 0.f;
        for (uint32_t s_ = 0; s_ < S; ++s_) {
            const float g = g_next;
            g_next = s_ + 1 < S ? dgain[8 * (uint64_t)(s_ + 1) + 4] : 0.f;
            // source s_ has landed when at most the group behind it is outstanding (compiler-issued loads in between only make
            // the wait stricter)
            if (s_ + 1 < S) wait_vm<KV>();
            else wait_vm<0>();
            const lds_u8 *buf = lds + st * kStage;
            v4f v[KV];
#pragma unroll
            for (int k = 0; k < KV; ++k) v[k] = *(const lds_f4 *)(buf + k * 1024 + lane * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the chunk is in registers: its stage is free ...
            if (s_ + 2 < S) stage_source(s_ + 2, st);           // ... for the source after next, requested before this one is summed
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                acc[k].x = fma_(g, v[k].x, acc[k].x);
                acc[k].y = fma_(g, v[k].y, acc[k].y);
                acc[k].z = fma_(g, v[k].z, acc[k].z);
                acc[k].w = fma_(g, v[k].w, acc[k].w);
            }
            st ^= 1u;
        }
    }
    RH_CPH_DECL
#if defined(RH_CHUNK_DIAG) && RH_CHUNK_DIAG < 3  // diagnostics builds (tools/build_variant.sh): the source loop alone (+ the tap prologue unless RH_CHUNK_DIAG == 2)
    {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < KV; ++k) t += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
#pragma unroll
        for (int rr = 0; rr < R + 2; ++rr) t += wgt[rr] + (float)offA[rr];
        p.out[(uint64_t)tile * 64 + lane] = t + lM[0] + b15[0] + b31[0] + kM[0] + (float)nfl + (first ? 1.f : 0.f);
        return;
    }
#endif
    // ---- the chunk's last 4 mixed frames to the tile behind (first: it is waiting for them); the mixed chunk into the LDS; the
    // last 4 frames of the tile in front ----
    if (lane >= 64 - NHV) {
        unsigned long long *hp = (unsigned long long *)(uintptr_t)a_halo_pub;
        const float e[4] = {acc[KV - 1].x, acc[KV - 1].y, acc[KV - 1].z, acc[KV - 1].w};
#pragma unroll
        for (int w = 0; w < 4; ++w) __hip_atomic_store(hp + w, ((unsigned long long)epoch_v << 32) | __float_as_uint(e[w]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int k = 0; k < KV; ++k) *(lds_f4 *)(lds + MB + (uint32_t)(k * 64 + lane) * 16) = acc[k];
    if (lane == 0) *(lds_f4 *)(lds + MB + kStage) = v4f{0.f, 0.f, 0.f, 0.f};  // the second tap of a verbatim last frame at the end of a chunk: finite, weight 0
    bool dead = false;
    if (tile == 0) {
        if (lane < H * C) *(RH_LDS float *)(lds + MB - H * FB + lane * 4) = 0.0f;
    } else {
        const bool want = lane < H * C;
        const unsigned long long *hp = (const unsigned long long *)(uintptr_t)a_halo_poll;
        unsigned long long hv = 0;
        bool ok = false;
        uint32_t spins = 0;
        while (true) {
            if (want && !ok) {
                hv = __hip_atomic_load(hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (uint32_t)(hv >> 32) == epoch_v;
            }
            if (__all(ok || !want)) break;
            if (++spins > kSpinLimit) {
                if (lane == 0) atomicOr(p.status, 1u);
                dead = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (want) *(RH_LDS float *)(lds + MB - H * FB + lane * 4) = dead ? __builtin_nanf("") : __uint_as_float((uint32_t)hv);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    RH_CPH(0)  // LDS image of the mixed chunk + the halo of the tile in front

    // ---- the lane's run of the mixed stream: lerp, zero-state biquad; the run-end state after nfl frames ----
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
        V x2 = first ? CH::zero() : tap(0);
        V x1 = first ? CH::zero() : tap(1);
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
    RH_CPH(1)  // taps, lerp, zero-state run
    // ---- scan of the run-end states (scan basis), as in k_rlm_fast ----
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
#define RH_CSCAN(K, N)                                                                             \
    {                                                                                              \
        float sq[2 * C];                                                                           \
        _Pragma("unroll") for (int k = 0; k < 2 * C; ++k) sq[k] = dpp0<kDppRowShr + N, 0xf>(Pq[k]); \
        const float sM[4] = {readlane_f(U, 9 + 4 * K), readlane_f(U, 10 + 4 * K), readlane_f(U, 11 + 4 * K), readlane_f(U, 12 + 4 * K)}; \
        _Pragma("unroll") for (int ch = 0; ch < C; ++ch) mat_acc(sM, sq[2 * ch], sq[2 * ch + 1], Pq[2 * ch], Pq[2 * ch + 1]); \
    }
    RH_CSCAN(0, 1)
    RH_CSCAN(1, 2)
    RH_CSCAN(2, 4)
    RH_CSCAN(3, 8)
#undef RH_CSCAN
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
    {  // the tile aggregate: the short last run on top of the inclusive prefix of the lane before it
        const uint32_t n_t = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_t_v);
        const int nl = (int)((n_t + R - 1) / R);  // lanes with frames (uniform)
        float A[2 * C];
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
            __hip_atomic_store((unsigned long long *)(uintptr_t)a_gran_pub, ((unsigned long long)epoch_v << 32) | __float_as_uint(ev), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    float Q[2 * C];
#pragma unroll
    for (int k = 0; k < 2 * C; ++k) Q[k] = dpp0<kDppWaveShr1, 0xf>(Pq[k]);  // exclusive: the prefix of the lanes before (all of them whole runs)
    RH_CPH(2)  // scan, aggregate, publish
    // ---- the tile carry: lane j < J polls tile-1-j, weights it with B^(m_lo[tile] - m_lo[tile-j]) ----
    float c[2 * C];
#pragma unroll
    for (int k = 0; k < 2 * C; ++k) c[k] = 0.f;
    if (tile > 0) {
        const bool want = want_look != 0;
        const unsigned long long *gp = (const unsigned long long *)(uintptr_t)a_gran_poll;
        unsigned long long gv[2 * C];
#pragma unroll
        for (int k = 0; k < 2 * C; ++k) gv[k] = 0;
        bool ok = false;
        uint32_t spins = 0;
        while (true) {
            if (want && !ok) {
                bool all = true;
#pragma unroll
                for (int k = 0; k < 2 * C; ++k) {
                    gv[k] = __hip_atomic_load(gp + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    all = all && ((uint32_t)(gv[k] >> 32) == epoch_v);
                }
                ok = all;
            }
            if (__all(ok || !want)) break;
            if (++spins > kSpinLimit) {
                if (lane == 0) atomicOr(p.status, 1u);
                dead = true;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        if (lane == 0 && spins) atomicAdd(p.status + 1, spins);
        if (want && ok && !dead) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) mat_acc(kM, __uint_as_float((uint32_t)gv[2 * ch]), __uint_as_float((uint32_t)gv[2 * ch + 1]), c[2 * ch], c[2 * ch + 1]);
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
    if (dead) {  // a hand-off that never arrived: the status word fails the call, the tile is poisoned
#pragma unroll
        for (int k = 0; k < 2 * C; ++k) c[k] = __builtin_nanf("");
    }
    RH_CPH(3)  // look-back
#pragma unroll
    for (int ch = 0; ch < C; ++ch) mat_acc(lM, c[2 * ch], c[2 * ch + 1], Q[2 * ch], Q[2 * ch + 1]);  // start state of the lane's run = Q + B^(R*lane) * carry
    // A lane's run is R * FB contiguous bytes, so a store of one frame per lane touches 64 different lines.  The runs go through
    // the (now idle) ring stages -- rows padded by one frame -- and leave as whole lines: lane l stores frame k * 64 + l of the tile.
    constexpr uint32_t kRow = (R + 1) * FB;
    {
        lds_u8 *row = lds + (uint32_t)lane * kRow;
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
        const uint32_t n_t = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_t_v);
        float *ot = (float *)(uintptr_t)a_out;  // this lane's frame of every group of 64
        for (uint32_t f0 = 0; f0 < n_t; f0 += 64) {
            const uint32_t f = f0 + (uint32_t)lane;
            if (f < n_t) {
                const lds_u8 *src2 = lds + (f / R) * kRow + (f % R) * FB;
                if (C == 2) {
                    const v2f a = *(const lds_f2 *)src2;
                    *reinterpret_cast<float2 *>(ot + (uint64_t)f0 * 2) = make_float2(a.x, a.y);
                } else {
                    ot[f0] = *(const RH_LDS float *)src2;
                }
            }
        }
    }
    RH_CPH(4)  // correction + stores (issue)
}
template <int R, int C, int KV>
__global__ __launch_bounds__(64, 2) void k_rlm_chunk(const Params p, const ChunkArgs q) {
    rlm_chunk_tile<R, C, KV>(p, q, blockIdx.x);
}
// Several batches in ONE launch (the filter classes of a mixer, rh_rlm_set_filters): workgroups first[k] .. first[k+1]-1 are class k's launch of
// k_rlm_chunk, with that class's arguments -- its own sources, tables, hand-off words, ticket counters and output row.  Nothing crosses
// between classes; what the one launch saves is the drain and the ramp between launches that each want the whole chip (four classes of the
// benchmark batch: 0.368 ms as four launches).  The arguments are read where they lie, in the kernarg segment: a class index that is not a
// constant would otherwise make the compiler copy the whole block to scratch.
template <int R, int C, int KV>
__global__ __launch_bounds__(64, 2) void k_rlm_chunk_multi(const ChunkMulti m) {
    uint32_t k = 0;
#pragma unroll
    for (uint32_t i = 1; i < kChunkMultiMax; ++i) k += (i < m.n && blockIdx.x >= m.first[i]) ? 1u : 0u;
    uint32_t b0 = 0;
#pragma unroll
    for (uint32_t i = 1; i < kChunkMultiMax; ++i) b0 = (i == k) ? m.first[i] : b0;
    k = __builtin_amdgcn_readfirstlane(k);
    typedef const __attribute__((address_space(4))) unsigned char *cbytes;
    cbytes e = (cbytes)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ChunkMulti, e) + (size_t)k * sizeof(ChunkMulti::Entry);
    Params p;
    ChunkArgs q;
    typedef const __attribute__((address_space(4))) uint32_t *cwords;
    static_assert(sizeof(Params) % 4 == 0 && sizeof(ChunkArgs) % 4 == 0 && offsetof(ChunkMulti::Entry, q) % 4 == 0, "copied word by word");
    {
        cwords wp = (cwords)(e + offsetof(ChunkMulti::Entry, p)), wq = (cwords)(e + offsetof(ChunkMulti::Entry, q));
        uint32_t *dp = reinterpret_cast<uint32_t *>(&p), *dq = reinterpret_cast<uint32_t *>(&q);
#pragma unroll
        for (uint32_t i = 0; i < sizeof(Params) / 4; ++i) dp[i] = wp[i];  // (what the tile does not read is never loaded: the copy dissolves into scalar loads)
#pragma unroll
        for (uint32_t i = 0; i < sizeof(ChunkArgs) / 4; ++i) dq[i] = wq[i];
    }
    rlm_chunk_tile<R, C, KV>(p, q, blockIdx.x - b0);
}

// The classes of a mixer in one launch, TWO waves a tile, ONE output (round 6).  A class of 64 sources is a quarter of the headline's bytes, and
// every way of running four of them measured the same 0.36-0.37 ms against the headline's 0.31 for the same bytes: a launch per class,
// the launches lined up in one grid (k_rlm_chunk_multi), a wave that loads while another converts and filters.  The loaders alone take
// 0.308 ms; stamps per tile and class (tools/cls_stamps.py) showed where the rest goes: each class's mix was STORED -- 36 MB of writes in
// all, interleaved with 2 GiB of reads, took 40-55 us of the launch (a run's stores: 55 us a class while the loaders run, 1.5 us once they
// are done) -- and then read again by rh_mix_sum.  So: a workgroup owns chunk `tile` of EVERY class.  Wave 0 only ever loads: it sums the
// chunk over class k's sources through its ring -- the ring does not know about classes, the source after next goes out across a class's
// end -- and leaves the mixed chunk in one of two LDS images (two counters in the LDS, no barrier: it may be two classes ahead).  Wave 1
// converts and filters class k's chunk exactly as k_rlm_chunk's lone wave does (the same operations in the same order) and ADDS the result
// to a run of registers, 0.0 + class 0 + class 1 + ... -- rh_mix_sum's order, the same bits -- which leaves once, as whole lines, when the
// last class is done and the loaders have nothing left to read.  0.317 ms: the classes cost what the headline costs.  For classes of one
// geometry (equal lengths, one rate pair: one table of tile bounds) whose tiles are all resident at once (the host checks: tile =
// workgroup, no ticket); anything else takes k_rlm_chunk_multi and the classes' rows.
#if defined(RH_CLS_DIAG) && RH_CLS_DIAG == 9  // diagnostics builds: wall-clock stamps per (tile, class): loader {sum done, image written}, wave 1 {image seen, halo there, look-back there, done}
__device__ unsigned long long g_cls_stamp[2048 * 8 * 8];
#define RH_CLS_STAMP(i) { if (lane == 0 && tile < 2048 && k < 8) g_cls_stamp[((uint64_t)tile * 8 + k) * 8 + (i)] = wall_clock64(); }
#else
#define RH_CLS_STAMP(i)
#endif
#define RH_PARG(T, e, f) (*(const __attribute__((address_space(4))) T *)((e) + offsetof(ChunkMulti::Entry, p) + offsetof(Params, f)))
#define RH_QARG(T, e, f) (*(const __attribute__((address_space(4))) T *)((e) + offsetof(ChunkMulti::Entry, q) + offsetof(ChunkArgs, f)))
template <int R, int C, int KV>
__global__ __launch_bounds__(128, 2) void k_rlm_chunk_classes(const ChunkMulti m) {
    typedef Chan<C> CH;
    typedef typename CH::V V;
    constexpr int NS = 2, H = 4;
    constexpr uint32_t FB = CH::kFB;
    constexpr uint32_t kStage = KV * 1024, P = kStage / FB;
    // LDS: the loader's ring | two words the waves meet at | TWO images of a mixed chunk, [64 bytes: the 4 frames in front | the chunk | 64 bytes].
    // Two images and counters instead of barriers: the loader may be two classes ahead of wave 1 -- a tile's wave 1 waits for its NEIGHBOURS'
    // loaders (the frames in front of its chunk, their aggregates), and with one image and a barrier a class every loader was tied to within
    // a class of the slowest loader near it: measured 0.361 ms against 0.308 for the loaders alone.
    constexpr uint32_t kCtl = NS * kStage;
    constexpr uint32_t kImg = kStage + 128;
    constexpr uint32_t MB0 = kCtl + 64 + 64;         // the chunk of image 0 (its 4 frames in front at MB0 - H * FB)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kCtl + 64 + 2 * kImg];
    lds_u8 *const lds = (lds_u8 *)smem;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
    typedef __attribute__((address_space(4))) const uint64_t cu64;
    typedef __attribute__((address_space(4))) const float cf32;
    typedef __attribute__((address_space(4))) const uint32_t cu32;
    typedef const __attribute__((address_space(4))) unsigned char *cbytes;
    const cbytes kseg = (cbytes)__builtin_amdgcn_kernarg_segment_ptr();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t n_cls = m.n;
    const Params &p0 = m.e[0].p;  // the geometry the classes share
    const uint32_t tile = blockIdx.x;
    const uint32_t Ns = p0.eq_frames;
    auto entry = [&](uint32_t k) -> cbytes { return kseg + offsetof(ChunkMulti, e) + (size_t)k * sizeof(ChunkMulti::Entry); };
    // the two counters: classes handed over by the loader / finished by wave 1 (LDS words; a wave's LDS operations complete in order)
    auto ctl_read = [&](uint32_t which) -> uint32_t {
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds0 + kCtl + which * 4) : "memory");
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    };
    auto ctl_write = [&](uint32_t which, uint32_t v) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) *(RH_LDS uint32_t *)(lds + kCtl + which * 4) = v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    if (wave == 0) {
        if (lane < 2) *(RH_LDS uint32_t *)(lds + kCtl + lane * 4) = 0u;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (wave == 0) {
        // ---- the loader: the sum of chunk `tile` over every class's sources, class after class (k_rlm_chunk's source loop) ----
        const uint32_t nvec = Ns * C / 4;
        const uint32_t v0 = tile * (KV * 64);
        uint32_t goff[KV];
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            uint32_t j = v0 + (uint32_t)k * 64 + lane;
            j = j < nvec ? j : nvec - 1;
            goff[k] = j * 16;
        }
        const bool lin = v0 + (uint32_t)(KV * 64) <= nvec;
        auto stage_source = [&](cu64 *desc, uint32_t s_, uint32_t stage) {
            const void *data = (const void *)(uintptr_t)desc[4 * (uint64_t)s_];
            if (lin) {
                glds16_run<KV>(data, goff[0], lds0 + stage * kStage);
            } else {
#pragma unroll
                for (int k = 0; k < KV; ++k) glds16(data, goff[k], lds0 + stage * kStage + k * 1024);
            }
        };
        constexpr int NHV = H * FB / 16;
        // The ring does not know about classes: the source after next goes out as soon as a stage is free, across a class's end too (its last
        // two sources are summed while the next class's first two are on their way) -- only the hand-over of the mixed chunk happens per class.
        __builtin_amdgcn_s_setprio(3);  // (this wave's few instructions are the memory side's pace: in front of wave 1's arithmetic on the same SIMD)
        uint32_t kc = 0, sc = 0;        // the next source to request: source sc of class kc
        cu64 *ndesc = (cu64 *)(uintptr_t)RH_PARG(uint64_t, entry(0), srcs);
        uint32_t nS = RH_PARG(uint32_t, entry(0), n_sources);
        uint32_t nstage = 0;
        auto request_next = [&]() {  // (classes without sources are the host's to leave out)
            if (kc >= n_cls) return false;
            stage_source(ndesc, sc, nstage);
            nstage ^= 1u;
            if (++sc == nS) {
                sc = 0;
                if (++kc < n_cls) {
                    ndesc = (cu64 *)(uintptr_t)RH_PARG(uint64_t, entry(kc), srcs);
                    nS = RH_PARG(uint32_t, entry(kc), n_sources);
                }
            }
            return true;
        };
        uint32_t ahead = 0;  // groups of KV fetches in flight
        if (request_next()) ++ahead;
        if (request_next()) ++ahead;
        uint32_t st = 0;
        for (uint32_t k = 0; k < n_cls; ++k) {
            const cbytes e = entry(k);
            cf32 *const dgain = (cf32 *)(uintptr_t)RH_PARG(uint64_t, e, srcs);
            const uint32_t S = RH_PARG(uint32_t, e, n_sources);
            v4f acc[KV];
#pragma unroll
            for (int i = 0; i < KV; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
            float g_next = S ? dgain[4] : 0.f;
            for (uint32_t s_ = 0; s_ < S; ++s_) {
                const float g = g_next;
                g_next = s_ + 1 < S ? dgain[8 * (uint64_t)(s_ + 1) + 4] : 0.f;
                // this source has landed when at most the group behind it is outstanding (the few stores of a hand-over in between only make
                // the wait stricter)
                if (ahead > 1) wait_vm<KV>();
                else wait_vm<0>();
                --ahead;
                const lds_u8 *buf = lds + st * kStage;
                v4f v[KV];
#pragma unroll
                for (int i = 0; i < KV; ++i) v[i] = *(const lds_f4 *)(buf + i * 1024 + lane * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the chunk is in registers: its stage is free ...
                if (request_next()) ++ahead;                        // ... for the source after next, of this class or the next one
#pragma unroll
                for (int i = 0; i < KV; ++i) {
                    acc[i].x = fma_(g, v[i].x, acc[i].x);
                    acc[i].y = fma_(g, v[i].y, acc[i].y);
                    acc[i].z = fma_(g, v[i].z, acc[i].z);
                    acc[i].w = fma_(g, v[i].w, acc[i].w);
                }
                st ^= 1u;
            }
            RH_CLS_STAMP(0)
            // the chunk's last 4 mixed frames to the tile behind (first: its wave 1 is waiting for them) ...
            if (lane >= 64 - NHV) {
                unsigned long long *hp = (unsigned long long *)(uintptr_t)RH_QARG(uint64_t, e, halo) + (uint64_t)tile * 8 + (uint32_t)(lane - (64 - NHV)) * 4;
                const uint32_t epoch = RH_PARG(uint32_t, e, epoch);
                const float ev[4] = {acc[KV - 1].x, acc[KV - 1].y, acc[KV - 1].z, acc[KV - 1].w};
#pragma unroll
                for (int w = 0; w < 4; ++w) __hip_atomic_store(hp + w, ((unsigned long long)epoch << 32) | __float_as_uint(ev[w]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // ... and the mixed chunk into the image wave 1 is not reading: it has finished class k - 2 (it may still be on class k - 1)
            if (k >= 2) {
                uint32_t spins = 0;
                while (ctl_read(1) + 2 <= k) {
                    if (++spins > kSpinLimit) break;  // (wave 1 reports what it waits for)
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            {
                lds_u8 *img = lds + MB0 + (k & 1u) * kImg;
#pragma unroll
                for (int i = 0; i < KV; ++i) *(lds_f4 *)(img + (uint32_t)(i * 64 + lane) * 16) = acc[i];
                if (lane == 0) *(lds_f4 *)(img + kStage) = v4f{0.f, 0.f, 0.f, 0.f};
            }
            ctl_write(0, k + 1);
            RH_CLS_STAMP(1)
        }
        return;
    }
    // ---- wave 1: what k_rlm_chunk does behind its source loop, for every class in turn ----
    const ChunkArgs &q0 = m.e[0].q;
    const uint32_t m_lo = ((cu32 *)(uintptr_t)q0.m_lo)[tile], m_hi = ((cu32 *)(uintptr_t)q0.m_lo)[tile + 1];
    const uint32_t m0 = m_lo + (uint32_t)lane * R;
    const uint32_t n_t = m_hi - m_lo;  // <= 64 * R (host)
    const int nfl = (int)n_t - lane * R < 0 ? 0 : ((int)n_t - lane * R > R ? R : (int)n_t - lane * R);
    const uint32_t nl0 = (n_t + R - 1) / R;
    const uint32_t vlast = nl0 ? n_t - (nl0 - 1) * R : 0;  // frames of the last lane's run, 1 .. R
    const bool first = (m0 == 0);
    int offA[R + 2];
    float wgt[R + 2];
    {
        const int64_t fbase = (int64_t)tile * P;
        Cursor c = cursor_at(first ? 0 : m0 - 2, p0);
#pragma unroll
        for (int rr = 0; rr < R + 2; ++rr) {
            const bool dummy = first && rr < 2;
            uint64_t i;
            uint32_t num;
            cursor_resolve(c, p0, i, num);
            if (i + 1 >= Ns) {
                i = Ns - 1;
                num = 0;
            }
            int64_t f = (int64_t)i - fbase;
            f = f < -H ? -H : (f > (int64_t)P - 1 ? (int64_t)P - 1 : f);
            offA[rr] = dummy ? 0 : (int)f * (int)FB;  // (relative to the chunk in its image)
            wgt[rr] = dummy ? 0.0f : (float)num / p0.Tf;
            if (!dummy) cursor_next(c, p0);
        }
    }
    // The classes' mixes are ADDED here, in registers, in the classes' order (0.0 + class 0 + class 1 + ...: rh_mix_sum's order over the classes'
    // rows, bit for bit) and leave once, behind the last class -- when the loaders are done.  Measured with a row per class (tools/cls_stamps.py):
    // 36 MB of output stores interleaved with the 2 GiB of reads cost 40-55 us of a 310 us launch; written at the end, like k_rlm_chunk's, nothing.
    V ytot[R];
#pragma unroll
    for (int r = 0; r < R; ++r) ytot[r] = CH::zero();
    for (uint32_t k = 0; k < n_cls; ++k) {
        const cbytes e = entry(k);
        // the class's tables and addresses: fetched here, a class's loads away from their use
        const Tables *__restrict__ tb = (const Tables *)(uintptr_t)RH_PARG(uint64_t, e, tabs);
        const uint32_t J = RH_PARG(uint32_t, e, J), epoch = RH_PARG(uint32_t, e, epoch);
        const uint32_t Jc = J < tile ? J : tile;
        const bool want_look = (uint32_t)lane < Jc;
        unsigned long long *const halo = (unsigned long long *)(uintptr_t)RH_QARG(uint64_t, e, halo);
        unsigned long long *const gran = (unsigned long long *)(uintptr_t)RH_PARG(uint64_t, e, gran);
        uint32_t *const status = (uint32_t *)(uintptr_t)RH_PARG(uint64_t, e, status);
        const float U = ((const float *)(uintptr_t)RH_QARG(uint64_t, e, uni))[lane < 61 ? lane : 60];
        float lM[4], b15[4], b31[4], kM[4], pwv[4];
        {
            const float *kp = (const float *)(uintptr_t)RH_QARG(uint64_t, e, lookT) + ((uint64_t)tile * J + (want_look ? lane : 0)) * 4;
            const float *pw = (const float *)(uintptr_t)RH_QARG(uint64_t, e, powM) + 4 * vlast;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lM[i] = tb->laneM[lane][i];
                b15[i] = tb->bc15M[lane][i];
                b31[i] = tb->bc31M[lane][i];
                kM[i] = kp[i];
                pwv[i] = pw[i];
            }
        }
        lds_u8 *const img = lds + MB0 + (k & 1u) * kImg;  // the class's mixed chunk, when the loader says so
        {
            uint32_t spins = 0;
            while (ctl_read(0) <= k) {
                if (++spins > kSpinLimit) {
                    if (lane == 0) atomicOr(status, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
#if defined(RH_CLS_DIAG) && RH_CLS_DIAG == 1  // diagnostics builds (wrong results): the loader's pace with nothing behind the hand-over
        CH::set(ytot[0], 0, CH::get(ytot[0], 0) + lM[0] + b15[0] + b31[0] + kM[0] + pwv[0] + U + (float)offA[0] + wgt[1] + (float)nfl + (want_look ? 1.f : 0.f) + (float)(uintptr_t)halo + (float)(uintptr_t)gran + (float)(uintptr_t)status + (float)epoch);
        ctl_write(1, k + 1);
        continue;
#endif
        RH_CLS_STAMP(2)
        // ---- the last 4 frames of the tile in front ----
        bool dead = false;
        if (tile == 0) {
            if (lane < H * C) *(RH_LDS float *)(img - H * FB + lane * 4) = 0.0f;
        } else {
            const bool want = lane < H * C;
            const unsigned long long *hp = halo + (uint64_t)(tile - 1) * 8 + (want ? lane : 0);
            unsigned long long hv = 0;
            bool ok = false;
            uint32_t spins = 0;
            while (true) {
                if (want && !ok) {
                    hv = __hip_atomic_load(hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (uint32_t)(hv >> 32) == epoch;
                }
                if (__all(ok || !want)) break;
                if (++spins > kSpinLimit) {
                    if (lane == 0) atomicOr(status, 1u);
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (want) *(RH_LDS float *)(img - H * FB + lane * 4) = dead ? __builtin_nanf("") : __uint_as_float((uint32_t)hv);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        RH_CLS_STAMP(3)
        // ---- the lane's run of the mixed stream: lerp, zero-state biquad; the run-end state after nfl frames ----
        const float b0 = readlane_f(U, 0), c1 = readlane_f(U, 1), c2 = readlane_f(U, 2), na1 = -readlane_f(U, 3), na2 = -readlane_f(U, 4);
        V out[R];
        V E1 = CH::zero(), E2 = CH::zero();
        {
            V ta[R + 2], tb2[R + 2];
#pragma unroll
            for (int rr = 0; rr < R + 2; ++rr) {
                ta[rr] = CH::ld_lds(img + offA[rr]);
                tb2[rr] = CH::ld_lds(img + offA[rr] + FB);
            }
            auto tap = [&](int rr) -> V {
                V x;
#pragma unroll
                for (int ch = 0; ch < C; ++ch) CH::set(x, ch, fma_(CH::get(tb2[rr], ch) - CH::get(ta[rr], ch), wgt[rr], CH::get(ta[rr], ch)));
                return x;
            };
            V x2 = first ? CH::zero() : tap(0);
            V x1 = first ? CH::zero() : tap(1);
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
        for (int i = 0; i < 2 * C; ++i) Pq[i] = 0.f;
        {
            const float Tm[4] = {readlane_f(U, 5), readlane_f(U, 6), readlane_f(U, 7), readlane_f(U, 8)};
#pragma unroll
            for (int ch = 0; ch < C; ++ch) mat_acc(Tm, CH::get(E1, ch), CH::get(E2, ch), Pq[2 * ch], Pq[2 * ch + 1]);
        }
        float own[2 * C];
#pragma unroll
        for (int i = 0; i < 2 * C; ++i) own[i] = Pq[i];
#define RH_CSCAN(K, N)                                                                             \
    {                                                                                              \
        float sq[2 * C];                                                                           \
        _Pragma("unroll") for (int i = 0; i < 2 * C; ++i) sq[i] = dpp0<kDppRowShr + N, 0xf>(Pq[i]); \
        const float sM[4] = {readlane_f(U, 9 + 4 * K), readlane_f(U, 10 + 4 * K), readlane_f(U, 11 + 4 * K), readlane_f(U, 12 + 4 * K)}; \
        _Pragma("unroll") for (int ch = 0; ch < C; ++ch) mat_acc(sM, sq[2 * ch], sq[2 * ch + 1], Pq[2 * ch], Pq[2 * ch + 1]); \
    }
        RH_CSCAN(0, 1)
        RH_CSCAN(1, 2)
        RH_CSCAN(2, 4)
        RH_CSCAN(3, 8)
#undef RH_CSCAN
        {
            float sq[2 * C];
#pragma unroll
            for (int i = 0; i < 2 * C; ++i) sq[i] = dpp0<kDppBcast15, 0xa>(Pq[i]);
#pragma unroll
            for (int ch = 0; ch < C; ++ch) mat_acc(b15, sq[2 * ch], sq[2 * ch + 1], Pq[2 * ch], Pq[2 * ch + 1]);
        }
        {
            float sq[2 * C];
#pragma unroll
            for (int i = 0; i < 2 * C; ++i) sq[i] = dpp0<kDppBcast31, 0xc>(Pq[i]);
#pragma unroll
            for (int ch = 0; ch < C; ++ch) mat_acc(b31, sq[2 * ch], sq[2 * ch + 1], Pq[2 * ch], Pq[2 * ch + 1]);
        }
        {  // the tile aggregate: the short last run on top of the inclusive prefix of the lane before it
            const int nl = (int)nl0;
            float A[2 * C];
#pragma unroll
            for (int i = 0; i < 2 * C; ++i) A[i] = 0.f;
            if (nl >= 1) {
#pragma unroll
                for (int i = 0; i < 2 * C; ++i) A[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(own[i]), nl - 1));
            }
            if (nl >= 2) {
                const float M[4] = {readfirstlane_f(pwv[0]), readfirstlane_f(pwv[1]), readfirstlane_f(pwv[2]), readfirstlane_f(pwv[3])};  // B^v
                float xp[2 * C];
#pragma unroll
                for (int i = 0; i < 2 * C; ++i) xp[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Pq[i]), nl - 2));
#pragma unroll
                for (int ch = 0; ch < C; ++ch) mat_acc(M, xp[2 * ch], xp[2 * ch + 1], A[2 * ch], A[2 * ch + 1]);
            }
            if (lane < 2 * C) {
                float ev = A[0];
#pragma unroll
                for (int i = 1; i < 2 * C; ++i) ev = lane == i ? A[i] : ev;
                __hip_atomic_store(gran + (uint64_t)tile * 4 + (uint32_t)lane, ((unsigned long long)epoch << 32) | __float_as_uint(ev), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        float Q[2 * C];
#pragma unroll
        for (int i = 0; i < 2 * C; ++i) Q[i] = dpp0<kDppWaveShr1, 0xf>(Pq[i]);
        // ---- the tile carry: lane j < J polls tile-1-j, weights it with B^(m_lo[tile] - m_lo[tile-j]) ----
        float c[2 * C];
#pragma unroll
        for (int i = 0; i < 2 * C; ++i) c[i] = 0.f;
        if (tile > 0) {
            const unsigned long long *gp = gran + (uint64_t)(tile - 1 - (want_look ? (uint32_t)lane : 0u)) * 4;
            unsigned long long gv[2 * C];
#pragma unroll
            for (int i = 0; i < 2 * C; ++i) gv[i] = 0;
            bool ok = false;
            uint32_t spins = 0;
            while (true) {
                if (want_look && !ok) {
                    bool all = true;
#pragma unroll
                    for (int i = 0; i < 2 * C; ++i) {
                        gv[i] = __hip_atomic_load(gp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        all = all && ((uint32_t)(gv[i] >> 32) == epoch);
                    }
                    ok = all;
                }
                if (__all(ok || !want_look)) break;
                if (++spins > kSpinLimit) {
                    if (lane == 0) atomicOr(status, 1u);
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            if (lane == 0 && spins) atomicAdd(status + 1, spins);
            if (want_look && ok && !dead) {
#pragma unroll
                for (int ch = 0; ch < C; ++ch) mat_acc(kM, __uint_as_float((uint32_t)gv[2 * ch]), __uint_as_float((uint32_t)gv[2 * ch + 1]), c[2 * ch], c[2 * ch + 1]);
            }
#pragma unroll
            for (int i = 0; i < 2 * C; ++i) {  // sum over lanes 0..31 -> uniform
                c[i] += dpp0<kDppRowShr + 1, 0xf>(c[i]);
                c[i] += dpp0<kDppRowShr + 2, 0xf>(c[i]);
                c[i] += dpp0<kDppRowShr + 4, 0xf>(c[i]);
                c[i] += dpp0<kDppRowShr + 8, 0xf>(c[i]);
                c[i] = readlane_f(c[i], 15) + readlane_f(c[i], 31);
            }
        }
        if (dead) {
#pragma unroll
            for (int i = 0; i < 2 * C; ++i) c[i] = __builtin_nanf("");
        }
#pragma unroll
        for (int ch = 0; ch < C; ++ch) mat_acc(lM, c[2 * ch], c[2 * ch + 1], Q[2 * ch], Q[2 * ch + 1]);
        RH_CLS_STAMP(4)
        // the taps are in registers: the image is the loader's again
        ctl_write(1, k + 1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                const float y = fma_(readlane_f(U, 25 + 2 * r), Q[2 * ch], fma_(readlane_f(U, 26 + 2 * r), Q[2 * ch + 1], CH::get(out[r], ch)));
                CH::set(ytot[r], ch, CH::get(ytot[r], ch) + y);
            }
        }
        RH_CLS_STAMP(5)
    }
    // ---- the sum leaves as whole lines through the ring (the loader has summed its last chunk: nothing is in flight into it) ----
    {
        constexpr uint32_t kRow = (R + 1) * FB;
        static_assert(64 * kRow <= kCtl + 64 + 2 * kImg, "the rows fit the LDS (the ring, and behind it the images: everything is free now)");
        lds_u8 *row = lds + (uint32_t)lane * kRow;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (C == 2) *(lds_f2 *)(row + r * FB) = v2f{CH::get(ytot[r], 0), CH::get(ytot[r], C - 1)};
            else *(RH_LDS float *)(row + r * FB) = CH::get(ytot[r], 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        float *ot = m.sum_out + ((uint64_t)m_lo + (uint32_t)lane) * C;  // this lane's frame of every group of 64
        for (uint32_t f0 = 0; f0 < n_t; f0 += 64) {
            const uint32_t f = f0 + (uint32_t)lane;
            if (f < n_t) {
                const lds_u8 *src2 = lds + (f / R) * kRow + (f % R) * FB;
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
#undef RH_PARG
#undef RH_QARG

// =================================================================================================
// The (tile, source) pairs of a ragged filtered batch in which the source is NOT stable, i.e. ends inside the tile or within the J
// tiles after it.  There are at most J+2 such tiles per source, so this part is small however large the batch: a tile finds its
// pairs with a ballot over the descriptors (rag_find_pairs) and handles each one start to finish (rag_run_pairs) -- stage, lerp,
// zero-state run (masked past the end of the source), scan, publish the source's own aggregate, poll the predecessors that hold
// one, correct frame by frame -- adding onto the tile's mix of the stable sources.  Tile geometry, tables and aggregate rows are
// those of k_rlm_wave.  Two callers: k_rlm_fast<RAG, SUMF> itself, behind its own part of the tile (Params::rag_merge: the mix is
// still in registers, and the pairs fill in behind the stable sources of the lighter tiles); k_rlm_resid, a launch of its own
// behind a first half that has stored its tiles (RH_RAG_TWO_KERNELS, and the per-source first half of diagnostics builds).
// =================================================================================================
// The tile's pairs: their source indices, in source order, as a list in the LDS at `list_off` (the host keeps batches with more than
// 24 such sources in one tile on k_rlm_wave).  Returns how many.
constexpr uint32_t kMaxPairs = 32;
__device__ __forceinline__ uint32_t rag_find_pairs(const Params &p, const uint32_t m_tile0, const uint32_t m_stable, const int lane, lds_u8 *lds, const uint32_t list_off) {
    const uint32_t S = p.n_sources;
    const uint32_t *const dsrc = reinterpret_cast<const uint32_t *>(p.srcs);
    uint32_t n_pairs = 0;
    for (uint32_t b = 0; b < S; b += 64) {
        const uint32_t q = b + lane;
        const uint32_t ms = q < S ? dsrc[8 * (uint64_t)q + 3] : 0u;
        const bool mine = ms > m_tile0 && ms < m_stable;  // still here, not stable
        const unsigned long long mask = __ballot(mine);
        const uint32_t rank = n_pairs + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (mine && rank < kMaxPairs) *(RH_LDS uint32_t *)(lds + list_off + rank * 4u) = q;
        n_pairs += (uint32_t)__popcll(mask);
    }
    if (n_pairs > kMaxPairs && lane == 0) atomicOr(p.status, 1u);  // (never: the host counted them.  The status word fails the call)
    n_pairs = __builtin_amdgcn_readfirstlane(n_pairs < kMaxPairs ? n_pairs : kMaxPairs);
    if (n_pairs) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    return n_pairs;
}
// The pairs one after the other, the span of the NEXT one already on its way (two stages of KV KiB at lds + 0 and lds + KV KiB): a
// pair costs a memory round trip to stage and another to poll, and nothing else of this tile could hide them.  `rows`: the
// per-source aggregate rows.  offA / wgt: taps and weights of the lane's R + 2 frames (the caller's: the tile geometry is shared).
template <int R, int KV, int C>
__device__ __forceinline__ void rag_run_pairs(const Params &p, unsigned long long *const rows, const uint32_t tile, const uint32_t n_pairs, const int lane, lds_u8 *lds, const uint32_t lds0,
                                              const uint32_t list_off, const uint32_t i_base, const uint32_t nvec, const uint32_t (&goff)[KV], const int (&offA)[R + 2], const float (&wgt)[R + 2],
                                              const float (&lM)[4], const float (&b15)[4], const float (&b31)[4], const float (&kM)[4], typename Chan<C>::V (&acc)[R]) {
    typedef Chan<C> CH;
    typedef typename CH::V V;
    constexpr uint32_t FB = CH::kFB, VF = CH::kVF;
    constexpr uint32_t L = 64u * R, kStage = KV * 1024;
    const uint32_t m_tile0 = tile * L, m0 = m_tile0 + (uint32_t)lane * R;
    const bool first = (m0 == 0);
    const uint32_t ncol = p.n_tiles;
    const uint32_t Jc = p.J < tile ? p.J : tile;
    typedef __attribute__((address_space(4))) const uint32_t cu32;
    cu32 *const desc = (cu32 *)(uintptr_t)p.srcs;
    const float b0 = p.u.b0, c1 = p.u.c1, c2 = p.u.c2, na1 = -p.u.a1, na2 = -p.u.a2;
    bool dead = false;
    auto stage_pair = [&](uint32_t s, uint32_t stage_off) {  // (lanes past the source's end re-fetch its last vector: finite data nothing valid reads)
        const uint64_t plo = desc[8 * (uint64_t)s], phi = desc[8 * (uint64_t)s + 1];
        const void *data = (const void *)(uintptr_t)(plo | (phi << 32));
        const uint32_t Ns = desc[8 * (uint64_t)s + 2];
        if (i_base + VF * nvec <= Ns) {
#pragma unroll
            for (int k = 0; k < KV; ++k) glds16(data, goff[k], lds0 + stage_off + k * 1024);
        } else {
            const uint32_t lastoff = ((Ns - 1) & ~(VF - 1u)) * FB;
#pragma unroll
            for (int k = 0; k < KV; ++k) glds16(data, goff[k] < lastoff ? goff[k] : lastoff, lds0 + stage_off + k * 1024);
        }
    };
    auto pair_at = [&](uint32_t i) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const RH_LDS uint32_t *)(lds + list_off + i * 4u)); };
    stage_pair(pair_at(0), 0);
    for (uint32_t i = 0; i < n_pairs; ++i) {
        const uint32_t s = pair_at(i);
        const lds_u8 *const buf = lds + (i & 1u) * kStage;
        if (i + 1 < n_pairs) {
            stage_pair(pair_at(i + 1), ((i + 1) & 1u) * kStage);
            wait_vm<KV>();  // this pair's span has landed when only the next one's is outstanding
        } else {
            wait_vm<0>();
        }
        const uint32_t Ms = desc[8 * (uint64_t)s + 3];
        const uint32_t Ns = desc[8 * (uint64_t)s + 2];
        const float g = __uint_as_float(desc[8 * (uint64_t)s + 4]);
        const uint32_t dthr = Ns - 1 - i_base;  // Ms > m_tile0 => i_base <= Ns-1
        const int thr = (int)((dthr < (1u << 27) ? dthr : (1u << 27)) * FB);
        const int nvalid = Ms >= m0 + R ? R : (Ms > m0 ? (int)(Ms - m0) : 0);
        auto tap = [&](int rr) -> V {
            const V a = CH::ld_lds(buf + offA[rr]), b = CH::ld_lds(buf + offA[rr] + FB);
            const bool last = offA[rr] >= thr;  // the source's last frame is emitted verbatim (sample_rate.rs:193-200)
            V x;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                const float ac = CH::get(a, ch), bc = CH::get(b, ch);
                const float l = fma_(bc - ac, wgt[rr], ac);
                CH::set(x, ch, last ? ac : l);
            }
            return x;
        };
        V x2 = first ? CH::zero() : tap(0);
        V x1 = first ? CH::zero() : tap(1);
        V w1 = CH::zero(), w2 = CH::zero();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const V x = tap(r + 2);
            V w;
            const bool v = r < nvalid;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                const float wc = fma_(na1, CH::get(w1, ch), fma_(na2, CH::get(w2, ch), fma_(c2, CH::get(x2, ch), c1 * CH::get(x1, ch))));
                CH::set(w, ch, wc);
                const float y = v ? fma_(b0, CH::get(x, ch), wc) : 0.0f;
                CH::set(acc[r], ch, fma_(g, y, CH::get(acc[r], ch)));
            }
            w2 = w1;
            w1 = w;
            x2 = x1;
            x1 = x;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the stage is free for the pair after next
        float P[4] = {0.f, 0.f, 0.f, 0.f};  // (mono: the second channel's words stay zero and travel as zeros, as in k_rlm_wave)
        mat_acc(p.u.Tm, CH::get(w1, 0) * g, CH::get(w2, 0) * g, P[0], P[1]);
        if (C == 2) mat_acc(p.u.Tm, CH::get(w1, C - 1) * g, CH::get(w2, C - 1) * g, P[2], P[3]);
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
        {
            const float q0 = dpp0<kDppBcast15, 0xa>(P[0]), q1 = dpp0<kDppBcast15, 0xa>(P[1]);
            const float q2 = dpp0<kDppBcast15, 0xa>(P[2]), q3 = dpp0<kDppBcast15, 0xa>(P[3]);
            mat_acc(b15, q0, q1, P[0], P[1]);
            mat_acc(b15, q2, q3, P[2], P[3]);
        }
        {
            const float q0 = dpp0<kDppBcast31, 0xc>(P[0]), q1 = dpp0<kDppBcast31, 0xc>(P[1]);
            const float q2 = dpp0<kDppBcast31, 0xc>(P[2]), q3 = dpp0<kDppBcast31, 0xc>(P[3]);
            mat_acc(b31, q0, q1, P[0], P[1]);
            mat_acc(b31, q2, q3, P[2], P[3]);
        }
        unsigned long long *const row = rows + (uint64_t)s * ncol * 4;
        {  // this source's own aggregate for the tile (a later tile of it polls for it)
            const float e0 = readlane_f(P[0], 63), e1 = readlane_f(P[1], 63);
            const float e2 = readlane_f(P[2], 63), e3 = readlane_f(P[3], 63);
            if (lane < 4) {
                const float ev = lane == 0 ? e0 : lane == 1 ? e1 : lane == 2 ? e2 : e3;
                const unsigned long long word = ((unsigned long long)p.epoch << 32) | __float_as_uint(ev);
                __hip_atomic_store(row + (uint64_t)tile * 4 + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        float Q[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) Q[q] = dpp0<kDppWaveShr1, 0xf>(P[q]);  // exclusive: lane 0 gets 0
        // predecessors in which the source was already on its own (in the others it was stable: its state came with the sums)
        const uint32_t tM = Ms / L;
        uint32_t have = tile + p.J > tM ? tile + p.J - tM : 0u;
        have = have < Jc ? have : Jc;
        float c[4] = {0.f, 0.f, 0.f, 0.f};
        if (have > 0) {
            const bool want = (uint32_t)lane < have;
            const unsigned long long *gp = row + (uint64_t)(tile - 1 - (want ? lane : 0)) * 4;
            unsigned long long gv[4] = {0, 0, 0, 0};
            bool ok = false;
            uint32_t spins = 0;
            while (!dead) {
                if (want && !ok) {
                    bool all = true;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        gv[q] = __hip_atomic_load(gp + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        all = all && ((uint32_t)(gv[q] >> 32) == p.epoch);
                    }
                    ok = all;
                }
                if (__all(ok || !want)) break;
                if (++spins > kSpinLimit) {
                    if (lane == 0) atomicOr(p.status, 1u);
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            if (want && ok && !dead) {
                mat_acc(kM, __uint_as_float((uint32_t)gv[0]), __uint_as_float((uint32_t)gv[1]), c[0], c[1]);
                mat_acc(kM, __uint_as_float((uint32_t)gv[2]), __uint_as_float((uint32_t)gv[3]), c[2], c[3]);
            }
            if (dead) c[0] = c[1] = c[2] = c[3] = __builtin_nanf("");  // poison: see k_rlm_fast
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // sum over lanes 0..31 -> uniform
                c[q] += dpp0<kDppRowShr + 1, 0xf>(c[q]);
                c[q] += dpp0<kDppRowShr + 2, 0xf>(c[q]);
                c[q] += dpp0<kDppRowShr + 4, 0xf>(c[q]);
                c[q] += dpp0<kDppRowShr + 8, 0xf>(c[q]);
                c[q] = readlane_f(c[q], 15) + readlane_f(c[q], 31);
            }
        }
        mat_acc(lM, c[0], c[1], Q[0], Q[1]);
        mat_acc(lM, c[2], c[3], Q[2], Q[3]);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool v = r < nvalid;
            const float hx = fma_(p.u.g[r][0], Q[0], p.u.g[r][1] * Q[1]), hy = fma_(p.u.g[r][0], Q[2], p.u.g[r][1] * Q[3]);
            CH::set(acc[r], 0, CH::get(acc[r], 0) + (v ? hx : 0.0f));
            if (C == 2) CH::set(acc[r], C - 1, CH::get(acc[r], C - 1) + (v ? hy : 0.0f));
        }
    }
}

// k_rlm_resid -- the pairs as a launch of their own, on top of what a first half has stored.
template <int R, int KV>
__global__ __launch_bounds__(64) void k_rlm_resid(const Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_u8 *const lds = (lds_u8 *)smem;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
    const int lane = threadIdx.x;
    // tiles are numbered in arrival order, like everywhere else: a pair polls predecessor tiles, which then hold earlier tickets
    const uint32_t tile = __builtin_amdgcn_readfirstlane(lane == 0 ? atomicAdd(p.ticket, 1u) - p.ticket_base : 0u);
    constexpr uint32_t L = 64u * R;
    const uint32_t m_tile0 = tile * L;
    const uint32_t m0 = m_tile0 + (uint32_t)lane * R;
    const bool first = (m0 == 0);
    const uint32_t Mout = (uint32_t)p.out_frames;
    const uint32_t m_far = m_tile0 + (p.J + 1u) * L;
    const uint32_t m_stable = m_far < Mout ? m_far : Mout;  // sources that last as long as the mix are the first half's to the end
    constexpr uint32_t kList = 2 * KV * 1024;
    const uint32_t n_pairs = rag_find_pairs(p, m_tile0, m_stable, lane, lds, kList);
    if (!n_pairs) return;  // what k_rlm_fast<RAG> stored for this tile is final

    uint32_t i_base, nvec;
    {
        uint64_t ib, ie;
        uint32_t nn;
        cursor_resolve(cursor_at(m_tile0 >= 2 ? m_tile0 - 2 : 0, p), p, ib, nn);
        ib &= ~15ull;
        cursor_resolve(cursor_at((uint64_t)m_tile0 + L - 1, p), p, ie, nn);
        ie += 1;
        uint32_t nv = (uint32_t)((ie - ib + 2) / 2);
        if (nv > (uint32_t)(KV * 64)) nv = KV * 64;
        i_base = __builtin_amdgcn_readfirstlane((uint32_t)ib);
        nvec = __builtin_amdgcn_readfirstlane(nv);
    }
    uint32_t goff[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        uint32_t j = lane + k * 64;
        j = j < nvec ? j : nvec - 1;
        goff[k] = (i_base + 2u * j) * 8u;
    }
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
            offA[rr] = dummy ? 0 : (int)(((uint32_t)i - i_base) * 8u);
            wgt[rr] = dummy ? 0.0f : (float)num / p.Tf;
            if (!dummy) cursor_next(c, p);
        }
    }
    const Tables *__restrict__ tb = p.tabs;
    float lM[4], b15[4], b31[4], kM[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        lM[q] = tb->laneM[lane][q];
        b15[q] = tb->bc15M[lane][q];
        b31[q] = tb->bc31M[lane][q];
        kM[q] = tb->lookM[lane & (kMaxLook - 1)][q];
    }
    v2f acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        acc[r] = v2f{0.0f, 0.0f};
        if (m0 + r < Mout) {
            const float2 v = *reinterpret_cast<const float2 *>(p.out + (uint64_t)(m0 + r) * 2);
            acc[r] = v2f{v.x, v.y};
        }
    }
    rag_run_pairs<R, KV, 2>(p, p.gran, tile, n_pairs, lane, lds, lds0, kList, i_base, nvec, goff, offA, wgt, lM, b15, b31, kM, acc);
    float *o = p.out + (uint64_t)m0 * 2;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (m0 + r < Mout) *reinterpret_cast<float2 *>(o + r * 2) = make_float2(acc[r].x, acc[r].y);
}

// Block streaming with per-source filter states (k_rlm_wave): after a block of n_tiles whole tiles, the state of
// source s at the block's end is what tile n_tiles would have received as its carry -- the look-back over the
// aggregates the block has just published (and, for short blocks, the previous block-start state in column 0).
// One lane per source; the result becomes column 0 of the next launch (tagged with ITS epoch).  col_next = 0
// writes the zero state a stream starts from.
__global__ __launch_bounds__(64) void k_rlm_state(unsigned long long *gran, const Tables *__restrict__ tabs, uint32_t n_sources, uint32_t ncol, uint32_t col_next, uint32_t J,
                                                  uint32_t epoch, uint32_t next_epoch) {
    const uint32_t s = blockIdx.x * 64u + threadIdx.x;
    if (s >= n_sources) return;
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    const uint32_t Jc = J < col_next ? J : col_next;
    unsigned long long *row = gran + (uint64_t)s * ncol * 4;
    for (uint32_t j = 0; j < Jc; ++j) {
        const unsigned long long *g = row + (uint64_t)(col_next - 1 - j) * 4;
        const unsigned long long g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
        // a source that had ended before that tile published nothing there: it has no state to carry either
        if ((uint32_t)(g0 >> 32) != epoch || (uint32_t)(g1 >> 32) != epoch || (uint32_t)(g2 >> 32) != epoch || (uint32_t)(g3 >> 32) != epoch) continue;
        const float kM[4] = {tabs->lookM[j][0], tabs->lookM[j][1], tabs->lookM[j][2], tabs->lookM[j][3]};
        mat_acc(kM, __uint_as_float((uint32_t)g0), __uint_as_float((uint32_t)g1), c[0], c[1]);
        mat_acc(kM, __uint_as_float((uint32_t)g2), __uint_as_float((uint32_t)g3), c[2], c[3]);
    }
    for (int q = 0; q < 4; ++q) row[q] = ((unsigned long long)next_epoch << 32) | __float_as_uint(c[q]);
}

// A stream with a state per source whose live sources run together again (the others have ended and given everything): the summed
// state the fused kernel streams on is the sum of the live sources' states -- column 0 of their rows, which k_rlm_state has just written
// (tagged `tag`).  A source the table marks with gain 0 is gone (or mute: its state is zero then) and stays out.  One workgroup.
__global__ __launch_bounds__(256) void k_rlm_state_sum(const unsigned long long *__restrict__ gran, const SrcDesc *__restrict__ srcs, uint32_t n_sources, uint32_t ncol, uint32_t tag,
                                                      float *__restrict__ w_out) {
    __shared__ float part[4][256];
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    for (uint32_t s = threadIdx.x; s < n_sources; s += 256u) {
        if (srcs[s].gain == 0.0f) continue;
        const unsigned long long *row = gran + (uint64_t)s * ncol * 4;
        const unsigned long long g0 = row[0], g1 = row[1], g2 = row[2], g3 = row[3];
        if ((uint32_t)(g0 >> 32) != tag || (uint32_t)(g1 >> 32) != tag || (uint32_t)(g2 >> 32) != tag || (uint32_t)(g3 >> 32) != tag) continue;
        c[0] += __uint_as_float((uint32_t)g0), c[1] += __uint_as_float((uint32_t)g1), c[2] += __uint_as_float((uint32_t)g2), c[3] += __uint_as_float((uint32_t)g3);
    }
    for (int q = 0; q < 4; ++q) part[q][threadIdx.x] = c[q];
    __syncthreads();
    for (uint32_t w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w)
            for (int q = 0; q < 4; ++q) part[q][threadIdx.x] += part[q][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < 4) w_out[threadIdx.x] = part[threadIdx.x][0];
}

// ------------------------------------------------------------------ the instances ----
#define RH_FAST(r, kv, ns) Variant{r, kv, ns, &k_rlm_fast<r, kv, ns, true>, &k_rlm_fast<r, kv, ns, false>}
#define RH_WAVE(r, kv, ns) Variant{r, kv, ns, &k_rlm_wave<r, kv, ns, true>, &k_rlm_wave<r, kv, ns, false>}
// KV KiB of LDS per stage must hold the input span of 64*R output frames: ~R/2 vectors per lane
// when upsampling (from <= to), up to R+1 for from <= 2*to.  The host picks the smallest KV that fits.
// Per R: the stage sizes for from/to <= ~0.93 (44.1->48 k), <= 1 and <= 2; R = 3, 4, 6 also for from/to <= 4.5 (192 -> 44.1 k).
const Variant kFast[] = {
#ifdef RH_DEV_VARIANTS  // quick development builds
    RH_FAST(4, 2, 2), RH_FAST(4, 2, 3), RH_FAST(5, 3, 2), RH_FAST(6, 3, 2), RH_FAST(6, 3, 3), RH_FAST(8, 4, 2), RH_FAST(8, 4, 3), RH_FAST(9, 5, 2), RH_FAST(9, 5, 3), RH_FAST(12, 6, 3),
#else
    RH_FAST(3, 2, 2),   RH_FAST(3, 2, 3),   RH_FAST(3, 4, 2),  RH_FAST(3, 7, 2),
    RH_FAST(4, 2, 2),   RH_FAST(4, 2, 3),   RH_FAST(4, 3, 2),  RH_FAST(4, 3, 3),  RH_FAST(4, 5, 2),  RH_FAST(4, 9, 2),
    RH_FAST(5, 3, 2),   RH_FAST(5, 3, 3),   RH_FAST(5, 6, 2),
    RH_FAST(6, 3, 2),   RH_FAST(6, 3, 3),   RH_FAST(6, 4, 2),  RH_FAST(6, 4, 3),  RH_FAST(6, 7, 2),  RH_FAST(6, 14, 2),
    RH_FAST(7, 4, 2),   RH_FAST(7, 4, 3),   RH_FAST(7, 8, 2),
    RH_FAST(8, 4, 2),   RH_FAST(8, 4, 3),   RH_FAST(8, 4, 4),  RH_FAST(8, 5, 2),  RH_FAST(8, 5, 3),  RH_FAST(8, 9, 2),
    RH_FAST(9, 5, 2),   RH_FAST(9, 5, 3),   RH_FAST(9, 10, 2),
    RH_FAST(10, 5, 2),  RH_FAST(10, 5, 3),  RH_FAST(10, 6, 2), RH_FAST(10, 6, 3), RH_FAST(10, 11, 2),
    RH_FAST(12, 6, 2),  RH_FAST(12, 6, 3),  RH_FAST(12, 7, 2), RH_FAST(12, 7, 3), RH_FAST(12, 13, 2),
    RH_FAST(14, 7, 2),  RH_FAST(14, 8, 2),
    RH_FAST(16, 8, 2),  RH_FAST(16, 8, 3),  RH_FAST(16, 9, 2), RH_FAST(16, 9, 3),
    RH_FAST(18, 9, 2),  RH_FAST(18, 9, 3),  RH_FAST(18, 10, 2),
    RH_FAST(20, 10, 2), RH_FAST(20, 10, 3), RH_FAST(20, 11, 2),
#endif
};
// The general kernel (ragged batches) is heavier; it ships in two tile sizes.
const Variant kWave[] = {
    RH_WAVE(6, 3, 2), RH_WAVE(6, 4, 2), RH_WAVE(6, 7, 2), RH_WAVE(6, 14, 2), RH_WAVE(8, 4, 2), RH_WAVE(8, 4, 3), RH_WAVE(8, 5, 2), RH_WAVE(8, 9, 2),
    RH_WAVE(9, 5, 2), RH_WAVE(9, 5, 3), RH_WAVE(10, 5, 2), RH_WAVE(10, 5, 3), RH_WAVE(10, 6, 2), RH_WAVE(12, 6, 2), RH_WAVE(12, 6, 3), RH_WAVE(12, 7, 2),
};
// k_rlm_fast<.., RAG>: the first half of a ragged filtered batch, in the tile sizes of the general kernel (both halves
// share the tile geometry, the tables and the aggregate rows)
#ifdef RH_RAG_NO_SUMF  // diagnostics builds: the per-source first half
#define RH_RAGN(r, kv, ns) Variant{r, kv, ns, &k_rlm_fast<r, kv, ns, true, true>, &k_rlm_resid<r, kv>}
#else
#define RH_RAGN(r, kv, ns) Variant{r, kv, ns, &k_rlm_fast<r, kv, ns, true, true, 2, true>, &k_rlm_resid<r, kv>}
#endif
#define RH_RAG(r, kv) RH_RAGN(r, kv, 2)
// (Rings of 3 and 4 stages for the long runs were built and measured in round 4: no change -- a heavy tile is not held back by what
// ONE wave keeps in flight, DESIGN.md 4.4 -- and dropped again: the overrides of rh_rlm_config address the fast plan, which has no
// such geometries.)
const Variant kRag[] = {
    RH_RAG(6, 3), RH_RAG(6, 4), RH_RAG(6, 7), RH_RAG(6, 14), RH_RAG(8, 4), RH_RAG(8, 5), RH_RAG(8, 9), RH_RAG(9, 5), RH_RAG(10, 5), RH_RAG(10, 6), RH_RAG(12, 6), RH_RAG(12, 7),
    RH_RAG(14, 7), RH_RAG(14, 8), RH_RAG(18, 9), RH_RAG(18, 10),
};
// ... and for mono sources (no launch of its own for the pairs: they run inside the kernel)
const Variant kRag1[] = {
#define RH_RAG1(r, kv) Variant{r, kv, 2, &k_rlm_fast<r, kv, 2, true, true, 1, true>, nullptr}
    RH_RAG1(6, 2), RH_RAG1(8, 2), RH_RAG1(8, 3), RH_RAG1(10, 3), RH_RAG1(12, 3), RH_RAG1(12, 4), RH_RAG1(16, 4), RH_RAG1(16, 5), RH_RAG1(18, 5),
#undef RH_RAG1
};
// mono (C = 1): a frame is 4 bytes, so a stage holds twice the frames per KiB
#define RH_FAST1(r, kv, ns) Variant{r, kv, ns, &k_rlm_fast<r, kv, ns, true, false, 1>, &k_rlm_fast<r, kv, ns, false, false, 1>}
#define RH_WAVE1(r, kv, ns) Variant{r, kv, ns, &k_rlm_wave<r, kv, ns, true, 1>, &k_rlm_wave<r, kv, ns, false, 1>}
const Variant kFast1[] = {
    RH_FAST1(4, 2, 2),  RH_FAST1(4, 3, 2),  RH_FAST1(4, 5, 2),  RH_FAST1(6, 2, 2),  RH_FAST1(6, 4, 2),  RH_FAST1(6, 7, 2),  RH_FAST1(8, 2, 2),   RH_FAST1(8, 3, 2),
    RH_FAST1(8, 5, 2),  RH_FAST1(8, 10, 2), RH_FAST1(9, 3, 2),  RH_FAST1(9, 5, 2),  RH_FAST1(10, 3, 2), RH_FAST1(10, 6, 2), RH_FAST1(12, 3, 2),  RH_FAST1(12, 4, 2),
    RH_FAST1(12, 7, 2), RH_FAST1(16, 4, 2), RH_FAST1(16, 5, 2), RH_FAST1(16, 9, 2), RH_FAST1(18, 5, 2), RH_FAST1(18, 5, 3), RH_FAST1(18, 10, 2), RH_FAST1(20, 5, 2),
    RH_FAST1(20, 6, 2), RH_FAST1(20, 11, 2),
};
const Variant kWave1[] = {
    RH_WAVE1(6, 2, 2), RH_WAVE1(6, 4, 2), RH_WAVE1(6, 7, 2), RH_WAVE1(8, 2, 2),  RH_WAVE1(8, 3, 2),  RH_WAVE1(8, 5, 2),  RH_WAVE1(8, 10, 2),
    RH_WAVE1(9, 3, 2), RH_WAVE1(9, 5, 2), RH_WAVE1(10, 3, 2), RH_WAVE1(10, 6, 2), RH_WAVE1(12, 3, 2), RH_WAVE1(12, 4, 2), RH_WAVE1(12, 7, 2),
};
#undef RH_FAST1
#undef RH_WAVE1
#undef RH_RAG
#undef RH_RAGN
#undef RH_FAST
#undef RH_WAVE
}  // namespace

namespace rhp {

VariantTab variant_tab(TabKind kind, bool mono) {
    switch (kind) {
    case kTabFast: return mono ? tab_of(kFast1) : tab_of(kFast);
    case kTabWave: return mono ? tab_of(kWave1) : tab_of(kWave);
    default: return mono ? tab_of(kRag1) : tab_of(kRag);
    }
}

const void *chunk_kernel(int R, uint32_t channels, int KV) {
    if (channels == 2 && R == 9 && KV == 4) return reinterpret_cast<const void *>(&k_rlm_chunk<9, 2, 4>);
    if (channels == 2 && R == 18 && KV == 8) return reinterpret_cast<const void *>(&k_rlm_chunk<18, 2, 8>);
    if (channels == 2 && R == 18 && KV == 4) return reinterpret_cast<const void *>(&k_rlm_chunk<18, 2, 4>);
    if (channels == 1 && R == 18 && KV == 4) return reinterpret_cast<const void *>(&k_rlm_chunk<18, 1, 4>);
    if (channels == 1 && R == 18 && KV == 2) return reinterpret_cast<const void *>(&k_rlm_chunk<18, 1, 2>);
    return nullptr;
}

static const void *chunk_multi_kernel(int R, uint32_t channels, int KV) {
    if (channels == 2 && R == 9 && KV == 4) return reinterpret_cast<const void *>(&k_rlm_chunk_multi<9, 2, 4>);
    if (channels == 2 && R == 18 && KV == 8) return reinterpret_cast<const void *>(&k_rlm_chunk_multi<18, 2, 8>);
    if (channels == 2 && R == 18 && KV == 4) return reinterpret_cast<const void *>(&k_rlm_chunk_multi<18, 2, 4>);
    if (channels == 1 && R == 18 && KV == 4) return reinterpret_cast<const void *>(&k_rlm_chunk_multi<18, 1, 4>);
    if (channels == 1 && R == 18 && KV == 2) return reinterpret_cast<const void *>(&k_rlm_chunk_multi<18, 1, 2>);
    return nullptr;
}

static const void *chunk_classes_kernel(int R, uint32_t channels, int KV) {
    if (channels == 2 && R == 18 && KV == 8) return reinterpret_cast<const void *>(&k_rlm_chunk_classes<18, 2, 8>);
    if (channels == 2 && R == 18 && KV == 4) return reinterpret_cast<const void *>(&k_rlm_chunk_classes<18, 2, 4>);
    if (channels == 1 && R == 18 && KV == 4) return reinterpret_cast<const void *>(&k_rlm_chunk_classes<18, 1, 4>);
    if (channels == 1 && R == 18 && KV == 2) return reinterpret_cast<const void *>(&k_rlm_chunk_classes<18, 1, 2>);
    return nullptr;
}

rh_status chunk_launch_classes(rh_rlm *const *classes, float *const *rows, uint64_t row_capacity_frames, uint32_t n, float *dst_sum, rh_stream stream, bool *taken, bool *summed) {
    *taken = false;
    *summed = false;
    if (n < 2 || n > kChunkMultiMax || rh::knob(rh::K_CLASSES_ONE_BY_ONE)) return RH_OK;
    const void *fn = nullptr;
    for (uint32_t k = 0; k < n; ++k) {
        const rh_rlm *h = classes[k];
        if (!h || !h->cls.empty() || !h->plan || h->out_frames == 0 || row_capacity_frames < h->out_frames) return RH_OK;
        if (!h->chunk.ok || !mix_first_applies(h, *h->plan, h->n_sources, false, false)) return RH_OK;
        const void *f = chunk_multi_kernel(h->chunk.R, h->cfg.channels, h->chunk.KV);
        if (!f || (fn && f != fn)) return RH_OK;
        fn = f;
    }
    ChunkMulti m;
    memset(&m, 0, sizeof m);
    for (uint32_t k = 0; k < n; ++k) {
        rh_rlm *h = classes[k];
        h->collect = &m;
        const rh_status st = rlm_launch(h, 0, h->n_sources, rows[k], row_capacity_frames, nullptr, stream, 0, 0, StreamArgs{});
        h->collect = nullptr;
        if (st != RH_OK) return st;
        if (m.n != k + 1) return RH_ERR_UNSUPPORTED;  // (the class took another path after all: its launch is queued, the others' tickets are not)
    }
    void *args[] = {&m};
    hipStream_t s = rh::as_stream(stream);
    // Classes of ONE geometry (equal lengths, one rate pair, one chunk size: one table of tile bounds) whose tiles fit the chip at once: a
    // workgroup of two waves per tile walks the classes itself -- one wave loads, the other converts and filters (k_rlm_chunk_classes)
    const void *fn2 = rh::knob(rh::K_CLASSES_ONE_WAVE) ? nullptr : chunk_classes_kernel(classes[0]->chunk.R, classes[0]->cfg.channels, classes[0]->chunk.KV);
    bool same = fn2 != nullptr;
    for (uint32_t k = 0; k < n && same; ++k) {
        const rh_rlm *h = classes[k], *h0 = classes[0];
        same = h->exclusive && h->eq_frames == h0->eq_frames && h->out_frames == h0->out_frames && h->F == h0->F && h->T == h0->T && h->chunk_in == h0->chunk_in &&
               h->chunk_out == h0->chunk_out && h->chunk.n_tiles == h0->chunk.n_tiles && h->chunk.R == h0->chunk.R && h->chunk.KV == h0->chunk.KV && h->n_sources >= 1;
    }
    if (same) {
        int per_cu = 0;
        same = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn2, 128, 0) == hipSuccess && per_cu >= 1 &&
               (uint64_t)classes[0]->chunk.n_tiles <= (uint64_t)per_cu * (uint64_t)rh::g_num_cus;
    }
    hipError_t e;
    if (same) {
        m.sum_out = dst_sum;
        e = hipLaunchKernel(fn2, dim3(classes[0]->chunk.n_tiles), dim3(128), args, 0, s);
        *summed = e == hipSuccess;
    } else {
        e = hipLaunchKernel(fn, dim3(m.first[m.n]), dim3(64), args, 0, s);
        if (e == hipSuccess)
            for (uint32_t k = 0; k < n; ++k) classes[k]->shard_base += (m.first[k + 1] - m.first[k]) / 8u;  // every counter of a class has handed out this many tickets
    }
    if (e != hipSuccess) {
        rh::set_hip_error(e, "k_rlm_chunk_multi / k_rlm_chunk_classes launch");
        return RH_ERR_HIP;
    }
    for (uint32_t k = 0; k < n; ++k) mark_launch(classes[k], s);
    *taken = true;
    return RH_OK;
}

#if defined(RH_CLS_DIAG) && RH_CLS_DIAG == 9
extern "C" int rh_debug_cls_stamps(unsigned long long *dst, size_t n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_cls_stamp), n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif
void launch_state(hipStream_t s, unsigned long long *gran, const Tables *tabs, uint32_t n_sources, uint32_t cols, uint32_t last_col, uint32_t J, uint32_t epoch, uint32_t next_epoch) {
    hipLaunchKernelGGL(k_rlm_state, dim3((n_sources + 63) / 64), dim3(64), 0, s, gran, tabs, n_sources, cols, last_col, J, epoch, next_epoch);
}

void launch_state_sum(hipStream_t s, const unsigned long long *gran, const SrcDesc *srcs, uint32_t n_sources, uint32_t cols, uint32_t tag, float *w_out) {
    hipLaunchKernelGGL(k_rlm_state_sum, dim3(1), dim3(256), 0, s, gran, srcs, n_sources, cols, tag, w_out);
}

// Mix first (k_mix_rows / k_mix_ring in front of a one-source fused launch): one-shot runs of filtered equal-length batches.
// ... and blocks of a stream that carries ONE summed state (rh_rlm_stream_block; rh_rlm_stream_block_v while its sources run
// together): the state of the sum is the sum of the states, so the block is summed first and the one mixed row streams through
// the fused kernel with the stream's state words.  `per_source_states`: streams whose blocks carry a state per source.
bool mix_first_applies(const rh_rlm *p, const Plan &pl, uint32_t count, bool per_source_states, bool batch) {
    return &pl == &p->fast && p->filt && p->mix_first_on && !per_source_states && !batch && count >= 2 && !rh::knob(rh::K_NO_MIX_FIRST);
}

rh_status rlm_launch(rh_rlm *p, uint32_t first, uint32_t count, float *dst, uint64_t out_capacity_frames, uint64_t *out_frames, rh_stream stream, uint32_t batch_streams, uint64_t out_stride,
                            const StreamArgs &sa) {
    RH_REQUIRE_INIT();
    if (!p) return RH_ERR_INVALID;
    if (!p->cls.empty()) return RH_ERR_UNSUPPORTED;  // per-source filters: whole one-shot runs only (rh_rlm_run)
    if (first > p->n_sources || count > p->n_sources - first) return RH_ERR_INVALID;
    if (out_frames) *out_frames = p->out_frames;
    if (p->out_frames == 0) return RH_OK;
    if (!dst || (reinterpret_cast<uintptr_t>(dst) & 15u)) return RH_ERR_INVALID;
    if (out_capacity_frames < p->out_frames) return RH_ERR_CAPACITY;
    hipStream_t s = rh::as_stream(stream);
    {
        const rh_status w = pre_launch(p, s);
        if (w != RH_OK) return w;
    }
    const Plan &pl = *p->plan;
    p->epoch += 1;
    if (p->epoch == 0) {  // tag wrap: old tags could alias, start over from a clean table
        if (p->d_gran) RH_HIP_TRY(hipMemsetAsync(p->d_gran, 0, p->gran_words * 8, s));
        if (p->chunk.d_halo) RH_HIP_TRY(hipMemsetAsync(p->chunk.d_halo, 0, p->chunk.cap_tiles * 64, s));
        if (p->chunk.d_gran) RH_HIP_TRY(hipMemsetAsync(p->chunk.d_gran, 0, p->chunk.cap_tiles * 32, s));
        p->epoch = 1;
    }
    Params k;
    k.srcs = p->d_srcs + first;
    k.tabs = pl.d_tabs;
    k.out = dst;
    k.gran = p->d_gran;
    k.ticket = p->d_ctl;
    k.status = p->d_ctl + 1;
    k.out_frames = p->out_frames;
    k.chunk_in = sa.mode ? p->st_chunk_in : p->chunk_in;  // a stream knows its spans whatever the block size
    k.chunk_out = sa.mode ? p->st_chunk_out : p->chunk_out;
    k.n_sources = count;
    k.n_tiles = p->n_tiles;
    k.F = p->F;
    k.T = p->T;
    k.qF = p->F / p->T;
    k.rF = p->F % p->T;
    k.Tf = (float)p->T;
    k.rcpT = rh::lerp_rcp(p->T);  // 0: this T did not pass the exhaustive check of the short division (rh_common.h)
    k.epoch = p->epoch;
    k.J = pl.J;
    k.ticket_base = p->ticket_base;
    k.direct = 0;
    k.rag_merge = k.rag_pairs_from = k.rag_pairs_to = 0;
    k.prof = p->d_prof;
    k.eq_frames = p->eq_frames;
    k.batch_streams = batch_streams;
    k.shards = (batch_streams >= 16 && batch_streams % 8 == 0 && !rh::knob(rh::K_NO_TICKET_SHARDS)) ? 8u : 1u;
    k.shard_base = p->shard_base;
    k.out_stride = out_stride;
    k.st_mode = sa.mode;
    k.st_active = sa.active;
    k.st_m0 = sa.m0;
    k.st_g0 = sa.g0;
    k.st_win = sa.win;
    k.st_wout = sa.wout;
    k.gran_cols = sa.gran_cols ? sa.gran_cols : p->n_tiles;
    k.col0 = sa.gran_cols ? 1u : 0u;
    k.u = pl.uni;
    void *args[] = {&k};
    const uint64_t grid = (uint64_t)p->n_tiles * (batch_streams ? batch_streams : 1);
    if (grid > 0x7fffffffull) return RH_ERR_UNSUPPORTED;
    if (&pl == &p->pair && !sa.mode && !batch_streams) {
        // first half: the stable pairs, summed aggregates into the row behind the `count` per-source rows; second half:
        // the few pairs in which a source is about to end, on top of the first (k_rlm_resid)
        Params k1 = k;
        k1.gran = p->d_gran + (uint64_t)count * p->n_tiles * 4;
        k1.eq_frames = p->rag_frames;  // the sources that last as long as the mix (one length): the end-of-source handling is theirs
        void *args1[] = {&k1};
#ifdef RH_RAG_NO_SUMF
        const bool merge = false;
#else
        const bool merge = !rh::knob(rh::K_RAG_TWO_KERNELS) || !pl.v->plain;  // the pairs inside the first kernel (Params::rag_merge; mono: always)
#endif
        k1.rag_merge = merge ? 1u : 0u;
        k1.rag_pairs_from = p->rag_pairs_from;
        k1.rag_pairs_to = p->rag_pairs_to;
        uint32_t lds1 = pl.lds_bytes + 128u;  // (the pair list behind the ring)
        if (const char *w = rh::knob(rh::K_RAG_RESIDENT)) {  // tuning aid: at most this many tiles of the first half on a CU at once
            const int want = atoi(w);
            if (want >= 2 && want < 8) lds1 = std::max(lds1, (kLdsGranules / (uint32_t)want) * kLdsGranule);
        }
        const uint32_t grid8 = ((uint32_t)grid + 7u) & ~7u;  // tiles by ticket from eight counters (k_rlm_fast): whole rounds
        k1.shards = 8;
        k1.shard_base = p->shard_base;
        hipError_t e1 = hipLaunchKernel(reinterpret_cast<const void *>(pl.v->filt), dim3(grid8), dim3(64), args1, lds1, s);
        if (e1 == hipSuccess) p->shard_base += grid8 / 8u;
        if (e1 == hipSuccess && !merge) {
            k.ticket_base = p->ticket_base;
            e1 = hipLaunchKernel(reinterpret_cast<const void *>(pl.v->plain), dim3((uint32_t)grid), dim3(64), args, 2u * (uint32_t)pl.v->KV * 1024u + 128u /* two stages + the pair list */, s);
            p->ticket_base += (uint32_t)grid;
        }
        if (e1 != hipSuccess) {
            rh::set_hip_error(e1, "ragged batch launch");
            return RH_ERR_HIP;
        }
        return mark_launch(p, s);
    }
    // Mix first: a filtered batch of equal-length sources is summed at the input rate (k_mix_rows: the one pass over the input),
    // and the fused kernel converts and filters that ONE stream.
    if (p->chunk.ok && !sa.mode && mix_first_applies(p, pl, count, false, batch_streams != 0) && count == p->n_sources && first == 0) {
        // mix first in one kernel: every tile sums its aligned chunk of every source, then converts and filters its part of the mix
        const ChunkPlan &c = p->chunk;
        k.tabs = c.d_tabs;
        k.gran = c.d_gran;
        k.n_tiles = c.n_tiles;
        k.J = c.J;
        k.u = c.uni;
        ChunkArgs ca;
        ca.m_lo = c.d_mlo;
        ca.halo = c.d_halo;
        ca.lookT = c.d_look;
        ca.powM = c.d_pow;
        ca.uni = c.d_uni;
        k.direct = (c.direct && p->exclusive) ? 1u : 0u;
        if (p->collect) {  // one launch for several classes (chunk_launch_classes): this one's arguments behind the others', tiles by ticket
            ChunkMulti &m = *static_cast<ChunkMulti *>(p->collect);
            if (m.n >= kChunkMultiMax) return RH_ERR_UNSUPPORTED;
            k.direct = 0;
            const uint32_t g8 = (c.n_tiles + 7u) & ~7u;
            m.e[m.n].p = k;
            m.e[m.n].q = ca;
            m.first[m.n + 1] = m.first[m.n] + g8;
            m.n += 1;
            return RH_OK;  // (the ticket counters advance in chunk_launch_classes, if the launch it decides on takes tickets)
        }
        const uint32_t cgrid = k.direct ? c.n_tiles : (c.n_tiles + 7u) & ~7u;  // by ticket: whole rounds of the eight counters (k_rlm_chunk)
        void *cargs[] = {&k, &ca};
        const hipError_t ce = hipLaunchKernel(c.fn, dim3(cgrid), dim3(64), cargs, 0, s);
        if (ce != hipSuccess) {
            rh::set_hip_error(ce, "k_rlm_chunk launch");
            return RH_ERR_HIP;
        }
        if (!k.direct) p->shard_base += cgrid / 8u;  // every counter has handed out this many tickets
        return mark_launch(p, s);
    }
    const bool pre = p->pre_filter;
    if (pre && (&pl != &p->fast || sa.mode || batch_streams)) return RH_ERR_UNSUPPORTED;  // filter_first: one-shot runs of equal-length batches (rodio_hip.h)
    if (pre || mix_first_applies(p, pl, count, sa.gran_cols != 0, batch_streams != 0)) {
        const uint64_t n_floats = (uint64_t)p->eq_frames * p->cfg.channels;
        const size_t row = (size_t)(((sa.mode ? (uint64_t)p->cfg.max_in_frames * p->cfg.channels : n_floats) + 3) & ~3ull);  // a stream: sized once, for its largest block
        // how the row is cut: vectors per lane, workgroups, and -- short rows -- groups of sources side by side (see k_mix_rows)
        constexpr uint32_t kMixGroups = 16;
        const uint64_t nvec = n_floats / 4;
        int U = 4;  // measured (256 x 1 Mi stereo frames): 0.410 / 0.409 / 0.342 ms for 1 / 2 / 4 vectors per lane
        // ... where the row fills the chip.  A stream's block is a short row (64 Ki frames: 128 workgroups at U = 4): fewer vectors per lane, more
        // workgroups, the same loads in flight per lane (8: the kernel takes 8 / U sources per step)
        while (U > 1 && (nvec + 256ull * U - 1) / (256ull * U) < 2ull * (uint64_t)rh::g_num_cus) U /= 2;
        if (const char *u = rh::knob(rh::K_MIX_U)) U = atoi(u);
        const uint32_t per = 256u * (uint32_t)(U == 1 ? 1 : U == 2 ? 2 : 4);
        const uint32_t wgs = (uint32_t)std::max<uint64_t>(1, (nvec + per - 1) / per);
        const uint64_t ring_waves = (nvec + 511) / 512;  // 8 KiB chunks
        int ring = ring_waves >= 2ull * rh::g_num_cus ? 2 : 0;  // ring depth; 0: the vector-load kernel (short rows: more, smaller pieces)
        if (const char *u = rh::knob(rh::K_MIX_U)) ring = atoi(u) >= 10 ? atoi(u) - 10 : 0;  // tuning aid: 12 / 13 = ring of 2 / 3 stages, 1 / 2 / 4 = vector loads
        // short rows: groups of sources side by side until the launch holds two workgroups per CU (every group keeps at least 8 sources: the
        // kernel's pipeline of descriptor fetches and loads)
        uint32_t groups = 1;
        const uint64_t per_cu = rh::knob(rh::K_MIX_GROUPS) ? (uint64_t)std::max(1, atoi(rh::knob(rh::K_MIX_GROUPS))) : 2ull;  // tuning aid: workgroups per CU the cut aims at
        if (!pre && !ring && !rh::knob(rh::K_MIX_U))
            while (groups < kMixGroups && (uint64_t)wgs * groups < per_cu * (uint64_t)rh::g_num_cus && count / (groups * 2) >= 8) groups *= 2;
        // the mixed row (16-byte vectors) [, the filtered row] -- or the groups' partial rows --, then the descriptors (32 bytes each) at the very end
        const size_t rows_needed = pre ? 2 : (sa.mode ? kMixGroups : groups);  // (a stream: sized once, for whatever its blocks will need)
        const size_t need = row * rows_needed + 64 + kMixGroups * 8;
        if (need > p->mix_floats) {
            const rh_status w = wait_idle(p);
            if (w != RH_OK) return w;
            if (p->d_mix) RH_HIP_TRY(hipFree(p->d_mix));
            p->d_mix = nullptr;
            RH_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_mix), need * sizeof(float)));
            p->mix_floats = need;
        }
        SrcDesc *const ydesc = reinterpret_cast<SrcDesc *>(p->d_mix + (p->mix_floats - 32 - kMixGroups * 8));
        float *const frow = pre ? p->d_mix + row : p->d_mix;  // the row the fused launch reads
        const uint32_t nf = p->eq_frames, mf = (uint32_t)p->out_frames;
        if (ring >= 3) hipLaunchKernelGGL(k_mix_ring<3>, dim3((uint32_t)ring_waves), dim3(64), 0, s, k.srcs, count, p->d_mix, n_floats, ydesc, nf, mf, frow, sa.src_off);
        else if (ring == 2) hipLaunchKernelGGL(k_mix_ring<2>, dim3((uint32_t)ring_waves), dim3(64), 0, s, k.srcs, count, p->d_mix, n_floats, ydesc, nf, mf, frow, sa.src_off);
        else if (U == 1) hipLaunchKernelGGL(k_mix_rows<1>, dim3(wgs, groups), dim3(256), 0, s, k.srcs, count, p->d_mix, n_floats, ydesc, nf, mf, frow, (uint64_t)row, sa.src_off);
        else if (U == 2) hipLaunchKernelGGL(k_mix_rows<2>, dim3(wgs, groups), dim3(256), 0, s, k.srcs, count, p->d_mix, n_floats, ydesc, nf, mf, frow, (uint64_t)row, sa.src_off);
        else hipLaunchKernelGGL(k_mix_rows<4>, dim3(wgs, groups), dim3(256), 0, s, k.srcs, count, p->d_mix, n_floats, ydesc, nf, mf, frow, (uint64_t)row, sa.src_off);
        RH_CHECK_LAUNCH();
        if (pre) {  // the filter of `src.low_pass(f)`, at from_rate, on the mix (time-parallel: rh_biquad mode 1, zero state)
            const rh_status fs = rh_biquad(frow, p->d_mix, p->eq_frames, p->cfg.channels, 1, p->pre_coeffs, nullptr, 1, stream);
            if (fs != RH_OK) return fs;
        }
        k.srcs = ydesc;
        k.n_sources = groups;  // (the partial rows, added in order by the fused launch; one row when the list was not cut)
        // every tile of the one-stream launch resident at once: no tickets (see Params::direct)
        k.direct = (p->exclusive && (uint64_t)p->n_tiles <= (uint64_t)rh::g_num_cus * (uint64_t)std::max(pl.resident_per_cu, 0)) ? 1u : 0u;
    }
    // batch mode fills the chip many times over: no residency shaping, the bare LDS request
    hipError_t e = hipLaunchKernel(pl.kernel, dim3((uint32_t)grid), dim3(64), args, batch_streams ? std::max((uint32_t)pl.v->KV * 1024u, 64u * ((uint32_t)pl.v->R * 8u + 8u)) /* one source per tile: one stage of the ring, reused by the output transpose */ : p->launch_lds, s);
    if (e != hipSuccess) {
        rh::set_hip_error(e, "k_rlm launch");
        return RH_ERR_HIP;
    }
    if (k.direct) {}  // no tickets taken
    else if (k.shards > 1) p->shard_base += (uint32_t)(grid / k.shards);
    else p->ticket_base += (uint32_t)grid;  // every launch takes exactly one ticket per workgroup
    return mark_launch(p, s);
}

}  // namespace rhp

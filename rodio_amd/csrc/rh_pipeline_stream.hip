// rh_pipeline_stream.hip -- block streaming of the fused path: the state that crosses a block boundary and the launches of a block
// (the kernels and rlm_launch: rh_pipeline.hip; handles and plans: rh_pipeline_plan.hip).
#include "rh_pipeline_internal.h"

extern "C" {

// ---- block streaming of the fused path (equal-length blocks, the same sources in every block) -------
// What crosses a block boundary: the converter's position (st_g0: global index of frame 0 of the caller's
// buffers; st_m: output frames emitted) and the SUM over the sources of the filter state at st_m (4 floats,
// scan basis) -- the merged-state kernel never needs a per-source state.  A block emits whole lane runs only
// (a multiple of R output frames), so that the state at its end is a lane's start state; the frames that are
// left over stay with the caller: *consumed tells how many of the frames it passed are done with.
// Closed forms of sample_rate.rs:131-201 for a stream whose converter restarts every `cin` input frames (`cout` output frames
// per whole span; cin == 0: one continuous conversion).
static uint64_t lerp_ready(uint64_t n, uint64_t F, uint64_t T) {  // #m with floor(m*F/T) <= n-2: both taps have arrived
    return n ? (uint64_t)((((unsigned __int128)(n - 1) * T) + F - 1) / F) : 0;
}
static uint64_t run_total(uint64_t n, uint64_t F, uint64_t T) {  // ... plus the verbatim last frame of a run that is complete
    const uint64_t c1 = lerp_ready(n, F, T);
    return n && (unsigned __int128)c1 * F < (unsigned __int128)n * T ? c1 + 1 : c1;
}
static uint64_t stream_ready(uint64_t N, uint64_t F, uint64_t T, uint64_t cin, uint64_t cout) {  // a source that will deliver more
    if (!cin) return lerp_ready(N, F, T);
    return N / cin * cout + lerp_ready(N % cin, F, T);  // whole spans are complete, verbatim frame included
}
static uint64_t stream_total(uint64_t N, uint64_t F, uint64_t T, uint64_t cin, uint64_t cout) {  // a source that has ended with N frames
    if (!cin) return run_total(N, F, T);
    return N / cin * cout + run_total(N % cin, F, T);
}
static uint64_t stream_first_tap(uint64_t m, uint64_t F, uint64_t T, uint64_t cin, uint64_t cout) {  // the input frame output frame m reads first
    if (!cin) return (uint64_t)(((unsigned __int128)m * F) / T);
    const uint64_t k = m / cout, il = (uint64_t)(((unsigned __int128)(m % cout) * F) / T);
    return k * cin + (il < cin - 1 ? il : cin - 1);
}

rh_status rh_rlm_stream_begin(rh_rlm *p) {
    RH_REQUIRE_INIT();
    if (!p) return RH_ERR_INVALID;
    if (p->pre_filter) return RH_ERR_UNSUPPORTED;  // filter_first: one-shot runs only (rodio_hip.h)
    if (!p->filters.empty()) return RH_ERR_UNSUPPORTED;  // per-source filters: one-shot runs (a streaming host keeps one handle per filter: rodio_hip.hpp)
    p->st_tab_version = ~0ull;  // (the first summed block uploads its table)
    p->st_chunk_in = p->st_chunk_out = 0;
    if (p->cfg.span_len != 0) {  // sources that report spans of span_len samples: the converter restarts every min(span_len, 32768) samples (uniform.rs:56-67)
        const uint64_t span = p->cfg.span_len < 32768 ? p->cfg.span_len : 32768;
        if (span % p->cfg.channels != 0) return RH_ERR_UNSUPPORTED;  // a span that splits a frame
        p->st_chunk_in = span / p->cfg.channels;
        const rh_status st = rh_resample_out_frames(p->st_chunk_in, p->cfg.from_rate, p->cfg.to_rate, p->cfg.channels, 0, &p->st_chunk_out);
        if (st != RH_OK) return st;
        if (p->F == p->T) p->st_chunk_in = p->st_chunk_out = 0;  // the converter passes through: its restarts leave no trace
    }
    {
        const rh_status w = wait_idle(p);  // a previous stream's last block may still read its state words
        if (w != RH_OK) return w;
    }
    for (int k = 0; k < 2; ++k) {
        if (!p->d_w[k]) RH_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_w[k]), 4 * sizeof(float)));
        RH_HIP_TRY(rh::fill_now(p->d_w[k], 0, 4 * sizeof(float)));
    }
    p->st_on = true;
    p->st_done = false;
    p->st_g0 = p->st_m = 0;
    p->st_nsrc = 0;
    p->st_cur = 0;
    p->st_total.clear();
    p->st_cols = 0;
    p->st_together = p->st_decided = false;
    p->st_n_summed = p->st_n_each = p->st_n_recover = p->st_n_rejoin = p->st_n_sblk = 0;
    p->st_prev_ptrs.clear();
    p->st_gone.clear();
    p->st_prev_gone.clear();
    p->st_prev_avail = p->st_prev_g0 = p->st_prev_m = p->st_prev_out = 0;
    sblk_other_block(p, true);
    return RH_OK;
}

rh_status rh_rlm_stream_overlap(rh_rlm *p, int32_t on) {
    if (!p) return RH_ERR_INVALID;
    p->st_overlap = on != 0;
    return RH_OK;
}

rh_status rh_rlm_stream_one_launch_blocks(rh_rlm *p, uint32_t *blocks) {
    if (!p || !blocks) return RH_ERR_INVALID;
    *blocks = p->st_n_sblk;
    return RH_OK;
}

rh_status rh_rlm_stream_overlapped_blocks(rh_rlm *p, uint32_t *blocks) {
    if (!p || !blocks) return RH_ERR_INVALID;
    *blocks = sblk_chained_blocks(p);
    return RH_OK;
}

rh_status rh_rlm_stream_stats(rh_rlm *p, uint32_t *summed_blocks, uint32_t *per_source_blocks, uint32_t *recoveries) {
    if (!p) return RH_ERR_INVALID;
    if (summed_blocks) *summed_blocks = p->st_n_summed;
    if (per_source_blocks) *per_source_blocks = p->st_n_each;
    if (recoveries) *recoveries = p->st_n_recover;
    return RH_OK;
}

rh_status rh_rlm_stream_keep_history(rh_rlm *p, int32_t on) {
    if (!p) return RH_ERR_INVALID;
    if (p->st_on && (p->st_nsrc || p->st_decided)) return RH_ERR_INVALID;  // before the stream's first block
    p->st_history = on != 0;
    return RH_OK;
}

static rh_status stream_block_summed(rh_rlm *p, const float *const *srcs_host, uint32_t n_sources, uint64_t avail_frames, int32_t flush, float *dst, uint64_t out_capacity_frames,
                                     uint64_t *out_frames, uint64_t *consumed_frames, rh_stream stream, const std::vector<uint8_t> *gone = nullptr);

rh_status rh_rlm_stream_block(rh_rlm *p, const float *const *srcs_host, uint32_t n_sources, uint64_t avail_frames, int32_t flush, float *dst, uint64_t out_capacity_frames,
                              uint64_t *out_frames, uint64_t *consumed_frames, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (!p || !p->st_on || p->st_done || !out_frames || !consumed_frames) return RH_ERR_INVALID;
    if (p->st_together || p->st_decided) return RH_ERR_INVALID;  // a stream uses one of the two block entries throughout
    return stream_block_summed(p, srcs_host, n_sources, avail_frames, flush, dst, out_capacity_frames, out_frames, consumed_frames, stream);
}

// `gone` (rh_rlm_stream_block_v, a stream back on the summed state after sources have ended): sources that have given everything.  Their table
// entries point at a live source's row with gain 0 -- exact zeros in the sum, the row count of the launch unchanged, and the entries move on
// with the live rows (the table stays on the device).
static rh_status stream_block_summed(rh_rlm *p, const float *const *srcs_host, uint32_t n_sources, uint64_t avail_frames, int32_t flush, float *dst, uint64_t out_capacity_frames,
                              uint64_t *out_frames, uint64_t *consumed_frames, rh_stream stream, const std::vector<uint8_t> *gone) {
    RH_REQUIRE_INIT();
    if (!p || !p->st_on || p->st_done || !out_frames || !consumed_frames) return RH_ERR_INVALID;
    if (n_sources == 0 || n_sources > p->cfg.max_sources || avail_frames > p->cfg.max_in_frames) return RH_ERR_CAPACITY;
    if ((p->st_cols && !gone) || (p->st_nsrc && p->st_nsrc != n_sources)) return RH_ERR_INVALID;  // the summed state belongs to one set of sources (and one kind of stream)
    *out_frames = 0;
    *consumed_frames = 0;
    const uint64_t F = p->F, T = p->T, R = p->fast.v->R, L = 64 * R;
    const uint64_t N = p->st_g0 + avail_frames;  // input frames of the stream that exist so far
    const uint64_t cin = p->st_chunk_in, cout = p->st_chunk_out;
    // output frames computable from them: every m whose two taps have arrived; at the end also the verbatim last frame
    const uint64_t m_end = flush ? stream_total(N, F, T, cin, cout) : stream_ready(N, F, T, cin, cout);
    uint64_t out = m_end > p->st_m ? m_end - p->st_m : 0;
    if (!flush) {
        // whole lane runs (the state at the block's end is a lane's start state) that are whole 16-byte vectors too: a caller that writes
        // the blocks one behind the other (`dst + emitted so far`) stays aligned whatever R the plan took (odd R: pairs of runs)
        const uint64_t vec = 4u / p->cfg.channels;  // frames per 16 bytes: 2 (stereo), 4 (mono)
        const uint64_t unit = R / std::gcd(R, vec) * vec;
        out = out / unit * unit;
    }
    if (out >= (1ull << 31)) return RH_ERR_UNSUPPORTED;
    if (out > out_capacity_frames) return RH_ERR_CAPACITY;
    if (out > 0 || flush) {
        if (out > 0) {
            if (!srcs_host || !dst || (reinterpret_cast<uintptr_t>(dst) & 15u)) return RH_ERR_INVALID;
            std::vector<SrcDesc> &h = p->h_desc;
            // A block that is summed first reads nothing of the table but the pointers and the gains (k_mix_rows / k_mix_ring; the frame
            // counts travel as kernel arguments), and they take a common offset: when every source has moved on by the same number of
            // bytes since the table was uploaded -- rows of one staging block, resident rows read at `row + consumed` -- the table stays
            // where it is and the distance goes with the launch.  One copy kernel and two dependent boundaries less per block
            // (4 + ~3 us of the 49 us a 64 Ki-frame block of 256 sources takes: profiles/r05_stream_kernel_trace.txt).
            // Sources that are gone (a stream back on the summed state) are not in the table at all: it holds the live ones, in order (an
            // entry with gain 0 that points at a live row turns that row's Inf into NaN -- rodio's mixer has no contribution there at all -- and
            // reads the row once more).
            std::vector<uint32_t> live;
            for (uint32_t s = 0; s < n_sources; ++s)
                if (!(gone && (*gone)[s])) live.push_back(s);
            const uint32_t nl = (uint32_t)live.size();
            if (nl == 0) return RH_ERR_INVALID;
            const bool sum_first = mix_first_applies(p, p->fast, nl, false, false);
            uint64_t src_off = 0;
            bool reuse = sum_first && p->st_tab_version == p->srcs_version && p->st_tab_ptrs.size() == nl && !rh::knob(rh::K_STREAM_UPLOAD_ALWAYS);
            auto row_of = [&](uint32_t k) { return srcs_host[live[k]]; };
            if (reuse) {
                src_off = (uint64_t)(reinterpret_cast<uintptr_t>(row_of(0)) - reinterpret_cast<uintptr_t>(p->st_tab_ptrs[0]));
                for (uint32_t k = 0; k < nl && reuse; ++k)
                    reuse = row_of(k) && !(reinterpret_cast<uintptr_t>(row_of(k)) & 15u) &&
                            (uint64_t)(reinterpret_cast<uintptr_t>(row_of(k)) - reinterpret_cast<uintptr_t>(p->st_tab_ptrs[k])) == src_off;
            }
            if (!reuse) {
                src_off = 0;
                h.resize(nl);
                p->st_tab_ptrs.resize(nl);
                for (uint32_t k = 0; k < nl; ++k) {
                    const float *row = row_of(k);
                    if (!row || (reinterpret_cast<uintptr_t>(row) & 15u)) return RH_ERR_INVALID;
                    const uint32_t s = live[k];
                    h[k] = SrcDesc{row, (uint32_t)avail_frames, (uint32_t)out, s < p->gains.size() ? p->gains[s] : 1.0f, {0, 0, 0}};
                    p->st_tab_ptrs[k] = row;
                }
                const rh_status up = upload_descriptors(p, nl, rh::as_stream(stream));
                if (up != RH_OK) return up;
                p->st_tab_version = p->srcs_version;
            }
            p->equal = true;
            p->eq_frames = (uint32_t)avail_frames;
            p->n_sources = nl;
            p->out_frames = out;
            p->chunk.ok = false;  // (the tile tables of k_rlm_chunk belong to a one-shot batch)
            // ONE launch per block where the block is k_rlm_sblk's (rh_pipeline_sblk.hip): the sum, the conversion and the filter in one kernel
            bool taken = false;
            if (sum_first) {
                StreamArgs sb;
                sb.mode = flush ? 2u : 1u;
                sb.active = (uint32_t)out;
                sb.m0 = p->st_m;
                sb.g0 = p->st_g0;
                sb.win = p->d_w[p->st_cur];
                sb.wout = p->d_w[p->st_cur ^ 1];
                sb.src_off = src_off;
                const rh_status sk = sblk_try(p, nl, avail_frames, out, dst, sb, rh::as_stream(stream), &taken);
                if (sk != RH_OK) return sk;
            }
            if (taken) {
                p->st_n_sblk += 1;
            } else {
                sblk_other_block(p);
                rh_status st = activate_plan(p, &p->fast);
                if (st != RH_OK) return st;
                const uint64_t tiles = flush ? (out + L - 1) / L : out / L + 1;  // + the tile that holds the end-state lane
                p->n_tiles = (uint32_t)tiles;
                if (p->filt && (size_t)tiles * 4 > p->gran_words) return RH_ERR_CAPACITY;  // activate_plan sized it for ceil(out/L)+... never smaller
                StreamArgs sa;
                sa.mode = flush ? 2u : 1u;
                sa.active = (uint32_t)out;
                sa.m0 = p->st_m;
                sa.g0 = p->st_g0;
                sa.win = p->d_w[p->st_cur];
                sa.wout = p->d_w[p->st_cur ^ 1];
                sa.src_off = src_off;
                st = rlm_launch(p, 0, nl, dst, out_capacity_frames, nullptr, stream, 0, 0, sa);
                if (st != RH_OK) return st;
            }
            p->st_n_summed += 1;
            if (!flush && p->filt) p->st_cur ^= 1;
        }
        p->st_m += out;
        p->st_nsrc = n_sources;
    }
    if (flush) {
        p->st_done = true;
        *consumed_frames = avail_frames;
    } else {
        // the next block needs the taps of output frames st_m-2 onwards: input frame floor((st_m-2)*F/T) (of their span)
        const uint64_t keep_from = p->st_m >= 2 ? stream_first_tap(p->st_m - 2, F, T, cin, cout) : 0;
        const uint64_t cons = keep_from > p->st_g0 ? keep_from - p->st_g0 : 0;
        *consumed_frames = cons < avail_frames ? cons : avail_frames;
        *consumed_frames -= *consumed_frames % (4u / p->cfg.channels);  // whole 16-byte vectors: `row + consumed` is a row the next block can take as it is
        p->st_g0 += *consumed_frames;
    }
    *out_frames = out;
    return RH_OK;
}

// ---- block streaming with per-source filter states (ragged batches: sources of one clock that end at
// different times) -- k_rlm_wave, whose look-back is per source anyway.  A block emits whole tiles, so the state
// that crosses the boundary is a tile carry: k_rlm_state folds the block's aggregates into column 0 of the
// aggregate rows, where the next launch finds it as the aggregate of a virtual predecessor tile.

static rh_status stream_block_v_impl(rh_rlm *p, const float *const *srcs_host, const uint64_t *avail_frames_host, const uint8_t *ended_host, uint32_t n_sources, float *dst,
                                     uint64_t out_capacity_frames, uint64_t *out_frames, uint64_t *consumed_frames, rh_stream stream);

rh_status rh_rlm_stream_block_v(rh_rlm *p, const float *const *srcs_host, const uint64_t *avail_frames_host, const uint8_t *ended_host, uint32_t n_sources, float *dst,
                                uint64_t out_capacity_frames, uint64_t *out_frames, uint64_t *consumed_frames, rh_stream stream) {
    if (p) p->st_dirty = false;
    const rh_status st = stream_block_v_impl(p, srcs_host, avail_frames_host, ended_host, n_sources, dst, out_capacity_frames, out_frames, consumed_frames, stream);
    // An error after the stream's state was touched (the switch from the summed state to per-source states: rows sized, the replay launched)
    // leaves a state no later block can continue from: the stream is over, and says so (RH_ERR_INVALID on every later call) instead of
    // wedging half-way.  Errors found while the arguments are checked leave the stream as it was.
    if (st != RH_OK && p && p->st_dirty) {
        p->st_done = true;
        p->st_prev_ptrs.clear();
    }
    return st;
}

static rh_status stream_block_v_impl(rh_rlm *p, const float *const *srcs_host, const uint64_t *avail_frames_host, const uint8_t *ended_host, uint32_t n_sources, float *dst,
                                     uint64_t out_capacity_frames, uint64_t *out_frames, uint64_t *consumed_frames, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (!p || !p->st_on || p->st_done || !out_frames || !consumed_frames || !avail_frames_host || !ended_host) return RH_ERR_INVALID;
    if (n_sources == 0 || n_sources > p->cfg.max_sources) return RH_ERR_CAPACITY;
    if (p->st_nsrc && (p->st_nsrc != n_sources || (!p->st_cols && !p->st_together))) return RH_ERR_INVALID;  // one set of sources, one kind of stream
    *out_frames = 0;
    *consumed_frames = 0;
    const uint64_t F = p->F, T = p->T, L = 64ull * p->wave.v->R;
    const uint64_t cin = p->st_chunk_in, cout = p->st_chunk_out;
    hipStream_t hs = rh::as_stream(stream);
    // ---- the sources run TOGETHER (all live, the same frames each): one summed state, the block summed first -------------------
    // What a recovery needs of the block before: K = J tiles of the per-source kernel, replayed from a zero state (the filter has
    // forgotten what lies further back: ||B^K|| < 2^-40).  So a block stays on the summed state only if it emits at least K frames;
    // a stream whose first block does not never starts on it.
    const uint64_t K = (uint64_t)(p->filt ? p->wave.J : 0) * L;
    if (!p->st_decided) {
        p->st_decided = true;
        p->st_together = p->st_history && p->filt && p->mix_first_on && n_sources >= 2 && K > 0 && !rh::knob(rh::K_NO_MIX_FIRST);
    }
    const uint64_t Rf = p->fast.v->R;
    auto would_emit = [&](uint64_t avail) {  // output frames a summed block of `avail` frames per source would emit: stream_block_summed's own rounding
        const uint64_t ready = stream_ready(p->st_g0 + avail, F, T, cin, cout);
        const uint64_t vec = 4u / p->cfg.channels, unit = Rf / std::gcd(Rf, vec) * vec;  // (whole lane runs that are whole 16-byte vectors: ADVICE r5 -- rounding to R alone let a
        return ready > p->st_m ? (ready - p->st_m) / unit * unit : 0;                     // block pass the K test and then emit fewer than K frames, which a later recovery relies on)
    };
    // ---- ... and TOGETHER AGAIN: a stream with a state per source whose sources have either given everything or still run, the running ones
    // with the same frames.  The summed state is the sum of their states (column 0 of their rows: k_rlm_state_sum); the ones that are gone
    // stay in the table with gain 0 (stream_block_summed).  From there the stream is what it was before the first source ended -- one summed
    // launch per block instead of a wave per tile walking every source (~360 us per block of 256 sources whatever its length) -- until the
    // next source ends.  RH_STREAM_NO_REJOIN=1: a state per source to the end, as before round 5.
    bool leaving = false;
    if (!p->st_together && p->st_cols && p->st_history && p->filt && p->mix_first_on && K > 0 && !p->st_total.empty() && !rh::knob(rh::K_NO_MIX_FIRST) && !rh::knob(rh::K_STREAM_NO_REJOIN)) {
        bool ok = true;
        uint32_t live = 0;
        uint64_t av = 0;
        std::vector<uint8_t> gone(n_sources, 0);
        for (uint32_t s = 0; s < n_sources && ok; ++s) {
            if (p->st_total[s] != ~0ull) {  // it has ended: gone only when it has given everything
                const uint64_t M = stream_total(p->st_total[s], F, T, cin, cout);
                ok = M <= p->st_m;
                gone[s] = 1;
                continue;
            }
            ok = !ended_host[s] && (live == 0 || avail_frames_host[s] == av);
            av = avail_frames_host[s];
            live += 1;
        }
        if (ok && live >= 2 && would_emit(av) >= K) {
            p->st_dirty = true;
            // the table of the summed blocks first (the sum reads who is gone from it), then the sum of the live states into the summed state's words
            std::vector<SrcDesc> &h = p->h_desc;
            h.resize(n_sources);
            uint32_t lead = 0;
            while (lead + 1 < n_sources && gone[lead]) ++lead;
            p->st_tab_ptrs.resize(n_sources);
            for (uint32_t s = 0; s < n_sources; ++s) {
                const float *row = gone[s] ? srcs_host[lead] : srcs_host[s];
                if (!row || (reinterpret_cast<uintptr_t>(row) & 15u)) return RH_ERR_INVALID;
                h[s] = SrcDesc{row, (uint32_t)av, 0u, gone[s] ? 0.0f : (s < p->gains.size() ? p->gains[s] : 1.0f), {0, 0, 0}};
                p->st_tab_ptrs[s] = row;
            }
            {
                const rh_status up = upload_descriptors(p, n_sources, hs);
                if (up != RH_OK) return up;
            }
            p->st_tab_version = p->srcs_version;
            {
                const rh_status pw = pre_launch(p, hs);
                if (pw != RH_OK) return pw;
            }
            launch_state_sum(hs, p->d_gran, p->d_srcs, n_sources, p->st_cols, p->epoch + 1, p->d_w[p->st_cur]);  // (column 0 carries the tag of the launch that would have read it)
            RH_CHECK_LAUNCH();
            p->epoch += 1;  // ... a tag the summed launch that comes next must not share: its tiles' words lie in the same table
            {
                const rh_status mk = mark_launch(p, hs);
                if (mk != RH_OK) return mk;
            }
            p->st_gone.swap(gone);
            p->st_together = true;
            p->st_n_rejoin += 1;
        }
    }
    if (p->st_together) {
        const std::vector<uint8_t> *gone = p->st_gone.empty() ? nullptr : &p->st_gone;
        bool same = true, any_ended = false, all_ended = true;
        uint32_t lead = 0;
        while (gone && lead + 1 < n_sources && (*gone)[lead]) ++lead;
        for (uint32_t s = 0; s < n_sources; ++s) {
            if (gone && (*gone)[s]) continue;
            same = same && avail_frames_host[s] == avail_frames_host[lead];
            any_ended = any_ended || ended_host[s] != 0;
            all_ended = all_ended && ended_host[s] != 0;
        }
        if (same && all_ended) {  // they end together too: the summed stream's last block
            const rh_status st = stream_block_summed(p, srcs_host, n_sources, avail_frames_host[lead], 1, dst, out_capacity_frames, out_frames, consumed_frames, stream, gone);
            if (st == RH_OK) p->st_together = false;  // (st_done is set: nothing follows)
            return st;
        }
        if (same && !any_ended && would_emit(avail_frames_host[lead]) >= K) {
            const uint64_t g0 = p->st_g0, m0 = p->st_m;
            const rh_status st = stream_block_summed(p, srcs_host, n_sources, avail_frames_host[lead], 0, dst, out_capacity_frames, out_frames, consumed_frames, stream, gone);
            if (st != RH_OK) return st;
            p->st_prev_ptrs.assign(srcs_host, srcs_host + n_sources);
            p->st_prev_gone = p->st_gone;
            p->st_prev_avail = avail_frames_host[lead];
            p->st_prev_g0 = g0;
            p->st_prev_m = m0;
            p->st_prev_out = *out_frames;
            return RH_OK;
        }
        p->st_dirty = true;
        p->st_together = false;  // a source ends or falls behind, or the block is short: one state per source from here on
        leaving = true;
    }
    sblk_other_block(p);  // (a block with a state per source: the summed state, if it comes back, comes back in the plain words)
    const bool recover = (leaving || !p->st_cols) && p->st_prev_out >= K && K > 0 && !p->st_prev_ptrs.empty();
    const bool rows_exist = p->st_cols != 0;  // (the stream has had a state per source before: its rows are sized, their column 0 carries old tags)
    if (!p->st_cols) {  // first block of the per-source stream: size the aggregate rows once (the states live in them), zero states
        p->st_dirty = true;
        rh::ResampleGeom g;
        rh_status st = rh::make_resample_geom(p->cfg.max_in_frames, p->cfg.from_rate, p->cfg.to_rate, p->cfg.channels, 0, &g);
        if (st != RH_OK) return st;
        const uint64_t span_extra = cin ? p->cfg.max_in_frames / cin + 2 : 0;  // every span a block touches adds its verbatim frame
        const uint64_t cols = (g.out_frames + span_extra + L - 1) / L + 2;
        if (cols > 0x7fffffffull) return RH_ERR_UNSUPPORTED;
        const size_t words = (size_t)(p->cfg.max_sources + 1) * cols * 4;  // as activate_plan counts: one row per source + the row of summed aggregates
        if (p->filt && words > p->gran_words) {
            const rh_status w = wait_idle(p);
            if (w != RH_OK) return w;
            if (p->d_gran) RH_HIP_TRY(hipFree(p->d_gran));
            p->d_gran = nullptr;
            RH_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_gran), words * 8));
            RH_HIP_TRY(rh::fill_now(p->d_gran, 0, words * 8));
            p->gran_words = words;
        }
        p->st_cols = (uint32_t)cols;
        p->st_total.assign(n_sources, ~0ull);
        if (p->filt && p->epoch >= 0xf0000000u) {  // keep the epoch tag from wrapping inside a stream (its states live in the table)
            RH_HIP_TRY(hipMemsetAsync(p->d_gran, 0, p->gran_words * 8, hs));
            p->epoch = 0;
        }
        if (p->filt) {
            const rh_status pw = pre_launch(p, hs);
            if (pw != RH_OK) return pw;
            launch_state(hs, p->d_gran, p->wave.d_tabs, n_sources, p->st_cols, 0u, (uint32_t)p->wave.J, p->epoch, p->epoch + 1);
            RH_CHECK_LAUNCH();
            const rh_status mk = mark_launch(p, hs);
            if (mk != RH_OK) return mk;
        }
    }
    if (recover) {
        // The states the summed stream never kept: replay the last K output frames of the block before through the per-source kernel
        // from a zero state (its rows are still there: rh_rlm_stream_keep_history), mix discarded, and fold the replay's aggregates
        // into column 0 -- exactly what the end of a per-source block does.
        const uint64_t m0 = p->st_m - K;  // >= st_prev_m: that block emitted at least K frames
        if (rows_exist) {  // zero states tagged for the replay (the first time round, the sizing above has just written them)
            const rh_status pw = pre_launch(p, hs);
            if (pw != RH_OK) return pw;
            launch_state(hs, p->d_gran, p->wave.d_tabs, n_sources, p->st_cols, 0u, (uint32_t)p->wave.J, p->epoch, p->epoch + 1);
            RH_CHECK_LAUNCH();
            const rh_status mk = mark_launch(p, hs);
            if (mk != RH_OK) return mk;
        }
        const size_t need = (size_t)K * p->cfg.channels + 64;
        if (need > p->replay_floats) {
            const rh_status w = wait_idle(p);
            if (w != RH_OK) return w;
            if (p->d_replay) RH_HIP_TRY(hipFree(p->d_replay));
            p->d_replay = nullptr;
            RH_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_replay), need * sizeof(float)));
            p->replay_floats = need;
        }
        std::vector<SrcDesc> &h = p->h_desc;
        h.resize(n_sources);
        for (uint32_t s = 0; s < n_sources; ++s) {
            const bool was_gone = s < p->st_prev_gone.size() && p->st_prev_gone[s];  // (it had given everything before that block: nothing to replay, no state)
            h[s] = SrcDesc{was_gone ? nullptr : p->st_prev_ptrs[s], was_gone ? 0u : (uint32_t)p->st_prev_avail, was_gone ? 0u : (uint32_t)K, s < p->gains.size() ? p->gains[s] : 1.0f, {0, 0, 0}};
        }
        {
            const rh_status up = upload_descriptors(p, n_sources, hs);
            if (up != RH_OK) return up;
        }
        p->equal = false;
        p->n_sources = n_sources;
        p->out_frames = K;
        rh_status st = activate_plan(p, &p->wave);
        if (st != RH_OK) return st;
        StreamArgs sa;
        sa.mode = 1u;
        sa.active = (uint32_t)K;
        sa.m0 = m0;
        sa.g0 = p->st_prev_g0;
        sa.gran_cols = p->st_cols;
        st = rlm_launch(p, 0, n_sources, p->d_replay, K, nullptr, stream, 0, 0, sa);
        if (st != RH_OK) return st;
        launch_state(hs, p->d_gran, p->wave.d_tabs, n_sources, p->st_cols, (uint32_t)(K / L) + 1u, (uint32_t)p->wave.J, p->epoch, p->epoch + 1);
        RH_CHECK_LAUNCH();
        const rh_status mk = mark_launch(p, hs);
        if (mk != RH_OK) return mk;
        p->st_prev_ptrs.clear();
        p->st_n_recover += 1;
    }
    // what every source can still give: a live one every frame whose two taps have arrived, an ended one all it has left
    uint64_t live_min = ~0ull, ended_max = 0;
    bool any_live = false;
    for (uint32_t s = 0; s < n_sources; ++s) {
        if (avail_frames_host[s] > p->cfg.max_in_frames) return RH_ERR_CAPACITY;
        if (p->st_total[s] == ~0ull && ended_host[s]) p->st_total[s] = p->st_g0 + avail_frames_host[s];
        if (p->st_total[s] == ~0ull) {
            const uint64_t N = p->st_g0 + avail_frames_host[s];
            const uint64_t m_end = stream_ready(N, F, T, cin, cout);  // every m whose two taps have arrived
            const uint64_t can = m_end > p->st_m ? m_end - p->st_m : 0;
            live_min = can < live_min ? can : live_min;
            any_live = true;
        } else {
            const uint64_t M = stream_total(p->st_total[s], F, T, cin, cout);
            const uint64_t rem = M > p->st_m ? M - p->st_m : 0;
            ended_max = rem > ended_max ? rem : ended_max;
        }
    }
    const bool final_block = !any_live;
    const uint64_t out = final_block ? ended_max : live_min / L * L;
    if (out >= (1ull << 31)) return RH_ERR_UNSUPPORTED;
    if (out > out_capacity_frames) return RH_ERR_CAPACITY;
    if (out > 0) {
        if (!srcs_host || !dst || (reinterpret_cast<uintptr_t>(dst) & 15u)) return RH_ERR_INVALID;
        const uint64_t tiles = (out + L - 1) / L;
        if (tiles + 1 > p->st_cols) return RH_ERR_CAPACITY;
        std::vector<SrcDesc> &h = p->h_desc;
        h.resize(n_sources);
        for (uint32_t s = 0; s < n_sources; ++s) {
            uint64_t ms = out;
            if (p->st_total[s] != ~0ull) {
                const uint64_t M = stream_total(p->st_total[s], F, T, cin, cout);
                const uint64_t rem = M > p->st_m ? M - p->st_m : 0;
                ms = rem < out ? rem : out;
            }
            if (ms && (!srcs_host[s] || (reinterpret_cast<uintptr_t>(srcs_host[s]) & 15u))) return RH_ERR_INVALID;
            h[s] = SrcDesc{ms ? srcs_host[s] : nullptr, ms ? (uint32_t)avail_frames_host[s] : 0u, (uint32_t)ms, s < p->gains.size() ? p->gains[s] : 1.0f, {0, 0, 0}};
        }
        {
            const rh_status up = upload_descriptors(p, n_sources, hs);
            if (up != RH_OK) return up;
        }
        p->equal = false;
        p->n_sources = n_sources;
        p->out_frames = out;
        rh_status st = activate_plan(p, &p->wave);  // never reallocates the rows: they were sized for the largest block
        if (st != RH_OK) return st;
        StreamArgs sa;
        sa.mode = final_block ? 2u : 1u;
        sa.active = (uint32_t)out;
        sa.m0 = p->st_m;
        sa.g0 = p->st_g0;
        sa.gran_cols = p->st_cols;
        st = rlm_launch(p, 0, n_sources, dst, out_capacity_frames, nullptr, stream, 0, 0, sa);
        if (st != RH_OK) return st;
        p->st_n_each += 1;
        if (!final_block && p->filt) {
            launch_state(hs, p->d_gran, p->wave.d_tabs, n_sources, p->st_cols, (uint32_t)tiles + 1u, (uint32_t)p->wave.J, p->epoch,
                               p->epoch + 1);
            RH_CHECK_LAUNCH();
            const rh_status mk = mark_launch(p, hs);
            if (mk != RH_OK) return mk;
        }
        p->st_m += out;
    }
    p->st_nsrc = n_sources;
    if (final_block) {
        p->st_done = true;
        uint64_t mx = 0;
        for (uint32_t s = 0; s < n_sources; ++s) mx = avail_frames_host[s] > mx ? avail_frames_host[s] : mx;
        *consumed_frames = mx;
    } else {
        // the next block needs the taps of output frames st_m-2 onwards: input frame floor((st_m-2)*F/T).  The
        // caller drops min(consumed, what it holds) frames of every source.
        const uint64_t keep_from = p->st_m >= 2 ? stream_first_tap(p->st_m - 2, F, T, cin, cout) : 0;
        *consumed_frames = keep_from > p->st_g0 ? keep_from - p->st_g0 : 0;
        *consumed_frames -= *consumed_frames % (4u / p->cfg.channels);  // whole 16-byte vectors (see rh_rlm_stream_block)
        p->st_g0 += *consumed_frames;
    }
    *out_frames = out;
    return RH_OK;
}

}  // extern "C"

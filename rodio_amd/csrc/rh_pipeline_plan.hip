// rh_pipeline_plan.hip -- the host side of the fused path (rh_pipeline.hip has the kernels and the launch): handles, launch plans and
// their tables, the sources / gains / filter classes of a batch, the one-shot entry points, the tuner and the diagnostics.
#include "rh_pipeline_internal.h"

namespace {

// Every live handle, for rh::rlm_stream_retired (rh_stream_destroy / rh_stream_release_scratch call it after they have
// synchronised the stream): a handle whose last launches went to a stream that is about to go must not record an event on it later.
std::mutex g_handles_mu;
std::vector<rh_rlm *> g_handles;
}  // namespace

namespace rhp {

// Block until every launch of this handle has completed.  The event is recorded HERE, behind everything the handle has queued on
// its stream (streams run in order), not behind every launch: a launch costs no API call and no marker on the device for it.
// The stream may be gone by now if it was a caller's own (a PyTorch stream destroyed behind the library's back; the library's
// streams tell the handle when they go): the record then fails, and the whole device is waited for instead.  Either way the handle
// comes out idle -- a failure here must not wedge every later set_sources / set_gains / stream_begin.
rh_status wait_idle(rh_rlm *p) {
    if (p->launched) {
        hipError_t e = hipSuccess;
        if (!p->idle_ev) e = hipEventCreateWithFlags(&p->idle_ev, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(p->idle_ev, p->last_stream);
        if (e == hipSuccess) e = hipEventSynchronize(p->idle_ev);
        if (e != hipSuccess) {
            (void)hipGetLastError();  // (the sticky error of the failed record)
            e = hipDeviceSynchronize();
        }
        p->launched = false;
        p->last_stream = nullptr;
        if (e != hipSuccess) {
            rh::set_hip_error(e, "wait_idle");
            return RH_ERR_HIP;
        }
    }
    return RH_OK;
}
// In front of the first kernel of a launch on `s`: a handle that moves to another stream waits for what it queued on the old one
// BEFORE anything is queued on the new one (the kernels of the two streams would otherwise share d_gran / d_ctl / d_mix).
rh_status pre_launch(rh_rlm *p, hipStream_t s) {
    if (p->launched && p->last_stream != s) return wait_idle(p);
    return RH_OK;
}
rh_status mark_launch(rh_rlm *p, hipStream_t s) {
    p->launched = true;
    p->last_stream = s;
    return RH_OK;
}

// Predecessor tiles a tile of L frames has to look back at: ||B^(L*J)|| < 2^-40 (older history is
// below f32 resolution of the state); 0 = pole radius too close to 1 for this tile length.
uint32_t look_tiles(const M2 &B, uint64_t L) {
    const M2 BL = mpow(B, L);
    M2 cur = BL;
    for (uint32_t j = 1; j <= (uint32_t)kMaxLook; ++j) {
        if (norm(cur) < 0x1p-40) return j;
        cur = mul(cur, BL);
    }
    return 0;
}
size_t lds_bytes_of(const Variant &v, bool general, uint32_t J) {
    size_t n = (size_t)v.NS * v.KV * 1024;
    if (general) n += 128 + 32 + kMaxLook * 16 + (size_t)((J + 3) / 4) * 1024;
    return n;
}

// Geometry + tables of one plan.  One wave per tile of L = 64*R output frames; the cost of a geometry
// is the most loaded SIMD: ceil(waves per CU / 4) waves, each issuing `per_frame` instructions per
// frame + `per_source` per source one after the other; the issue interval falls with occupancy
// (measured, tools/ubench/valu_rate.hip: 4.3 / 3.0 / 2.7 cycles per wave-instruction at 1 / 2 / 4
// waves per SIMD).
rh_status make_plan(rh_rlm *p, Plan &pl, VariantTab tab, bool general, const rh::ResampleGeom &g, uint32_t want_R, uint32_t want_NS) {
    const M2 A{-(double)p->coeffs[3], -(double)p->coeffs[4], 1.0, 0.0};
    M2 Tm, Ti;
    scan_basis((double)p->coeffs[3], (double)p->coeffs[4], Tm, Ti);
    const M2 B = mul(mul(Tm, A), Ti);
    const uint64_t M = g.out_frames ? g.out_frames : 1;
    const int cus = rh::g_num_cus;
    const double per_frame = 19.0, per_source = general ? 170.0 : 50.0;
    double best = 1e300;
    const Variant *bestV = nullptr;
    uint32_t bestJ = 0;
    for (int R = 1; R <= kMaxR; ++R) {
        if (want_R && (int)want_R != R) continue;
        const uint64_t L = 64ull * R;
        const uint32_t Jr = p->filt ? look_tiles(B, L) : 1;
        if (Jr == 0) continue;
        const uint64_t tiles = (M + L - 1) / L;
        const uint64_t per_cu = (tiles + cus - 1) / cus;
        for (int NS = 2; NS <= 4; ++NS) {
            if (want_NS && (int)want_NS != NS) continue;
            const Variant *v = find_variant(tab, R, kv_needed(L, g.F, g.T, p->cfg.channels), NS);
            if (!v) continue;
            const void *fn = reinterpret_cast<const void *>(p->filt ? v->filt : v->plain);
            const int resident = blocks_per_cu(fn, lds_bytes_of(*v, general, Jr));
            if (resident < 1) continue;
            // tiles beyond the resident set only start when earlier ones finish: legal, but they
            // run as a second pass
            const double passes = std::ceil((double)per_cu / resident);
            const uint64_t on_cu = std::min<uint64_t>(per_cu, resident);
            const double w = std::ceil(on_cu / 4.0);  // waves on the most loaded SIMD
            const double issue = 2.6 + 1.7 / std::pow(w, 1.5);
            double cost = passes * w * (per_frame * R + per_source) * issue;
            cost *= 1.0 + 0.01 * NS;  // among equals prefer the shallower ring (less LDS)
            if (general && R < 8) cost *= 1.3;  // measured: the per-source scan and hand-off want the longer runs
            if (cost < best) {
                best = cost;
                bestV = v;
                bestJ = Jr;
            }
        }
    }
    if (!bestV) return RH_ERR_UNSUPPORTED;
    pl.v = bestV;
    pl.general = general;
    pl.kernel = reinterpret_cast<const void *>(p->filt ? bestV->filt : bestV->plain);
    pl.J = p->filt ? bestJ : 0;
    pl.lds_bytes = (uint32_t)lds_bytes_of(*bestV, general, bestJ);
    pl.resident_per_cu = blocks_per_cu(pl.kernel, pl.lds_bytes);
    // ---- tables (all powers of B = Tm A Tm^-1 are taken in f64 on the host and rounded to f32 once)
    Tables *h = new Tables();
    std::memset(h, 0, sizeof(Tables));
    Uniforms &U = pl.uni;
    std::memset(&U, 0, sizeof(U));
    U.b0 = p->coeffs[0];
    U.c1 = (float)((double)p->coeffs[1] - (double)p->coeffs[0] * (double)p->coeffs[3]);
    U.c2 = (float)((double)p->coeffs[2] - (double)p->coeffs[0] * (double)p->coeffs[4]);
    U.a1 = p->coeffs[3];
    U.a2 = p->coeffs[4];
    put(U.Tm, Tm);
    const uint64_t R = bestV->R, L = 64ull * bestV->R;
    for (int k = 0; k < 4; ++k) put(U.scanM[k], mpow(B, R << k));
    for (int r = 0; r < bestV->R; ++r) {
        const M2 m = mul(mpow(A, r + 1), Ti);  // w[r] = row 0 of A^(r+1) applied to the companion state Ti*z
        U.g[r][0] = (float)m.a;
        U.g[r][1] = (float)m.b;
    }
    for (int l = 0; l < 64; ++l) {
        put(h->laneM[l], mpow(B, R * l));
        put(h->bc15M[l], mpow(B, R * ((l & 15) + 1)));
        put(h->bc31M[l], mpow(B, R * ((l & 31) + 1)));
    }
    {
        const M2 BL = mpow(B, L);
        M2 cur{1, 0, 0, 1};
        for (int j = 0; j < kMaxLook; ++j) {
            put(h->lookM[j], cur);
            cur = mul(cur, BL);
        }
    }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&pl.d_tabs), sizeof(Tables));
    if (e == hipSuccess) e = hipMemcpy(pl.d_tabs, h, sizeof(Tables), hipMemcpyHostToDevice);
    delete h;
    if (e != hipSuccess) {
        rh::set_hip_error(e, "rh_rlm_create tables");
        return e == hipErrorOutOfMemory ? RH_ERR_NOMEM : RH_ERR_HIP;
    }
    return RH_OK;
}

// k_rlm_chunk for the batch that is set (equal lengths): tile boundaries, look-back weights, residency.  Leaves chunk.ok false
// where the kernel does not apply -- the two-kernel form of mix first (or the per-source kernel) runs instead.
rh_status build_chunk(rh_rlm *p) {
    ChunkPlan &c = p->chunk;
    c.ok = false;
    constexpr uint64_t H = 4;
    const uint32_t C = p->cfg.channels;
    if (!p->filt || !p->equal || p->cfg.force_general || p->n_sources < 2 || rh::knob(rh::K_NO_CHUNK) || rh::knob(rh::K_NO_MIX_FIRST)) return RH_OK;
    // The instances, in the order they are tried: chunks of 1024 frames in runs of 18 (stereo: 8 KiB, mono: 4 KiB), then chunks of
    // 512 frames in runs of 18 for converters that make more than 1152 frames of 1024 (ratios up to 2.25: 22.05 -> 48 kHz).
    // RH_CHUNK_HALF (a tuning aid): stereo chunks of 512 frames in runs of 9.
    struct Inst {
        int R, KV;
        const void *fn;
    };
    std::vector<Inst> cand;
    if (C == 2 && rh::knob(rh::K_CHUNK_HALF)) cand.push_back({9, 4, chunk_kernel(9, 2, 4)});
    if (C == 2) {
        cand.push_back({18, 8, chunk_kernel(18, 2, 8)});
        cand.push_back({18, 4, chunk_kernel(18, 2, 4)});
    } else {
        cand.push_back({18, 4, chunk_kernel(18, 1, 4)});
        cand.push_back({18, 2, chunk_kernel(18, 1, 2)});
    }
    const uint64_t Ns = p->eq_frames, M = p->out_frames;
    if (Ns < 2 || (Ns * C) % 4 != 0 || M == 0) return RH_OK;  // (whole 16-byte vectors)
    // the input frame of an output frame (cursor_at / cursor_resolve, and the verbatim last frame)
    const uint64_t F = p->F, T = p->T, cin = p->chunk_in, cout = p->chunk_out;
    auto in_index = [&](uint64_t m) -> uint64_t {
        const uint64_t k = cout ? m / cout : 0, ml = m - k * cout;
        uint64_t il = ml * F / T;
        if (cout && il + 1 >= cin) il = cin - 1;
        const uint64_t i = k * cin + il;
        return i + 1 >= Ns ? Ns - 1 : i;
    };
    std::vector<uint32_t> mlo;
    uint64_t tiles = 0, n_min = ~0ull;
    int R = 0, KV = 0;
    const void *fn = nullptr;
    for (const Inst &in : cand) {
        const uint64_t P = (uint64_t)in.KV * 1024 / (4 * C);
        tiles = (Ns + P - 1) / P;
        if (tiles < 2ull * (uint64_t)rh::g_num_cus || tiles > 0x3fffffffull) continue;  // short rows: more, smaller pieces fill the chip better
        mlo.assign((size_t)tiles + 1, 0u);
        for (uint64_t t = 1; t < tiles; ++t) {  // the first frame whose second tap lies in chunk t or behind it
            uint64_t lo = mlo[(size_t)t - 1], hi = M;
            while (lo < hi) {
                const uint64_t mid = (lo + hi) / 2;
                if (in_index(mid) + 1 >= t * P) hi = mid;
                else lo = mid + 1;
            }
            mlo[(size_t)t] = (uint32_t)lo;
        }
        mlo[(size_t)tiles] = (uint32_t)M;
        bool fits = true;
        n_min = ~0ull;
        for (uint64_t t = 0; t < tiles && fits; ++t) {
            const uint64_t n = mlo[(size_t)t + 1] - mlo[(size_t)t];
            if (n == 0 || n > 64ull * in.R) fits = false;  // a ratio that puts more frames into a chunk than 64 runs hold (or none)
            if (t + 1 < tiles) n_min = std::min(n_min, n);
            if (t > 0 && fits) {  // the two frames the filter looks back at, and the first tap of the first frame: in the 4 frames in front of the chunk
                const uint64_t m = mlo[(size_t)t];
                if (m < 2 || in_index(m - 2) + H < t * P || in_index(m) + 1 < t * P) fits = false;
            }
        }
        if (fits) {
            R = in.R, KV = in.KV, fn = in.fn;
            break;
        }
    }
    if (!fn) return RH_OK;
    if (c.fn != fn) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 64, 0) != hipSuccess) return RH_OK;
        c.resident_per_cu = n < 1 ? -1 : n;
        c.fn = fn;
        c.R = R;
        c.KV = KV;
    }
    if (c.resident_per_cu < 1) return RH_OK;
    c.direct = tiles <= (uint64_t)rh::g_num_cus * (uint64_t)c.resident_per_cu;  // every tile resident at once: no tickets
    const M2 A{-(double)p->coeffs[3], -(double)p->coeffs[4], 1.0, 0.0};
    M2 Tm, Ti;
    scan_basis((double)p->coeffs[3], (double)p->coeffs[4], Tm, Ti);
    const M2 B = mul(mul(Tm, A), Ti);
    const uint32_t J = look_tiles(B, n_min);
    if (J == 0 || J > 32) return RH_OK;
    if ((uint64_t)tiles * J * 16 > (64ull << 20)) return RH_OK;  // (the per-tile look-back table: a filter that forgets slowly over very long rows)
    {
        const rh_status w = wait_idle(p);  // an earlier run may still read the tables
        if (w != RH_OK) return w;
    }
    if (c.tabs_R != R) {
        if (c.d_tabs) RH_HIP_TRY(hipFree(c.d_tabs));
        if (c.d_pow) RH_HIP_TRY(hipFree(c.d_pow));
        if (c.d_uni) RH_HIP_TRY(hipFree(c.d_uni));
        c.d_tabs = nullptr, c.d_pow = nullptr, c.d_uni = nullptr, c.tabs_R = 0;
        Tables *h = new Tables();
        std::memset(h, 0, sizeof(Tables));
        Uniforms &U = c.uni;
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
        float pw[kMaxR + 1][4];
        std::memset(pw, 0, sizeof(pw));
        for (int v = 0; v <= R; ++v) put(pw[v], mpow(B, (uint64_t)v));
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&c.d_tabs), sizeof(Tables));
        if (e == hipSuccess) e = hipMemcpy(c.d_tabs, h, sizeof(Tables), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c.d_pow), sizeof(pw));
        if (e == hipSuccess) e = hipMemcpy(c.d_pow, pw, sizeof(pw), hipMemcpyHostToDevice);
        static_assert(offsetof(Uniforms, Tm) == 20 && offsetof(Uniforms, scanM) == 36 && offsetof(Uniforms, g) == 100, "the kernel reads the Uniforms by float index");
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c.d_uni), sizeof(Uniforms));
        if (e == hipSuccess) e = hipMemcpy(c.d_uni, &U, sizeof(Uniforms), hipMemcpyHostToDevice);
        delete h;
        if (e != hipSuccess) {
            rh::set_hip_error(e, "k_rlm_chunk tables");
            return e == hipErrorOutOfMemory ? RH_ERR_NOMEM : RH_ERR_HIP;
        }
        c.tabs_R = R;
    }
    if ((size_t)tiles > c.cap_tiles) {
        if (c.d_mlo) RH_HIP_TRY(hipFree(c.d_mlo));
        if (c.d_halo) RH_HIP_TRY(hipFree(c.d_halo));
        if (c.d_gran) RH_HIP_TRY(hipFree(c.d_gran));
        c.d_mlo = nullptr, c.d_halo = nullptr, c.d_gran = nullptr, c.cap_tiles = 0;
        RH_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c.d_mlo), ((size_t)tiles + 1) * 4));
        RH_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c.d_halo), (size_t)tiles * 64));
        RH_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c.d_gran), (size_t)tiles * 32));
        RH_HIP_TRY(rh::fill_now(c.d_halo, 0, (size_t)tiles * 64));  // tag 0 = never written (launch tags start at 1)
        RH_HIP_TRY(rh::fill_now(c.d_gran, 0, (size_t)tiles * 32));
        c.cap_tiles = (size_t)tiles;
    }
    const size_t look_floats = (size_t)tiles * J * 4;
    if (look_floats > c.cap_look) {
        if (c.d_look) RH_HIP_TRY(hipFree(c.d_look));
        c.d_look = nullptr, c.cap_look = 0;
        RH_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c.d_look), look_floats * 4));
        c.cap_look = look_floats;
    }
    std::vector<float> look(look_floats, 0.0f);
    {
        // B^d for the few distances that occur (tiles of n_min or n_min + 1 frames, seams aside): memoised
        std::unordered_map<uint64_t, M2> memo;
        for (uint64_t t = 1; t < tiles; ++t)
            for (uint32_t j = 0; j < J && j < t; ++j) {
                const uint64_t d = (uint64_t)mlo[(size_t)t] - mlo[(size_t)(t - j)];
                auto it = memo.find(d);
                if (it == memo.end()) it = memo.emplace(d, mpow(B, d)).first;
                put(&look[((size_t)t * J + j) * 4], it->second);
            }
    }
    RH_HIP_TRY(hipMemcpy(c.d_mlo, mlo.data(), mlo.size() * 4, hipMemcpyHostToDevice));
    RH_HIP_TRY(hipMemcpy(c.d_look, look.data(), look_floats * 4, hipMemcpyHostToDevice));
    c.n_tiles = (uint32_t)tiles;
    c.J = J;
    c.frames = (uint32_t)Ns;
    c.ok = true;
    return RH_OK;
}

// Can the batch that is set take the kernel pair in the tile geometry of `pl`?  (1) the sources that last as long as the
// mix share one length (the lean kernel's end-of-source handling is uniform) and (2) no tile holds many sources that are
// about to end (k_rlm_resid takes a tile's pairs one after the other; batches whose sources all end within a few frames of
// each other stay with k_rlm_wave).
bool pair_ok(rh_rlm *p, const Plan &pl) {
    if (!pl.v || !p->filt || p->equal || p->cfg.force_general || p->h_desc.size() != p->n_sources || rh::knob(rh::K_NO_HYBRID)) return false;
    const uint64_t M = p->out_frames, L = 64ull * pl.v->R, J = pl.J;
    const uint64_t tiles = (M + L - 1) / L;
    if (!tiles) return false;
    uint32_t frames_of_longest = 0, most = 0, ends_from = 0xffffffffu, ends_to = 0;
    std::vector<uint32_t> pairs((size_t)tiles, 0u);
    for (const SrcDesc &d : p->h_desc) {
        if (d.out_frames == M) {
            if (frames_of_longest && frames_of_longest != d.frames) return false;
            frames_of_longest = d.frames;
        } else if (d.out_frames > 0) {
            ends_from = std::min(ends_from, d.out_frames);
            ends_to = std::max(ends_to, d.out_frames);
            const uint64_t t_end = (d.out_frames - 1) / L;                                              // the tile the source ends in
            const uint64_t t_lo = (uint64_t)d.out_frames / L > J ? (uint64_t)d.out_frames / L - J : 0;  // first tile with out_frames < (t+1+J)*L
            for (uint64_t t = t_lo; t <= t_end && t < tiles; ++t) most = std::max(most, ++pairs[(size_t)t]);
        }
    }
    if (!frames_of_longest || most > 24) return false;
    p->rag_frames = frames_of_longest;
    p->rag_pairs_from = ends_from;
    p->rag_pairs_to = ends_to;
    return true;
}

// Point the handle at a plan for the current batch: grid, aggregate table, LDS request.
rh_status activate_plan(rh_rlm *p, Plan *pl) {
    const uint64_t M = p->out_frames;
    const uint64_t L = 64ull * pl->v->R;
    const uint64_t tiles = (M + L - 1) / L;
    if (tiles > 0x7fffffffull) return RH_ERR_UNSUPPORTED;
    const size_t words = (size_t)(p->n_sources + 1) * (tiles + 1) * 4;  // per (source, tile): general kernel (+ its row of summed aggregates) and batch mode; +1: streaming's end-state tile
    if (p->filt && words > p->gran_words) {
        {
            const rh_status w = wait_idle(p);  // a queued launch may still read the old table
            if (w != RH_OK) return w;
        }
        if (p->d_gran) RH_HIP_TRY(hipFree(p->d_gran));
        p->d_gran = nullptr;
        RH_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_gran), words * 8));
        RH_HIP_TRY(rh::fill_now(p->d_gran, 0, words * 8));  // epoch 0 never matches a run
        p->gran_words = words;
    }
#ifdef RH_PHASE_PROFILE
    if (p->d_prof) RH_HIP_TRY(hipFree(p->d_prof));
    p->d_prof = nullptr;
    RH_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_prof), (tiles + 1) * 64));
    RH_HIP_TRY(rh::fill_now(p->d_prof, 0, (tiles + 1) * 64));
#endif
    // The most loaded CU sets the pace: pad the LDS request until the dispatcher cannot put more
    // than ceil(tiles/CUs) waves on any CU.
    p->launch_lds = pl->lds_bytes;
    if (tiles > 0 && !p->cfg.no_balance) {
        const uint64_t per_cu = (tiles + rh::g_num_cus - 1) / rh::g_num_cus;
        if ((int)per_cu <= pl->resident_per_cu) {
            uint32_t want = (uint32_t)(kLdsGranules / per_cu) * kLdsGranule;  // whole granules: exactly per_cu fit
            if (want > 64u * 1024u) want = 64u * 1024u;
            while (want > pl->lds_bytes && blocks_per_cu(pl->kernel, want) < (int)per_cu) want -= kLdsGranule;
            if (want > pl->lds_bytes) p->launch_lds = want;
        }
    }
    p->plan = pl;
    p->n_tiles = (uint32_t)tiles;
    return RH_OK;
}


// Upload h_desc[0..n) to d_srcs on `s` through the page-locked ring.
rh_status upload_descriptors(rh_rlm *p, uint32_t n, hipStream_t s) {
    const int k = p->h_ring_next;
    p->h_ring_next = (k + 1) % rh_rlm::kDescRing;
    if (!p->h_ring[k]) {
        RH_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&p->h_ring[k]), sizeof(SrcDesc) * p->cfg.max_sources, hipHostMallocDefault));
        RH_HIP_TRY(hipEventCreateWithFlags(&p->h_ring_ev[k], hipEventDisableTiming));
    } else {
        RH_HIP_TRY(hipEventSynchronize(p->h_ring_ev[k]));  // the copy that last read this table has run (normally long ago)
    }
    std::memcpy(p->h_ring[k], p->h_desc.data(), sizeof(SrcDesc) * n);
    RH_HIP_TRY(hipMemcpyAsync(p->d_srcs, p->h_ring[k], sizeof(SrcDesc) * n, hipMemcpyHostToDevice, s));
    p->srcs_version += 1;
    RH_HIP_TRY(hipEventRecord(p->h_ring_ev[k], s));
    return RH_OK;
}

}  // namespace rhp

namespace rh {
// `s` has been synchronised and is about to be destroyed (or its scratch released): handles whose launches went there are idle.
void rlm_stream_retired(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_handles_mu);
    for (rh_rlm *p : g_handles)
        if (p->launched && p->last_stream == s) {
            p->launched = false;
            p->last_stream = nullptr;
        }
}
}  // namespace rh

extern "C" {

rh_status rh_rlm_create(rh_rlm **out, const rh_rlm_config *cfg) {
    RH_REQUIRE_INIT();
    if (!out || !cfg || cfg->from_rate == 0 || cfg->to_rate == 0 || cfg->channels == 0 || cfg->max_sources == 0) return RH_ERR_INVALID;
    if (cfg->channels != 1 && cfg->channels != 2) return RH_ERR_UNSUPPORTED;  // mono and stereo frames inside the fused kernels; other layouts: rh_channels_convert / rh_uniform_segments in front
    // from_rate == to_rate: the converter passes through (sample_rate.rs:133-136): filter + ordered mix only
    if (cfg->max_in_frames >= (1ull << 29)) return RH_ERR_UNSUPPORTED;  // 32-bit byte offsets inside a source
    rh::ResampleGeom g;
    rh_status st = rh::make_resample_geom(cfg->max_in_frames, cfg->from_rate, cfg->to_rate, cfg->channels, cfg->span_len, &g);
    if (st != RH_OK) return st;
    if (2ull * g.F > 9ull * g.T) return RH_ERR_UNSUPPORTED;  // staging is sized for ratios <= 4.5: 192 kHz -> 44.1 kHz (the unfused ops cover the rest)
    if (g.out_frames >= (1ull << 31)) return RH_ERR_UNSUPPORTED;  // 32-bit frame indices in the kernels
    rh_rlm *p = new rh_rlm();
    p->cfg = *cfg;
    p->F = g.F;
    p->T = g.T;
    p->chunk_in = g.n_chunks > 1 ? g.chunk_in : 0;
    p->chunk_out = g.n_chunks > 1 ? g.chunk_out : 0;
    p->filt = cfg->filter_kind >= 0;
    if (cfg->filter_first && p->filt) {  // mixer.add(src.low_pass(f)): coefficients at the SOURCE rate; the converter behind it runs bare
        if (cfg->filter_kind == 2) {
            for (int k = 0; k < 5; ++k) p->pre_coeffs[k] = cfg->custom_coeffs[k];
        } else {
            st = rh_biquad_coeffs(cfg->filter_kind, cfg->filter_freq, cfg->filter_q, cfg->from_rate, p->pre_coeffs);
            if (st != RH_OK) {
                delete p;
                return st;
            }
        }
        p->pre_filter = true;
        p->filt = false;
    }
    if (p->pre_filter) {
        p->coeffs[0] = 1.f;
        p->coeffs[1] = p->coeffs[2] = p->coeffs[3] = p->coeffs[4] = 0.f;
    } else if (cfg->filter_kind == 2) {  // coefficients given ({b0,b1,b2,a1,a2}, already divided by a0)
        for (int k = 0; k < 5; ++k) p->coeffs[k] = cfg->custom_coeffs[k];
        const double a1 = p->coeffs[3], a2 = p->coeffs[4];  // stability triangle: the look-back needs a decaying filter
        if (!(std::fabs(a2) < 1.0 && std::fabs(a1) < 1.0 + a2)) {
            delete p;
            return RH_ERR_UNSUPPORTED;
        }
    } else if (p->filt) {
        st = rh_biquad_coeffs(cfg->filter_kind, cfg->filter_freq, cfg->filter_q, cfg->to_rate, p->coeffs);
        if (st != RH_OK) {
            delete p;
            return st;
        }
    } else {
        p->coeffs[0] = 1.f;
        p->coeffs[1] = p->coeffs[2] = p->coeffs[3] = p->coeffs[4] = 0.f;
    }
    // two plans: equal-length batches (k_rlm_fast) and ragged ones (k_rlm_wave).  The geometry
    // overrides of the config address the fast plan; the general plan follows them when it can.
    const bool mono = cfg->channels == 1;
    const VariantTab t_fast = variant_tab(kTabFast, mono), t_wave = variant_tab(kTabWave, mono);
    st = make_plan(p, p->fast, t_fast, false, g, cfg->frames_per_lane, cfg->ring_stages);
    if (st == RH_ERR_UNSUPPORTED && mono && (cfg->frames_per_lane || cfg->ring_stages)) st = make_plan(p, p->fast, t_fast, false, g, 0, 0);  // (fewer mono tile sizes are built)
    if (st == RH_OK) {
        st = make_plan(p, p->wave, t_wave, true, g, cfg->frames_per_lane, cfg->ring_stages);
        if (st == RH_ERR_UNSUPPORTED && (cfg->frames_per_lane || cfg->ring_stages)) st = make_plan(p, p->wave, t_wave, true, g, 0, 0);
    } else if (st == RH_ERR_UNSUPPORTED && (cfg->frames_per_lane || cfg->ring_stages)) {
        st = RH_ERR_INVALID;
    }
    if (st == RH_OK && p->filt) {  // optional.  It follows the overrides when it has that geometry; otherwise the longest runs that
        // still give every CU three tiles (measured on 256 sources of [N/2, N] frames: 0.366 / 0.352 / 0.336 / 0.328 / 0.319 / 0.319 ms for
        // 6 / 8 / 10 / 12 / 14 / 18 frames per lane -- the first half sums first, so a tile's fixed costs are all that is left to amortise)
        rh_status ps = RH_ERR_UNSUPPORTED;
        if (cfg->frames_per_lane || cfg->ring_stages) ps = make_plan(p, p->pair, variant_tab(kTabRag, mono), false, g, cfg->frames_per_lane, cfg->ring_stages);
        for (uint32_t R : {18u, 14u, 12u, 10u}) {
            if (ps == RH_OK) break;
            const uint64_t M = g.out_frames ? g.out_frames : 1, tiles = (M + 64ull * R - 1) / (64ull * R);
            if (tiles >= 3ull * (uint64_t)rh::g_num_cus) ps = make_plan(p, p->pair, variant_tab(kTabRag, mono), false, g, R, 2);
        }
        if (ps != RH_OK) ps = make_plan(p, p->pair, variant_tab(kTabRag, mono), false, g, 0, 0);
        if (ps != RH_OK) p->pair.v = nullptr;
    }
    hipError_t e = hipSuccess;
    if (st == RH_OK) {
        e = hipMalloc(reinterpret_cast<void **>(&p->d_srcs), sizeof(SrcDesc) * cfg->max_sources);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&p->d_ctl), 128 * 9);  // control words + 8 ticket counters on their own cache lines
        if (e == hipSuccess) e = rh::fill_now(p->d_ctl, 0, 128 * 9);  // the ticket counter: a late fill would renumber tiles in mid-launch
        if (e != hipSuccess) {
            rh::set_hip_error(e, "rh_rlm_create");
            st = e == hipErrorOutOfMemory ? RH_ERR_NOMEM : RH_ERR_HIP;
        }
    }
    if (st != RH_OK) {
        rh_rlm_destroy(p);
        return st;
    }
    p->plan = &p->fast;
    {
        std::lock_guard<std::mutex> lk(g_handles_mu);
        g_handles.push_back(p);
    }
    *out = p;
    return RH_OK;
}

rh_status rh_rlm_set_exclusive(rh_rlm *p, int32_t exclusive) {
    if (!p) return RH_ERR_INVALID;
    p->exclusive = exclusive != 0;
    return RH_OK;
}

rh_status rh_rlm_set_mix_first(rh_rlm *p, int32_t enable) {
    if (!p) return RH_ERR_INVALID;
    p->mix_first_on = enable != 0;
    return RH_OK;
}

rh_status rh_rlm_destroy(rh_rlm *p) {
    if (!p) return RH_OK;
    (void)wait_idle(p);  // nothing of this handle may still run when its tables go
    {
        std::lock_guard<std::mutex> lk(g_handles_mu);
        g_handles.erase(std::remove(g_handles.begin(), g_handles.end(), p), g_handles.end());
    }
#if defined(RH_CHUNK_DIAG) && RH_CHUNK_DIAG == 3
    if (p->d_ctl && p->chunk.ok) {
        uint32_t h[8] = {0};
        (void)hipMemcpy(h, p->d_ctl + 8, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "chunk phases (cycles summed over tiles and launches): image+halo %u  run %u  scan+publish %u  look-back %u  correction+stores %u\n", h[0], h[1], h[2], h[3], h[4]);
    }
#endif
    for (rh_rlm::FilterClass &c : p->cls)
        if (c.h) (void)rh_rlm_destroy(c.h);
    p->cls.clear();
    for (hipStream_t st : p->cls_streams) {
        (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
    }
    for (hipEvent_t e : p->cls_done) (void)hipEventDestroy(e);
    if (p->cls_fork) (void)hipEventDestroy(p->cls_fork);
    if (p->d_cls_rows) (void)hipFree(p->d_cls_rows);
    sblk_free(p);
    if (p->idle_ev) (void)hipEventDestroy(p->idle_ev);
    bool fast_in_tried = false, wave_in_tried = false, pair_in_tried = false;
    for (Plan &c : p->tried) {
        fast_in_tried = fast_in_tried || c.d_tabs == p->fast.d_tabs;
        wave_in_tried = wave_in_tried || c.d_tabs == p->wave.d_tabs;
        pair_in_tried = pair_in_tried || c.d_tabs == p->pair.d_tabs;
        if (c.d_tabs) (void)hipFree(c.d_tabs);
    }
    if (p->fast.d_tabs && !fast_in_tried) (void)hipFree(p->fast.d_tabs);
    if (p->wave.d_tabs && !wave_in_tried) (void)hipFree(p->wave.d_tabs);
    if (p->pair.d_tabs && !pair_in_tried) (void)hipFree(p->pair.d_tabs);
    if (p->d_srcs) (void)hipFree(p->d_srcs);
    if (p->d_gran) (void)hipFree(p->d_gran);
    if (p->d_ctl) (void)hipFree(p->d_ctl);
    if (p->d_mix) (void)hipFree(p->d_mix);
    if (p->chunk.d_tabs) (void)hipFree(p->chunk.d_tabs);
    if (p->chunk.d_pow) (void)hipFree(p->chunk.d_pow);
    if (p->chunk.d_uni) (void)hipFree(p->chunk.d_uni);
    if (p->chunk.d_mlo) (void)hipFree(p->chunk.d_mlo);
    if (p->chunk.d_look) (void)hipFree(p->chunk.d_look);
    if (p->chunk.d_halo) (void)hipFree(p->chunk.d_halo);
    if (p->chunk.d_gran) (void)hipFree(p->chunk.d_gran);
    if (p->d_prof) (void)hipFree(p->d_prof);
    if (p->d_replay) (void)hipFree(p->d_replay);
    for (int k = 0; k < 2; ++k)
        if (p->d_w[k]) (void)hipFree(p->d_w[k]);
    for (int k = 0; k < rh_rlm::kDescRing; ++k) {
        if (p->h_ring_ev[k]) (void)hipEventDestroy(p->h_ring_ev[k]);
        if (p->h_ring[k]) (void)hipHostFree(p->h_ring[k]);
    }
    delete p;
    return RH_OK;
}

// on_stream != nullptr: the table travels on that stream through the page-locked ring (ordered behind the launches already
// queued there, no host synchronisation) -- what rh_biquad mode 1 does per call; nullptr: the synchronous form of the C ABI.
static rh_status set_sources_impl(rh_rlm *p, const float *const *srcs_host, const uint64_t *in_frames_host, uint32_t n_sources, const hipStream_t *on_stream) {
    RH_REQUIRE_INIT();
    if (!p || (n_sources && (!srcs_host || !in_frames_host))) return RH_ERR_INVALID;
    if (n_sources > p->cfg.max_sources) return RH_ERR_CAPACITY;
    std::vector<SrcDesc> &h = p->h_desc;
    h.resize(n_sources);
    uint64_t M = 0;
    bool equal = true;
    for (uint32_t s = 0; s < n_sources; ++s) {
        if (in_frames_host[s] > p->cfg.max_in_frames) return RH_ERR_CAPACITY;
        if (in_frames_host[s] && (!srcs_host[s] || (reinterpret_cast<uintptr_t>(srcs_host[s]) & 15u))) return RH_ERR_INVALID;
        rh::ResampleGeom g;
        rh_status st = rh::make_resample_geom(in_frames_host[s], p->cfg.from_rate, p->cfg.to_rate, p->cfg.channels, p->cfg.span_len, &g);
        if (st != RH_OK) return st;
        if (g.out_frames >= (1ull << 31)) return RH_ERR_UNSUPPORTED;  // 32-bit frame indices in the kernels
        h[s] = SrcDesc{srcs_host[s], (uint32_t)in_frames_host[s], (uint32_t)g.out_frames, s < p->gains.size() ? p->gains[s] : 1.0f, {0, 0, 0}};
        if (g.out_frames > M) M = g.out_frames;
        equal = equal && in_frames_host[s] == in_frames_host[0];
    }
    if (on_stream && p->launched && p->last_stream == *on_stream) {
        if (n_sources) {
            const rh_status up = upload_descriptors(p, n_sources, *on_stream);
            if (up != RH_OK) return up;
        }
    } else {
        const rh_status w = wait_idle(p);  // an earlier run of this handle may still be reading the table
        if (w != RH_OK) return w;
        if (n_sources) RH_HIP_TRY(hipMemcpy(p->d_srcs, h.data(), sizeof(SrcDesc) * n_sources, hipMemcpyHostToDevice));
        p->srcs_version += 1;
    }
    p->equal = equal;
    p->eq_frames = n_sources ? (uint32_t)in_frames_host[0] : 0;
    p->n_sources = n_sources;
    p->out_frames = M;
    p->chunk.ok = false;
    // equal-length batch: the merged-state kernel; otherwise the general one
    if (equal && !p->cfg.force_general) {
        const rh_status st = activate_plan(p, &p->fast);
        if (st != RH_OK || on_stream) return st;
        return build_chunk(p);
    }
    // different lengths + filter: almost every (tile, source) pair is "stable" and goes through the lean kernel of the pair
    return activate_plan(p, pair_ok(p, p->pair) ? &p->pair : &p->wave);
}

// The sources dealt over the filter classes (rh_rlm_set_filters).  Classes are kept across calls (their tables and plans belong to
// their filter); a class that has no member this time keeps its handle and is skipped by the run.
static rh_status set_sources_classes(rh_rlm *p, const float *const *srcs_host, const uint64_t *in_frames_host, uint32_t n_sources) {
    if (n_sources > p->cfg.max_sources) return RH_ERR_CAPACITY;
    if (n_sources && (!srcs_host || !in_frames_host)) return RH_ERR_INVALID;
    for (rh_rlm::FilterClass &c : p->cls) c.members.clear();
    p->n_sources = 0;  // (an error below leaves a handle without sources, not one whose classes and counts disagree: a run then fails cleanly)
    p->out_frames = 0;
    const rh_rlm::FilterSpec own{p->cfg.filter_kind == 2 ? 0 : p->cfg.filter_kind, p->cfg.filter_freq, p->cfg.filter_q};
    for (uint32_t s = 0; s < n_sources; ++s) {
        const rh_rlm::FilterSpec f = s < p->filters.size() ? p->filters[s] : own;
        size_t k = 0;
        while (k < p->cls.size() && !(p->cls[k].spec == f)) ++k;
        if (k == p->cls.size()) {
            rh_rlm::FilterClass c;
            c.spec = f;
            rh_rlm_config cfg = p->cfg;
            cfg.filter_kind = f.kind < 0 ? -1 : f.kind;
            cfg.filter_freq = f.freq;
            cfg.filter_q = f.q;
            const rh_status st = rh_rlm_create(&c.h, &cfg);
            if (st != RH_OK) return st;
            p->cls.push_back(c);
        }
        p->cls[k].members.push_back(s);
    }
    uint64_t M = 0;
    std::vector<const float *> ptrs;
    std::vector<uint64_t> frames;
    std::vector<float> gains;
    for (rh_rlm::FilterClass &c : p->cls) {
        ptrs.clear(), frames.clear(), gains.clear();
        for (uint32_t s : c.members) {
            ptrs.push_back(srcs_host[s]);
            frames.push_back(in_frames_host[s]);
            gains.push_back(s < p->gains.size() ? p->gains[s] : 1.0f);
        }
        c.h->exclusive = p->exclusive;
        c.h->mix_first_on = p->mix_first_on;
        rh_status st = rh_rlm_set_gains(c.h, gains.data(), (uint32_t)gains.size());
        if (st == RH_OK) st = rh_rlm_set_sources(c.h, ptrs.data(), frames.data(), (uint32_t)ptrs.size());
        if (st != RH_OK) return st;
        c.out_frames = c.h->out_frames;
        if (c.out_frames > M) M = c.out_frames;
    }
    p->n_sources = n_sources;
    p->out_frames = M;
    p->chunk.ok = false;
    return RH_OK;
}

rh_status rh_rlm_set_sources(rh_rlm *p, const float *const *srcs_host, const uint64_t *in_frames_host, uint32_t n_sources) {
    if (p && !p->filters.empty()) {
        RH_REQUIRE_INIT();
        return set_sources_classes(p, srcs_host, in_frames_host, n_sources);
    }
    if (p && !p->cls.empty()) {  // back to the handle's one filter
        for (rh_rlm::FilterClass &c : p->cls)
            if (c.h) (void)rh_rlm_destroy(c.h);
        p->cls.clear();
    }
    return set_sources_impl(p, srcs_host, in_frames_host, n_sources, nullptr);
}

rh_status rh_rlm_set_filters(rh_rlm *p, const int32_t *kinds_host, const uint32_t *freqs_host, const float *qs_host, uint32_t n) {
    RH_REQUIRE_INIT();
    if (!p || (n && (!kinds_host || !freqs_host || !qs_host)) || n > p->cfg.max_sources) return RH_ERR_INVALID;
    if (p->pre_filter || p->st_on || p->cfg.filter_kind == 2) return RH_ERR_UNSUPPORTED;  // filter_first / custom-coefficient handles and running streams keep their one filter
    std::vector<rh_rlm::FilterSpec> f(n);
    for (uint32_t s = 0; s < n; ++s) {
        if (kinds_host[s] > 1) return RH_ERR_INVALID;  // -1 none, 0 low_pass, 1 high_pass
        f[s] = rh_rlm::FilterSpec{kinds_host[s] < 0 ? -1 : kinds_host[s], kinds_host[s] < 0 ? 0u : freqs_host[s], kinds_host[s] < 0 ? 0.f : qs_host[s]};
        if (f[s].kind >= 0) {  // refuse here what rh_rlm_create would refuse at the next set_sources
            float c5[5];
            const rh_status st = rh_biquad_coeffs(f[s].kind, f[s].freq, f[s].q, p->cfg.to_rate, c5);
            if (st != RH_OK) return st;
        }
    }
    p->filters = std::move(f);
    p->n_sources = 0;  // the sources are dealt over the classes by the next rh_rlm_set_sources
    p->out_frames = 0;
    return RH_OK;
}

rh_status rh_rlm_set_gains(rh_rlm *p, const float *gains_host, uint32_t n) {
    RH_REQUIRE_INIT();
    if (!p || (n && !gains_host) || n > p->cfg.max_sources) return RH_ERR_INVALID;
    p->gains.assign(gains_host, gains_host + n);
    p->srcs_version += 1;  // (a stream that reuses its table on the device uploads it again: the gains travel in it)
    if (!p->cls.empty()) {  // per-source filters: every class takes the factors of its members
        std::vector<float> g;
        for (rh_rlm::FilterClass &c : p->cls) {
            g.clear();
            for (uint32_t s : c.members) g.push_back(s < n ? gains_host[s] : 1.0f);
            const rh_status st = rh_rlm_set_gains(c.h, g.data(), (uint32_t)g.size());
            if (st != RH_OK) return st;
        }
        return RH_OK;
    }
    if (!p->st_on && p->n_sources && p->h_desc.size() == p->n_sources) {  // sources already set: refresh their descriptors
        for (uint32_t s = 0; s < p->n_sources; ++s) p->h_desc[s].gain = s < n ? gains_host[s] : 1.0f;
        {
            const rh_status w = wait_idle(p);
            if (w != RH_OK) return w;
        }
        RH_HIP_TRY(hipMemcpy(p->d_srcs, p->h_desc.data(), sizeof(SrcDesc) * p->n_sources, hipMemcpyHostToDevice));
        p->srcs_version += 1;
    }
    return RH_OK;
}

// One launch (or pair) per filter class into the class's row, then the classes' mixes summed in order of first appearance.
static rh_status run_classes(rh_rlm *p, float *dst, uint64_t out_capacity_frames, uint64_t *out_frames, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (out_frames) *out_frames = p->out_frames;
    if (p->out_frames == 0) return RH_OK;
    if (!dst || (reinterpret_cast<uintptr_t>(dst) & 15u)) return RH_ERR_INVALID;
    if (out_capacity_frames < p->out_frames) return RH_ERR_CAPACITY;
    const uint32_t C = p->cfg.channels;
    std::vector<rh_rlm::FilterClass *> live;
    for (rh_rlm::FilterClass &c : p->cls)
        if (!c.members.empty() && c.out_frames) live.push_back(&c);
    if (live.empty()) return RH_OK;
    p->cls_one_launch = false;
    if (live.size() == 1) return rh_rlm_run(live[0]->h, dst, out_capacity_frames, nullptr, stream);
    const size_t row = (size_t)((p->out_frames * C + 3) & ~3ull);
    if (row > p->cls_row_floats || live.size() > p->cls_rows) {
        const rh_status w = wait_idle(p);
        if (w != RH_OK) return w;
        for (rh_rlm::FilterClass *c : live) {  // (the rows are read by the sum behind the classes' launches: those first)
            const rh_status wc = wait_idle(c->h);
            if (wc != RH_OK) return wc;
        }
        if (p->d_cls_rows) RH_HIP_TRY(hipFree(p->d_cls_rows));
        p->d_cls_rows = nullptr;
        p->cls_row_floats = std::max(row, p->cls_row_floats);
        p->cls_rows = std::max(live.size(), p->cls_rows);
        RH_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_cls_rows), p->cls_row_floats * p->cls_rows * sizeof(float)));
    }
    std::vector<const float *> ptrs;
    std::vector<uint64_t> start, len;
    hipStream_t s0 = rh::as_stream(stream);
    // Measured (profiles/r05_per_class.txt: 4 classes x 64 sources x 1 Mi frames): one after the other 0.368 ms, side by side 0.40-0.42 -- four
    // launches that each want the whole chip's bandwidth get in each other's way (and their tiles go by ticket then).  So: one after the other;
    // RH_CLASSES_SIDE_BY_SIDE=1 keeps the other form selectable.
    const bool side_by_side = rh::knob(rh::K_CLASSES_SIDE_BY_SIDE) != nullptr;
    // Round 6: ONE launch that walks the classes (k_rlm_chunk_multi: class k's workgroups behind class k-1's, every class with its own arguments,
    // tables and tickets) where every class takes the k_rlm_chunk path: what four launches lose between them -- each drains the chip and ramps
    // up again -- stays inside one grid.  RH_CLASSES_ONE_BY_ONE=1: the launches of round 5.
    if (!side_by_side) {
        std::vector<rh_rlm *> hs;
        std::vector<float *> rows;
        for (size_t k = 0; k < live.size(); ++k) {
            hs.push_back(live[k]->h);
            rows.push_back(p->d_cls_rows + k * p->cls_row_floats);
        }
        bool taken = false, summed = false;
        const rh_status st = chunk_launch_classes(hs.data(), rows.data(), p->cls_row_floats / C, (uint32_t)hs.size(), dst, stream, &taken, &summed);
        if (st != RH_OK) return st;
        p->cls_one_launch = taken;
        if (summed) return mark_launch(p, rh::as_stream(stream));  // (the launch added the classes' mixes itself)
        if (taken) {
            for (size_t k = 0; k < live.size(); ++k) {
                ptrs.push_back(rows[k]);
                start.push_back(0);
                len.push_back(live[k]->out_frames * C);
            }
            const rh_status sm = rh_mix_sum(dst, p->out_frames * C, ptrs.data(), start.data(), len.data(), (uint32_t)ptrs.size(), stream);
            if (sm != RH_OK) return sm;
            return mark_launch(p, rh::as_stream(stream));
        }
    }
    if (side_by_side) {
        while (p->cls_streams.size() + 1 < live.size()) {
            hipStream_t ns = nullptr;
            hipEvent_t ne = nullptr;
            RH_HIP_TRY(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
            p->cls_streams.push_back(ns);
            RH_HIP_TRY(hipEventCreateWithFlags(&ne, hipEventDisableTiming));
            p->cls_done.push_back(ne);
        }
        if (!p->cls_fork) RH_HIP_TRY(hipEventCreateWithFlags(&p->cls_fork, hipEventDisableTiming));
        RH_HIP_TRY(hipEventRecord(p->cls_fork, s0));  // what the caller queued in front (the sources' samples) is in front of every class
    }
    for (size_t k = 0; k < live.size(); ++k) {
        float *r = p->d_cls_rows + k * p->cls_row_floats;
        hipStream_t sk = s0;
        if (side_by_side && k > 0) {
            sk = p->cls_streams[k - 1];
            RH_HIP_TRY(hipStreamWaitEvent(sk, p->cls_fork, 0));
        }
        if (side_by_side) live[k]->h->exclusive = false;  // other classes' kernels share the CUs: tiles by ticket (rh_rlm_set_exclusive)
        const rh_status st = rh_rlm_run(live[k]->h, r, p->cls_row_floats / C, nullptr, reinterpret_cast<rh_stream>(sk));
        if (st != RH_OK) return st;
        if (side_by_side && k > 0) RH_HIP_TRY(hipEventRecord(p->cls_done[k - 1], sk));
        ptrs.push_back(r);
        start.push_back(0);
        len.push_back(live[k]->out_frames * C);
    }
    if (side_by_side)
        for (size_t k = 1; k < live.size(); ++k) RH_HIP_TRY(hipStreamWaitEvent(s0, p->cls_done[k - 1], 0));
    const rh_status st = rh_mix_sum(dst, p->out_frames * C, ptrs.data(), start.data(), len.data(), (uint32_t)ptrs.size(), stream);
    if (st != RH_OK) return st;
    return mark_launch(p, rh::as_stream(stream));
}

rh_status rh_rlm_run(rh_rlm *p, float *dst, uint64_t out_capacity_frames, uint64_t *out_frames, rh_stream stream) {
    if (!p) return RH_ERR_INVALID;
    if (!p->cls.empty()) return run_classes(p, dst, out_capacity_frames, out_frames, stream);
    return rh_rlm_run_subset(p, 0, p->n_sources, dst, out_capacity_frames, out_frames, stream);
}

rh_status rh_rlm_run_subset(rh_rlm *p, uint32_t first, uint32_t count, float *dst, uint64_t out_capacity_frames, uint64_t *out_frames, rh_stream stream) {
    return rlm_launch(p, first, count, dst, out_capacity_frames, out_frames, stream, 0, 0);
}

rh_status rh_rlm_run_batch(rh_rlm *p, float *dst, uint64_t dst_stride_frames, uint64_t *out_frames, rh_stream stream) {
    if (!p) return RH_ERR_INVALID;
    if (p->plan != &p->fast || p->cfg.channels != 2) return RH_ERR_UNSUPPORTED;  // equal-length stereo sources only
    if (p->n_sources > 1 && (dst_stride_frames < p->out_frames || (dst_stride_frames * 2) % 4 != 0)) return RH_ERR_INVALID;
    return rlm_launch(p, 0, p->n_sources, dst, dst_stride_frames, out_frames, stream, p->n_sources, dst_stride_frames * 2);
}

rh_status rh_rlm_autotune(rh_rlm *p, float *dst, uint64_t out_capacity_frames, rh_stream stream, uint32_t *frames_per_lane, uint32_t *ring_stages) {
    RH_REQUIRE_INIT();
    if (!p || !dst) return RH_ERR_INVALID;
    if (!p->cls.empty()) {  // per-source filters: every class finds its own geometry (the last one's is reported)
        for (rh_rlm::FilterClass &c : p->cls) {
            if (c.members.empty()) continue;
            const rh_status st = rh_rlm_autotune(c.h, dst, out_capacity_frames, stream, frames_per_lane, ring_stages);
            if (st != RH_OK) return st;
        }
        return RH_OK;
    }
    if (p->out_frames > 0 && out_capacity_frames >= p->out_frames) {
        const bool general = p->plan == &p->wave, is_pair = p->plan == &p->pair;
        Plan &slot = is_pair ? p->pair : general ? p->wave : p->fast;
        rh::ResampleGeom g;
        rh_status st = rh::make_resample_geom((general || is_pair) ? p->cfg.max_in_frames : p->eq_frames, p->cfg.from_rate, p->cfg.to_rate, p->cfg.channels, p->cfg.span_len, &g);
        if (st != RH_OK) return st;
        hipStream_t s = rh::as_stream(stream);
        hipEvent_t e0, e1;
        RH_HIP_TRY(hipEventCreate(&e0));
        RH_HIP_TRY(hipEventCreate(&e1));
        auto time_current = [&](float &ms) -> rh_status {
            rh_status r = rh_rlm_run(p, dst, out_capacity_frames, nullptr, stream);  // warm-up
            if (r != RH_OK) return r;
            ms = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                RH_HIP_TRY(hipEventRecord(e0, s));
                r = rh_rlm_run(p, dst, out_capacity_frames, nullptr, stream);
                if (r != RH_OK) return r;
                RH_HIP_TRY(hipEventRecord(e1, s));
                RH_HIP_TRY(hipEventSynchronize(e1));
                float t = 0.f;
                RH_HIP_TRY(hipEventElapsedTime(&t, e0, e1));
                if (t < ms) ms = t;
            }
            return RH_OK;
        };
        float best_ms = 0.f;
        st = time_current(best_ms);
        Plan best = slot;
        p->tried.reserve(128);
        {  // every table the handle ever owned is freed through `tried`
            bool have = false;
            for (const Plan &c : p->tried) have = have || c.d_tabs == slot.d_tabs;
            if (!have) p->tried.push_back(slot);
        }
        for (int R = 2; R <= kMaxR && st == RH_OK; ++R) {
            for (int NS = 2; NS <= 3; ++NS) {
                if (R == best.v->R && NS == best.v->NS) continue;
                Plan cand;
                const bool mono = p->cfg.channels == 1;
                if ((is_pair   ? make_plan(p, cand, variant_tab(kTabRag, mono), false, g, (uint32_t)R, (uint32_t)NS)
                     : general ? make_plan(p, cand, variant_tab(kTabWave, mono), true, g, (uint32_t)R, (uint32_t)NS)
                               : make_plan(p, cand, variant_tab(kTabFast, mono), false, g, (uint32_t)R, (uint32_t)NS)) != RH_OK)
                    continue;
                if (is_pair && !pair_ok(p, cand)) {  // this tile size would put too many ending sources into one tile
                    p->tried.push_back(cand);
                    continue;
                }
                p->tried.push_back(cand);
                const uint64_t tiles = (p->out_frames + 64ull * R - 1) / (64ull * R);
                const uint64_t per_cu = (tiles + rh::g_num_cus - 1) / rh::g_num_cus;
                if ((int)per_cu > cand.resident_per_cu && !is_pair) continue;  // would run in passes: never the fastest (a ragged batch's tiles are
                                                                                // unequal: there the later ones fill in behind the heavy ones)
                slot = cand;
                if ((st = activate_plan(p, &slot)) != RH_OK) break;
                float ms = 0.f;
                if ((st = time_current(ms)) != RH_OK) break;
                if (rh::knob(rh::K_AUTOTUNE_LOG)) std::fprintf(stderr, "rh_rlm_autotune: %d frames per lane, %d KiB x %d stages: %.4f ms (best so far %.4f)\n", R, cand.v->KV, NS, ms, best_ms);
                if (ms < best_ms * 0.99f) {  // a candidate has to win by more than the run-to-run noise (ties keep the earlier, shallower one)
                    best_ms = ms;
                    best = cand;
                }
            }
        }
        slot = best;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (st != RH_OK) return st;
        st = activate_plan(p, &slot);
        if (st != RH_OK) return st;
    }
    if (frames_per_lane) *frames_per_lane = (uint32_t)p->plan->v->R;
    if (ring_stages) *ring_stages = (uint32_t)p->plan->v->NS;
    return RH_OK;
}

rh_status rh_rlm_last_status(rh_rlm *p) {
    RH_REQUIRE_INIT();
    if (!p) return RH_ERR_INVALID;
    for (rh_rlm::FilterClass &c : p->cls) {  // per-source filters: the kernels ran on the classes' handles
        const rh_status st = rh_rlm_last_status(c.h);
        if (st != RH_OK) return st;
    }
    {
        const rh_status w = wait_idle(p);  // every launch of this handle has completed: the words below are final
        if (w != RH_OK) return w;
    }
    uint32_t ctl[2] = {0, 0};
    RH_HIP_TRY(hipMemcpy(ctl, p->d_ctl, 8, hipMemcpyDeviceToHost));
    if (ctl[1]) {  // sticky until read
        RH_HIP_TRY(rh::fill_now(p->d_ctl + 1, 0, 4));
        return RH_ERR_TIMEOUT;
    }
    return RH_OK;
}

rh_status rh_rlm_late_carries(rh_rlm *p, uint64_t *count) {
    RH_REQUIRE_INIT();
    if (!p || !count) return RH_ERR_INVALID;
    {
        const rh_status w = wait_idle(p);
        if (w != RH_OK) return w;
    }
    uint32_t v[2] = {0, 0};
    RH_HIP_TRY(hipMemcpy(v, p->d_ctl + 2, 8, hipMemcpyDeviceToHost));
    RH_HIP_TRY(rh::fill_now(p->d_ctl + 2, 0, 8));
    *count = v[0] | ((uint64_t)v[1] << 32);  // high word: empty polls (RH_PHASE_PROFILE builds)
    return RH_OK;
}

rh_status rh_rlm_phase_cycles(rh_rlm *p, double out8[8]) {
    RH_REQUIRE_INIT();
    if (!p || !out8) return RH_ERR_INVALID;
    if (!p->d_prof) return RH_ERR_UNSUPPORTED;  // not an RH_PHASE_PROFILE build
    std::vector<unsigned long long> h((size_t)p->n_tiles * 8);
    RH_HIP_TRY(hipMemcpy(h.data(), p->d_prof, h.size() * 8, hipMemcpyDeviceToHost));
    if (const char *path = rh::knob(rh::K_PROF_DUMP)) {  // raw [tiles][8] u64 for tools/prof_tiles.py
        if (FILE *f = fopen(path, "wb")) {
            fwrite(h.data(), 8, h.size(), f);
            fclose(f);
        }
    }
    for (int i = 0; i < 8; ++i) out8[i] = 0.0;
    for (uint32_t t = 0; t < p->n_tiles; ++t)
        for (int i = 0; i < 8; ++i) out8[i] += (double)h[(size_t)t * 8 + i];
    for (int i = 0; i < 8; ++i) out8[i] /= p->n_tiles ? p->n_tiles : 1;
    return RH_OK;
}

rh_status rh_rlm_geometry(rh_rlm *p, rh_rlm_geometry_info *info) {
    if (!p || !info) return RH_ERR_INVALID;
    if (!p->cls.empty()) {  // per-source filters: the geometry of the largest class
        const rh_rlm::FilterClass *best = nullptr;
        for (const rh_rlm::FilterClass &c : p->cls)
            if (!best || c.members.size() > best->members.size()) best = &c;
        const rh_status st = rh_rlm_geometry(best->h, info);
        if (st == RH_OK && p->cls_one_launch) info->mix_first = 3u;
        return st;
    }
    const Plan &pl = *p->plan;
    info->threads = 64;
    info->frames_per_lane = (uint32_t)pl.v->R;
    info->ring_stages = (uint32_t)pl.v->NS;
    info->stage_kib = (uint32_t)pl.v->KV;
    info->lds_bytes = p->launch_lds ? p->launch_lds : pl.lds_bytes;
    info->lookback_tiles = pl.J;
    info->resident_waves_per_cu = (uint32_t)pl.resident_per_cu;
    info->n_tiles = p->n_tiles;
    info->general_kernel = (pl.general || p->plan == &p->pair) ? 1u : 0u;
    info->ragged_pair = p->plan == &p->pair ? 1u : 0u;
    info->mix_first = (p->pre_filter && p->plan == &p->fast) ? 1u : mix_first_applies(p, pl, p->n_sources, false, false) ? (p->chunk.ok ? 2u : 1u) : 0u;
    return RH_OK;
}

}  // extern "C"

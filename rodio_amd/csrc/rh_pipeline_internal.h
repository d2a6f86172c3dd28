// rh_pipeline_internal.h -- what the translation units of the fused path share (rh_pipeline.hip: the kernels, their instances and
// the launch; rh_pipeline_plan.hip: handles, plans, tables, the one-shot entry points; rh_pipeline_stream.hip: block streaming).
// Not part of the C ABI: nothing outside rodio_amd/csrc/rh_pipeline*.hip includes it.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <type_traits>
#include <unordered_map>
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

namespace rhp {

#ifndef RH_CARRY_DEFER
#define RH_CARRY_DEFER 2  // see DESIGN.md: slack of the tile-to-tile hand-off, in groups of 8 sources
#endif
constexpr int kMaxR = 20;
constexpr int kGroupLag = RH_CARRY_DEFER;  // a source group's carries are fetched kGroupLag groups (of 8 sources) after its own
constexpr int kMaxLook = 32;              // 2 lanes x 16 B of LDS-DMA per predecessor tile: 64 lanes
constexpr uint32_t kSpinLimit = 1u << 22;  // x (~1 us load + s_sleep): seconds, then give up for good.  Waits end by construction (a tile only
                                           // waits for tiles with earlier tickets); the bound must outlast a GPU that is time-sliced with other processes

struct SrcDesc {          // 32 bytes, read with s_load (constant address space)
    const float *data;
    uint32_t frames;      // N_s   (< 2^29)
    uint32_t out_frames;  // M_s   (< 2^31)
    float gain;           // Amplify factor of the source (amplify.rs:64); the chain is linear, so it scales the mix term
    uint32_t pad[3];
};
static_assert(sizeof(SrcDesc) == 32, "descriptor stride");

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
    float Tm[4];          // (w1,w2) -> z
    float scanM[4][4];    // B^(R*2^k), k = 0..3   (row_shr 1,2,4,8)
    float g[kMaxR][2];    // row 0 of A^(r+1) Tm^-1: homogeneous response of w inside a run
};
struct Tables {            // per-lane tables (loaded once per lane)
    float bc15M[64][4];    // B^(R*((lane&15)+1))   (row_bcast:15 step)
    float bc31M[64][4];    // B^(R*((lane&31)+1))   (row_bcast:31 step)
    float laneM[64][4];    // B^(R*lane)
    float lookM[64][4];    // B^(L*j), j < kMaxLook
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
    uint32_t ticket_base;  // value of *ticket when this launch starts (the counter is never reset)
    uint32_t direct;       // k_rlm_fast: tile = blockIdx.x, no ticket.  Only for launches whose workgroups are all resident at once (the host
                           // checks): then no tile can wait for one that has no slot yet.  One counter hands out ~85 tickets per microsecond,
                           // which a one-source launch (mix first) cannot hide.
    unsigned long long *prof;  // RH_PHASE_PROFILE builds: [tiles][8] cycles per phase
    uint32_t eq_frames;        // k_rlm_fast: the common length of all sources
    uint32_t batch_streams;    // k_rlm_fast: > 0 = no mixing: ticket k is tile k / batch_streams of source k % batch_streams
    uint32_t shards;           // batch mode: > 1 = the streams are dealt over this many ticket counters (stream s -> counter s % shards)
    uint32_t shard_base;       // ... whose common start value for this launch this is (every counter hands out n_tiles * batch_streams / shards tickets)
    uint64_t out_stride;       // ... whose output row starts out_stride floats after the previous one
    // k_rlm_fast, block streaming (st_mode: 0 off, 1 block of a running stream, 2 its last block):
    uint32_t st_mode, st_active;  // st_active: output frames this block emits (a multiple of R in mode 1)
    uint64_t st_m0, st_g0;        // global index of the block's first output frame / of input frame 0 of the buffers
    // k_rlm_wave keeps one aggregate row per source: gran_cols columns, tile t in column t + col0.  Streaming sets
    // col0 = 1: column 0 then holds the source's filter state at the block start, i.e. the aggregate of a virtual
    // predecessor tile -- the look-back needs no other change.
    uint32_t gran_cols, col0;
    const float *st_win;          // summed filter state (scan basis) at output frame st_m0
    float *st_wout;               // ... at st_m0 + st_active, written by the lane that would come next
    // k_rlm_fast<RAG, SUMF>: 1 = a tile also handles its own (tile, source) pairs in which the source is about to end (rag_run_pairs),
    // on top of its mix of the stable sources and before it stores; the per-source aggregate rows lie in front of `gran`.  Sources
    // that are not among the longest end between output frames rag_pairs_from and rag_pairs_to: only tiles near that range look.
    uint32_t rag_merge, rag_pairs_from, rag_pairs_to;
    Uniforms u;
};

// ---- host side: 2x2 matrices in f64 for the tables
struct M2 {
    double a, b, c, d;
};
inline M2 mul(const M2 &x, const M2 &y) { return {x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d}; }
inline M2 mpow(M2 base, uint64_t e) {
    M2 r{1, 0, 0, 1};
    while (e) {
        if (e & 1) r = mul(r, base);
        base = mul(base, base);
        e >>= 1;
    }
    return r;
}
inline void put(float *dst, const M2 &m) {
    dst[0] = (float)m.a;
    dst[1] = (float)m.b;
    dst[2] = (float)m.c;
    dst[3] = (float)m.d;
}
inline double norm(const M2 &m) { return std::fabs(m.a) + std::fabs(m.b) + std::fabs(m.c) + std::fabs(m.d); }

// Scan basis for the companion matrix of z^2 + a1 z + a2 (see the comment above Uniforms).
inline void scan_basis(double a1, double a2, M2 &T, M2 &Tinv) {
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

using KernelFn = void (*)(const Params);
struct Variant {
    int R, KV, NS;
    KernelFn filt, plain;
};
struct VariantTab {
    const Variant *v;
    size_t n;
};
template <size_t N>
constexpr VariantTab tab_of(const Variant (&t)[N]) { return VariantTab{t, N}; }
inline const Variant *find_variant(VariantTab tab, int R, int kv_needed, int NS) {  // smallest KV >= kv_needed
    const Variant *best = nullptr;
    for (size_t i = 0; i < tab.n; ++i) {
        const Variant &v = tab.v[i];
        if (v.R == R && v.NS == NS && v.KV >= kv_needed && (!best || v.KV < best->KV)) best = &v;
    }
    return best;
}
// Single-wave workgroups with `lds` dynamic bytes the hardware co-schedules on one CU.
// LDS is handed out in 1 280-byte granules (160 KiB / 128), which the occupancy query does not round to:
// measured with tools/prof_simd.py -- a 27 136-byte request fits 5 times per CU, not 6.
constexpr uint32_t kLdsGranule = 1280, kLdsGranules = 128;
inline int blocks_per_cu(const void *fn, size_t lds) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess) return 0;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 64, lds) != hipSuccess) return 0;
    const int by_lds = lds ? (int)(kLdsGranules / ((lds + kLdsGranule - 1) / kLdsGranule)) : n;
    return n < by_lds ? n : by_lds;
}
// Vectors (16 B = 2 stereo or 4 mono frames) per lane a stage must hold for a tile of L output frames.
inline int kv_needed(uint64_t L, uint32_t F, uint32_t T, uint32_t channels) {
    const uint64_t fb = 4ull * channels, vf = 16 / fb;
    const uint64_t span = ((L + 1) * F) / T + 5 + (128 / fb - 2);  // i(m0+L-1) - i(m0-2) + tap + alignment to a 128-byte line
    const uint64_t nvec = span / vf + 2;
    return (int)((nvec + 63) / 64);
}

// One launch plan: a kernel variant with its tables.
// k_rlm_chunk: tables for runs of 18 frames, the tile boundaries of the batch that is set, the hand-off tables.
struct ChunkPlan {
    int R = 18, KV = 8;                // frames per lane; KiB per chunk (stereo: 18 / 8; mono: 18 / 4: 1024 frames per chunk either way)
    const void *fn = nullptr;          // the instance for (R, channels, KV)
    int tabs_R = 0;                    // the R the filter tables were built for (0: none yet)
    bool ok = false;                   // the batch that is set can take the kernel
    Uniforms uni;
    Tables *d_tabs = nullptr;
    float *d_pow = nullptr;            // [R + 1][4]
    float *d_uni = nullptr;            // `uni` as floats, for the kernel's lanes
    uint32_t *d_mlo = nullptr;         // [n_tiles + 1]
    float *d_look = nullptr;           // [n_tiles][J][4]
    unsigned long long *d_halo = nullptr;  // [n_tiles][8]
    unsigned long long *d_gran = nullptr;  // [n_tiles][4]
    size_t cap_tiles = 0, cap_look = 0;
    uint32_t n_tiles = 0, J = 0, frames = 0;
    int resident_per_cu = 0;
    bool direct = true;
};
struct Plan {
    const Variant *v = nullptr;
    const void *kernel = nullptr;
    bool general = false;
    uint32_t J = 0, lds_bytes = 0;
    int resident_per_cu = 0;
    Uniforms uni;
    Tables *d_tabs = nullptr;
};

struct StreamArgs {
    uint32_t mode = 0, active = 0;
    uint64_t m0 = 0, g0 = 0;
    const float *win = nullptr;
    float *wout = nullptr;
    uint32_t gran_cols = 0;  // != 0: per-source states (k_rlm_wave), aggregate rows of this many columns, tile 0 in column 1
    uint64_t src_off = 0;    // mix first: bytes added to every source pointer of the table on the device (a stream that did not upload it again)
};
}  // namespace rhp
using namespace rhp;  // (an internal header: see the top)

struct rh_rlm {
    rh_rlm_config cfg;
    uint32_t F, T;
    uint64_t chunk_in, chunk_out;  // chunk_out == 0: unchunked
    bool filt;
    float coeffs[5];
    Plan fast, wave;        // equal-length batches / ragged batches
    Plan pair;              // ragged filtered one-shot batches: k_rlm_fast<RAG> (v->filt) + k_rlm_resid (v->plain); v == nullptr: none
    Plan *plan = nullptr;   // chosen by set_sources
    uint32_t launch_lds = 0;  // lds_bytes, padded so that a CU admits exactly ceil(tiles/CUs) waves
    uint32_t rag_frames = 0;  // pair plan: the length of the sources that last as long as the mix
    uint32_t rag_pairs_from = 0, rag_pairs_to = 0;  // ... and the output frames between which the other sources end
    uint32_t eq_frames = 0;
    bool equal = true;
    std::vector<Plan> tried;  // autotune candidates (their tables are freed with the handle)
    SrcDesc *d_srcs = nullptr;
    uint64_t srcs_version = 0;  // counts the uploads of the table (whoever makes them): a stream that reuses the table checks it is still its own
    std::vector<const float *> st_tab_ptrs;  // a stream's summed blocks: the pointers of the table it uploaded last ...
    uint64_t st_tab_version = ~0ull;         // ... and srcs_version right after that upload
    unsigned long long *d_gran = nullptr;
    size_t gran_words = 0;
    uint32_t *d_ctl = nullptr;  // [0] ticket, [1] status, [2] late carries, [3] empty polls
    float *d_mix = nullptr;     // mix first (k_mix_rows): the batch summed at the input rate, and behind it its one-entry descriptor table
    size_t mix_floats = 0;
    ChunkPlan chunk;            // mix first in one kernel (k_rlm_chunk)
    bool cls_one_launch = false;  // per-source filters: the last run walked the classes in one launch (k_rlm_chunk_multi)
    void *collect = nullptr;    // chunk_launch_classes: the launch being put together (rlm_launch then adds this handle's arguments to it instead of launching)
    void *sblk = nullptr;       // a stream's summed blocks in one kernel (k_rlm_sblk: rh_pipeline_sblk.hip owns the type)
    bool pre_filter = false;    // cfg.filter_first: the filter runs at from_rate in front of the converter (the fused kernels then run without one)
    float pre_coeffs[5] = {1.f, 0.f, 0.f, 0.f, 0.f};
    unsigned long long *d_prof = nullptr;
    uint32_t n_sources = 0, n_tiles = 0;
    uint64_t out_frames = 0;
    uint32_t epoch = 0;
    uint32_t ticket_base = 0;
    uint32_t shard_base = 0;  // batch mode with sharded ticket counters (d_ctl + 32*(1+x)): tickets each of them has handed out
    // block streaming (rh_rlm_stream_*)
    bool st_on = false, st_done = false;
    uint64_t st_g0 = 0, st_m = 0;
    uint64_t st_chunk_in = 0, st_chunk_out = 0;  // a stream of spanned sources: input / output frames per span (0: continuous)
    uint32_t st_nsrc = 0;
    float *d_w[2] = {nullptr, nullptr};
    int st_cur = 0;
    std::vector<SrcDesc> h_desc;  // host copy of the descriptor table
    // Block streaming uploads a descriptor table per block while earlier blocks may still be queued: the copies go through
    // a ring of page-locked tables, and a table is rewritten only after the copy that read it has run (an asynchronous
    // copy from pageable memory may read its source later than the call -- seen as a block mixed with the next block's
    // descriptors when the device was busy).
    static constexpr int kDescRing = 4;
    SrcDesc *h_ring[kDescRing] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t h_ring_ev[kDescRing] = {nullptr, nullptr, nullptr, nullptr};
    int h_ring_next = 0;
    std::vector<float> gains;     // per-source Amplify factors (1.0 when unset)
    // block streaming with per-source states (rh_rlm_stream_block_v)
    std::vector<uint64_t> st_total;  // input frames of a source that has ended (~0: still live)
    uint32_t st_cols = 0;            // columns of an aggregate row for the stream (0: no such stream yet)
    // rh_rlm_stream_block_v while its sources RUN TOGETHER (all live, equal frames per block): the summed state of
    // rh_rlm_stream_block -- so the block is summed first -- until a source ends or falls behind.  Then the per-source states are
    // recovered from the rows of the block before (rh_rlm_stream_keep_history: the caller has kept them) and the stream goes on
    // with one state per source.
    bool st_history = false;     // rh_rlm_stream_keep_history
    bool st_together = false;    // the stream still runs on the summed state
    bool st_dirty = false;       // rh_rlm_stream_block_v has touched the stream's state in this call (an error then ends the stream)
    bool st_decided = false;     // ... or has decided not to
    std::vector<const float *> st_prev_ptrs;  // the block before: its rows,
    uint64_t st_prev_avail = 0, st_prev_g0 = 0, st_prev_m = 0, st_prev_out = 0;  // frames per row, global index of frame 0, first output frame, output frames
    float *d_replay = nullptr;   // where the recovery's replay of the last tiles writes its (unused) mix
    size_t replay_floats = 0;
    uint32_t st_n_summed = 0, st_n_each = 0, st_n_recover = 0;  // rh_rlm_stream_stats
    uint32_t st_n_rejoin = 0;                 // times the stream went back to the summed state
    uint32_t st_n_sblk = 0;                   // summed blocks that ran as ONE launch (k_rlm_sblk)
    bool st_overlap = false;                  // rh_rlm_stream_overlap: consecutive one-launch blocks run side by side (rows resident)
    std::vector<uint8_t> st_gone, st_prev_gone;  // sources that have given everything (the summed blocks behind a return; the block before, for a recovery)
    // Recorded (by wait_idle) behind what the handle has queued.  The library's streams are hipStreamNonBlocking: a null-stream
    // hipMemcpy / hipMemset does NOT wait for them, so everything on the host side that rewrites device state a queued
    // kernel may still read (descriptors, control words, aggregate table, stream states) waits for this event first.
    hipEvent_t idle_ev = nullptr;
    bool launched = false;
    hipStream_t last_stream = nullptr;  // the stream of the launches idle_ev covers
    // rh_rlm_set_exclusive: may a launch assume that nothing else occupies CUs while it runs?  Only then does a launch whose tiles
    // all fit at once take tile = workgroup index (Params::direct); otherwise tiles are handed out by ticket, which needs neither
    // residency nor in-order dispatch: a tile only ever waits for tiles that already hold a wave slot.
    bool exclusive = true;
    bool mix_first_on = true;  // rh_rlm_set_mix_first
    // rh_rlm_set_filters: a filter per source.  Sources of one (kind, freq, q) form a CLASS; every class is a handle of its own
    // (`cls[c].h`, this handle's configuration with that filter) holding the class's sources in insertion order, so that each
    // class keeps everything a one-filter batch has -- mix first where its sources share a length, the ragged pair, the ordered
    // sum where it has no filter -- and the classes' mixes are summed in order of first appearance.
    struct FilterSpec {
        int32_t kind;
        uint32_t freq;
        float q;
        bool operator==(const FilterSpec &o) const { return kind == o.kind && (kind < 0 || (freq == o.freq && q == o.q)); }
    };
    struct FilterClass {
        FilterSpec spec;
        rh_rlm *h = nullptr;
        std::vector<uint32_t> members;  // indices into the parent's source list
        uint64_t out_frames = 0;
    };
    std::vector<FilterSpec> filters;  // per source; empty: the handle's one filter
    std::vector<FilterClass> cls;     // classes of the sources that are set (empty: one filter, this handle runs itself)
    float *d_cls_rows = nullptr;      // [classes][row] partial mixes
    size_t cls_row_floats = 0, cls_rows = 0;
    // RH_CLASSES_SIDE_BY_SIDE=1 (a measured alternative, slower: see run_classes): the classes' launches side by side -- class 0 on the caller's
    // stream, the others on streams of the handle's, forked from and joined to the caller's by events
    std::vector<hipStream_t> cls_streams;
    std::vector<hipEvent_t> cls_done;
    hipEvent_t cls_fork = nullptr;
};

namespace rhp {
// rh_pipeline.hip
enum TabKind { kTabFast, kTabWave, kTabRag };
VariantTab variant_tab(TabKind kind, bool mono);             // the instances of k_rlm_fast / k_rlm_wave / k_rlm_fast<RAG> + k_rlm_resid
const void *chunk_kernel(int R, uint32_t channels, int KV);  // the instance of k_rlm_chunk, or nullptr
// The classes of a mixer with per-source filters as ONE launch of k_rlm_chunk_multi (class k's mix into rows[k]): *taken = false and nothing
// done when a class does not take the k_rlm_chunk path, the classes' instances differ, or there are more than fit one kernarg segment
// (*summed: the launch added the classes' mixes itself, into dst_sum -- k_rlm_chunk_classes; the rows are then unwritten)
rh_status chunk_launch_classes(rh_rlm *const *classes, float *const *rows, uint64_t row_capacity_frames, uint32_t n, float *dst_sum, rh_stream stream, bool *taken, bool *summed);
// k_rlm_state on `s`: folds a block's aggregates into column 0 of the per-source rows (see rh_pipeline_stream.hip)
void launch_state(hipStream_t s, unsigned long long *gran, const Tables *tabs, uint32_t n_sources, uint32_t cols, uint32_t last_col, uint32_t J, uint32_t epoch, uint32_t next_epoch);
// k_rlm_state_sum on `s`: the sum of the live sources' states (column 0 of their rows, tagged `tag`) -> the 4 words of a summed state
void launch_state_sum(hipStream_t s, const unsigned long long *gran, const SrcDesc *srcs, uint32_t n_sources, uint32_t cols, uint32_t tag, float *w_out);
bool mix_first_applies(const rh_rlm *p, const Plan &pl, uint32_t count, bool per_source_states, bool batch);
rh_status rlm_launch(rh_rlm *p, uint32_t first, uint32_t count, float *dst, uint64_t out_capacity_frames, uint64_t *out_frames, rh_stream stream, uint32_t batch_streams, uint64_t out_stride_floats,
                     const StreamArgs &sa = StreamArgs());
// rh_pipeline_sblk.hip: a block of a stream on the summed state in ONE launch (k_rlm_sblk); *taken = false: not this kernel's block
rh_status sblk_try(rh_rlm *p, uint32_t n_sources, uint64_t avail_frames, uint64_t out_frames, float *dst, const StreamArgs &sa, hipStream_t s, bool *taken);
void sblk_free(rh_rlm *p);
void sblk_other_block(rh_rlm *p, bool stream_begins = false);
uint32_t sblk_chained_blocks(const rh_rlm *p);  // blocks of the current stream launched without a barrier behind the block in front  // a block of the stream ran elsewhere (or the stream begins)
// rh_pipeline_plan.hip
rh_status wait_idle(rh_rlm *p);
rh_status pre_launch(rh_rlm *p, hipStream_t s);
rh_status mark_launch(rh_rlm *p, hipStream_t s);
uint32_t look_tiles(const M2 &B, uint64_t L);
rh_status make_plan(rh_rlm *p, Plan &pl, VariantTab tab, bool general, const rh::ResampleGeom &g, uint32_t want_R, uint32_t want_NS);
rh_status build_chunk(rh_rlm *p);
bool pair_ok(rh_rlm *p, const Plan &pl);
rh_status activate_plan(rh_rlm *p, Plan *pl);
rh_status upload_descriptors(rh_rlm *p, uint32_t n, hipStream_t s);
}  // namespace rhp

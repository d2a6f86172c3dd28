// rh_limit.hip -- the limiter (src/source/limit.rs:94-130, :853-873, :903-916, :927-988) as a TIME-PARALLEL kernel.
//
// Per channel the reference runs, sample after sample,
//     g_n = gain computer (log2, soft knee)                          limit.rs:853-873     elementwise
//     I_n = max(g_n, r*I_{n-1} + (1-r)*g_n)                          limit.rs:909-912     release-smoothed peak hold
//     P_n = a*P_{n-1} + (1-a)*I_n                                    limit.rs:913         attack smoothing
//     y_n = x_n * 2^(-max_c P_c * 0.05*log2 10)                      limit.rs:927-988     gain coupled over the channels
// (r, a = exp(-1/(t*fs)), math.rs:110-113).  Both recurrences are scannable:
//   * I: the step maps I -> max(g, r*I + (1-r)g) belong to the family I -> max(A, c*I + B), closed under composition:
//        (A2,c2,B2) o (A1,c1,B1) = (max(A2, c2*A1 + B2), c2*c1, c2*B1 + B2).  c = r^(steps) depends on the POSITION only,
//        so a segment is two floats (A, B) and every c is a host-computed constant (f64, rounded once).
//        All values are >= 0 and A >= B, hence (0, 0) acts as the identity and DPP zero-fill needs no special case.
//   * P: linear, driven by the (now known) I_n: zero-state run + a^(steps) * (start state).
//
// Decomposition (wave64, no MFMA -- there is no contraction):
//   * a workgroup of NW waves = one tile of LW = NW*64*R frames of ONE stream; a wave owns 64*R consecutive frames, a lane R
//     of them (all channels).  Tiles are handed out by an atomic ticket, tile-major over the streams (ticket k -> tile k / S
//     of stream k % S), to a persistent grid: a tile only ever waits for tiles with smaller tickets, which run or are done.
//   * HBM traffic = 4 B in + 4 B out per sample.  The samples of the NEXT tile are requested by LDS-DMA (global_load_lds,
//     no VGPR round trip) while the current one is worked on; a lane reads its run from LDS (swizzled slots: conflict-free
//     without padding), the result goes back to the same slots and leaves with coalesced 16-byte stores.
//   * per tile: gain computer -> lane-local (A,B) -> wave scan (DPP Kogge-Stone) -> the waves' aggregates meet in LDS
//     (barrier 1) -> workgroup aggregate published -> look-back over the workgroup tiles in front (a wave polls up to 64
//     predecessors at once: their zero-state aggregates compose in one wave scan; the walk ends at the state the block starts
//     from or where r^(LW*j) < 2^-30 -- never at a predecessor's end state, so the result does not depend on timing) -> true I per sample + zero-state P run -> wave scan -> LDS (barrier 2) -> published ->
//     look-back for P -> per-sample P, channel-coupled max, exp2, multiply -> store.
//   * hand-off words are plain f32 in a table filled with 0xFF on the stream in front of the launch: all-ones = "not yet",
//     anything else is the value (the data is the flag, cdna_hip_programming.md G16 form R2); consumers fetch whole sections
//     with 8/16-byte sc1 loads.
// What the first versions taught (64 streams x 1 Mi stereo frames, 8 B of traffic per sample; the one-lane-per-stream
// kernel of round 1 took 3 628 ms):
//     single-wave tiles, one ticket per tile                      0.99 ms   -- the ticket counter: one device-scope word hands
//                                                                              out ~85 tickets/us (MI355X_MICROARCH.md "dequeue")
//     ... 16 tickets per atomic, 8-byte {tag,value} granules      0.78 ms   -- 384 sc1 loads per tile and look-back: L2 request rate
//     ... f32 words with a NaN-pattern sentinel, 16-byte loads    0.61 ms   -- look-back latency x lockstep (every wave of the chip
//                                                                              is in the same phase: the phases' times add up)
//     workgroup tiles (8 waves, LDS exchange, 2 barriers)         0.38 ms
//     ... 16 frames per lane, LDS-DMA prefetch                    0.33 ms = 3.2 TB/s = 40 % of 8 TB/s (SQ counters: waves wait
//                                                                              56 % of their cycles, VALU busy 19 %: sync-bound)
//     ... two channels per instruction (v_pk_*_f32, -23 % VALU)   0.31 ms = 43 %; 2048 streams x 32 Ki: 0.305 ms
//     ... one poll point per tile when streams > workgroups       0.31 ms = 43 %; 2048 streams x 32 Ki: 0.280 ms = 48 %
//     ... the next tile's DMA at the top of every tile            0.286 ms = 47 %; 2048 streams x 32 Ki: 0.288 ms = 47 %
//         (the next tile's integrator look-back rides on this tile's peak look-back; with FEW streams that chains the
//          workgroups -- 0.76 ms -- so there the walk stays inside the tile)
// What is left per tile (~9 us for 128 KiB of traffic, one workgroup of 8 waves per CU because of the 128 KiB of LDS the
// double-buffered 16-frame runs take): ~3 us of vector arithmetic, the rest is the poll round trip(s) and the two barriers
// with nothing else resident to fill them; 8 frames per lane and two workgroups per CU measures slower (0.35 ms: the scans
// and look-backs amortise over half as many samples).
// The sequential kernel of rh_recurrence.hip stays as the reference-order path for layouts this one does not take
// (rows that are not 16-byte aligned).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "rh_common.h"

namespace rh {
rh_status limit_seq_launch(float *dst, const float *src, uint64_t frames, uint32_t channels, uint32_t n_streams, const float k5[5], float *state, hipStream_t s);
float duration_to_coefficient_f32(uint64_t ns, uint32_t sample_rate);
}  // namespace rh

namespace {

#include "rh_scan_common.h"

constexpr float LOG2_10 = 3.32192809488736234787f;
constexpr float LOG10_2 = 0.301029995663981195214f;
constexpr int kMaxR = 16, kMaxNW = 16;
constexpr uint32_t kSpinLimit = 1u << 22;
constexpr float kNegligible = 0x1p-30f;  // a weight below this no longer moves an f32 state of the same magnitude

struct LimitTabs {      // per-lane / per-predecessor constants, f64 on the host, rounded once; travels as a kernel argument
    float r15[64], r31[64], rlane[64];  // r^(R*((l&15)+1)), r^(R*((l&31)+1)), r^(R*l)
    float a15[64], a31[64], alane[64];  // the same powers of the attack coefficient
    float rlook[64], alook[64];         // r^(LW*j), a^(LW*j): weight of the j-th workgroup tile in front (j = 0: the nearest)
};
struct LimitArgs {
    float *dst;
    const float *src;
    float *gran;               // [S][tiles][Rec<C>::stride] hand-off words: A[C] B[C] | Iend[C] | Pz[C] | Pend[C] (see Rec)
    float *gran_other;         // the table of the NEXT launch on this stream: every tile sets its record there back to "not yet" when it is done
                               // (nullptr: the next launch initialises its own table)
    uint32_t ticket_base;      // value of ctl[0] when this launch starts (the counter is never reset)
    const float *state_in;     // [S][C][2] {integrator, peak} snapshot taken in front of the launch, or nullptr
    float *state_out;          // the caller's state, or nullptr
    uint32_t *ctl;             // [0] ticket
    uint32_t *status;          // the library's sticky failure word (rh_async_status)
    uint32_t spin;             // polls of one hand-off before the tile gives up (kSpinLimit; RH_SCAN_SPIN_LIMIT overrides: tests of the failure path)
    uint32_t dma_top;          // 1: the next tile's samples are requested at the top of every tile
    uint64_t frames;           // per stream
    uint64_t stride;           // floats between streams
    uint32_t n_streams, tiles; // tiles per stream
    float threshold, knee_width, inv_knee_8, attack, release;
    float one_minus_attack, one_minus_release;  // 1.0f - attack, 1.0f - release in f32, as limit.rs:911,913 compute them (gfx950 has no scalar float ALU: a subtraction in the kernel would be a vector instruction per use)
    float rscan[4], ascan[4];  // r^(R*2^k), a^(R*2^k), k = 0..3 (row_shr 1,2,4,8)
    float rL, aL, rLW, aLW;    // r^L, a^L (one wave's share), r^LW, a^LW (one workgroup tile)
    float rwave[kMaxNW], awave[kMaxNW];  // r^(L*k), a^(L*k): k waves in front inside the workgroup tile
    float rL64, aL64;          // r^(LW*64), a^(LW*64): one full look-back window
    uint32_t jI, jP;           // predecessors beyond these are below kNegligible (<= 64)
    LimitTabs t;
};

// limit.rs:853-873 (f32::MIN_POSITIVE = 2^-126).  The argument of the logarithm is a normal number (>= 2^-126) and the
// argument of the exponential below lies in [-50, 0]: the bare v_log_f32 / v_exp_f32 (1 ulp) are what log2f / exp2f reduce
// to on this range, without their denormal pre- and post-scaling.  The dB scale (log10(2) * 20) and the knee test on
// 2 * bias are folded into one FMA each: the gain computer is continuous, a last-bit difference in bias_db moves the
// output by parts in 1e-8.
struct GainK {
    float db_per_log2, neg_thr, half_knee, knee_width, inv_knee_8;
};
__device__ __forceinline__ float gain_computer(float sample, const GainK &k) {
    const float bias_db = fma_(__builtin_amdgcn_logf(fabsf(sample) + 1.17549435e-38f), k.db_per_log2, k.neg_thr);
    const float x = fma_(2.0f, bias_db, k.knee_width);
    const float soft = x * x * k.inv_knee_8;
    return bias_db < -k.half_knee ? 0.0f : (fabsf(bias_db) <= k.half_knee ? soft : bias_db);
}

// Inclusive wave64 scan of max-affine segments (A, B) whose slopes are position constants (see the header).
__device__ __forceinline__ void scan_maxaff(float &A, float &B, const float (&cs)[4], float c15, float c31) {
#define RH_STEP(K, N)                                                      \
    {                                                                      \
        const float a1 = dpp0<kRowShr + N, 0xf>(A), b1 = dpp0<kRowShr + N, 0xf>(B); \
        A = fmaxf(A, fma_(cs[K], a1, B));                                  \
        B = fma_(cs[K], b1, B);                                            \
    }
    RH_STEP(0, 1)
    RH_STEP(1, 2)
    RH_STEP(2, 4)
    RH_STEP(3, 8)
#undef RH_STEP
    {
        const float a1 = dpp0<kBcast15, 0xa>(A), b1 = dpp0<kBcast15, 0xa>(B);
        A = fmaxf(A, fma_(c15, a1, B));
        B = fma_(c15, b1, B);
    }
    {
        const float a1 = dpp0<kBcast31, 0xc>(A), b1 = dpp0<kBcast31, 0xc>(B);
        A = fmaxf(A, fma_(c31, a1, B));
        B = fma_(c31, b1, B);
    }
}
// Inclusive wave64 scan of a linear 1-pole: V_l = sum_{k<=l} c^(R*(l-k)) v_k.
__device__ __forceinline__ void scan_lin(float &V, const float (&cs)[4], float c15, float c31) {
    V = fma_(cs[0], dpp0<kRowShr + 1, 0xf>(V), V);
    V = fma_(cs[1], dpp0<kRowShr + 2, 0xf>(V), V);
    V = fma_(cs[2], dpp0<kRowShr + 4, 0xf>(V), V);
    V = fma_(cs[3], dpp0<kRowShr + 8, 0xf>(V), V);
    V = fma_(c15, dpp0<kBcast15, 0xa>(V), V);
    V = fma_(c31, dpp0<kBcast31, 0xc>(V), V);
}
// ---- hand-off words ---------------------------------------------------------------------------------------------------
// A tile publishes its aggregates / end states as plain f32 words in a table that is filled with 0xFF bytes on the stream
// in front of the launch: the all-ones pattern (a NaN no arithmetic produces; NaNs are published in canonical form) means
// "not there yet", anything else is the value -- the data is the flag at 4-byte granularity (cdna_hip_programming.md G16,
// form R2), so a consumer may fetch a whole section with one 16-byte load and a torn load merely reads "not yet".
// Per tile: [A[C] B[C] | (unused) | Pz[C] | (unused)], sections aligned for the widest load that fits them.
template <int C>
struct Rec {
    static constexpr int up(int v, int m) { return (v + m - 1) / m * m; }
    static constexpr int wA = (2 * C) % 4 == 0 ? 4 : ((2 * C) % 2 == 0 ? 2 : 1);  // vector width of the A/B section
    static constexpr int wS = C % 4 == 0 ? 4 : (C % 2 == 0 ? 2 : 1);              // ... of the one-per-channel sections
    static constexpr int oA = 0;
    static constexpr int oI = up(2 * C, 4);
    static constexpr int oZ = oI + up(C, wS);
    static constexpr int oE = oZ + up(C, wS);
    static constexpr int stride = up(oE + C, 4);
};
#ifdef RH_LIMIT_PROFILE  // diagnostic builds (tools/build_variant.sh): shader cycles per phase of a tile, summed over all tiles
__device__ unsigned long long g_limit_prof[16];
#define RH_LP_DECL unsigned long long lp_last = __builtin_readcyclecounter();
#define RH_LP(i)                                                                  \
    {                                                                             \
        const unsigned long long lp_now = __builtin_readcyclecounter();           \
        lp_acc[i] += lp_now - lp_last;                                            \
        lp_last = lp_now;                                                         \
    }
#define RH_LP_PARAM , unsigned long long (&lp_acc)[8]
#define RH_LP_ARG , lp_acc
#else
#define RH_LP_DECL
#define RH_LP(i)
#define RH_LP_PARAM
#define RH_LP_ARG
#endif

// One window of a look-back walk.  Lane j looks at tile base-j and fetches its zero-state aggregate (all loads of a poll leave
// together: one round trip; the poll repeats until every lane has its words).  Returns j* = the nearest lane that stands for a
// KNOWN STATE instead of a tile: the state the block starts from (tile -1, `init`), a tile before it, or a predecessor whose
// weight is negligible (taken as zero).  Real predecessors never contribute anything but their aggregate, whatever else they
// may have published by now: which tiles have finished when this one polls depends on timing, and f32 composition is not
// associative -- this way the result does not depend on it (the same input gives the same bits, run after run).
template <int C, int NAGG, int WAGG>
__device__ __forceinline__ uint32_t poll_window(const float *gran_stream, int64_t base, int lane, uint32_t reach, uint32_t stride, uint32_t off_agg,
                                                const float *init, float (&agg)[NAGG], float (&inc)[C], bool &dead, const uint32_t spin_limit) {
    const int64_t idx = base - lane;
    const bool real = idx >= 0 && (uint32_t)lane < reach;
    const float *pr = gran_stream + (real ? (uint64_t)idx : 0) * stride;
    bool have = !real;
#pragma unroll
    for (int c = 0; c < NAGG; ++c) agg[c] = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) inc[c] = (idx == -1 && init) ? init[2 * c] : 0.0f;
    const unsigned long long virt = __ballot(!real);
    const uint32_t jstar = virt ? (uint32_t)__builtin_ctzll(virt) : 64u;
    uint32_t spins = 0;
    while (true) {
        if (!have) {
            float ga[NAGG];
            load_words<NAGG, WAGG>(pr + off_agg, ga);
            wait_loads(ga);
            bool ok = true;
#pragma unroll
            for (int c = 0; c < NAGG; ++c) ok = ok && word_ok(ga[c]);
            if (ok) {
                have = true;
#pragma unroll
                for (int c = 0; c < NAGG; ++c) agg[c] = ga[c];
            }
        }
        if (__all(have || (uint32_t)lane > jstar)) return jstar;  // lanes behind j* are not needed
        if (++spins > spin_limit) {
            dead = true;
            return 64u;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// The look-ahead poll of a tile (see limit_tile; more streams than workgroups): the first windows of BOTH look-backs of the workgroup's
// NEXT tile -- its predecessors hold older tickets and their zero-state aggregates depend on nothing this workgroup still owes --
// requested behind this tile's last publish and collected behind its output pass: one round trip to L2 for both, and no tile waits
// for it (the pass hides it).  issue() puts the loads on the wire; complete() is the poll loop entered with its first loads in flight.
template <int C>
struct PollPair {
    typedef Rec<C> RC;
    float Z[C], E[C], AB[2 * C], Ij[C];
    uint32_t jstarP, jstarI;
    float gz[C], ga[2 * C];  // in flight between issue() and complete(): nothing may read them before complete()'s wait
    bool haveP, haveI;
    // (the addresses are worked out again where a poll has to be repeated: kept across the tile they cost the widest instance its last registers)
    static __device__ __forceinline__ void where(const float *gs, int64_t base, uint32_t reachP, uint32_t reachI, int lane, const float *&pP, const float *&pI, bool &realP, bool &realI) {
        constexpr uint32_t G = RC::stride;
        const int64_t idx = base - lane;
        realP = idx >= 0 && (uint32_t)lane < reachP, realI = idx >= 0 && (uint32_t)lane < reachI;
        pP = gs + (realP ? (uint64_t)idx : 0) * G + RC::oZ, pI = gs + (realI ? (uint64_t)idx : 0) * G + RC::oA;
    }
    __device__ __forceinline__ void issue(const float *gs, int64_t base, uint32_t reachP, uint32_t reachI, int lane) {
        const float *pP, *pI;
        bool realP, realI;
        where(gs, base, reachP, reachI, lane, pP, pI, realP, realI);
        if (realP) load_words<C, RC::wS>(pP, gz);
        if (realI) load_words<2 * C, RC::wA>(pI, ga);
    }
    // YOUNGER: vector-memory instructions this wave has certainly issued behind issue() (the LDS-DMA of the next tile's samples, or 0).
    // vmcnt retires in order: waiting for "at most YOUNGER outstanding" collects the poll and leaves those fetches in flight.
    template <int YOUNGER>
    __device__ __forceinline__ bool complete(const bool younger_there, const float *gs, int64_t base, uint32_t reachP, uint32_t reachI, const float *init, int lane,
                                             const uint32_t spin_limit) {  // false: a hand-off never arrived
        // (everything but the words in flight is worked out here, not kept across the tile: the widest instance has no registers to spare)
        const float *pP, *pI;
        bool realP, realI;
        where(gs, base, reachP, reachI, lane, pP, pI, realP, realI);
        const int64_t idx = base - lane;
        haveP = !realP, haveI = !realI;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            Z[c] = 0.0f, AB[c] = AB[C + c] = 0.0f;
            E[c] = (idx == -1 && init) ? init[2 * c + 1] : 0.0f;
            Ij[c] = (idx == -1 && init) ? init[2 * c] : 0.0f;
        }
        const unsigned long long virtP = __ballot(!realP), virtI = __ballot(!realI);
        jstarP = virtP ? (uint32_t)__builtin_ctzll(virtP) : 64u;
        jstarI = virtI ? (uint32_t)__builtin_ctzll(virtI) : 64u;
        uint32_t spins = 0;
        bool first = true;
        while (true) {
            if (!first) {
                if (!haveP) load_words<C, RC::wS>(pP, gz);
                if (!haveI) load_words<2 * C, RC::wA>(pI, ga);
            }
            if (first && younger_there) {
                wait_loads_behind<YOUNGER>(gz);  // (both are waited for whether or not this lane fetched: the registers are only read if it did)
                wait_loads_behind<YOUNGER>(ga);
            } else {
                wait_loads(gz);
                wait_loads(ga);
            }
            first = false;
            if (!haveP) {
                bool ok = true;
#pragma unroll
                for (int c = 0; c < C; ++c) ok = ok && word_ok(gz[c]);
                if (ok) {
                    haveP = true;
#pragma unroll
                    for (int c = 0; c < C; ++c) Z[c] = gz[c];
                }
            }
            if (!haveI) {
                bool ok = true;
#pragma unroll
                for (int c = 0; c < 2 * C; ++c) ok = ok && word_ok(ga[c]);
                if (ok) {
                    haveI = true;
#pragma unroll
                    for (int c = 0; c < 2 * C; ++c) AB[c] = ga[c];
                }
            }
            if (__all((haveP || (uint32_t)lane > jstarP) && (haveI || (uint32_t)lane > jstarI))) return true;
            if (++spins > spin_limit) return false;
            __builtin_amdgcn_s_sleep(2);
        }
    }
};

// (two channels per instruction: f2, Pk<C>, splat / vfma / vmax / comp / pair_of / vdpp / vsel -- rh_scan_common.h)
// limit.rs:853-873 for one sample or a pair (see gain_computer above: the same expression, with the lower knee branch
// expressed as a clamp: 2*bias + knee < 0 exactly when bias < -knee/2, and then the square is the reference's 0.0).
__device__ __forceinline__ float log2_mag(float s) { return __builtin_amdgcn_logf(fabsf(s) + 1.17549435e-38f); }
__device__ __forceinline__ f2 log2_mag(f2 s) {
    f2 r;
    r.x = log2_mag(s.x), r.y = log2_mag(s.y);
    return r;
}
template <class T>
__device__ __forceinline__ T gain_computer_v(T sample, const GainK &k) {
    const T bias_db = vfma(log2_mag(sample), splat<T>(k.db_per_log2), splat<T>(k.neg_thr));
    const T x = vmax(vfma(splat<T>(2.0f), bias_db, splat<T>(k.knee_width)), splat<T>(0.0f));
    const T soft = x * x * splat<T>(k.inv_knee_8);
    if constexpr (sizeof(T) == 8) {
        f2 r;
        r.x = !(bias_db.x <= k.half_knee) ? bias_db.x : soft.x;  // (a NaN sample stays a NaN, as in the reference)
        r.y = !(bias_db.y <= k.half_knee) ? bias_db.y : soft.y;
        return r;
    } else
        return !(bias_db <= k.half_knee) ? bias_db : soft;
}
template <class T>
__device__ __forceinline__ void scan_maxaff_v(T &A, T &B, const float (&cs)[4], float c15, float c31) {
#define RH_STEP(CS, CTRL, MASK)                                            \
    {                                                                      \
        const T a1 = vdpp<CTRL, MASK>(A), b1 = vdpp<CTRL, MASK>(B);       \
        A = vmax(A, vfma(splat<T>(CS), a1, B));                            \
        B = vfma(splat<T>(CS), b1, B);                                     \
    }
    RH_STEP(cs[0], kRowShr + 1, 0xf)
    RH_STEP(cs[1], kRowShr + 2, 0xf)
    RH_STEP(cs[2], kRowShr + 4, 0xf)
    RH_STEP(cs[3], kRowShr + 8, 0xf)
    RH_STEP(c15, kBcast15, 0xa)
    RH_STEP(c31, kBcast31, 0xc)
#undef RH_STEP
}
template <class T>
__device__ __forceinline__ void scan_lin_v(T &V, const float (&cs)[4], float c15, float c31) {
    V = vfma(splat<T>(cs[0]), vdpp<kRowShr + 1, 0xf>(V), V);
    V = vfma(splat<T>(cs[1]), vdpp<kRowShr + 2, 0xf>(V), V);
    V = vfma(splat<T>(cs[2]), vdpp<kRowShr + 4, 0xf>(V), V);
    V = vfma(splat<T>(cs[3]), vdpp<kRowShr + 8, 0xf>(V), V);
    V = vfma(splat<T>(c15), vdpp<kBcast15, 0xa>(V), V);
    V = vfma(splat<T>(c31), vdpp<kBcast31, 0xc>(V), V);
}

// One wave's share (L = 64*R frames) of a workgroup tile (NW waves, LW = NW*L frames of one stream).
//   waves of a workgroup exchange their aggregates through LDS (two barriers per tile); only the workgroup-level
//   aggregates and end states travel through HBM, one record per LW frames, and every wave walks the (short) look-back over
//   them on its own -- redundant polls are cheaper than two more barriers.
// One share's fetch when it is short (the end of a stream): guarded vector loads into the share's LDS slots, zeros behind the end.
template <int C, int V>
__device__ __forceinline__ void load_share_guarded(const float *src, v4f *lds, const uint32_t nfloat, int lane) {
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
// One share's results, LDS rows -> whole lines.  Streaming (nt) stores: every output byte is written once, and a 1:1 stream of
// reads and writes moves 5-8 % faster with them (tools/ubench/write_bw.hip; the limiter 0.278 -> 0.257 ms, the biquad 0.212 -> 0.204).
template <int V, bool FULL>
__device__ __forceinline__ void store_share(float *dst, const v4f *lds, const uint32_t nfloat, int lane) {
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const uint32_t q = k * 64 + lane, o = 4u * q;
        const v4f v = lds[slot_of<V>(q / V, q % V)];
        if (FULL || o + 4 <= nfloat) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(dst + o));
        else if (o < nfloat) {
            dst[o] = v.x;
            if (o + 1 < nfloat) dst[o + 1] = v.y;
            if (o + 2 < nfloat) dst[o + 2] = v.z;
        }
    }
}

// NIO > 0: the workgroup has NIO more waves that do nothing but move samples (k_limit_scan): this function then neither fetches
// nor stores, and its polls are the only vector-memory loads of its wave -- a poll retires when ITS data arrives, not behind
// 16 KiB of LDS-DMA and stores in the in-order vmcnt queue.
// The kernel's argument block, read where it lies: in the constant address space (the kernarg segment; it is the kernel's only argument, so it
// sits at offset 0 of __builtin_amdgcn_kernarg_segment_ptr()).  See limit_tile for why the kernels do not read their by-value parameter.
typedef const __attribute__((address_space(4))) LimitArgs *ArgsC;
#if defined(RH_LIMIT_NO_ARITH) && RH_LIMIT_NO_ARITH
#define RH_LIMIT_NO_ARITH_ 1
#else
#define RH_LIMIT_NO_ARITH_ 0
#endif

template <int C, int R, int NW, bool FULL, bool SKEW, int NIO = 0>
__device__ __forceinline__ void limit_tile(ArgsC kargs, v4f *lds, float (*xI)[2 * C], float (*xP)[C], const int lane_, const int wave, const uint32_t tile, const uint32_t stream,
                                           const float (*tab)[64], const uint32_t nf, float (&Icarry)[C], float (&Pcarry)[C], bool &have_I, const bool has_next, const uint32_t ntile,
                                           const uint32_t nstream, const float *next_src, v4f *next_buf, const uint32_t ticket_ahead, uint32_t *ticket_slot RH_LP_PARAM) {
    // The lane id is made opaque per tile: everything derived from it (LDS slots, global offsets) is then recomputed here, a
    // few VALU operations, instead of being hoisted out of the persistent loop into registers that stay occupied for the
    // whole kernel (which spilled).
    int lane = lane_;
    asm volatile("" : "+v"(lane));
    // The same for the kernel's ARGUMENTS.  They live in the constant address space (the kernarg segment) and cost one scalar load each; read
    // through `a_in` the compiler loads all ~50 of them once, in front of the persistent loop, and keeps them in SGPRs for the whole kernel --
    // more than the SGPR file holds beside the loop's own state, so a tenth of the kernel's vector instructions were v_readlane / v_writelane
    // moving spilled scalars (round 4: "9 % of VALU").  Read through a pointer that is made opaque at the top of every phase, each phase
    // loads the few constants it uses where it uses them (scalar cache hits, under the phase's first LDS reads) and nothing outlives it.
    // (Taking the address of the by-value parameter instead would make the compiler copy the 2 KB block to scratch.)
#define RH_ARGS_FRESH() asm volatile("" : "+s"(kargs))
    RH_ARGS_FRESH();
#define a (*kargs)
    constexpr int V = C * R / 4;
    constexpr uint32_t L = 64u * R, LW = L * NW;
    typedef Rec<C> RC;
    constexpr uint32_t G = RC::stride;
    typedef typename Pk<C>::T T;
    constexpr int W = Pk<C>::W, N = Pk<C>::N;
    static_assert(W == 1 || 4 % W == 0, "a pair never straddles two 16-byte vectors");
    // the loops over the waves' aggregates: where registers allow (the variants that run 2 waves per SIMD) all their LDS reads
    // leave together, one latency instead of NW
    constexpr int kPrefixUnroll = (C <= 2 && C * R > 16) ? NW : 1;
#define att (a.attack)
#define relT splat<T>(a.release)
#define omrT splat<T>(a.one_minus_release)
#define attT splat<T>(a.attack)
#define omaT splat<T>(a.one_minus_attack)
    const GainK gk{LOG10_2 * 20.0f, -a.threshold, 0.5f * a.knee_width, a.knee_width, a.inv_knee_8};
    const uint64_t f0 = (uint64_t)tile * LW + (uint64_t)wave * L;  // first frame of this wave's share; nf = valid frames in it (FULL: L)
    const uint32_t nfl = FULL ? (uint32_t)R : (nf > (uint32_t)lane * R ? (nf - lane * R < (uint32_t)R ? nf - lane * R : R) : 0u);
    const float *src = a.src + stream * a.stride + f0 * C;
    float *dst = a.dst + stream * a.stride + f0 * C;
    const uint32_t nfloat = nf * C;
    const float *const gstream = a.gran + (uint64_t)stream * a.tiles * G;
    float *const rec = a.gran + ((uint64_t)stream * a.tiles + tile) * G;
    const float *const init = a.state_in ? a.state_in + (uint64_t)stream * C * 2 : nullptr;
    RH_LP_DECL
    // The next tile's samples (LDS-DMA into the other buffer, whose last reader was the previous tile's store) are requested in
    // front of everything: vmcnt retires in order, so a poll waits for every fetch issued before it, and the top of the tile
    // is as far ahead of the polls as a fetch can be (a.dma_top = 0, the round-2 first version, put it behind the integrator
    // look-back: that poll was then free, but the peak poll paid the whole fetch latency).
    const bool had_I = SKEW && have_I;
#if defined(RH_LIMIT_NO_LOOKBACK) && RH_LIMIT_NO_LOOKBACK
    constexpr bool kLookback = false;
#else
    constexpr bool kLookback = true;
#endif
    // ---- the look-ahead poll (SKEW: more streams than workgroups): the first windows of both look-backs of the workgroup's NEXT tile leave
    //      here, in front of every other fetch of this tile, and are collected behind its output pass.  Ticket order: that tile's
    //      predecessors hold tickets a whole round of the streams older -- they published long ago, as a rule -- and what they publish
    //      depends on nothing this workgroup still owes (waits go to strictly smaller tickets: no cycle).  In FRONT of the DMA because
    //      vmcnt retires in order: a poll behind it comes home when the 8 KiB in front of it have, and on a chip that is busy moving
    //      samples that is a tile's time later (measured: the polls cost 0.04 ms of 0.26 wherever they stood behind the DMA) ---------
    const bool ahead = kLookback && SKEW && has_next;
    PollPair<C> q;
    if (ahead) q.issue(a.gran + (uint64_t)nstream * a.tiles * G, (int64_t)ntile - 1, a.jP, a.jI, lane);
    const uint32_t dma_where = a.dma_top;  // 1: at the top of the tile; 0: behind the integrator look-back; 2: behind the poll point (no poll of this tile ever waits for it)
    const bool dma_first = dma_where == 1 || (had_I && dma_where != 2);
    if (NIO == 0 && dma_first && next_src) dma_share<V>(next_src, next_buf, lane);
    // ---- the samples: whole shares were put into LDS by the DMA issued a tile ago; a short share (end of a stream) is
    //      fetched here, guarded, into the same slots.  (FULL is a property of the TILE -- every wave of the workgroup runs the same
    //      instantiation, barriers included; a whole share inside a short tile arrived by DMA like any other)
    if (NIO == 0 && !FULL && nf != L) load_share_guarded<C, V>(src, lds, nfloat, lane);
    __builtin_amdgcn_wave_barrier();
    RH_LP(0)
    // ---- this lane's run: gain computer + lane-local max-affine segment per channel (the samples stay in the LDS row) ----
    // (per-lane constants live in LDS, not in registers, between their uses; each is read a phase ahead of its use)
    const float r15 = tab[0][lane], r31 = tab[1][lane];
    T g[R][N], A[N], B[N];
#pragma unroll
    for (int p = 0; p < N; ++p) A[p] = B[p] = splat<T>(0.0f);
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const v4f v = lds[slot_of<V>(lane, j)];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; i += W) {
            const int r = (4 * j + i) / C, p = ((4 * j + i) % C) / W;
#if defined(RH_LIMIT_NO_ARITH) && RH_LIMIT_NO_ARITH  // timing experiment only (wrong results): the tile's skeleton without its per-sample arithmetic
            T gv = pair_of(e + i, T());
#else
            T gv = gain_computer_v<T>(pair_of(e + i, T()), gk);
#endif
            if (!FULL) gv = vsel((uint32_t)r < nfl, gv, splat<T>(0.0f));
            g[r][p] = gv;
        }
    }
#pragma unroll
    for (int r = 0; r < (RH_LIMIT_NO_ARITH_ ? 1 : R); ++r)
#pragma unroll
        for (int p = 0; p < N; ++p) {
            const T bn = omrT * g[r][p];
            const T An = vmax(g[r][p], vfma(relT, A[p], bn)), Bn = vfma(relT, B[p], bn);
            if (FULL || (uint32_t)r < nfl) {  // frames past the end of the stream are the identity
                A[p] = An;
                B[p] = Bn;
            }
        }
    const float rscan[4] = {a.rscan[0], a.rscan[1], a.rscan[2], a.rscan[3]};
    T Ax[N], Bx[N];  // exclusive prefixes inside the wave
#pragma unroll
    for (int p = 0; p < N; ++p) {
        scan_maxaff_v<T>(A[p], B[p], rscan, r15, r31);
        Ax[p] = vdpp<kWaveShr1, 0xf>(A[p]);
        Bx[p] = vdpp<kWaveShr1, 0xf>(B[p]);
    }
    if (lane == 63) {
#pragma unroll
        for (int c = 0; c < C; ++c) xI[wave][c] = comp(A[c / W], c % W), xI[wave][C + c] = comp(B[c / W], c % W);
    }
    const float wI = tab[6][lane], rlane = tab[2][lane];
    RH_LP(1)
    __syncthreads();  // (1) the waves' aggregates are in LDS
    RH_ARGS_FRESH();
    // prefix over the waves in front of this one, and the workgroup aggregate (uniform; LDS broadcast reads)
    T Ap[N], Bp[N], AT[N], BT[N];
#pragma unroll
    for (int p = 0; p < N; ++p) Ap[p] = Bp[p] = AT[p] = BT[p] = splat<T>(0.0f);
    const T rLT = splat<T>(a.rL);
#pragma unroll kPrefixUnroll
    for (int k = 0; k < NW; ++k) {
#pragma unroll
        for (int p = 0; p < N; ++p) {
            if (k == wave) Ap[p] = AT[p], Bp[p] = BT[p];
            const T Ak = pair_of(&xI[k][p * W], T()), Bk = pair_of(&xI[k][C + p * W], T());
            AT[p] = vmax(Ak, vfma(rLT, AT[p], Bk));
            BT[p] = vfma(rLT, BT[p], Bk);
        }
    }
    if (wave == 0 && lane < 2 * C) {
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) v = lane == c ? comp(AT[c / W], c % W) : (lane == C + c ? comp(BT[c / W], c % W) : v);
        word_store(rec + RC::oA + lane, v);
    }
    // ---- look-back for the integrator over the workgroup tiles in front: I_in = f_{t-1}(f_{t-2}(... )) -----------------------
    // With w_j = r^(LW*j) and S_j = sum_{i<j} w_i B_i the composition of a window is
    //     max_{j<j*} (w_j A_j + S_j)  v  (w_j* Iend_j* + S_j*).
    // No j* in the window: fold it into (Ao, Bo, Co) and slide on, unless Co has decayed below f32 resolution.
    // A workgroup's first tile walks this look-back here; every later tile received Iin from the poll point of the tile before.
    bool dead = false;
    float Iin[C];
#pragma unroll
    for (int c = 0; c < C; ++c) Iin[c] = Icarry[c];
    // folds one polled window (aggregates AB, known state Ij at lane j*) into (Ao, Bo): true when the walk is over
    auto fold_I = [&](float (&Ao)[C], float (&Bo)[C], float &Co, const float (&AB)[2 * C], const float (&Ij)[C], uint32_t jstar) {
        const bool front = (uint32_t)lane < jstar, star = (uint32_t)lane == jstar;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float tot;
            const float S = wave_excl_sum(front ? wI * AB[C + c] : 0.0f, tot);
            const float cand = front ? fma_(wI, AB[c], S) : (star ? fma_(wI, Ij[c], S) : 0.0f);
            const float Aw = wave_max(cand);
            Ao[c] = fmaxf(Ao[c], fma_(Co, Aw, Bo[c]));
            Bo[c] = fma_(Co, tot, Bo[c]);
        }
        if (jstar < 64) return true;  // reached a known state: Ao contains it
        Co *= a.rL64;
        return Co < kNegligible;
    };
#if defined(RH_LIMIT_NO_LOOKBACK) && RH_LIMIT_NO_LOOKBACK  // timing experiment only (wrong results): what do the polls cost?
    if (false) {
#else
    if (!have_I) {
#endif
        float Ao[C], Bo[C], Co = 1.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) Ao[c] = Bo[c] = 0.0f;
        int64_t base = (int64_t)tile - 1;
        while (true) {
            float AB[2 * C], Ij[C];
            const uint32_t jstar = poll_window<C, 2 * C, RC::wA>(gstream, base, lane, a.jI, G, RC::oA, init, AB, Ij, dead, a.spin);
            if (dead) break;
            if (fold_I(Ao, Bo, Co, AB, Ij, jstar)) break;
            base -= 64;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) Iin[c] = Ao[c];
    }
    if (NIO == 0 && !dma_first && dma_where != 2 && next_src) dma_share<V>(next_src, next_buf, lane);
    RH_LP(2)
    RH_ARGS_FRESH();
    // ---- true integrator per sample, zero-state attack run (g becomes the zero-state peak) ----------------------------------
    T I[N], Pz[N];
    const float rwave = a.rwave[__builtin_amdgcn_readfirstlane(wave)];
    const float a15 = tab[3][lane], a31 = tab[4][lane];
#pragma unroll
    for (int p = 0; p < N; ++p) {
        const T Iw = vmax(Ap[p], vfma(splat<T>(rwave), pair_of(Iin + p * W, T()), Bp[p]));  // this wave's start state
        I[p] = vmax(Ax[p], vfma(splat<T>(rlane), Iw, Bx[p]));                               // this lane's
        Pz[p] = splat<T>(0.0f);
    }
#pragma unroll
    for (int r = 0; r < (RH_LIMIT_NO_ARITH_ ? 1 : R); ++r)
#pragma unroll
        for (int p = 0; p < N; ++p) {
            // limit.rs:909-913 (P from a zero state), each recurrence step as one multiply and one FMA -- the product r*I (a*P) is not rounded on
            // its own, which moves a value by at most one ulp where the reference rounds twice; everything around it is compared at 1e-5
            const T In = vmax(g[r][p], vfma(relT, I[p], omrT * g[r][p]));
            const T Pn = vfma(attT, Pz[p], omaT * In);
            if (FULL || (uint32_t)r < nfl) {
                I[p] = In;
                Pz[p] = Pn;
            }
            g[r][p] = Pz[p];
        }
    const float ascan[4] = {a.ascan[0], a.ascan[1], a.ascan[2], a.ascan[3]};
    T Px[N];
#pragma unroll
    for (int p = 0; p < N; ++p) {
        T Pi = Pz[p];
        scan_lin_v<T>(Pi, ascan, a15, a31);
        Px[p] = vdpp<kWaveShr1, 0xf>(Pi);
        if (lane == 63) {
#pragma unroll
            for (int w = 0; w < W; ++w) xP[wave][p * W + w] = comp(Pi, w);
        }
    }
    const float wP = tab[7][lane], alane = tab[5][lane];
    // the ticket taken at the top of the tile (for the tile after next) goes to LDS only here: its atomic has had two phases to
    // return, instead of holding wave 0 -- and with it the whole workgroup at barrier (1) -- for a device-scope round trip
    if (threadIdx.x == 0) *ticket_slot = ticket_ahead;
    RH_LP(3)
    __syncthreads();  // (2) the waves' zero-state peak aggregates are in LDS
    RH_ARGS_FRESH();
    T Pp[N], PT[N];
#pragma unroll
    for (int p = 0; p < N; ++p) Pp[p] = PT[p] = splat<T>(0.0f);
    const T aLT = splat<T>(a.aL);
#pragma unroll kPrefixUnroll
    for (int k = 0; k < NW; ++k) {
#pragma unroll
        for (int p = 0; p < N; ++p) {
            if (k == wave) Pp[p] = PT[p];
            PT[p] = vfma(aLT, PT[p], pair_of(&xP[k][p * W], T()));
        }
    }
    if (wave == 0 && lane < C) {
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) v = lane == c ? comp(PT[c / W], c % W) : v;
        word_store(rec + RC::oZ + lane, v);
    }
    // ---- look-back for the peak of this tile, P_in = sum_{j<j*} a^(LW*j) Pz_{t-1-j} + a^(LW*j*) Pend_j*.  More streams than workgroups
    //      (SKEW): it was walked a tile ago, with the integrator's, by the look-ahead poll below; a workgroup's first tile and the
    //      other shapes walk it here ------------------------------------------------------------------------------------------------
    float Pin[C];
    auto fold_P = [&](float (&Po)[C], float &Co, const float (&Zj)[C], const float (&Ej)[C], uint32_t jstar) {
        const bool front = (uint32_t)lane < jstar, star = (uint32_t)lane == jstar;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float tot;
            (void)wave_excl_sum(front ? wP * Zj[c] : (star ? wP * Ej[c] : 0.0f), tot);
            Po[c] = fma_(Co, tot, Po[c]);
        }
        if (jstar < 64) return true;
        Co *= a.aL64;
        return Co < kNegligible;
    };
#pragma unroll
    for (int c = 0; c < C; ++c) Pin[c] = had_I ? Pcarry[c] : 0.0f;
    if (kLookback && !had_I && !dead) {
        float Co = 1.0f;
        int64_t baseP = (int64_t)tile - 1;
        while (true) {
            float Zj[C], Ej[C];
            const uint32_t jstar = poll_window<C, C, RC::wS>(gstream, baseP, lane, a.jP, G, RC::oZ, init ? init + 1 : nullptr, Zj, Ej, dead, a.spin);
            if (dead) break;
            if (fold_P(Pin, Co, Zj, Ej, jstar)) break;
            baseP -= 64;
        }
    }
    if (dead) {  // a hand-off never arrived: fail the call (status word) and poison the tile
        if (lane == 0) atomicOr(a.status, 1u);
#pragma unroll
        for (int c = 0; c < C; ++c) Pin[c] = __builtin_nanf("");
    }
    if (NIO == 0 && dma_where == 2 && next_src) dma_share<V>(next_src, next_buf, lane);
    RH_LP(4)
    RH_ARGS_FRESH();
    // ---- per-sample peak, gain coupled over the channels (limit.rs:946-960, :983-986); the result goes back to the LDS row ----
    T Ps[N];
    float Pcur[C];
    const float awave = a.awave[__builtin_amdgcn_readfirstlane(wave)];  // (a scalar load: a vector one would wait behind the look-ahead poll)
#pragma unroll
    for (int p = 0; p < N; ++p) {
        const T Pw = vfma(splat<T>(awave), pair_of(Pin + p * W, T()), Pp[p]);  // this wave's start state
        Ps[p] = vfma(splat<T>(alane), Pw, Px[p]);                              // this lane's
#pragma unroll
        for (int w = 0; w < W; ++w) Pcur[p * W + w] = comp(Ps[p], w);
    }
    const float kexp = -0.05f * LOG2_10;  // math.rs:51-56: 2^(dB * 0.05 * log2 10), the two constants folded
    __builtin_amdgcn_wave_barrier();
    float ap = 1.0f;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        v4f v = lds[slot_of<V>(lane, j)];
        float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; i += W) {
            const int r = (4 * j + i) / C, c0 = (4 * j + i) % C, p = c0 / W;
            if (c0 == 0) ap *= att;  // a^(r+1)
#if defined(RH_LIMIT_NO_ARITH) && RH_LIMIT_NO_ARITH
            if (r != 0) {
                const T out = pair_of(e + i, T()) * Ps[p];
#pragma unroll
                for (int w = 0; w < W; ++w) e[i + w] = comp(out, w);
                continue;
            }
#endif
            const T Pnew = vfma(splat<T>(ap), Ps[p], g[r][p]);
            // the reference advances one sample at a time: the gain of channel c sees this frame's peaks of the channels up to
            // c and the previous frame's of the others
            float mpw[W];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                Pcur[c0 + w] = comp(Pnew, w);
                float mp = C > 2 ? 0.0f : Pcur[0];  // LimitMulti folds from 0.0
#pragma unroll
                for (int k = (C > 2 ? 0 : 1); k < C; ++k) mp = fmaxf(mp, Pcur[k]);
                mpw[w] = mp;
            }
            const T gain_db = pair_of(mpw, T()) * splat<T>(kexp);
            T gain;
            if constexpr (W == 2) gain.x = __builtin_amdgcn_exp2f(gain_db.x), gain.y = __builtin_amdgcn_exp2f(gain_db.y);
            else gain = __builtin_amdgcn_exp2f(gain_db);
            const T out = pair_of(e + i, T()) * gain;
#pragma unroll
            for (int w = 0; w < W; ++w) e[i + w] = comp(out, w);
        }
        v.x = e[0], v.y = e[1], v.z = e[2], v.w = e[3];
        lds[slot_of<V>(lane, j)] = v;
    }
    // the block's end state: the lane that holds the stream's last frame (a stream's last wave share may be short or empty)
    if (a.state_out && nfl > 0 && f0 + (uint64_t)lane * R + nfl == a.frames) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float ps = comp(Ps[c / W], c % W);
            float pe = ps, apw = 1.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                apw *= att;
                pe = (uint32_t)r + 1 == nfl ? fma_(apw, ps, comp(g[r][c / W], c % W)) : pe;
            }
            a.state_out[((uint64_t)stream * C + c) * 2] = comp(I[c / W], c % W);
            a.state_out[((uint64_t)stream * C + c) * 2 + 1] = pe;
        }
    }
    RH_LP(5)
    // ---- the look-ahead poll comes home (it had the whole output pass to do so); further windows -- coefficients that do not forget
    //      within 64 tiles -- are walked here.  A hand-off that never arrives is left to the next tile, which then walks in place
    //      and fails the call ---------------------------------------------------------------------------------------------------------
    have_I = false;
    RH_ARGS_FRESH();
    const float *const gnext = a.gran + (uint64_t)nstream * a.tiles * G;
    const float *const ninit = a.state_in ? a.state_in + (uint64_t)nstream * C * 2 : nullptr;
    if (ahead && q.template complete<V>(NIO == 0 && next_src != nullptr, gnext, (int64_t)ntile - 1, a.jP, a.jI, ninit, lane, a.spin) && !dead) {
        float Ao[C], Bo[C], Pn[C], CoI = 1.0f, CoP = 1.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) Ao[c] = Bo[c] = Pn[c] = 0.0f;
        bool doneP = fold_P(Pn, CoP, q.Z, q.E, q.jstarP), doneI = fold_I(Ao, Bo, CoI, q.AB, q.Ij, q.jstarI), lost = false;
        int64_t baseP = (int64_t)ntile - 1 - 64, baseI = baseP;
        while (!doneP && !lost) {
            float Zj[C], Ej[C];
            const uint32_t jstar = poll_window<C, C, RC::wS>(gnext, baseP, lane, a.jP, G, RC::oZ, ninit ? ninit + 1 : nullptr, Zj, Ej, lost, a.spin);
            if (lost) break;
            doneP = fold_P(Pn, CoP, Zj, Ej, jstar);
            baseP -= 64;
        }
        while (!doneI && !lost) {
            float AB[2 * C], Ij[C];
            const uint32_t jstar = poll_window<C, 2 * C, RC::wA>(gnext, baseI, lane, a.jI, G, RC::oA, ninit, AB, Ij, lost, a.spin);
            if (lost) break;
            doneI = fold_I(Ao, Bo, CoI, AB, Ij, jstar);
            baseI -= 64;
        }
        if (!lost) {
            have_I = true;
#pragma unroll
            for (int c = 0; c < C; ++c) Icarry[c] = Ao[c], Pcarry[c] = Pn[c];
        }
    }
    // ---- LDS rows -> coalesced store -----------------------------------------------------------------------------------
    __builtin_amdgcn_wave_barrier();
    if (NIO == 0) {
        if (FULL || nf == L) store_share<V, true>(dst, lds, nfloat, lane);
        else store_share<V, false>(dst, lds, nfloat, lane);
    }
    __builtin_amdgcn_wave_barrier();  // the rows are free for the next tile
    // the hand-off table of the NEXT launch on this stream (the two alternate): this tile's record there back to "not yet" -- the launch
    // in front of this one used it, nothing reads it now, and the next launch then needs no kernel in front of it to clear it
    if (a.gran_other && wave == 0 && (uint32_t)lane < G) a.gran_other[((uint64_t)stream * a.tiles + tile) * G + lane] = __uint_as_float(kNotYet);
    RH_LP(6)
}
#undef a
#undef att
#undef relT
#undef omrT
#undef attT
#undef omaT
#undef RH_ARGS_FRESH

// NIO = 0: every wave fetches (LDS-DMA, a tile ahead) and stores its own share.  NIO > 0: the workgroup has NIO I/O waves behind
// its NW computing waves; they own the samples' way in and out, the computing waves touch vector memory for their hand-offs only.
//
// Why (round 4, tools/ubench/write_bw.hip): the kernel without its look-backs takes 0.214 ms, with them 0.245 -- not because a
// poll's round trip is long but because `s_waitcnt vmcnt(0)` behind a poll retires the wave's whole in-order queue first: the
// 8 KiB of LDS-DMA for the next tile and the 8 KiB of stores of the last one.  Every poll was a drain of the wave's I/O, twice per
// tile, and a drained queue is bandwidth not used.  And the I/O skeleton itself -- 8 waves each moving 8 KiB shares through their
// own ring -- copies at 5.2 TB/s, where TWO waves moving the whole 64 KiB tile reach 6.07 TB/s (the memory system likes fewer,
// longer queues: 4 waves 5.99, 8 waves 5.28).  So: B0 at the top of a tile = "the tile's samples have landed AND the tile before
// is complete in its rows"; behind it the I/O waves store the tile before (LDS rows -> whole lines, nt) and request the tile
// after into the buffer that just became free -- one burst per tile period, a full period ahead of its use -- then sit out
// barriers (1) and (2).  The computing waves' vmcnt holds polls and publishes, nothing else.
template <int C, int R, int NW, bool SKEW, int NIO = 0>
// (occupancy bound: 4 workgroups per CU where the registers allow it without a spill -- 4 channels x 4 frames do not: a spill is a
// vector-memory operation of the compiler's own inside the counted waits of the LDS-DMA; tests/test_code_objects.py checks)
__global__ __launch_bounds__(64 * (NW + NIO), (NW + NIO >= 4 ? (C * R <= 16 && C <= 3 ? 4 : 2) : 1)) void k_limit_scan(const LimitArgs a_by_value) {
    (void)a_by_value;
    ArgsC kargs = (ArgsC)__builtin_amdgcn_kernarg_segment_ptr();
#define a (*kargs)
    static_assert((C * R) % 4 == 0 && R <= kMaxR && NW <= kMaxNW, "a lane's run is whole 16-byte vectors");
    static_assert(NIO == 0 || NW % NIO == 0, "every I/O wave moves the same number of shares");
    constexpr int V = C * R / 4;  // 16-byte vectors per lane; a wave's share of a tile is V KiB
    constexpr uint32_t L = 64u * R, LW = L * NW;
    __shared__ __attribute__((aligned(1024))) v4f bufs[NW][2][64 * V];  // per wave: the tile being worked on and the one being fetched
    __shared__ float xI[NW][2 * C], xP[NW][C];
    __shared__ uint32_t s_ticket[3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t total = a.n_streams * a.tiles;  // host: < 2^32
    __shared__ float tab[8][64];  // r15 r31 rlane a15 a31 alane rlook alook per lane (LimitTabs)
    if (wave == 0) {
        tab[0][lane] = a.t.r15[lane], tab[1][lane] = a.t.r31[lane], tab[2][lane] = a.t.rlane[lane];
        tab[3][lane] = a.t.a15[lane], tab[4][lane] = a.t.a31[lane], tab[5][lane] = a.t.alane[lane];
        tab[6][lane] = a.t.rlook[lane], tab[7][lane] = a.t.alook[lane];
    }
#ifdef RH_LIMIT_PROFILE
    unsigned long long lp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // Tiles are handed out by one ticket per workgroup tile, tile-major over the streams (ticket k -> tile k / S of stream
    // k % S): a tile only ever waits for tiles with smaller tickets, which run or have finished.  Tickets are taken two
    // tiles ahead (thread 0, through LDS, published by the barriers of the tile in between): the tile after the current one
    // is known when the current one starts, so its samples are requested (LDS-DMA) before any of the current tile's work.
    auto share_of = [&](uint32_t ticket, int w, const float *&src, float *&dst, uint32_t &nf) {
        const uint32_t tile = ticket / a.n_streams, stream = ticket - tile * a.n_streams;
        const uint64_t f0 = (uint64_t)tile * LW + (uint64_t)w * L;
        nf = f0 >= a.frames ? 0u : (a.frames - f0 < L ? (uint32_t)(a.frames - f0) : L);
        src = a.src + stream * a.stride + f0 * C;
        dst = a.dst + stream * a.stride + f0 * C;
    };
    auto share = [&](uint32_t ticket, const float *&src, uint32_t &nf) {
        float *dst;
        share_of(ticket, wave, src, dst, nf);
    };
    auto tile_full = [&](uint32_t ticket) { return ((uint64_t)(ticket / a.n_streams) + 1) * LW <= a.frames; };  // every share of the tile is whole
    if (threadIdx.x == 0) {
        s_ticket[0] = atomicAdd(a.ctl, 1u) - a.ticket_base;
        s_ticket[1] = atomicAdd(a.ctl, 1u) - a.ticket_base;
    }
    __syncthreads();
    uint32_t cur = s_ticket[0], nxt = s_ticket[1];
    uint32_t n = 0;
    bool prev_full = false, have_I = false;
    float Icarry[C], Pcarry[C];
#pragma unroll
    for (int c = 0; c < C; ++c) Icarry[c] = Pcarry[c] = 0.0f;
    if constexpr (NIO > 0) {
        constexpr int PER = NW / NIO;  // shares per I/O wave
        const bool io = wave >= NW;
        const int w0 = (wave - NW) * PER;
        // a tile's way in: whole shares by LDS-DMA, the short ones at the end of a stream guarded (zeros behind the end)
        auto io_fetch = [&](uint32_t ticket, int b) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const float *src;
                float *dst;
                uint32_t nf;
                share_of(ticket, w0 + i, src, dst, nf);
                if (nf == L) dma_share<V>(src, bufs[w0 + i][b], lane);
                else load_share_guarded<C, V>(src, bufs[w0 + i][b], nf * C, lane);
            }
        };
        auto io_store = [&](uint32_t ticket, int b) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const float *src;
                float *dst;
                uint32_t nf;
                share_of(ticket, w0 + i, src, dst, nf);
                if (nf == L) store_share<V, true>(dst, bufs[w0 + i][b], nf * C, lane);
                else if (nf) store_share<V, false>(dst, bufs[w0 + i][b], nf * C, lane);
            }
        };
        if (io && cur < total) io_fetch(cur, 0);
        uint32_t prev = 0;
        while (cur < total) {
            if (io) {
                wait_vm<0>();                                      // this tile's samples have landed (and the stores of two tiles ago are done)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (a guarded share went through registers into its slots)
                __syncthreads();                                   // B0: ... and every computing wave has left the tile before
                // the I/O waves stand at the computing waves' barriers too: their work is dealt over the intervals so that they are
                // never the last to arrive -- the stores in the short interval up to (1) (the gain computer), the requests in the
                // long one up to (2) (integrator look-back + run), nothing behind (2)
                if (n > 0) io_store(prev, (n - 1) & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rows have been read: their buffer is free
                __syncthreads();                                   // (1)
                if (nxt < total) io_fetch(nxt, (n + 1) & 1);
                __syncthreads();                                   // (2)
            } else {
                uint32_t ticket_ahead = 0;
                if (threadIdx.x == 0) ticket_ahead = atomicAdd(a.ctl, 1u) - a.ticket_base;  // stored by limit_tile in front of its second barrier
                uint32_t *const ticket_slot = &s_ticket[(n + 2) % 3];
                const uint32_t tile = cur / a.n_streams, stream = cur - tile * a.n_streams;
                const float *src;
                uint32_t nf;
                share(cur, src, nf);
                const bool has_next = SKEW && nxt < total && nxt - cur < a.n_streams;
                const uint32_t ntile = has_next ? nxt / a.n_streams : 0u, nstream = has_next ? nxt - ntile * a.n_streams : 0u;
                __syncthreads();  // B0
                if (tile_full(cur)) limit_tile<C, R, NW, true, SKEW, NIO>(kargs, bufs[wave][n & 1], xI, xP, lane, wave, tile, stream, tab, nf, Icarry, Pcarry, have_I, has_next, ntile, nstream, nullptr, nullptr, ticket_ahead, ticket_slot RH_LP_ARG);
                else limit_tile<C, R, NW, false, SKEW, NIO>(kargs, bufs[wave][n & 1], xI, xP, lane, wave, tile, stream, tab, nf, Icarry, Pcarry, have_I, has_next, ntile, nstream, nullptr, nullptr, ticket_ahead, ticket_slot RH_LP_ARG);
            }
            prev = cur;
            cur = nxt;
            nxt = s_ticket[(n + 2) % 3];  // written before barrier (2) of the tile just done
            ++n;
        }
        __syncthreads();  // the last tile is complete in its rows
        if (io && n > 0) io_store(prev, (n - 1) & 1);
        wait_vm<0>();
    } else {
    if (cur < total) {
        const float *src;
        uint32_t nf;
        share(cur, src, nf);
        if (nf == L) dma_share<V>(src, bufs[wave][0], lane);
    }
    while (cur < total) {
        asm volatile("" : "+s"(kargs));  // (the arguments are re-read where they are used: see limit_tile)
        uint32_t ticket_ahead = 0;
        if (threadIdx.x == 0) ticket_ahead = atomicAdd(a.ctl, 1u) - a.ticket_base;  // stored by limit_tile in front of its second barrier
        uint32_t *const ticket_slot = &s_ticket[(n + 2) % 3];
        const uint32_t tile = cur / a.n_streams, stream = cur - tile * a.n_streams;
        const float *src;
        uint32_t nf;
        share(cur, src, nf);
        // this tile's DMA is older than everything else in flight; what may still be pending behind it are the V output stores
        // of the previous tile (if it was a whole share -- otherwise drain)
        if (prev_full) wait_vm<V>();
        else wait_vm<0>();
        const float *src2 = nullptr;
        if (nxt < total) {
            uint32_t nf2;
            share(nxt, src2, nf2);
            if (nf2 != L) src2 = nullptr;  // a short share is fetched by its own tile, guarded
        }
        // SKEW (the host's choice when there are more streams than workgroups): the integrator look-back of the next tile is
        // walked at THIS tile's poll point.  Its predecessors are then old tickets (nearest: nxt - S < cur) that published long
        // ago.  With few streams the predecessor of the next tile is some workgroup's own next tile; waiting for it here would
        // chain the workgroups (measured: 64 streams, 0.31 -> 0.76 ms).
        const bool has_next = SKEW && nxt < total && nxt - cur < a.n_streams;
        const uint32_t ntile = has_next ? nxt / a.n_streams : 0u, nstream = has_next ? nxt - ntile * a.n_streams : 0u;
        v4f *const buf2 = bufs[wave][(n + 1) & 1];
        // FULL is the TILE's property, the same for every wave of the workgroup: all of them run one instantiation, barriers included
        if (tile_full(cur)) limit_tile<C, R, NW, true, SKEW>(kargs, bufs[wave][n & 1], xI, xP, lane, wave, tile, stream, tab, nf, Icarry, Pcarry, have_I, has_next, ntile, nstream, src2, buf2, ticket_ahead, ticket_slot RH_LP_ARG);
        else limit_tile<C, R, NW, false, SKEW>(kargs, bufs[wave][n & 1], xI, xP, lane, wave, tile, stream, tab, nf, Icarry, Pcarry, have_I, has_next, ntile, nstream, src2, buf2, ticket_ahead, ticket_slot RH_LP_ARG);
        prev_full = nf == L;
        cur = nxt;
        nxt = s_ticket[(n + 2) % 3];  // written before barrier (2) of the tile just done
        ++n;
    }
    wait_vm<0>();
    }
#ifdef RH_LIMIT_PROFILE
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_limit_prof[i], lp_acc[i]);
#endif
#undef a
}

// Everything the launch finds in its scratch, written by ONE kernel of this library in front of it: the control words (ticket
// counter, status) zeroed, a snapshot of the caller's states (the last tile of a stream rewrites the state while early tiles
// may still read it), every hand-off word "not yet".  The scratch itself is the stream's own buffer (rh::stream_scratch).
// Neither hipMallocAsync/hipFreeAsync per call nor hipMemsetAsync: with them ~7 % of short GpuSource chains carried a wrong
// state into one tile (a zero aggregate or a zeroed state snapshot where the launch had written something else); either
// change alone lowered the rate, only both removed it (profiles/r02_limit_flake.md).
__global__ void k_limit_init(uint32_t *ctl, float *snap, const float *state, uint32_t n_state, uint32_t *words, uint64_t n_words) {
    const uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (uint64_t)gridDim.x * blockDim.x;
    if (i0 < 16) ctl[i0] = 0u;
    for (uint64_t i = i0; i < n_state; i += step) snap[i] = state[i];
    for (uint64_t i = i0; i < n_words; i += step) words[i] = 0xffffffffu;
}

using LimitFn = void (*)(const LimitArgs);
struct LimitVariant {
    int C, R, NW;
    LimitFn fn, fn_skew;  // fn_skew: the variant with the skewed integrator look-back, or nullptr
    int NIO;              // I/O waves behind the NW computing waves (0: every wave moves its own share)
};
#define RH_LV(c, r, nw) LimitVariant{c, r, nw, &k_limit_scan<c, r, nw, false>, (nw) >= 4 ? &k_limit_scan<c, r, nw, ((nw) >= 4)> : nullptr, 0}
#define RH_LVIO(c, r, nw, nio) LimitVariant{c, r, nw, &k_limit_scan<c, r, nw, false, nio>, &k_limit_scan<c, r, nw, true, nio>, nio}
const LimitVariant kVariants[] = {
    RH_LVIO(2, 16, 6, 2),  // 6 computing + 2 I/O waves, tiles of 6144 frames: only with RH_LIMIT_NIO=1 (rh_limit)
    RH_LV(1, 8, 8),  RH_LV(1, 16, 8), RH_LV(1, 16, 16), RH_LV(1, 8, 1),
    RH_LV(2, 8, 8),  RH_LV(2, 8, 16), RH_LV(2, 8, 4),  RH_LV(2, 8, 1), RH_LV(2, 16, 8), RH_LV(2, 16, 4),
    RH_LV(3, 4, 8),  RH_LV(3, 4, 1),  RH_LV(4, 4, 8),   RH_LV(4, 4, 1), RH_LV(5, 4, 8), RH_LV(5, 4, 1),
    RH_LV(6, 4, 8),  RH_LV(6, 4, 1),  RH_LV(7, 4, 8),   RH_LV(7, 4, 1), RH_LV(8, 4, 8), RH_LV(8, 4, 1),
};
#undef RH_LV
#undef RH_LVIO

size_t rec_stride(uint32_t channels) {
    switch (channels) {
        case 1: return Rec<1>::stride;
        case 2: return Rec<2>::stride;
        case 3: return Rec<3>::stride;
        case 4: return Rec<4>::stride;
        case 5: return Rec<5>::stride;
        case 6: return Rec<6>::stride;
        case 7: return Rec<7>::stride;
        default: return Rec<8>::stride;
    }
}

double ipow(double b, uint64_t e) {
    double r = 1.0;
    while (e) {
        if (e & 1) r *= b;
        b *= b;
        e >>= 1;
    }
    return r;
}

}  // namespace

extern "C" rh_status rh_limit(float *dst, const float *src, uint64_t frames, uint32_t channels, uint32_t sample_rate, uint32_t n_streams, const rh_limit_params *p, float *state, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (channels == 0 || sample_rate == 0 || !p) return RH_ERR_INVALID;
    if (channels > 8) return RH_ERR_UNSUPPORTED;
    if (frames == 0 || n_streams == 0) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    hipStream_t s = rh::as_stream(stream);
    const float attack = rh::duration_to_coefficient_f32(p->attack_ns, sample_rate);    // limit.rs:96-97
    const float release = rh::duration_to_coefficient_f32(p->release_ns, sample_rate);
    const float k5[5] = {p->threshold_db, p->knee_width_db, 1.0f / (8.0f * p->knee_width_db) /* limit.rs:877 */, attack, release};
    const uint64_t stride = frames * channels;
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0 && (n_streams == 1 || stride % 4 == 0);
    // coefficients outside [0, 1) (zero or negative durations give exp(-inf) = 0 or exp(+x) > 1) leave the scan's premises
    const bool scannable = release >= 0.0f && release < 1.0f && attack >= 0.0f && attack < 1.0f && p->knee_width_db > 0.0f;
    const char *force_seq = rh::knob(rh::K_LIMIT_SEQ);  // diagnostics: the reference-order kernel
    if (!aligned || !scannable || (force_seq && force_seq[0] == '1')) return rh::limit_seq_launch(dst, src, frames, channels, n_streams, k5, state, s);

    // Geometry: the LONGEST tile (64 * R * NW frames) that a stream fills at least half of; among equals, more frames per lane.
    // Long tiles amortise the scans and the look-backs, and a slowly decaying recurrence (100 ms of release) reaches back
    // over MANY short tiles: measured (profiles/r02_scan_geometry_midsize.txt) 8192-frame tiles win from 64 x 1 Mi frames down to
    // 256 x 8192 (21 us against 49 us with single-wave tiles), although the short batches then have fewer tiles than the chip
    // has CUs.  Single-wave tiles are for blocks of a few hundred frames (a pull shim's).
    const LimitVariant *v = nullptr;
    if (rh::knob(rh::K_LIMIT_R) || rh::knob(rh::K_LIMIT_NW)) {  // tuning aids: the variant closest to the request
        const int want_R = rh::knob(rh::K_LIMIT_R) ? atoi(rh::knob(rh::K_LIMIT_R)) : 16, want_NW = rh::knob(rh::K_LIMIT_NW) ? atoi(rh::knob(rh::K_LIMIT_NW)) : 8;
        for (const LimitVariant &c : kVariants) {
            if (c.C != (int)channels || c.NIO) continue;
            auto score = [&](const LimitVariant &x) { return 10 * std::abs(x.NW - want_NW) + std::abs(x.R - want_R); };
            if (!v || score(c) < score(*v)) v = &c;
        }
    } else {
        auto tile_of = [](const LimitVariant &x) { return (uint64_t)64 * x.R * x.NW; };
        const char *nio_knob = rh::knob(rh::K_LIMIT_NIO);
        const bool want_io = nio_knob && nio_knob[0] == '1';  // RH_LIMIT_NIO=1: the I/O-wave variant (measured SLOWER: DESIGN.md 5.1; kept selectable so that the
        for (const LimitVariant &c : kVariants) {             // measurement can be repeated, and tested: tests/test_gpu_limit.py)
            if (want_io && c.NIO && c.C == (int)channels && tile_of(c) <= 2 * frames) v = &c;
        }
        for (const LimitVariant &c : kVariants) {
            if (v && v->NIO) break;
            if (c.C != (int)channels || c.NW > 8 || c.NIO) continue;  // (16-wave tiles: only on request)
            if (!v) {
                v = &c;
                continue;
            }
            const bool fits_c = tile_of(c) <= 2 * frames, fits_v = tile_of(*v) <= 2 * frames;
            const bool better = fits_c != fits_v ? fits_c
                                : (fits_c ? (tile_of(c) > tile_of(*v) || (tile_of(c) == tile_of(*v) && c.R > v->R))   // the longest that fits
                                          : tile_of(c) < tile_of(*v));                                                // nothing fits: the shortest
            if (better) v = &c;
        }
    }
    if (!v) return RH_ERR_UNSUPPORTED;
    const uint32_t R = (uint32_t)v->R, NW = (uint32_t)v->NW, L = 64u * R, LW = L * NW;
    const uint64_t tiles64 = (frames + LW - 1) / LW;
    if (tiles64 > 0x7fffffffull || tiles64 * n_streams >= 0xfff00000ull) return RH_ERR_UNSUPPORTED;  // tickets are 32-bit

    LimitArgs a;
    std::memset(&a, 0, sizeof(a));
    a.dst = dst;
    a.src = src;
    a.frames = frames;
    a.stride = stride;
    a.n_streams = n_streams;
    a.tiles = (uint32_t)tiles64;
    a.threshold = k5[0];
    a.knee_width = k5[1];
    a.inv_knee_8 = k5[2];
    a.attack = attack;
    a.release = release;
    a.one_minus_attack = 1.0f - attack;
    a.one_minus_release = 1.0f - release;
    // every power of the two coefficients the kernel uses (f64, rounded once): ~600 of them, the same for every block of a
    // stream -- a pull shim calls this once per block, so the last set is kept per host thread
    struct Consts {
        float release = -1.0f, attack = -1.0f;
        uint32_t R = 0, NW = 0;
        LimitArgs a;
    };
    static thread_local Consts cache;
    if (cache.release != release || cache.attack != attack || cache.R != R || cache.NW != NW) {
        LimitArgs &c = cache.a;
        std::memset(&c, 0, sizeof(c));
        const double r = release, al = attack;
        for (int k = 0; k < 4; ++k) {
            c.rscan[k] = (float)ipow(r, (uint64_t)R << k);
            c.ascan[k] = (float)ipow(al, (uint64_t)R << k);
        }
        c.rL = (float)ipow(r, L);
        c.aL = (float)ipow(al, L);
        c.rLW = (float)ipow(r, LW);
        c.aLW = (float)ipow(al, LW);
        for (uint32_t k = 0; k < NW; ++k) {
            c.rwave[k] = (float)ipow(r, (uint64_t)L * k);
            c.awave[k] = (float)ipow(al, (uint64_t)L * k);
        }
        c.rL64 = (float)ipow(r, (uint64_t)LW * 64);
        c.aL64 = (float)ipow(al, (uint64_t)LW * 64);
        c.jI = c.jP = 64;
        for (int l = 0; l < 64; ++l) {
            c.t.r15[l] = (float)ipow(r, (uint64_t)R * ((l & 15) + 1));
            c.t.r31[l] = (float)ipow(r, (uint64_t)R * ((l & 31) + 1));
            c.t.rlane[l] = (float)ipow(r, (uint64_t)R * l);
            c.t.a15[l] = (float)ipow(al, (uint64_t)R * ((l & 15) + 1));
            c.t.a31[l] = (float)ipow(al, (uint64_t)R * ((l & 31) + 1));
            c.t.alane[l] = (float)ipow(al, (uint64_t)R * l);
            c.t.rlook[l] = (float)ipow(r, (uint64_t)LW * l);
            c.t.alook[l] = (float)ipow(al, (uint64_t)LW * l);
            if (c.jI == 64 && c.t.rlook[l] < kNegligible) c.jI = (uint32_t)l;
            if (c.jP == 64 && c.t.alook[l] < kNegligible) c.jP = (uint32_t)l;
        }
        if (c.jI == 0) c.jI = 1;
        if (c.jP == 0) c.jP = 1;
        cache.release = release, cache.attack = attack, cache.R = R, cache.NW = NW;
    }
    {
        const LimitArgs &c = cache.a;
        std::memcpy(a.rscan, c.rscan, sizeof(a.rscan));
        std::memcpy(a.ascan, c.ascan, sizeof(a.ascan));
        a.rL = c.rL, a.aL = c.aL, a.rLW = c.rLW, a.aLW = c.aLW, a.rL64 = c.rL64, a.aL64 = c.aL64;
        std::memcpy(a.rwave, c.rwave, sizeof(a.rwave));
        std::memcpy(a.awave, c.awave, sizeof(a.awave));
        a.jI = c.jI, a.jP = c.jP;
        a.t = c.t;
    }

    // scratch: control words + the carried-in states + TWO hand-off tables.  A launch works on one of them and sets the other one's
    // records back to "not yet" as its tiles finish, so the next launch of the same shape on this stream finds its table clean and its
    // ticket counter where the host knows it to be: no kernel in front of it (k_limit_init and its boundary were 5 % of a 0.27 ms call).
    // The first launch of a shape, a launch with a carried state (its snapshot is a kernel anyway) and a launch behind another user of
    // the stream's scratch (rh::ScratchAux) initialise both tables.
    const size_t n_state = (size_t)n_streams * channels * 2;
    const size_t gran_bytes = (((size_t)n_streams * tiles64 * rec_stride(channels) * sizeof(float)) + 63) & ~size_t(63);
    const size_t head = 64 + ((n_state * 4 + 63) & ~size_t(63));
    unsigned char *scratch = nullptr;
    std::unique_lock<std::mutex> scratch_hold;
    rh::ScratchAux *aux = nullptr;
    RH_HIP_TRY(rh::stream_scratch(s, head + 2 * gran_bytes, reinterpret_cast<void **>(&scratch), scratch_hold, &aux));
    uint64_t tag = 0x4c494d4954ull;  // "LIMIT", then the shape (FNV-1a)
    for (uint64_t v : {(uint64_t)n_streams, tiles64, (uint64_t)channels, (uint64_t)head, (uint64_t)gran_bytes, (uint64_t)reinterpret_cast<uintptr_t>(scratch)}) tag = (tag ^ v) * 0x100000001b3ull;
    tag |= 1;  // (never 0)
    const char *init_knob = rh::knob(rh::K_LIMIT_INIT);
    const bool clean = !state && aux->tag == tag && !(init_knob && init_knob[0] == '1');
    a.ctl = reinterpret_cast<uint32_t *>(scratch);
    a.status = rh::g_async_status;
    a.dma_top = rh::knob(rh::K_SCAN_DMA_TOP) ? (uint32_t)atoi(rh::knob(rh::K_SCAN_DMA_TOP)) : 1u;  // measured: 0.312 -> 0.286 ms (limiter), 0.234 -> 0.221 ms (biquad), 64 x 1 Mi frames
    a.spin = rh::knob(rh::K_SCAN_SPIN_LIMIT) ? (uint32_t)strtoul(rh::knob(rh::K_SCAN_SPIN_LIMIT), nullptr, 10) : kSpinLimit;
    float *snap = reinterpret_cast<float *>(scratch + 64);
    hipError_t e = hipSuccess;
    if (!clean) {
        const uint64_t n_words = 2 * gran_bytes / 4;
        const unsigned init_wgs = (unsigned)std::min<uint64_t>(1024, (std::max<uint64_t>(n_words, n_state) + 255) / 256);
        hipLaunchKernelGGL(k_limit_init, dim3(init_wgs), dim3(256), 0, s, a.ctl, snap, state, state ? (uint32_t)n_state : 0u, reinterpret_cast<uint32_t *>(scratch + head), n_words);
        e = hipGetLastError();
        aux->tag = state ? 0 : tag;
        aux->ticket_base = 0;
        aux->parity = 0;
    }
    a.gran = reinterpret_cast<float *>(scratch + head + (state ? 0 : aux->parity) * gran_bytes);
    a.gran_other = state ? nullptr : reinterpret_cast<float *>(scratch + head + (aux->parity ^ 1u) * gran_bytes);
    a.ticket_base = aux->ticket_base;
    if (state) a.state_in = snap, a.state_out = state;
    if (e == hipSuccess) {
        static int occupancy[sizeof(kVariants) / sizeof(kVariants[0])];  // asked once per variant (both instantiations share registers and LDS)
        int &per_cu_cached = occupancy[v - kVariants];
        if (per_cu_cached == 0) {
            int q = 0;
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, reinterpret_cast<const void *>(v->fn), 64 * (int)(NW + v->NIO), 0);
            per_cu_cached = q < 1 ? 1 : q;
        }
        int per_cu = per_cu_cached;
        if (per_cu * (int)(NW + v->NIO) > 16) per_cu = 16 / (int)(NW + v->NIO) > 0 ? 16 / (int)(NW + v->NIO) : 1;
        if (const char *w = rh::knob(rh::K_LIMIT_WGS)) per_cu = atoi(w) > 0 ? atoi(w) : per_cu;  // tuning aid: resident workgroups per CU
        uint64_t grid = (uint64_t)rh::g_num_cus * (uint64_t)per_cu;
        const uint64_t total = tiles64 * n_streams;
        if (grid > total) grid = total;
        if (const char *g = rh::knob(rh::K_LIMIT_GRID)) grid = atoi(g) > 0 ? (uint64_t)atoi(g) : grid;  // diagnostics
        if (e == hipSuccess) {
            void *args[] = {&a};
            bool skew = v->fn_skew && grid < n_streams;  // see k_limit_scan: only with more streams than workgroups
            if (const char *k = rh::knob(rh::K_LIMIT_SKEW)) skew = v->fn_skew && k[0] == '1';  // tuning aid
            e = hipLaunchKernel(reinterpret_cast<const void *>(skew ? v->fn_skew : v->fn), dim3((uint32_t)grid), dim3(64 * (NW + (uint32_t)v->NIO)), args, 0, s);
            if (e == hipSuccess && !state) {  // every workgroup takes two tickets ahead and one per tile it works on
                aux->ticket_base += (uint32_t)(total + 2 * grid);
                aux->parity ^= 1u;
            }
        }
    }
    if (e != hipSuccess) {
        aux->tag = 0;  // (whatever state the tables are in: the next call starts over)
        rh::set_hip_error(e, "rh_limit launch");
        return RH_ERR_HIP;
    }
    return RH_OK;
}

// Diagnostics, only in -DRH_LIMIT_PROFILE builds (RH_ERR_UNSUPPORTED otherwise): shader cycles per phase {load, gain+segment,
// look-back I, integrator run, look-back P, gain stage, store} summed over all tiles since the last call.  Not in rodio_hip.h.
extern "C" rh_status rh_limit_phase_cycles(double out8[8]) {
#ifdef RH_LIMIT_PROFILE
    unsigned long long h[16];
    RH_HIP_TRY(hipDeviceSynchronize());
    RH_HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_limit_prof), sizeof(h)));
    for (int i = 0; i < 8; ++i) out8[i] = (double)h[i];
    std::memset(h, 0, sizeof(h));
    RH_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_limit_prof), h, sizeof(h)));
    return RH_OK;
#else
    (void)out8;
    return RH_ERR_UNSUPPORTED;
#endif
}

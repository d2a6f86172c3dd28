// rh_agc.hip -- AutomaticGainControl (src/source/agc.rs:133-171, :397-504) taken apart along its dependency chains.
//
// process_sample (agc.rs:433-504) per sample, one state for all interleaved channels of a stream:
//     s      = |x|
//     peak   = peak * c + s * (1 - c),  c = s > peak ? 0 : release                      update_peak_level   :397-407
//     sum    = sum - old + s*s          (old: the square that leaves the 8192-sample window)   CircularBuffer::push :152-163
//     rms    = sqrt(sum / 8192);  rms_gain = rms > 0 ? target / rms : max_gain
//     peak_gain = peak > 0 ? min(target / peak, max_gain) : max_gain;  desired = max(min(rms_gain, peak_gain), floor)
//     gain   = clamp(gain * a + desired * (1 - a), 0.1, max_gain),  a = desired > gain ? attack : release
//     y      = x * gain
// Only THREE short chains run along time: the window sum (two dependent adds), the peak follower (with the default release of 0
// it is no chain at all: peak = s) and the gain (multiply, add, select, clamp).  Everything expensive -- the square root and the
// two IEEE divides -- depends on the chains' RESULTS only.  Round 2 ran all of it on one lane per stream, 73 vector
// instructions per sample one after the other (0.03-0.9 % of the HBM roofline).  Here:
//     k_agc_chain<SumOp>     one lane per stream: the squares and 2 dependent adds per sample                 -> sum[n]      (into dst)
//     k_agc_chain<PeakOp>    only when release != 0: the reference's select / multiply / add                  -> peak[n]     (scratch)
//     k_agc_desired          every lane of the chip: sqrt, divides, min / max                                -> desired[n]  (in place)
//     k_agc_chain<GainOp>    release != 0: both candidates, select, clamp, output multiply                    -> y[n]        (in place)
//     k_agc_chain<GainOp0>   release == 0 (the default): the release candidate is `desired` itself; the chain is multiply, add,
//                            max, compare, select on values the parallel pass prepared                         -> gain[n]     (in place)
//     k_agc_apply            y = x * gain, every lane of the chip
// and, since round 4, for EVERY aligned out-of-place call unless RH_AGC_SEGMENTS=1 asks for the form above (k_agc_fused<true>: any parameters,
// the peak follower as a third chain wave; more streams than the chip holds workgroups of 16 simply queue):
//     k_agc_fused            ALL of the above in one workgroup per 16 streams: the two chains on a wave each, the square roots and
//                            divides on four more, loaders and storers around them -- nothing but x in and y out   (see there)
//     k_agc_fused0           round 5, the default parameters (release == 0): the same pipeline with everything that is not a chain
//                            operation taken off the two chain waves (a wave's time is its instruction count x 6-7 cycles)   (see there)
// Both chains MUST round like the reference, step by step: the window sum drifts 7e-5 relative over 2 Mi samples when
// re-associated, and the gain -- a one-pole with a 4 s time constant, 192 000 samples at 48 kHz -- integrates its own rounding
// noise to 5e-6 relative (x gain 7 = 4e-5 on the output): a scan over composed gain maps (g -> min(H, max(L, c*g + B)) is
// closed under composition) was built and measured 7e-5 from the reference, outside the bound.  So the operations and their
// order are the reference's.  The general path (GainOp: any release) is bit-identical to the reference-order kernel (k_agc_seq in
// rh_recurrence.hip, which stays for unaligned rows and in-place calls); the default-parameter path (GainOp0, release == 0) folds
// the select and the clamp into one median and is bit-identical EXCEPT where the attack candidate and `desired` lie within a rounding
// of each other, where it differs by that rounding (one ulp, not accumulating: the recurrence contracts) -- measured 0.0 on every
// test signal, compared at 2e-7.  Non-finite samples (round 6, derived from agc.rs in tests/golden/derive_traces.py): the reference's GAIN
// never becomes NaN -- a NaN sample poisons the window sum and the peak level for good, both `> 0.0` tests fail from then on, `desired` is
// absolute_max_gain and the gain climbs to the ceiling and stays there (the NaN sample itself comes out NaN).  The kernels do the same: the sums
// run in the reference's order (NaN stays), and the default-parameter paths, which take |x| for the peak level, poison it with the sum
// (agc_peak0); test_gpu_agc_after_a_nan holds them against the derived trace.
//
// What a chain costs: ONE wave walks time for 64 streams, and a lone wave issues one instruction every 5-7 cycles whatever its
// kind (measured: SQ_WAVE_CYCLES / instructions), so the chain wave carries nothing but the chain -- wave 0 issues LDS reads,
// the chain's arithmetic and LDS writes; waves 1 to 4 keep the next six chunks (32 samples per stream each) of the two inputs
// coming by LDS-DMA, half an array each (an LDS-DMA issue costs its wave 100-300 cycles, and a chain is also bound by memory
// LATENCY: 1.6 us per round trip when one workgroup is all that runs); wave 5 writes the previous chunk's results back.  Transfers are whole 128-byte lines through the swizzled tile image of
// rh_scan_common.h: the chain wave reads and writes its 16-byte vectors without bank conflicts.  One barrier per chunk.
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "rh_common.h"

namespace {
#include "rh_scan_common.h"

constexpr uint32_t kRmsWindow = 8192;              // agc.rs:51
constexpr size_t kAgcStateFloats = 4 + kRmsWindow;  // {sum, index(bits), peak_level, current_gain, ring[8192]} (as in rh_recurrence.hip)
constexpr int kCS = 32;                             // samples per chunk and stream
constexpr int kV = kCS / 4;                         // 16-byte vectors per lane and chunk
constexpr int kGS = 64;                             // streams per workgroup: one per lane of the chain wave
constexpr int kDma = kGS * kV / 64;                 // LDS-DMA instructions per chunk and array: 8
constexpr int kRing = 8;                            // chunks of input in LDS: one in use, up to 6 in flight (6 x 8 = 48 outstanding per loader, vmcnt counts to 63)
constexpr int kAhead = kRing - 2;
constexpr uint32_t kSlotVecs = kGS * kV;            // one chunk of one array: 512 vectors = 8 KiB
constexpr uint32_t kSlotBytes = kSlotVecs * 16;
constexpr uint32_t kHeadChunks = kRmsWindow / kCS;
// LDS: ring slot r = [first input | second input] at r * 2 * kSlotBytes (the second input is an immediate offset away from the
// first), then the two result images
constexpr uint32_t kOutBase = 2 * kRing * kSlotBytes;
constexpr size_t kChainLds = (size_t)kOutBase + 2 * kSlotBytes;  // 144 KiB

struct AgcK {
    float target_level, attack_coeff, release_coeff, absolute_max_gain, floor;
};

struct ChainArgs {
    const float *in0, *in1;  // rows of n samples, stride0 / stride1 floats apart (in1: SumOp's samples 8192 back, GainOp's desired gains)
    const float *in1_head;   // SumOp: what leaves the window during the first 8192 samples -- the carried window, [streams][8192] in time order; null: zeros
    float *out;
    uint64_t n, stride0, stride1, stride_out;
    uint32_t n_streams;
    uint32_t head_chunks;    // chunks whose second input is in1_head (SumOp: 256; the others: 0)
    float *state;            // per stream `state_stride` floats {sum, -, peak, gain, ...}: the caller's, or four scratch words per stream
    uint32_t state_stride;
    AgcK k;
};

// ---- the chains: the reference's operations in the reference's order (this TU is built with -ffp-contract=off) -------------
struct SumOp {  // CircularBuffer::push, agc.rs:152-163: sum = sum - old + new
    static constexpr bool kTwoIn = true;
    float sum;
    __device__ void init(const ChainArgs &, const float *st) { sum = st[0]; }
    // HEAD: `o` is a square already (the carried window); otherwise the sample 8192 back
    __device__ __forceinline__ float one(float x, float o, bool head) {
        const float nw = x * x, od = head ? o : o * o;  // |x| * |x| (agc.rs:414) == x * x
        sum = sum - od + nw;
        return sum;
    }
    // (component by component: a vector multiply becomes v_pk_mul_f32, and a packed f32 instruction holds a lone wave ~20 cycles
    // longer than a plain one -- measured 22.5 against 13.5 ns per sample for the gain chain of the same instruction count)
    template <bool HEAD>
    __device__ __forceinline__ v4f step4(v4f x, v4f o) {
        v4f nw, od;
#ifdef RH_AGC_DIAG_NOSQ  // diagnostics builds (wrong results): what do the squares cost the chain wave?
        nw = x, od = o;
#else
        nw.x = x.x * x.x, nw.y = x.y * x.y, nw.z = x.z * x.z, nw.w = x.w * x.w;
        if (HEAD) od = o;
        else od.x = o.x * o.x, od.y = o.y * o.y, od.z = o.z * o.z, od.w = o.w * o.w;
#endif
        v4f r;
        sum = sum - od.x + nw.x, r.x = sum;
        sum = sum - od.y + nw.y, r.y = sum;
        sum = sum - od.z + nw.z, r.z = sum;
        sum = sum - od.w + nw.w, r.w = sum;
        return r;
    }
    __device__ void finish(float *st) { st[0] = sum; }
};
struct SumSqOp {  // the same on squares a parallel pass prepared (rows apart by a stride that is NOT a power of two): two adds
    static constexpr bool kTwoIn = true;
    float sum;
    __device__ void init(const ChainArgs &, const float *st) { sum = st[0]; }
    __device__ __forceinline__ float one(float nw, float od, bool) {
        sum = sum - od + nw;
        return sum;
    }
    template <bool HEAD>
    __device__ __forceinline__ v4f step4(v4f nw, v4f od) {
        v4f r;
        sum = sum - od.x + nw.x, r.x = sum;
        sum = sum - od.y + nw.y, r.y = sum;
        sum = sum - od.z + nw.z, r.z = sum;
        sum = sum - od.w + nw.w, r.w = sum;
        return r;
    }
    __device__ void finish(float *st) { st[0] = sum; }
};
struct PeakOp {  // update_peak_level, agc.rs:397-407
    static constexpr bool kTwoIn = false;
    float peak, rel;
    __device__ void init(const ChainArgs &a, const float *st) {
        peak = st[2];
        rel = a.k.release_coeff;
    }
    __device__ __forceinline__ float one(float x, float, bool) {
        const float s = fabsf(x);
        const float c = s > peak ? 0.0f : rel;
        peak = peak * c + s * (1.0f - c);
        return peak;
    }
    template <bool HEAD>
    __device__ __forceinline__ v4f step4(v4f x, v4f) {
        v4f r;
        r.x = one(x.x, 0.f, false), r.y = one(x.y, 0.f, false), r.z = one(x.z, 0.f, false), r.w = one(x.w, 0.f, false);
        return r;
    }
    __device__ void finish(float *st) { st[2] = peak; }
};
struct GainOp {  // agc.rs:486-499, and the output multiply :503
    static constexpr bool kTwoIn = true;
    float gain, att, rel, oma, omr, maxg;
    __device__ void init(const ChainArgs &a, const float *st) {
        gain = st[3];
        att = a.k.attack_coeff;
        rel = a.k.release_coeff;
        oma = 1.0f - att;
        omr = 1.0f - rel;
        maxg = a.k.absolute_max_gain;
    }
    // gain * speed + desired * (1 - speed) for both speeds side by side (mul, mul, add: not contracted); what waits for the
    // previous gain is one multiply, one add, the select and the clamp -- the other four operations fill the gaps.  Plain
    // scalar operations on purpose: the packed forms (v_pk_mul_f32 / v_pk_add_f32) stall a dependent chain.
    __device__ __forceinline__ float one(float x, float d, bool) {
        const float da = d * oma, dr = d * omr;
        const float ca = gain * att + da, cr = gain * rel + dr;
        const float g = d > gain ? ca : cr;             // attack_speed = desired > current ? attack : release
        gain = __builtin_amdgcn_fmed3f(g, 0.1f, maxg);  // f32::clamp(0.1, absolute_max_gain) of a number
        return x * gain;
    }
    template <bool HEAD>
    __device__ __forceinline__ v4f step4(v4f x, v4f d) {
        v4f r;
        r.x = one(x.x, d.x, false), r.y = one(x.y, d.y, false), r.z = one(x.z, d.z, false), r.w = one(x.w, d.w, false);
        return r;
    }
    __device__ void finish(float *st) { st[3] = gain; }
};

struct GainOp0 {  // release == 0 (agc.rs:73-82's default): the release candidate gain * 0 + desired * 1 IS desired.
    // in0 = dc = clamp(desired, 0.1, max), in1 = da = desired * (1 - attack), both from the parallel pass.  With 0.1 <= gain <= max
    // (clamped since the first sample; a fresh AGC starts at 1.0):
    //     desired >  gain: the attack candidate A = gain * attack + da lies below desired, its clamp is clamp(A, 0.1, dc)
    //     desired <= gain: A >= desired, and the reference takes clamp(desired) = dc = clamp(A, 0.1, dc)
    // so the select and both clamps are ONE median: gain' = med3(A, 0.1, dc).  A is the reference's mul, mul, add (da carries the
    // second product); the result differs from the reference's only when A and desired are within a rounding of each other, by
    // that rounding (one ulp, once: the recurrence contracts) -- the tests compare against the reference-order kernel at 2e-7.
    static constexpr bool kTwoIn = true;
    float gain, att;
    __device__ void init(const ChainArgs &a, const float *st) {
        gain = st[3];
        att = a.k.attack_coeff;
    }
    __device__ __forceinline__ float one(float dc, float da, bool) {
        gain = __builtin_amdgcn_fmed3f(gain * att + da, 0.1f, dc);
        return gain;
    }
    template <bool HEAD>
    __device__ __forceinline__ v4f step4(v4f dc, v4f da) {
        v4f r;
        r.x = one(dc.x, da.x, false), r.y = one(dc.y, da.y, false), r.z = one(dc.z, da.z, false), r.w = one(dc.w, da.w, false);
        return r;
    }
    __device__ void finish(float *st) { st[3] = gain; }
};

__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- the skeleton ---------------------------------------------------------------------------------------------------------
// waves: 0 the chain; 1, 2 load the first input (an LDS-DMA instruction costs its wave 100-200 cycles of issue: four per wave and
// chunk keep the loaders ahead of the chain); 3, 4 load the second input; 5 stores the results.  Barrier c (one per chunk, all
// six waves) says: chunk c is in LDS, and the chain is done with chunk c - 1 (its ring slot is free, its results are complete).
template <class Op>
__device__ __forceinline__ void chain_body(const ChainArgs &a, uint32_t group) {
    constexpr bool TWO = Op::kTwoIn;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_u8 *const lds = (lds_u8 *)smem;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x / 64), lane = (int)threadIdx.x & 63;
    const uint32_t g0 = group * (uint32_t)kGS;                                                  // first stream of the group
    const uint32_t live = a.n_streams - g0 < (uint32_t)kGS ? a.n_streams - g0 : (uint32_t)kGS;   // streams in it
    const uint64_t nch = a.n / kCS;                                                             // whole chunks; the rest is the chain wave's epilogue
    if (wave == 0) {  // ---- the chain: lane = stream; LDS reads, the chain's arithmetic, LDS writes, nothing else ----
        const bool mine = (uint32_t)lane < live;
        const uint32_t stream = g0 + (mine ? (uint32_t)lane : live - 1);
        float *st = a.state + (uint64_t)stream * a.state_stride;
        Op op;
        op.init(a, st);
        uint32_t sl[kV];  // byte offset of this lane's vector j inside a chunk image
#pragma unroll
        for (int j = 0; j < kV; ++j) sl[j] = slot_of<kV>((uint32_t)lane, (uint32_t)j) * 16u;
        // one chunk: all reads first, then the chain.  HEAD / ZERO are per-chunk facts: separate bodies, no per-sample selects
        auto chunk = [&](uint64_t c, auto head_tag, auto zero_tag) {
            constexpr bool HEAD = decltype(head_tag)::value, ZERO = decltype(zero_tag)::value;
            const lds_u8 *in = lds + (uint32_t)(c % kRing) * (2u * kSlotBytes);
            lds_u8 *img = lds + kOutBase + (uint32_t)(c & 1) * kSlotBytes;
            v4f x[kV], o[kV];
#pragma unroll
            for (int j = 0; j < kV; ++j) {
                x[j] = *(const __attribute__((address_space(3))) v4f *)(in + sl[j]);
                o[j] = ZERO ? v4f{0.f, 0.f, 0.f, 0.f} : *(const __attribute__((address_space(3))) v4f *)(in + sl[j] + kSlotBytes);
            }
#pragma unroll
            for (int j = 0; j < kV; ++j) *(__attribute__((address_space(3))) v4f *)(img + sl[j]) = op.template step4<HEAD>(x[j], o[j]);
        };
        const uint64_t nhead = TWO ? (nch < a.head_chunks ? nch : (uint64_t)a.head_chunks) : 0;
        uint64_t c = 0;
        if (a.in1_head) {
            for (; c < nhead; ++c) {
                barrier_lds();
                chunk(c, std::true_type{}, std::false_type{});
            }
        } else {
            for (; c < nhead; ++c) {
                barrier_lds();
                chunk(c, std::true_type{}, std::true_type{});
            }
        }
        for (; c < nch; ++c) {
            barrier_lds();
            if (TWO) chunk(c, std::false_type{}, std::false_type{});
            else chunk(c, std::false_type{}, std::true_type{});
        }
        barrier_lds();
        if (mine) {  // the last n % 32 samples: straight from memory, one at a time
            const float *r0 = a.in0 + (uint64_t)stream * a.stride0;
            float *ro = a.out + (uint64_t)stream * a.stride_out;
            for (uint64_t i = nch * kCS; i < a.n; ++i) {
                const bool head = TWO && i < (uint64_t)a.head_chunks * kCS;
                float ov = 0.0f;
                if (TWO) ov = head ? (a.in1_head ? a.in1_head[(uint64_t)stream * kRmsWindow + i] : 0.0f) : a.in1[(uint64_t)stream * a.stride1 + i];
                const float xv = r0[i];
                ro[i] = op.one(xv, ov, head);
            }
            op.finish(st);
        }
        return;
    }
    // per-lane geometry of the line-wise transfers: slot q = k*64 + lane of a chunk image holds vector vec_in_slot(q) = o*kV + j,
    // i.e. samples 4j..4j+3 of stream o's chunk (surplus streams of the last group repeat the last live one and are never stored)
    uint32_t so[kDma], sj[kDma];
#pragma unroll
    for (int k = 0; k < kDma; ++k) {
        const uint32_t vec = vec_in_slot<kV>((uint32_t)k * 64u + (uint32_t)lane);
        so[k] = vec / kV;
        sj[k] = vec % kV;
    }
    if (wave == 5) {  // ---- the storer: the previous chunk's results, whole lines ----
        for (uint64_t c = 0; c <= nch; ++c) {
            barrier_lds();
            if (c == 0) continue;
            const lds_u8 *img = lds + kOutBase + (uint32_t)((c - 1) & 1) * kSlotBytes;
            float *ob = a.out + (uint64_t)g0 * a.stride_out + (c - 1) * kCS;
#pragma unroll
            for (int k = 0; k < kDma; ++k) {
                const v4f v = *(const __attribute__((address_space(3))) v4f *)(img + (k * 64 + lane) * 16);
                if (so[k] < live) *reinterpret_cast<v4f *>(ob + (uint64_t)so[k] * a.stride_out + sj[k] * 4u) = v;
            }
        }
        return;
    }
    // ---- the loaders: kAhead chunks ahead of the chain; wave 1 + h / 3 + h issues instructions h*kDma/2 .. of a chunk ----
    const int second = wave >= 3 ? 1 : 0, half = (wave - 1) & 1;
    if (second && !TWO) {
        for (uint64_t c = 0; c <= nch; ++c) barrier_lds();
        return;
    }
    constexpr int kMine = kDma / 2;
    const uint32_t lbase = (uint32_t)(uintptr_t)lds + (second ? kSlotBytes : 0u);
    uint32_t voff0[kDma], voff1[kDma];  // host: kGS * stride * 4 < 2^32
#pragma unroll
    for (int k = 0; k < kDma; ++k) {
        const uint32_t o = so[k] < live ? so[k] : live - 1;
        voff0[k] = (uint32_t)(((uint64_t)o * a.stride0 + sj[k] * 4u) * 4u);
        voff1[k] = (uint32_t)(((uint64_t)o * a.stride1 + sj[k] * 4u) * 4u);
    }
    auto issue = [&](uint64_t c) {
        const uint32_t slot = lbase + (uint32_t)(c % kRing) * (2u * kSlotBytes);
        if (second && c < a.head_chunks && a.in1_head) {  // rows of 8192 floats: the carried window
            const float *bh = a.in1_head + (uint64_t)g0 * kRmsWindow + c * kCS;
#pragma unroll
            for (int k = 0; k < kDma; ++k)
                if (k / kMine == half) glds16(bh, (((so[k] < live ? so[k] : live - 1) * kRmsWindow + sj[k] * 4u) * 4u), slot + k * 1024);
            return;
        }
        // (a fresh window -- zeros, the chain does not read the slot -- still fetches, the first input again: every chunk counts
        // the same in vmcnt)
        const bool first = !second || c < a.head_chunks;
        const float *b = first ? a.in0 + (uint64_t)g0 * a.stride0 + c * kCS : a.in1 + (uint64_t)g0 * a.stride1 + c * kCS;
#pragma unroll
        for (int k = 0; k < kDma; ++k)
            if (k / kMine == half) glds16(b, first ? voff0[k] : voff1[k], slot + k * 1024);
    };
    for (uint64_t c = 0; c < nch && c < (uint64_t)kAhead; ++c) issue(c);
    for (uint64_t c = 0; c < nch; ++c) {
        // chunk c has landed when only the chunks issued after it are outstanding (vmcnt retires in order)
        const uint64_t left = nch - 1 - c;
        if (left >= (uint64_t)(kAhead - 1)) wait_vm<(kAhead - 1) * kMine>();
        else if (left >= 3) wait_vm<3 * kMine>();  // the last chunks of a row: coarser steps (waiting for more than necessary is harmless)
        else if (left == 2) wait_vm<2 * kMine>();
        else if (left == 1) wait_vm<kMine>();
        else wait_vm<0>();
        barrier_lds();
        if (c + kAhead < nch) issue(c + kAhead);  // into the slot of chunk c - 2, which the chain left two barriers ago
    }
    barrier_lds();
}
template <class Op>
__global__ __launch_bounds__(384) void k_agc_chain(const ChainArgs a) {
    chain_body<Op>(a, blockIdx.x);
}
// Two chains side by side in one launch: even workgroups walk chain A, odd ones chain B (other arrays, other state words) -- the
// window sum of one time segment next to the gain of the segment before it.
template <class OpA, class OpB>
__global__ __launch_bounds__(384) void k_agc_chain2(const ChainArgs a, const ChainArgs b) {
    if (blockIdx.x & 1) chain_body<OpB>(b, blockIdx.x >> 1);
    else chain_body<OpA>(a, blockIdx.x >> 1);
}

// ---- the default parameters (release == 0) in ONE kernel -------------------------------------------------------------------
// With release == 0 the peak follower is no chain (peak = |x|), and what is left -- window sum, desired gain, gain, output -- is a
// four-stage pipeline over chunks of samples that fits one workgroup: every sample is fetched twice (as itself and, 8192 samples
// later, as what leaves the window) and written once -- 12 bytes against the 8 of the algorithm, where the segment-by-segment form
// above moves 60 (squares, sums, desired gains, gains: all through memory, a launch per stage and segment).
//     wave 0 (S)           window sum of chunk t:      x, x_old  -> sum                        (SumOp: the squares ride along)
//     waves 2 3 6 7 (D)    desired gain of chunk t-1:  sum, |x|  -> desired   (IEEE sqrt and two IEEE divides: ~48 instructions a sample)
//     wave 1 (G)           gain of chunk t-2:          desired   -> gain      (GainOp0, its two prepared operands made in place)
//     waves 10 11 (Y)      output of chunk t-3:        x * gain  -> memory, whole 128-byte lines
//     waves 4 5 / 8 9      LDS-DMA of x / x_old, kFAhead chunks ahead of S, half a chunk's instructions each
// One barrier per chunk.  The two chains keep a SIMD each to themselves and the loaders, which only issue and wait (wave w of a
// workgroup runs on SIMD w % 4); everything with arithmetic in it runs on the other two.  What sizes the workgroup is D: a chain
// wave walks 64 streams as fast as 16, but the desired gains of 64 streams x 32 samples are 1500 wave instructions = 6100 SIMD
// cycles on two SIMDs, against the 1100 cycles the chains take for them (measured: 46 ns per sample with 64 streams per
// workgroup, the chains alone 16).  So a workgroup takes 16 STREAMS and chunks of 128 samples (the same 8 KiB images; a quarter of
// the barriers per sample): 16 of a chain wave's lanes work, and D's 3100 cycles fit the 4400 of a chunk.  The chip has the CUs:
// 64 streams are 4 workgroups, 2048 are 128; larger batches queue as further workgroups (16 384 x 32 Ki: 3.7 ms against the segment form's 10.2).
// LDS: x stays until Y has used it (kFAhead + 4 chunks), x_old until S has (kFAhead + 1), the value image -- sum, then desired,
// then gain, in place -- 4 chunks.  Same operations in the same order as the stages above (and as k_agc_seq, up to GainOp0's tie:
// see the header), so the same bits.
constexpr int kFS = 16;                      // streams per workgroup
constexpr int kFCS = 128;                    // samples per chunk and stream
constexpr int kFV = kFCS / 4;                // 16-byte vectors per stream and chunk: 32
constexpr int kFSub = 8;                     // vectors a chain lane holds at a time
constexpr uint32_t kFHeadChunks = kRmsWindow / kFCS;
constexpr int kFAhead = 3;
constexpr int kFRX = kFAhead + 4, kFRO = kFAhead + 1, kFRA = 4;
static_assert(kFS * kFV * 16 == (int)kSlotBytes, "a chunk image is 8 KiB");
constexpr uint32_t kFOBase = kFRX * kSlotBytes, kFABase = kFOBase + kFRO * kSlotBytes, kFFin = kFABase + kFRA * kSlotBytes;
constexpr size_t kFusedLds = (size_t)kFFin + kFS * 4;  // 120 KiB + 64 B
constexpr int kFWaves = 12;
// Any other parameters (release != 0): the peak follower is a third chain (wave 8, on the window sum's SIMD: two chains leave each
// other's issue slots alone; wave 9 then loads all of x_old), its levels go to D through an image of their own (4 more chunks), and
// the gain takes both candidates (GainOp).  Everything else is the same pipeline, in the same twelve waves (a thirteenth would cut
// the registers of all of them to 128 a lane: spills).
constexpr uint32_t kFPBase = kFFin;                                      // the peak images (GEN only)
constexpr uint32_t kFFinG = kFPBase + kFRA * kSlotBytes;                 // final gain [16] | final peak [16]
constexpr size_t kFusedLdsG = (size_t)kFFinG + 2 * kFS * 4;              // 152 KiB + 128 B
constexpr int kFWavesG = 12;
// the image: slot o*32 + (j ^ o) holds vector j of stream o's chunk -- the 16 chain lanes, each on vector j of its own stream, then
// hit 16 different bank groups although the rows are not padded
__device__ __forceinline__ constexpr uint32_t fslot_of(uint32_t o, uint32_t j) { return o * kFV + (j ^ (o & (kFV - 1))); }
struct FusedArgs {
    const float *in;        // rows of n samples, `stride` floats apart
    const float *in1_head;  // the carried window, [streams][8192] squares in time order; null: a fresh window (zeros)
    float *out;
    uint64_t n, stride, stride_out;
    uint32_t n_streams;
    float *state;           // per stream `state_stride` floats {sum, -, peak, gain, ...}
    uint32_t state_stride;
    AgcK k;
};
// release == 0: the peak follower peak * c + |x| * (1 - c) with c = 0 either way (agc.rs:397-407) IS |x| -- for numbers.  A NaN sample makes the
// reference's peak level NaN for good (NaN * 0 + v), an infinite one from the sample after it (inf * 0); from then on `peak_level > 0.0` is false
// and the peak gain is absolute_max_gain (:421-427).  The window sum tells: it is NaN exactly when the peak level is (sum - old + NaN; inf - inf
// when the infinite square leaves the window: the peak has been NaN since the sample after it, and while the sum is still +inf the rms gain 0
// decides whatever the peak says).  So: the peak level of the default parameters, poisoned where the reference's is.
__device__ __forceinline__ float agc_peak0(float sum, float x) { return sum != sum ? sum : fabsf(x); }
__device__ __forceinline__ float agc_desired(float sum, float p, const AgcK &k) {  // agc.rs:416-430, :169 (k_agc_desired's `one`)
    const float rms = sqrtf(sum / (float)kRmsWindow);
    const float rms_gain = rms > 0.0f ? k.target_level / rms : k.absolute_max_gain;
    const float peak_gain = p > 0.0f ? fminf(k.target_level / p, k.absolute_max_gain) : k.absolute_max_gain;
    return fmaxf(fminf(rms_gain, peak_gain), k.floor);
}
// The same value with ONE division: a correctly rounded quotient is monotonic in its divisor, so of target / rms and target / p the
// smaller is the one by the larger divisor -- min(target / rms, min(target / p, max)) == min(target / max(rms, p), max) bit for bit -- and
// a level that is zero (or a NaN: the square root of a window sum that rounding drove below zero; `rms > 0` is false for it and the
// reference takes `max`) drops out of both forms: v_max_f32 returns the other operand.  k_agc_fused0's D waves: 11 instructions of 48 less.
__device__ __forceinline__ float agc_desired1(float sum, float p, const AgcK &k) {
    const float rms = sqrtf(sum / (float)kRmsWindow);
    const float m = fmaxf(rms, p);  // (p = agc_peak0(): NaN with the sum, and then so is m)
    const float g = m > 0.0f ? fminf(k.target_level / m, k.absolute_max_gain) : k.absolute_max_gain;
    return fmaxf(g, k.floor);
}
template <bool GEN>
__global__ __launch_bounds__(64 * (GEN ? kFWavesG : kFWaves)) void k_agc_fused(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_u8 *const lds = (lds_u8 *)smem;
    typedef __attribute__((address_space(3))) v4f lds_v4;
    typedef __attribute__((address_space(3))) float lds_f;
    constexpr uint32_t kFin = GEN ? kFFinG : kFFin;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x / 64), lane = (int)threadIdx.x & 63;
    const uint32_t g0 = blockIdx.x * (uint32_t)kFS;
    const uint32_t live = a.n_streams - g0 < (uint32_t)kFS ? a.n_streams - g0 : (uint32_t)kFS;
    const uint32_t nch = (uint32_t)(a.n / kFCS);  // whole chunks (host: n < 2^24); the rest is wave 0's epilogue
    const uint32_t nsteps = nch + 3;
    const bool chain_lane = lane < kFS;            // the lanes of S and G that walk a stream
    const bool mine = (uint32_t)lane < live;
    const uint32_t stream = g0 + (mine ? (uint32_t)lane : live - 1);
    if (wave == 0) {  // ---- S: the window sum, lane = stream (agc.rs:152-163) ----
        float *st = a.state + (uint64_t)stream * a.state_stride;
        SumOp op;
        op.sum = st[0];
        auto chunk = [&](uint32_t sx, uint32_t so_, uint32_t sa, auto head_tag, auto zero_tag) {
            constexpr bool HEAD = decltype(head_tag)::value, ZERO = decltype(zero_tag)::value;
            const lds_u8 *inx = lds + sx * kSlotBytes, *ino = lds + kFOBase + so_ * kSlotBytes;
            lds_u8 *img = lds + kFABase + sa * kSlotBytes;
            for (int b = 0; b < kFV / kFSub; ++b) {
                v4f x[kFSub], o[kFSub];
                uint32_t sl[kFSub];
#pragma unroll
                for (int j = 0; j < kFSub; ++j) {
                    sl[j] = fslot_of((uint32_t)lane & (kFS - 1), (uint32_t)(b * kFSub + j)) * 16u;
                    x[j] = *(const lds_v4 *)(inx + sl[j]);
                    o[j] = ZERO ? v4f{0.f, 0.f, 0.f, 0.f} : *(const lds_v4 *)(ino + sl[j]);
                }
#pragma unroll
                for (int j = 0; j < kFSub; ++j) *(lds_v4 *)(img + sl[j]) = op.template step4<HEAD>(x[j], o[j]);
            }
        };
        uint32_t sx = 0, so_ = 0, sa = 0;
        for (uint32_t t = 0; t < nsteps; ++t) {
            barrier_lds();
            if (t < nch && chain_lane) {
                if (t < kFHeadChunks) {
                    if (a.in1_head) chunk(sx, so_, sa, std::true_type{}, std::false_type{});
                    else chunk(sx, so_, sa, std::true_type{}, std::true_type{});
                } else {
                    chunk(sx, so_, sa, std::false_type{}, std::false_type{});
                }
            }
            sx = sx + 1 == (uint32_t)kFRX ? 0 : sx + 1;
            so_ = so_ + 1 == (uint32_t)kFRO ? 0 : so_ + 1;
            sa = (sa + 1) & (kFRA - 1);
        }
        barrier_lds();  // G's final gain is in the LDS
        if (mine) {  // the last n % 128 samples: every stage, one sample at a time, straight from memory
            const float *r0 = a.in + (uint64_t)stream * a.stride;
            float *ro = a.out + (uint64_t)stream * a.stride_out;
            const float oma = 1.0f - a.k.attack_coeff;
            if constexpr (GEN) {
                GainOp gop;
                gop.att = a.k.attack_coeff, gop.rel = a.k.release_coeff, gop.oma = 1.0f - gop.att, gop.omr = 1.0f - gop.rel, gop.maxg = a.k.absolute_max_gain;
                gop.gain = *(const lds_f *)(lds + kFin + lane * 4);
                PeakOp pop;
                pop.peak = *(const lds_f *)(lds + kFin + (kFS + lane) * 4);
                pop.rel = a.k.release_coeff;
                for (uint64_t i = (uint64_t)nch * kFCS; i < a.n; ++i) {
                    const bool head = i < (uint64_t)kRmsWindow;
                    const float ov = head ? (a.in1_head ? a.in1_head[(uint64_t)stream * kRmsWindow + i] : 0.0f) : r0[i - kRmsWindow];
                    const float xv = r0[i];
                    const float sm = op.one(xv, ov, head);
                    const float pk = pop.one(xv, 0.f, false);
                    ro[i] = gop.one(xv, agc_desired(sm, pk, a.k), false);
                }
                st[0] = op.sum;
                st[2] = pop.peak;
                st[3] = gop.gain;
            } else {
                float gain = *(const lds_f *)(lds + kFin + lane * 4);
                for (uint64_t i = (uint64_t)nch * kFCS; i < a.n; ++i) {
                    const bool head = i < (uint64_t)kRmsWindow;
                    const float ov = head ? (a.in1_head ? a.in1_head[(uint64_t)stream * kRmsWindow + i] : 0.0f) : r0[i - kRmsWindow];
                    const float xv = r0[i];
                    const float sm = op.one(xv, ov, head);
                    const float d = agc_desired(sm, agc_peak0(sm, xv), a.k);
                    const float dc = __builtin_amdgcn_fmed3f(d, 0.1f, a.k.absolute_max_gain), da = d * oma;
                    gain = __builtin_amdgcn_fmed3f(gain * a.k.attack_coeff + da, 0.1f, dc);
                    ro[i] = xv * gain;
                }
                st[0] = op.sum;
                st[3] = gain;
                if (a.n) st[2] = fabsf(r0[a.n - 1]);  // release == 0: the peak level is the last sample's magnitude
            }
        }
        return;
    }
    if (GEN && wave == 8) {  // ---- P: the peak follower, lane = stream (agc.rs:397-407) ----
        const float *st = a.state + (uint64_t)stream * a.state_stride;
        PeakOp pop;
        pop.peak = st[2];
        pop.rel = a.k.release_coeff;
        uint32_t sx = 0, sa = 0;
        for (uint32_t t = 0; t < nsteps; ++t) {
            barrier_lds();
            if (t < nch && chain_lane) {
                const lds_u8 *inx = lds + sx * kSlotBytes;
                lds_u8 *img = lds + kFPBase + sa * kSlotBytes;
                for (int b = 0; b < kFV / kFSub; ++b) {
                    v4f x[kFSub];
                    uint32_t sl[kFSub];
#pragma unroll
                    for (int j = 0; j < kFSub; ++j) {
                        sl[j] = fslot_of((uint32_t)lane & (kFS - 1), (uint32_t)(b * kFSub + j)) * 16u;
                        x[j] = *(const lds_v4 *)(inx + sl[j]);
                    }
#pragma unroll
                    for (int j = 0; j < kFSub; ++j) *(lds_v4 *)(img + sl[j]) = pop.template step4<false>(x[j], v4f{0.f, 0.f, 0.f, 0.f});
                }
            }
            sx = sx + 1 == (uint32_t)kFRX ? 0 : sx + 1;
            sa = (sa + 1) & (kFRA - 1);
        }
        if (chain_lane) *(lds_f *)(lds + kFin + (kFS + lane) * 4) = pop.peak;
        barrier_lds();
        return;
    }
    if (wave == 1) {  // ---- G: the gain, lane = stream (agc.rs:486-499 with release == 0: GainOp0) ----
        const float *st = a.state + (uint64_t)stream * a.state_stride;
        float gain = st[3];
        const float att = a.k.attack_coeff, oma = 1.0f - att, maxg = a.k.absolute_max_gain;
        uint32_t sa = (0u - 2u) & (kFRA - 1);  // the image of chunk t - 2
        for (uint32_t t = 0; t < nsteps; ++t) {
            barrier_lds();
            if (t >= 2 && t - 2 < nch && chain_lane) {
                lds_u8 *img = lds + kFABase + sa * kSlotBytes;
                for (int b = 0; b < kFV / kFSub; ++b) {
                    v4f d[kFSub];
                    uint32_t sl[kFSub];
#pragma unroll
                    for (int j = 0; j < kFSub; ++j) {
                        sl[j] = fslot_of((uint32_t)lane & (kFS - 1), (uint32_t)(b * kFSub + j)) * 16u;
                        d[j] = *(const lds_v4 *)(img + sl[j]);
                    }
#pragma unroll
                    for (int j = 0; j < kFSub; ++j) {
                        if constexpr (GEN) {  // both candidates, select, clamp (GainOp without its output multiply: Y's)
                            const float omr = 1.0f - a.k.release_coeff, rel = a.k.release_coeff;
                            const v4f da = d[j] * oma, dr = d[j] * omr;
                            v4f r;
                            auto one = [&](float dd, float daa, float drr) {
                                const float ca = gain * att + daa, cr = gain * rel + drr;
                                const float g = dd > gain ? ca : cr;
                                gain = __builtin_amdgcn_fmed3f(g, 0.1f, maxg);
                                return gain;
                            };
                            r.x = one(d[j].x, da.x, dr.x), r.y = one(d[j].y, da.y, dr.y), r.z = one(d[j].z, da.z, dr.z), r.w = one(d[j].w, da.w, dr.w);
                            *(lds_v4 *)(img + sl[j]) = r;
                            continue;
                        }
                        // what does not wait for the gain first: clamp(desired) and desired * (1 - attack), the two operands of the chain
#ifdef RH_AGC_DIAG_G  // diagnostics builds (wrong results): what do the two prepared operands cost the gain wave?
                        const v4f da = d[j];
                        v4f dc = d[j], r;
#else
                        const v4f da = d[j] * oma;
                        v4f dc, r;
                        dc.x = __builtin_amdgcn_fmed3f(d[j].x, 0.1f, maxg), dc.y = __builtin_amdgcn_fmed3f(d[j].y, 0.1f, maxg);
                        dc.z = __builtin_amdgcn_fmed3f(d[j].z, 0.1f, maxg), dc.w = __builtin_amdgcn_fmed3f(d[j].w, 0.1f, maxg);
#endif
                        gain = __builtin_amdgcn_fmed3f(gain * att + da.x, 0.1f, dc.x), r.x = gain;
                        gain = __builtin_amdgcn_fmed3f(gain * att + da.y, 0.1f, dc.y), r.y = gain;
                        gain = __builtin_amdgcn_fmed3f(gain * att + da.z, 0.1f, dc.z), r.z = gain;
                        gain = __builtin_amdgcn_fmed3f(gain * att + da.w, 0.1f, dc.w), r.w = gain;
                        *(lds_v4 *)(img + sl[j]) = r;
                    }
                }
            }
            sa = (sa + 1) & (kFRA - 1);
        }
        if (chain_lane) *(lds_f *)(lds + kFin + lane * 4) = gain;
        barrier_lds();
        return;
    }
    if (wave == 2 || wave == 3 || wave == 6 || wave == 7) {  // ---- D: the desired gain of chunk t - 1, 4 samples per lane and half chunk ----
        const uint32_t m = (uint32_t)(wave == 2 ? 0 : wave == 3 ? 1 : wave == 6 ? 2 : 3);
        const uint32_t q0 = (m * 64u + (uint32_t)lane) * 16u;  // this lane's slots: q0 and q0 + 4 KiB (the images share one layout)
        uint32_t sx = kFRX - 1, sa = kFRA - 1;                  // ring slot and image of chunk t - 1
        for (uint32_t t = 0; t < nsteps; ++t) {
            barrier_lds();
            if (t >= 1 && t - 1 < nch) {
                const lds_u8 *inx = lds + sx * kSlotBytes;
                lds_u8 *img = lds + kFABase + sa * kSlotBytes;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const v4f s4 = *(const lds_v4 *)(img + q0 + h * 4096);
                    v4f x4;  // the peak level: the follower's, or with release == 0 the sample's magnitude (fabsf below is then the identity or the level)
                    if constexpr (GEN) x4 = *(const lds_v4 *)(lds + kFPBase + sa * kSlotBytes + q0 + h * 4096);
                    else x4 = *(const lds_v4 *)(inx + q0 + h * 4096);
                    v4f r;
#ifdef RH_AGC_DIAG_D  // diagnostics builds (wrong results): what do the square roots and divides cost the pipeline?
                    r = s4 + x4;
#else
                    r.x = agc_desired(s4.x, agc_peak0(s4.x, x4.x), a.k), r.y = agc_desired(s4.y, agc_peak0(s4.y, x4.y), a.k);
                    r.z = agc_desired(s4.z, agc_peak0(s4.z, x4.z), a.k), r.w = agc_desired(s4.w, agc_peak0(s4.w, x4.w), a.k);
#endif
                    *(lds_v4 *)(img + q0 + h * 4096) = r;
                }
            }
            sx = sx + 1 == (uint32_t)kFRX ? 0 : sx + 1;
            sa = (sa + 1) & (kFRA - 1);
        }
        barrier_lds();
        return;
    }
    // per-lane geometry of the line-wise transfers: slot q = k*64 + lane of a chunk image holds vector j = (q % 32) ^ o of stream
    // o = q / 32, i.e. samples 4j..4j+3 of its chunk (surplus streams of the last group repeat the last live one and are never stored)
    uint32_t so[kDma], sj[kDma];
#pragma unroll
    for (int k = 0; k < kDma; ++k) {
        const uint32_t q = (uint32_t)k * 64u + (uint32_t)lane;
        so[k] = q / kFV;
        sj[k] = (q % kFV) ^ (so[k] & (kFV - 1));
    }
    constexpr int kMine = kDma / 2;
    if (wave >= 10) {  // ---- Y: chunk t - 3 leaves as x * gain (agc.rs:503), whole lines; half the lines each ----
        const int half = wave - 10;
        uint32_t sx = kFRX - 3, sa = kFRA - 3;
        for (uint32_t t = 0; t < nsteps; ++t) {
            barrier_lds();
            if (t >= 3) {  // (t - 3 < nch always: nsteps = nch + 3)
                const lds_u8 *inx = lds + sx * kSlotBytes, *img = lds + kFABase + sa * kSlotBytes;
                float *ob = a.out + (uint64_t)g0 * a.stride_out + (uint64_t)(t - 3) * kFCS;
#pragma unroll
                for (int k = 0; k < kDma; ++k) {
                    if (k / kMine != half) continue;
                    const v4f g4 = *(const lds_v4 *)(img + (k * 64 + lane) * 16), x4 = *(const lds_v4 *)(inx + (k * 64 + lane) * 16);
                    if (so[k] < live) __builtin_nontemporal_store(x4 * g4, reinterpret_cast<v4f *>(ob + (uint64_t)so[k] * a.stride_out + sj[k] * 4u));
                }
            }
            sx = sx + 1 == (uint32_t)kFRX ? 0 : sx + 1;
            sa = (sa + 1) & (kFRA - 1);
        }
        barrier_lds();
        return;
    }
    // ---- the loaders (waves 4 5: x; 8 9: what leaves the window), kFAhead chunks ahead of S ----
    const int second = wave >= 8 ? 1 : 0, half = wave & 1;
    const bool all = GEN && wave == 9;  // (its partner walks the peak follower: this wave issues the whole chunk)
    const uint32_t ring = second ? (uint32_t)kFRO : (uint32_t)kFRX;
    const uint32_t lbase = (uint32_t)(uintptr_t)lds + (second ? kFOBase : 0u);
    uint32_t voff[kDma];  // host: kFS * stride * 4 < 2^32
#pragma unroll
    for (int k = 0; k < kDma; ++k) {
        const uint32_t o = so[k] < live ? so[k] : live - 1;
        voff[k] = (uint32_t)(((uint64_t)o * a.stride + sj[k] * 4u) * 4u);
    }
    uint32_t slot_next = 0;  // ring slot of the next chunk to issue
    auto issue = [&](uint32_t c) {
        const uint32_t slot = lbase + slot_next * kSlotBytes;
        slot_next = slot_next + 1 == ring ? 0 : slot_next + 1;
        if (second && c < kFHeadChunks && a.in1_head) {  // rows of 8192 floats: the carried window
            const float *bh = a.in1_head + (uint64_t)g0 * kRmsWindow + (uint64_t)c * kFCS;
#pragma unroll
            for (int k = 0; k < kDma; ++k)
                if (all || k / kMine == half) glds16(bh, (((so[k] < live ? so[k] : live - 1) * kRmsWindow + sj[k] * 4u) * 4u), slot + k * 1024);
            return;
        }
        // (a fresh window -- zeros, S does not read the slot -- still fetches, x again: every chunk counts the same in vmcnt)
        const bool self = !second || c < kFHeadChunks;
        const float *b = a.in + (uint64_t)g0 * a.stride + (uint64_t)c * kFCS - (self ? 0 : kRmsWindow);
#pragma unroll
        for (int k = 0; k < kDma; ++k)
            if (all || k / kMine == half) glds16(b, voff[k], slot + k * 1024);
    };
    for (uint32_t c = 0; c < nch && c < (uint32_t)kFAhead; ++c) issue(c);
    for (uint32_t t = 0; t < nsteps; ++t) {
        if (t < nch) {  // chunk t has landed when only the chunks issued after it are outstanding (vmcnt retires in order)
            const uint32_t left = nch - 1 - t;
            if (left >= (uint32_t)(kFAhead - 1)) {
                if (all) wait_vm<(kFAhead - 1) * kDma>();
                else wait_vm<(kFAhead - 1) * kMine>();
            } else if (left == 1) {
                if (all) wait_vm<kDma>();
                else wait_vm<kMine>();
            } else {
                wait_vm<0>();
            }
        }
        barrier_lds();
        // x: into the slot of chunk t - 4, which Y left before this barrier; x_old: into the slot of chunk t - 1, which S did
        if (t + kFAhead < nch) issue(t + kFAhead);
    }
    barrier_lds();
}

// ---- round 5: the default parameters again, the chain waves carrying nothing but their chains -------------------------------
// Every wave of this kernel issues ONE instruction every 6-7 cycles, whatever the instruction and whether or not it depends on
// the one before (RH_AGC_PROFILE builds print the time each role spends between two barriers: profiles/r05_agc_roles.txt), so a
// chain wave's time is its instruction COUNT.  k_agc_fused<false> above spends 5.5 instructions a sample on each: the window sum
// squares both its inputs (2 multiplies beside the 2 dependent adds), the gain makes its two operands (a multiply and a median
// beside the 3 dependent operations), both add a ring base to every address and wait before every first use of a vector.  Here the
// waves that only move data or work in parallel take that over, one pipeline stage earlier:
//     role 4 5 (X, Q)      LDS-DMA of x, 3 chunks ahead; chunk t: x*x of the vectors the wave fetched itself -> the value image
//     role 0 (S)           window sum of chunk t-1:    x*x (image), x_old -> sum        (1 multiply, 2 dependent adds)
//     role 2 3 6 7 (D)     operands of chunk t-2:      sum, |x| -> desired * (1 - attack) (image), clamp(desired) (a ring of its own);
//                          `desired` with ONE division (agc_desired1)
//     role 1 (G)           gain of chunk t-3:          the two operands -> gain          (3 dependent operations, nothing else)
//     role 10 11 (Y)       output of chunk t-4:        x * gain -> memory
//     role 8 9             LDS-DMA of x_old, as before
// and the chain waves take their vectors in batches of four behind ONE wait each.  Per chunk of 128 samples: S 530 instructions
// (384 arithmetic, 96 LDS, 32 address adds, 18 waits and scalar), G 528 -- 4.1 a sample against 5.5 -- in 3300 cycles.
// Same operations on the same operands in the same order as k_agc_fused<false> -- who computes a square or a clamp does not change
// its bits (tests: equal to RH_AGC_FUSED_R4=1, to the segment form and to the reference-order kernel).
// LDS: x 8 chunks (3 in flight, Q, S's step, D, G's step, Y), x_old 4, the value image 5, clamp(desired) 2: 152 KiB.
constexpr int kQRX = 8, kQRO = 4, kQRA = 5, kQRC = 2;
constexpr uint32_t kQOBase = kQRX * kSlotBytes, kQABase = kQOBase + kQRO * kSlotBytes, kQCBase = kQABase + kQRA * kSlotBytes, kQFin = kQCBase + kQRC * kSlotBytes;
constexpr size_t kFused0Lds = (size_t)kQFin + kFS * 4;  // 152 KiB + 64 B
__device__ __forceinline__ void ring_next(uint32_t &s, uint32_t n) { s = s + 1 == n ? 0 : s + 1; }
#ifdef RH_AGC_PROFILE  // diagnostics builds: per role, the time between leaving a barrier and reaching the next (s_memtime), printed by workgroup 0
#define RH_AP_DECL unsigned long long ap_busy = 0, ap_t0 = 0; const unsigned long long ap_start = __builtin_readcyclecounter(), ap_rt0 = __builtin_amdgcn_s_memrealtime();
#define RH_AP_BEGIN ap_t0 = __builtin_readcyclecounter();
#define RH_AP_END ap_busy += __builtin_readcyclecounter() - ap_t0;
#define RH_AP_REPORT(name)                                                                                                                      \
    if (blockIdx.x == 0 && lane == 0)                                                                                                           \
        printf("agc hw wave %2d role %2d %s: busy %llu of %llu memtime ticks, %llu realtime ticks, %u steps\n", wave_hw, wave, name, ap_busy,   \
               __builtin_readcyclecounter() - ap_start, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - ap_rt0), nsteps);
#else
#define RH_AP_DECL
#define RH_AP_BEGIN
#define RH_AP_END
#define RH_AP_REPORT(name)
#endif
#ifndef RH_AGC_ND
#define RH_AGC_ND 4  // D waves: 4 (8 samples a lane and chunk) or 8 (4 samples)
#endif
constexpr int kQND = RH_AGC_ND, kQWaves = 8 + kQND;
__global__ __launch_bounds__(64 * kQWaves) void k_agc_fused0(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_u8 *const lds = (lds_u8 *)smem;
    typedef __attribute__((address_space(3))) v4f lds_v4;
    typedef __attribute__((address_space(3))) float lds_f;
    // Which role runs where (hardware wave w of a workgroup sits on SIMD w % 4) moves the time by 20 %: a chain wave issues FASTER the
    // busier the waves beside it are (6.1 cycles an instruction with two D waves beside it, 6.3 with one, 6.9 with two loaders) -- but
    // a D wave beside a chain takes 3200 cycles for its chunk instead of 2200, two of them 4000, more than the chain.  Measured
    // (profiles/r05_agc_placement_*.txt, 64 x 1 Mi frames): S G | D D | X X | O O | Y Y in hardware order 27.8 -> 25.0 ms with the
    // batched waits; both D pairs on the chains' SIMDs 27.4; one D beside each chain and Y behind it (below) 23.0; sixteen waves with
    // eight D waves (RH_AGC_ND=8) 26.3-27.1; s_setprio for the chains: nothing.  RH_AGC_ROLE_MAP (builds): the role of each hardware wave.
#ifndef RH_AGC_ROLE_MAP
#if RH_AGC_ND == 8
#define RH_AGC_ROLE_MAP 0, 1, 12, 14, 2, 6, 13, 15, 3, 7, 4, 5, 10, 11, 8, 9
#else
#define RH_AGC_ROLE_MAP 0, 1, 2, 3, 6, 7, 4, 5, 10, 11, 8, 9
#endif
#endif
    const int wave_hw = __builtin_amdgcn_readfirstlane((int)threadIdx.x / 64), lane = (int)threadIdx.x & 63;
    constexpr int kRoleMap[kQWaves] = {RH_AGC_ROLE_MAP};
    int wave = 0;
#pragma unroll
    for (int w = 0; w < kQWaves; ++w) wave = wave_hw == w ? kRoleMap[w] : wave;
    const uint32_t g0 = blockIdx.x * (uint32_t)kFS;
    const uint32_t live = a.n_streams - g0 < (uint32_t)kFS ? a.n_streams - g0 : (uint32_t)kFS;
    const uint32_t nch = (uint32_t)(a.n / kFCS);  // whole chunks (host: n < 2^24); the rest is wave 0's epilogue
    const uint32_t nsteps = nch + 4;              // step t: Q on chunk t, S on t - 1, D on t - 2, G on t - 3, Y on t - 4
    const bool chain_lane = lane < kFS;
    const bool mine = (uint32_t)lane < live;
    const uint32_t stream = g0 + (mine ? (uint32_t)lane : live - 1);
    RH_AP_DECL
    if (wave == 0 || wave == 1) {
        // a chain lane's 32 vectors of a chunk image: vector 16 h + l of stream o lies at pre[l] + 256 h (fslot_of: (16 h + l) ^ o = 16 h + (l ^ o), o < 16)
        const uint32_t o = (uint32_t)lane & (kFS - 1);
        uint32_t pre[16];
#pragma unroll
        for (int l = 0; l < 16; ++l) pre[l] = fslot_of(o, (uint32_t)l) * 16u;
        if (wave == 0) {  // ---- S: the window sum of chunk t - 1 (agc.rs:152-163), in place over the squares Q left in the image ----
            float *st = a.state + (uint64_t)stream * a.state_stride;
            SumOp op;
            op.sum = st[0];
            auto chunk = [&](uint32_t so_, uint32_t sa, auto head_tag, auto zero_tag) {
                constexpr bool HEAD = decltype(head_tag)::value, ZERO = decltype(zero_tag)::value;
                const lds_u8 *ino = lds + kQOBase + so_ * kSlotBytes;
                lds_u8 *img = lds + kQABase + sa * kSlotBytes;
                // (batches of 4 vectors, one wait per batch, the next batch's reads issued in the middle of this one's chain: see G)
                constexpr int kB = 4, kNB = kFV / kB;
                v4f nw[2][kB], od[2][kB], r[kB];
                auto slot_of = [&](int v) { return pre[v & 15] + (uint32_t)(v >> 4) * 256u; };
                auto fetch = [&](int b) {
#pragma unroll
                    for (int j = 0; j < kB; ++j) {
                        nw[b & 1][j] = *(const lds_v4 *)(img + slot_of(b * kB + j));
                        if (!ZERO) od[b & 1][j] = *(const lds_v4 *)(ino + slot_of(b * kB + j));
                    }
                };
                auto run = [&](int b, int j) {
                    v4f q = ZERO ? v4f{0.f, 0.f, 0.f, 0.f} : od[b & 1][j];
                    const v4f &n4 = nw[b & 1][j];
#ifndef RH_AGC_DIAG_NOSQ  // (diagnostics builds, wrong results: what does the remaining multiply cost the chain wave?)
                    if (!HEAD) q.x = q.x * q.x, q.y = q.y * q.y, q.z = q.z * q.z, q.w = q.w * q.w;  // (HEAD: the carried window holds squares)
#endif
                    op.sum = op.sum - q.x + n4.x, r[j].x = op.sum;
                    op.sum = op.sum - q.y + n4.y, r[j].y = op.sum;
                    op.sum = op.sum - q.z + n4.z, r[j].z = op.sum;
                    op.sum = op.sum - q.w + n4.w, r[j].w = op.sum;
                };
                fetch(0);
#pragma unroll
                for (int b = 0; b < kNB; ++b) {
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
                    __builtin_amdgcn_sched_barrier(0);
                    if (b > 0) {
                        *(lds_v4 *)(img + slot_of((b - 1) * kB + 2)) = r[2];
                        *(lds_v4 *)(img + slot_of((b - 1) * kB + 3)) = r[3];
                    }
                    run(b, 0), run(b, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (b + 1 < kNB) fetch(b + 1);
                    *(lds_v4 *)(img + slot_of(b * kB + 0)) = r[0];
                    *(lds_v4 *)(img + slot_of(b * kB + 1)) = r[1];
                    __builtin_amdgcn_sched_barrier(0);
                    run(b, 2), run(b, 3);
                }
                *(lds_v4 *)(img + slot_of(kFV - 2)) = r[2];
                *(lds_v4 *)(img + slot_of(kFV - 1)) = r[3];
            };
            uint32_t so_ = kQRO - 1, sa = kQRA - 1;  // the slots of chunk t - 1
            for (uint32_t t = 0; t < nsteps; ++t) {
                barrier_lds();
                RH_AP_BEGIN
                if (t >= 1 && t - 1 < nch && chain_lane) {
                    if (t - 1 < kFHeadChunks) {
                        if (a.in1_head) chunk(so_, sa, std::true_type{}, std::false_type{});
                        else chunk(so_, sa, std::true_type{}, std::true_type{});
                    } else {
                        chunk(so_, sa, std::false_type{}, std::false_type{});
                    }
                }
                ring_next(so_, kQRO);
                ring_next(sa, kQRA);
                RH_AP_END
            }
            RH_AP_REPORT("S (window sum)")
            barrier_lds();  // G's final gain is in the LDS
            if (mine) {  // the last n % 128 samples: every stage, one sample at a time, straight from memory
                const float *r0 = a.in + (uint64_t)stream * a.stride;
                float *ro = a.out + (uint64_t)stream * a.stride_out;
                const float oma = 1.0f - a.k.attack_coeff;
                float gain = *(const lds_f *)(lds + kQFin + lane * 4);
                for (uint64_t i = (uint64_t)nch * kFCS; i < a.n; ++i) {
                    const bool head = i < (uint64_t)kRmsWindow;
                    const float ov = head ? (a.in1_head ? a.in1_head[(uint64_t)stream * kRmsWindow + i] : 0.0f) : r0[i - kRmsWindow];
                    const float xv = r0[i];
                    const float sm = op.one(xv, ov, head);
                    const float d = agc_desired(sm, agc_peak0(sm, xv), a.k);
                    const float dc = __builtin_amdgcn_fmed3f(d, 0.1f, a.k.absolute_max_gain), da = d * oma;
                    gain = __builtin_amdgcn_fmed3f(gain * a.k.attack_coeff + da, 0.1f, dc);
                    ro[i] = xv * gain;
                }
                st[0] = op.sum;
                st[3] = gain;
                if (a.n) st[2] = fabsf(r0[a.n - 1]);  // release == 0: the peak level is the last sample's magnitude
            }
            return;
        }
        // ---- G: the gain of chunk t - 3 (agc.rs:486-499 with release == 0: GainOp0), in place over desired * (1 - attack) ----
        const float *st = a.state + (uint64_t)stream * a.state_stride;
        float gain = st[3];
        const float att = a.k.attack_coeff;
        uint32_t sa = kQRA - 3, sc = (0u - 3u) & (kQRC - 1);  // the slots of chunk t - 3
        for (uint32_t t = 0; t < nsteps; ++t) {
            barrier_lds();
            RH_AP_BEGIN
            if (t >= 3 && t - 3 < nch && chain_lane) {
                lds_u8 *img = lds + kQABase + sa * kSlotBytes;
                const lds_u8 *clp = lds + kQCBase + sc * kSlotBytes;
                // Batches of 4 vectors, the next batch's reads issued in the middle of this one's chain and ONE wait per batch (the compiler's
                // own schedule waits before every first use, 63 times a chunk -- and every instruction of a lone wave costs its 6 cycles); a
                // batch's last two results are written behind the next batch's wait, so that no wait ever stands behind a fresh LDS write.
                constexpr int kB = 4, kNB = kFV / kB;
                v4f da[2][kB], dc[2][kB], r[kB];
                auto slot_of = [&](int v) { return pre[v & 15] + (uint32_t)(v >> 4) * 256u; };
                auto fetch = [&](int b) {
#pragma unroll
                    for (int j = 0; j < kB; ++j) {
                        da[b & 1][j] = *(const lds_v4 *)(img + slot_of(b * kB + j));
                        dc[b & 1][j] = *(const lds_v4 *)(clp + slot_of(b * kB + j));
                    }
                };
                auto run = [&](int b, int j) {
                    const v4f &A = da[b & 1][j], &C = dc[b & 1][j];
#ifdef RH_AGC_DIAG_G  // diagnostics builds (wrong results): the chain with two dependent operations a sample instead of three
                    gain = gain * att + A.x, r[j].x = gain;
                    gain = gain * att + A.y, r[j].y = gain;
                    gain = gain * att + A.z, r[j].z = gain;
                    gain = gain * att + A.w + C.w, r[j].w = gain;
#else
                    gain = __builtin_amdgcn_fmed3f(gain * att + A.x, 0.1f, C.x), r[j].x = gain;
                    gain = __builtin_amdgcn_fmed3f(gain * att + A.y, 0.1f, C.y), r[j].y = gain;
                    gain = __builtin_amdgcn_fmed3f(gain * att + A.z, 0.1f, C.z), r[j].z = gain;
                    gain = __builtin_amdgcn_fmed3f(gain * att + A.w, 0.1f, C.w), r[j].w = gain;
#endif
                };
                fetch(0);
#pragma unroll
                for (int b = 0; b < kNB; ++b) {
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
                    __builtin_amdgcn_sched_barrier(0);
                    if (b > 0) {
                        *(lds_v4 *)(img + slot_of((b - 1) * kB + 2)) = r[2];
                        *(lds_v4 *)(img + slot_of((b - 1) * kB + 3)) = r[3];
                    }
                    run(b, 0), run(b, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (b + 1 < kNB) fetch(b + 1);
                    *(lds_v4 *)(img + slot_of(b * kB + 0)) = r[0];
                    *(lds_v4 *)(img + slot_of(b * kB + 1)) = r[1];
                    __builtin_amdgcn_sched_barrier(0);
                    run(b, 2), run(b, 3);
                }
                *(lds_v4 *)(img + slot_of(kFV - 2)) = r[2];
                *(lds_v4 *)(img + slot_of(kFV - 1)) = r[3];
            }
            ring_next(sa, kQRA);
            sc = (sc + 1) & (kQRC - 1);
            RH_AP_END
        }
        RH_AP_REPORT("G (gain)")
        if (chain_lane) *(lds_f *)(lds + kQFin + lane * 4) = gain;
        barrier_lds();
        return;
    }
    if (wave == 2 || wave == 3 || wave == 6 || wave == 7 || wave >= 12) {  // ---- D: the gain's two operands for chunk t - 2, 4 samples per lane and part of a chunk ----
        const uint32_t m = (uint32_t)(wave >= 12 ? wave - 8 : wave == 2 ? 0 : wave == 3 ? 1 : wave == 6 ? 2 : 3);
        constexpr int kParts = 8 / kQND;                        // 16-byte vectors per lane and chunk
        constexpr uint32_t kPart = kSlotBytes / kParts;
        const uint32_t q0 = (m * 64u + (uint32_t)lane) * 16u;  // this lane's slots: q0 (and q0 + 4 KiB: the images share one layout)
        const float oma = 1.0f - a.k.attack_coeff, maxg = a.k.absolute_max_gain;
        uint32_t sx = kQRX - 2, sa = kQRA - 2, sc = 0;  // the slots of chunk t - 2
        for (uint32_t t = 0; t < nsteps; ++t) {
            barrier_lds();
            RH_AP_BEGIN
            if (t >= 2 && t - 2 < nch) {
                const lds_u8 *inx = lds + sx * kSlotBytes;
                lds_u8 *img = lds + kQABase + sa * kSlotBytes, *clp = lds + kQCBase + sc * kSlotBytes;
#pragma unroll
                for (int h = 0; h < kParts; ++h) {
                    const v4f s4 = *(const lds_v4 *)(img + q0 + h * kPart), x4 = *(const lds_v4 *)(inx + q0 + h * kPart);
                    v4f d, dc;
#ifdef RH_AGC_DIAG_D  // diagnostics builds (wrong results): what do the square roots and divides cost the pipeline?
                    d = s4 + x4;
#elif defined(RH_AGC_TWO_DIVISIONS)
                    d.x = agc_desired(s4.x, agc_peak0(s4.x, x4.x), a.k), d.y = agc_desired(s4.y, agc_peak0(s4.y, x4.y), a.k);
                    d.z = agc_desired(s4.z, agc_peak0(s4.z, x4.z), a.k), d.w = agc_desired(s4.w, agc_peak0(s4.w, x4.w), a.k);
#else
                    d.x = agc_desired1(s4.x, agc_peak0(s4.x, x4.x), a.k), d.y = agc_desired1(s4.y, agc_peak0(s4.y, x4.y), a.k);
                    d.z = agc_desired1(s4.z, agc_peak0(s4.z, x4.z), a.k), d.w = agc_desired1(s4.w, agc_peak0(s4.w, x4.w), a.k);
#endif
                    // what does not wait for the gain: clamp(desired) and desired * (1 - attack), the two operands of the chain
                    dc.x = __builtin_amdgcn_fmed3f(d.x, 0.1f, maxg), dc.y = __builtin_amdgcn_fmed3f(d.y, 0.1f, maxg);
                    dc.z = __builtin_amdgcn_fmed3f(d.z, 0.1f, maxg), dc.w = __builtin_amdgcn_fmed3f(d.w, 0.1f, maxg);
                    *(lds_v4 *)(img + q0 + h * kPart) = d * oma;
                    *(lds_v4 *)(clp + q0 + h * kPart) = dc;
                }
            }
            ring_next(sx, kQRX);
            ring_next(sa, kQRA);
            sc = (sc + 1) & (kQRC - 1);
            RH_AP_END
        }
        RH_AP_REPORT("D (desired gain)")
        barrier_lds();
        return;
    }
    // per-lane geometry of the line-wise transfers (as in k_agc_fused)
    uint32_t so[kDma], sj[kDma];
#pragma unroll
    for (int k = 0; k < kDma; ++k) {
        const uint32_t q = (uint32_t)k * 64u + (uint32_t)lane;
        so[k] = q / kFV;
        sj[k] = (q % kFV) ^ (so[k] & (kFV - 1));
    }
    constexpr int kMine = kDma / 2;
    if (wave >= 10) {  // ---- Y: chunk t - 4 leaves as x * gain (agc.rs:503), whole lines; half the lines each ----
        const int half = wave - 10;
        uint32_t sx = kQRX - 4, sa = kQRA - 4;
        for (uint32_t t = 0; t < nsteps; ++t) {
            barrier_lds();
            RH_AP_BEGIN
            if (t >= 4) {  // (t - 4 < nch always: nsteps = nch + 4)
                const lds_u8 *inx = lds + sx * kSlotBytes, *img = lds + kQABase + sa * kSlotBytes;
                float *ob = a.out + (uint64_t)g0 * a.stride_out + (uint64_t)(t - 4) * kFCS;
#pragma unroll
                for (int k = 0; k < kDma; ++k) {
                    if (k / kMine != half) continue;
                    const v4f g4 = *(const lds_v4 *)(img + (k * 64 + lane) * 16), x4 = *(const lds_v4 *)(inx + (k * 64 + lane) * 16);
                    if (so[k] < live) __builtin_nontemporal_store(x4 * g4, reinterpret_cast<v4f *>(ob + (uint64_t)so[k] * a.stride_out + sj[k] * 4u));
                }
            }
            ring_next(sx, kQRX);
            ring_next(sa, kQRA);
            RH_AP_END
        }
        RH_AP_REPORT("Y (output)")
        barrier_lds();
        return;
    }
    const int second = wave >= 8 ? 1 : 0, half = wave & 1;
    uint32_t voff[kDma];  // host: kFS * stride * 4 < 2^32
#pragma unroll
    for (int k = 0; k < kDma; ++k) {
        const uint32_t o = so[k] < live ? so[k] : live - 1;
        voff[k] = (uint32_t)(((uint64_t)o * a.stride + sj[k] * 4u) * 4u);
    }
    if (!second) {  // ---- X, Q (waves 4 5): x, three chunks ahead of its squares; chunk t's squares go to the value image ----
        const uint32_t lbase = (uint32_t)(uintptr_t)lds;
        uint32_t slot_next = 0;
        auto issue = [&](uint32_t c) {
            const uint32_t slot = lbase + slot_next * kSlotBytes;
            ring_next(slot_next, kQRX);
            const float *b = a.in + (uint64_t)g0 * a.stride + (uint64_t)c * kFCS;
#pragma unroll
            for (int k = 0; k < kDma; ++k)
                if (k / kMine == half) glds16(b, voff[k], slot + k * 1024);
        };
        for (uint32_t c = 0; c < nch && c < 3u; ++c) issue(c);
        uint32_t sx = 0, sa = 0;
        for (uint32_t t = 0; t < nsteps; ++t) {
            barrier_lds();
            RH_AP_BEGIN
            // into the slot of chunk t - 5, which Y left before this barrier
            if (t + 3 < nch) issue(t + 3);
            if (t < nch) {  // chunk t has landed when only the chunks issued after it are outstanding (vmcnt retires in order)
                const uint32_t left = nch - 1 - t;
                if (left >= 3) wait_vm<3 * kMine>();
                else if (left == 2) wait_vm<2 * kMine>();
                else if (left == 1) wait_vm<kMine>();
                else wait_vm<0>();
                const lds_u8 *inx = lds + sx * kSlotBytes;
                lds_u8 *img = lds + kQABase + sa * kSlotBytes;
#pragma unroll
                for (int k = 0; k < kDma; ++k) {
                    if (k / kMine != half) continue;  // (the vectors this wave fetched itself: its own vmcnt covers them)
                    const v4f x4 = *(const lds_v4 *)(inx + (k * 64 + lane) * 16);
                    v4f q;
                    q.x = x4.x * x4.x, q.y = x4.y * x4.y, q.z = x4.z * x4.z, q.w = x4.w * x4.w;  // |x| * |x| (agc.rs:414) == x * x
                    *(lds_v4 *)(img + (k * 64 + lane) * 16) = q;
                }
            }
            ring_next(sx, kQRX);
            ring_next(sa, kQRA);
            RH_AP_END
        }
        RH_AP_REPORT("X, Q (x and its squares)")
        barrier_lds();
        return;
    }
    // ---- waves 8 9: what leaves the window (x 8192 samples back, or the carried window), landed before S's step ----
    const uint32_t lbase = (uint32_t)(uintptr_t)lds + kQOBase;
    uint32_t slot_next = 0;
    auto issue = [&](uint32_t c) {
        const uint32_t slot = lbase + slot_next * kSlotBytes;
        ring_next(slot_next, kQRO);
        if (c < kFHeadChunks && a.in1_head) {  // rows of 8192 floats: the carried window
            const float *bh = a.in1_head + (uint64_t)g0 * kRmsWindow + (uint64_t)c * kFCS;
#pragma unroll
            for (int k = 0; k < kDma; ++k)
                if (k / kMine == half) glds16(bh, (((so[k] < live ? so[k] : live - 1) * kRmsWindow + sj[k] * 4u) * 4u), slot + k * 1024);
            return;
        }
        // (a fresh window -- zeros, S does not read the slot -- still fetches, x again: every chunk counts the same in vmcnt)
        const float *b = a.in + (uint64_t)g0 * a.stride + (uint64_t)c * kFCS - (c < kFHeadChunks ? 0 : kRmsWindow);
#pragma unroll
        for (int k = 0; k < kDma; ++k)
            if (k / kMine == half) glds16(b, voff[k], slot + k * 1024);
    };
    for (uint32_t c = 0; c < nch && c < 2u; ++c) issue(c);
    for (uint32_t t = 0; t < nsteps; ++t) {
        RH_AP_BEGIN
        if (t >= 1 && t - 1 < nch) {  // S takes chunk t - 1 behind this barrier
            const uint32_t left = nch - t;  // chunks behind it
            if (left >= 2) wait_vm<2 * kMine>();
            else if (left == 1) wait_vm<kMine>();
            else wait_vm<0>();
        }
        RH_AP_END
        barrier_lds();
        if (t + 2 < nch) issue(t + 2);  // into the slot of chunk t - 2, which S left before this barrier
    }
    RH_AP_REPORT("x_old (its wait only)")
    barrier_lds();
}

// ---- everything that is not a chain: one lane per 4 samples, the whole chip.  A launch covers samples [off, off + len) of every row
// (rows are n floats apart).
struct SegArgs {
    uint64_t n, off, len;
    uint32_t n_streams;
};
// desired gain from the window sum (in `d`, replaced in place) and the peak level (|x| when release == 0: peak * 0 + s * 1).
// da != null (release == 0): d receives clamp(desired, 0.1, max) and da desired * (1 - attack), what GainOp0 reads
__global__ __launch_bounds__(256) void k_agc_desired(float *__restrict__ d, const float *__restrict__ x, const float *__restrict__ peak, float *__restrict__ da, SegArgs g, AgcK k,
                                                       float *__restrict__ peak_state) {
    const uint64_t vpr = (g.len + 3) / 4, nvec = vpr * g.n_streams;  // vectors per row (the last one may be cut)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto one = [&](float sum, float p) {
        const float rms = sqrtf(sum / (float)kRmsWindow);  // agc.rs:416, :169
        const float rms_gain = rms > 0.0f ? k.target_level / rms : k.absolute_max_gain;
        const float peak_gain = p > 0.0f ? fminf(k.target_level / p, k.absolute_max_gain) : k.absolute_max_gain;  // agc.rs:424-430
        return fmaxf(fminf(rms_gain, peak_gain), k.floor);
    };
    const float oma = 1.0f - k.attack_coeff;
    auto clampg = [&](float v) { return v < 0.1f ? 0.1f : (v > k.absolute_max_gain ? k.absolute_max_gain : v); };
    for (uint64_t v = i0; v < nvec; v += stride) {
        const uint64_t row = v / vpr, col = (v % vpr) * 4, at = row * g.n + g.off + col;
        if (col + 4 <= g.len) {
            const v4f s4 = *reinterpret_cast<const v4f *>(d + at);
            const v4f x4 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(x + at));
            v4f p4;
            if (peak) p4 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(peak + at));
            else p4 = v4f{agc_peak0(s4.x, x4.x), agc_peak0(s4.y, x4.y), agc_peak0(s4.z, x4.z), agc_peak0(s4.w, x4.w)};
            v4f r;
            r.x = one(s4.x, p4.x), r.y = one(s4.y, p4.y), r.z = one(s4.z, p4.z), r.w = one(s4.w, p4.w);
            if (da) {
                *reinterpret_cast<v4f *>(da + at) = r * oma;
                r.x = clampg(r.x), r.y = clampg(r.y), r.z = clampg(r.z), r.w = clampg(r.w);
            }
            *reinterpret_cast<v4f *>(d + at) = r;
        } else {
            for (uint64_t i = col; i < g.len; ++i) {
                const uint64_t q = row * g.n + g.off + i;
                const float r = one(d[q], peak ? peak[q] : agc_peak0(d[q], x[q]));
                if (da) da[q] = r * oma;
                d[q] = da ? clampg(r) : r;
            }
        }
    }
    if (peak_state && g.len)  // release == 0: the peak level the next block starts from is the last sample's magnitude
        for (uint64_t r = i0; r < g.n_streams; r += stride) peak_state[r * kAgcStateFloats + 2] = fabsf(x[r * g.n + g.off + g.len - 1]);
}
// y = x * gain (agc.rs:503), in place of the gain
__global__ __launch_bounds__(256) void k_agc_apply(float *__restrict__ gn, const float *__restrict__ x, SegArgs g) {
    const uint64_t vpr = (g.len + 3) / 4, nvec = vpr * g.n_streams;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t v = i0; v < nvec; v += stride) {
        const uint64_t row = v / vpr, col = (v % vpr) * 4, at = row * g.n + g.off + col;
        if (col + 4 <= g.len) {
            const v4f a = *reinterpret_cast<const v4f *>(gn + at), b = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(x + at));
            *reinterpret_cast<v4f *>(gn + at) = b * a;
        } else {
            for (uint64_t i = col; i < g.len; ++i) gn[row * g.n + g.off + i] = x[row * g.n + g.off + i] * gn[row * g.n + g.off + i];
        }
    }
}
// squares of samples [off, off + len) of every row into rows `pstride` floats apart (what SumSqOp walks)
__global__ __launch_bounds__(256) void k_agc_square(float *__restrict__ sq, uint64_t pstride, const float *__restrict__ x, SegArgs g) {
    const uint64_t vpr = (g.len + 3) / 4, nvec = vpr * g.n_streams;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t v = i0; v < nvec; v += stride) {
        const uint64_t row = v / vpr, col = (v % vpr) * 4;
        const float *xi = x + row * g.n + g.off + col;
        float *so = sq + row * pstride + g.off + col;
        if (col + 4 <= g.len) {
            const v4f a = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(xi));
            *reinterpret_cast<v4f *>(so) = a * a;
        } else {
            for (uint64_t i = 0; col + i < g.len; ++i) so[i] = xi[i] * xi[i];
        }
    }
}
// four scratch words per stream for a call without a state: a fresh AGC (agc.rs:209-236)
__global__ __launch_bounds__(256) void k_agc_fresh(float *__restrict__ st4, uint32_t n_streams) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_streams * 4u) st4[i] = (i & 3u) == 3u ? 1.0f : 0.0f;
}
// the carried window in time order: ordered[s][i] = ring[(index + i) & 8191]
__global__ __launch_bounds__(256) void k_agc_window_out(float *__restrict__ ordered, const float *__restrict__ state, uint32_t n_streams) {
    const uint64_t total = (uint64_t)n_streams * kRmsWindow, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
        const uint64_t s = q / kRmsWindow;
        const uint32_t i = (uint32_t)(q % kRmsWindow);
        const float *st = state + s * kAgcStateFloats;
        ordered[q] = st[4 + ((__float_as_uint(st[1]) + i) & (kRmsWindow - 1))];
    }
}
// the window the next block starts from: the last 8192 squares of (carried window ++ this block), stored in time order (index 0)
__global__ __launch_bounds__(256) void k_agc_window_in(float *__restrict__ state, const float *__restrict__ ordered, const float *__restrict__ x, uint64_t n, uint32_t n_streams) {
    const uint64_t total = (uint64_t)n_streams * kRmsWindow, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
        const uint64_t s = q / kRmsWindow, i = q % kRmsWindow;
        float v;
        if (n + i < kRmsWindow) v = ordered[s * kRmsWindow + n + i];
        else {
            const float xv = x[s * n + (n + i - kRmsWindow)];
            v = fabsf(xv) * fabsf(xv);
        }
        float *st = state + s * kAgcStateFloats;
        st[4 + i] = v;
        if (i == 0) st[1] = __uint_as_float(0u);
    }
}

template <class K>
rh_status chain_attr_n(K kernel, const char *what, size_t lds) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        rh::set_hip_error(e, what);
        return RH_ERR_HIP;
    }
    return RH_OK;
}
template <class K>
rh_status chain_attr(K kernel, const char *what) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChainLds);
    if (e != hipSuccess) {
        rh::set_hip_error(e, what);
        return RH_ERR_HIP;
    }
    return RH_OK;
}
template <class Op>
rh_status launch_chain(const ChainArgs &a, hipStream_t s) {
    static const rh_status attr = chain_attr(&k_agc_chain<Op>, "hipFuncSetAttribute(k_agc_chain)");
    if (attr != RH_OK) return attr;
    hipLaunchKernelGGL(k_agc_chain<Op>, dim3((a.n_streams + kGS - 1) / kGS), dim3(384), kChainLds, s, a);
    RH_CHECK_LAUNCH();
    return RH_OK;
}
template <class OpA, class OpB>
rh_status launch_chain2(const ChainArgs &a, const ChainArgs &b, hipStream_t s) {
    static const rh_status attr = chain_attr(&k_agc_chain2<OpA, OpB>, "hipFuncSetAttribute(k_agc_chain2)");
    if (attr != RH_OK) return attr;
    hipLaunchKernelGGL((k_agc_chain2<OpA, OpB>), dim3(2 * ((a.n_streams + kGS - 1) / kGS)), dim3(384), kChainLds, s, a, b);
    RH_CHECK_LAUNCH();
    return RH_OK;
}

}  // namespace

namespace rh {
// rh_agc for rows that start on 16-byte boundaries, dst and src apart.  k5 = {target, attack coefficient, release coefficient, max gain, floor}.
// RH_ERR_UNSUPPORTED: not this shape -- the caller takes the reference-order kernel.
rh_status agc_chain_launch(float *dst, const float *src, uint64_t n_samples, uint32_t n_streams, const float k5[5], float *state, hipStream_t s) {
    if (n_samples >= (1ull << 24) || n_streams == 0 || n_streams > 0x7fffffu) return RH_ERR_UNSUPPORTED;  // 32-bit byte offsets across the rows of a group
    AgcK k{k5[0], k5[1], k5[2], k5[3], k5[4]};
    // release == 0 (the default): peak = peak * 0 + s * 1 = s, no chain, and the gain's release candidate is `desired` itself
    // (GainOp0's shortcuts also want 0.1 <= max and floor <= max, so that desired <= max)
    const bool general = k.release_coeff != 0.0f || !(k.floor <= k.absolute_max_gain) || !(k.absolute_max_gain >= 0.1f);
    // scratch: four state words per stream for a call without a state | the carried windows in time order | one row per stream
    // for the peak levels (general) or for desired * (1 - attack)
    const size_t fresh_floats = state ? 0 : (((size_t)n_streams * 4 + 3) & ~(size_t)3);
    const size_t win_floats = state ? (size_t)n_streams * kRmsWindow : 0;
    // Few streams: a chain is all that runs, and what it costs per sample counts.  The squares then come from a parallel pass, into
    // rows that are NOT a power of two apart: the 64 rows a workgroup walks side by side otherwise sit on the same HBM channels
    // (measured: 22.6 against 14.5 ns per sample for rows of 2^21 and 2^21 + 128 samples).
    const bool presq = !general && n_streams <= 512 && n_samples >= 4 * kRmsWindow;
    const uint64_t pstride = ((n_samples + 31) & ~31ull) + 160;
    if (presq && pstride >= (1ull << 24)) return RH_ERR_UNSUPPORTED;
    std::unique_lock<std::mutex> hold;
    float *scr = nullptr;
    // the `rows` region rounded up to whole 16-byte vectors: the squares behind it are written with v4f stores and fetched by 16-byte
    // LDS-DMA (one stream of an odd length -- 1 x 40 001 -- would otherwise put them 4 to 12 bytes off)
    // release == 0 in one kernel (k_agc_fused): no rows of intermediates at all.  RH_AGC_SEGMENTS=1: the segment-by-segment form
    // (any number of streams: a workgroup takes 16, a CU one workgroup at a time)
    const bool fused = !rh::knob(rh::K_AGC_SEGMENTS);
    const size_t rows_floats = fused ? 0 : (((size_t)n_streams * n_samples + 3) & ~(size_t)3);
    const size_t scratch_floats = fresh_floats + win_floats + rows_floats + (presq && !fused ? (size_t)n_streams * pstride : 0);
    // The scratch is as large as the batch (twice with the squares) and stays with the stream: a batch that would pin more than 8 GiB
    // that way, or whose scratch cannot be had at all, takes the reference-order kernels instead -- they need none (RH_ERR_UNSUPPORTED
    // is the caller's cue; a failed allocation must not fail a call that worked before the chain existed)
    if (scratch_floats * sizeof(float) > (8ull << 30)) return RH_ERR_UNSUPPORTED;
    {
        const hipError_t e = rh::stream_scratch(s, scratch_floats * sizeof(float), reinterpret_cast<void **>(&scr), hold);
        if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
            (void)hipGetLastError();
            return RH_ERR_UNSUPPORTED;
        }
        if (e != hipSuccess) {
            rh::set_hip_error(e, "rh_agc scratch");
            return RH_ERR_HIP;
        }
    }
    float *ordered = state ? scr + fresh_floats : nullptr, *rows = scr + fresh_floats + win_floats, *sq = rows + rows_floats;
    const unsigned wgrid = rh::grid_for((size_t)n_streams * kRmsWindow);
    if (state) {
        hipLaunchKernelGGL(k_agc_window_out, dim3(wgrid), dim3(256), 0, s, ordered, state, n_streams);
    } else {
        hipLaunchKernelGGL(k_agc_fresh, dim3((n_streams * 4 + 255) / 256), dim3(256), 0, s, scr, n_streams);
    }
    RH_CHECK_LAUNCH();
    ChainArgs base;
    base.k = k;
    base.stride0 = base.stride1 = base.stride_out = n_samples;
    base.n_streams = n_streams;
    base.state = state ? state : scr;
    base.state_stride = state ? (uint32_t)kAgcStateFloats : 4u;
    // The two chains of a stream overlap in TIME SEGMENTS: one launch walks the window sum of segment i (even workgroups) next to
    // the gain of segment i - 1 (odd workgroups), the parallel passes in between prepare and finish what they need.  A segment
    // is at least the window long, so that only the first one reads the carried window.
    uint64_t nseg = general ? 1 : n_samples / (2 * kRmsWindow);
    nseg = nseg < 1 ? 1 : (nseg > 8 ? 8 : nseg);
    uint64_t seg_len = ((n_samples + nseg - 1) / nseg + kCS - 1) / kCS * kCS;
    auto seg = [&](uint64_t i, uint64_t &off, uint64_t &len) {
        off = i * seg_len;
        len = off >= n_samples ? 0 : (n_samples - off < seg_len ? n_samples - off : seg_len);
    };
    auto sum_args = [&](uint64_t i) {
        uint64_t off, len;
        seg(i, off, len);
        ChainArgs a = base;
        a.in0 = (presq ? sq : src) + off;
        a.in1 = a.in0 - kRmsWindow;  // the sample 8192 back (never touched in front of the row: those chunks read in1_head)
        if (presq) a.stride0 = a.stride1 = pstride;
        a.in1_head = i == 0 ? ordered : nullptr;
        a.head_chunks = i == 0 ? kHeadChunks : 0;
        a.out = dst + off;
        a.n = len;
        return a;
    };
    auto par_grid = [&](uint64_t len) { return dim3(rh::grid_tiles((size_t)n_streams * ((len + 3) / 4) + 1)); };
    rh_status st = RH_OK;
    if (fused) {
        static const rh_status attr0 = chain_attr_n(&k_agc_fused<false>, "hipFuncSetAttribute(k_agc_fused)", kFusedLds);
        static const rh_status attr1 = chain_attr_n(&k_agc_fused<true>, "hipFuncSetAttribute(k_agc_fused)", kFusedLdsG);
        static const rh_status attr2 = chain_attr_n(&k_agc_fused0, "hipFuncSetAttribute(k_agc_fused0)", kFused0Lds);
        if (attr0 != RH_OK) return attr0;
        if (attr1 != RH_OK) return attr1;
        if (attr2 != RH_OK) return attr2;
        FusedArgs f;
        f.in = src;
        f.in1_head = ordered;
        f.out = dst;
        f.n = n_samples;
        f.stride = f.stride_out = n_samples;
        f.n_streams = n_streams;
        f.state = base.state;
        f.state_stride = base.state_stride;
        f.k = k;
        if (general) hipLaunchKernelGGL(k_agc_fused<true>, dim3((n_streams + kFS - 1) / kFS), dim3(64 * kFWavesG), kFusedLdsG, s, f);
        else if (rh::knob(rh::K_AGC_FUSED_R4)) hipLaunchKernelGGL(k_agc_fused<false>, dim3((n_streams + kFS - 1) / kFS), dim3(64 * kFWaves), kFusedLds, s, f);
        else hipLaunchKernelGGL(k_agc_fused0, dim3((n_streams + kFS - 1) / kFS), dim3(64 * kQWaves), kFused0Lds, s, f);
        RH_CHECK_LAUNCH();
    } else if (general) {
        // window sum -> dst and peak follower -> rows side by side, desired gain in place, then the gain chain with both candidates
        ChainArgs a = sum_args(0), p = base;
        p.in0 = src;
        p.in1 = nullptr;
        p.in1_head = nullptr;
        p.head_chunks = 0;
        p.out = rows;
        p.n = n_samples;
        if ((st = launch_chain2<SumOp, PeakOp>(a, p, s)) != RH_OK) return st;
        const SegArgs g{n_samples, 0, n_samples, n_streams};
        hipLaunchKernelGGL(k_agc_desired, par_grid(n_samples), dim3(256), 0, s, dst, src, rows, (float *)nullptr, g, k, (float *)nullptr);
        RH_CHECK_LAUNCH();
        ChainArgs q = base;
        q.in0 = src;
        q.in1 = dst;
        q.in1_head = nullptr;
        q.head_chunks = 0;
        q.out = dst;
        q.n = n_samples;
        if ((st = launch_chain<GainOp>(q, s)) != RH_OK) return st;
    } else {
        for (uint64_t i = 0; i <= nseg; ++i) {
            uint64_t off, len, poff, plen;
            seg(i, off, len);
            seg(i ? i - 1 : 0, poff, plen);
            ChainArgs gq = base;  // the gain of segment i - 1: clamp(desired) in dst, desired * (1 - attack) in rows
            gq.in0 = dst + poff;
            gq.in1 = rows + poff;
            gq.in1_head = nullptr;
            gq.head_chunks = 0;
            gq.out = dst + poff;
            gq.n = plen;
            if (presq && i < nseg && len) {
                hipLaunchKernelGGL(k_agc_square, par_grid(len), dim3(256), 0, s, sq, pstride, src, SegArgs{n_samples, off, len, n_streams});
                RH_CHECK_LAUNCH();
            }
            if (i < nseg && len && i > 0) st = presq ? launch_chain2<SumSqOp, GainOp0>(sum_args(i), gq, s) : launch_chain2<SumOp, GainOp0>(sum_args(i), gq, s);
            else if (i < nseg && len) st = presq ? launch_chain<SumSqOp>(sum_args(i), s) : launch_chain<SumOp>(sum_args(i), s);
            else if (i > 0 && plen) st = launch_chain<GainOp0>(gq, s);
            if (st != RH_OK) return st;
            if (i > 0 && plen) {  // y = x * gain
                hipLaunchKernelGGL(k_agc_apply, par_grid(plen), dim3(256), 0, s, dst, src, SegArgs{n_samples, poff, plen, n_streams});
                RH_CHECK_LAUNCH();
            }
            if (i < nseg && len) {
                const bool last = off + len == n_samples;
                hipLaunchKernelGGL(k_agc_desired, par_grid(len), dim3(256), 0, s, dst, src, (const float *)nullptr, rows, SegArgs{n_samples, off, len, n_streams}, k,
                                   (state && last) ? state : (float *)nullptr);
                RH_CHECK_LAUNCH();
            }
        }
    }
    if (state) {
        hipLaunchKernelGGL(k_agc_window_in, dim3(wgrid), dim3(256), 0, s, state, ordered, src, n_samples, n_streams);
        RH_CHECK_LAUNCH();
    }
    return RH_OK;
}
}  // namespace rh

// rh_pipeline_dev.h -- device-side helpers the kernels of the fused path share (rh_pipeline.hip: k_rlm_fast / k_rlm_wave / k_rlm_chunk /
// k_mix_*; rh_pipeline_sblk.hip: k_rlm_sblk): the converter's cursor, the 2x2 products, DPP moves, the hand-counted LDS-DMA pipeline, the
// channel-count helpers.  Not part of the C ABI.  Everything sits in an unnamed namespace: every unit gets its own copies.
#pragma once
#include "rh_pipeline_internal.h"

namespace {

struct Cursor {
    uint64_t k, ml, il;
    uint32_t num;
};
__device__ __forceinline__ Cursor cursor_at(uint64_t m, const Params &p) {
    Cursor c;
    if (p.T == 1 && p.F == 1 && !p.chunk_out) {  // pass-through converter: no divisions
        c.k = 0;
        c.ml = c.il = m;
        c.num = 0;
        return c;
    }
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

// t / T of math.rs:25: rh::div_lerp (rh_common.h) -- three instructions and a select where that is the IEEE quotient (rcpT = rh::lerp_rcp(T): this
// T checked exhaustively; zeros and infinities handled), the IEEE sequence for a T that failed the check.  Until round 6's last session this was
// the bare three instructions: +0 for t = -0 (every frame that lands on a tap), NaN for t = Inf.  Nonzero |t| < 2^-120 may still be an ulp off.
__device__ __forceinline__ float div_T(float t, float Tf, float rcpT) { return rh::div_lerp(t, Tf, rcpT); }

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
// v_readlane / v_readfirstlane of a float (the builtins take int: pass the bits, not the value)
__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float readfirstlane_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
constexpr int kDppRowShr = 0x110;    // row_shr:n  = 0x110 + n
constexpr int kDppWaveShr1 = 0x138;  // wave_shr:1
constexpr int kDppBcast15 = 0x142;   // lane 15 of each row -> the next row
constexpr int kDppBcast31 = 0x143;   // lane 31 -> rows 2 and 3

#define RH_LDS __attribute__((address_space(3)))
typedef RH_LDS unsigned char lds_u8;
typedef float v2f __attribute__((ext_vector_type(2)));  // native vectors: HIP's float2/float4 classes
typedef float v4f __attribute__((ext_vector_type(4)));  // cannot be read through address-space pointers
typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
typedef RH_LDS v2f lds_f2;
typedef RH_LDS v4f lds_f4;
typedef RH_LDS v2u64 lds_u64x2;
#define RH_GLB __attribute__((address_space(1)))
typedef RH_GLB const float glb_cf32;  // a pointer loaded from a descriptor is generic: say it is global,
typedef RH_GLB const v2f glb_cf2;     // or every source load is a flat_load that also blocks lgkmcnt

// ---- hand-counted memory pipeline -------------------------------------------------------------
// One LDS-DMA instruction: 64 lanes x 16 bytes from base + voff (per lane) land at LDS byte
// address lds_dst + lane*16 (wave-uniform base in M0; M0 is compiler-reserved, so it is saved and
// restored inside the statement -- cdna_hip_programming.md 5.7).  hipcc does not see the load:
// nothing waits for it except the wait_vm<N>() calls below.
// "s" operands must be provably wave-uniform: rebuild the 64-bit base from two readfirstlanes.
__device__ __forceinline__ const void *uniform_ptr(const void *q) {
    const uint64_t v = (uint64_t)(uintptr_t)q;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const void *)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
#ifndef RH_GLDS_POL
#ifdef RH_GLDS_PLAIN
#define RH_GLDS_POL ""
#else
#define RH_GLDS_POL " nt"
#endif
#endif
__device__ __forceinline__ void glds16(const void *sbase_, uint32_t voff, uint32_t lds_dst_) {
    const void *sbase = uniform_ptr(sbase_);
    const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_);
    uint32_t keep;
#ifdef RH_GLDS_PLAIN  // diagnostics: without the streaming hint
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
    return;
#endif
    // nt: every source byte is read once per launch -- a streaming (non-temporal) fetch does not displace what the
    // caches could reuse and, measured, lifts the achievable read rate from 6.3 to 7.0 TB/s (tools/ubench/read_bw.hip)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" RH_GLDS_POL "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
// The same with sc1 (agent scope: served by L2, never by this CU's L1): the carry granules.
__device__ __forceinline__ void glds16_sc1(const void *sbase_, uint32_t voff, uint32_t lds_dst_) {
    const void *sbase = uniform_ptr(sbase_);
    const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
// N LDS-DMA instructions whose sources AND destinations lie 1 KiB apart (lane l of instruction k: sbase + voff + 1024 k -> lds_dst +
// 1024 k + 16 l): the instruction's immediate offset moves both addresses, so the run shares ONE M0 and ONE offset register.
// Groups of 8 (immediates -4096 .. 3072 around a base 4 KiB in) and of 4 (0 .. 3072).
__device__ __forceinline__ void glds16_x8(const void *sbase_, uint32_t voff, uint32_t lds_dst_) {
    const void *sbase = uniform_ptr(sbase_);
    const uint32_t lds_mid = __builtin_amdgcn_readfirstlane(lds_dst_ + 4096u);
    const uint32_t vmid = voff + 4096u;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:-4096" RH_GLDS_POL "\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:-3072" RH_GLDS_POL "\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:-2048" RH_GLDS_POL "\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:-1024" RH_GLDS_POL "\n\t"
                 "global_load_lds_dwordx4 %1, %2" RH_GLDS_POL "\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024" RH_GLDS_POL "\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048" RH_GLDS_POL "\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:3072" RH_GLDS_POL "\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(vmid), "s"(sbase), "s"(lds_mid)
                 : "memory");
}
__device__ __forceinline__ void glds16_x4(const void *sbase_, uint32_t voff, uint32_t lds_dst_) {
    const void *sbase = uniform_ptr(sbase_);
    const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2" RH_GLDS_POL "\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024" RH_GLDS_POL "\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048" RH_GLDS_POL "\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:3072" RH_GLDS_POL "\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void glds16_run(const void *sbase, uint32_t voff, uint32_t lds_dst) {
    if constexpr (N >= 8) {
        glds16_x8(sbase, voff, lds_dst);
        glds16_run<N - 8>(sbase, voff + 8192u, lds_dst + 8192u);
    } else if constexpr (N >= 4) {
        glds16_x4(sbase, voff, lds_dst);
        glds16_run<N - 4>(sbase, voff + 4096u, lds_dst + 4096u);
    } else if constexpr (N >= 1) {
        glds16(sbase, voff, lds_dst);
        glds16_run<N - 1>(sbase, voff + 1024u, lds_dst + 1024u);
    }
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// Wait until at most n*KV vector-memory operations are outstanding (n uniform, 0..NS-1).
template <int KV, int NS>
__device__ __forceinline__ void wait_groups(int n) {
    if (NS > 3 && n >= 3) wait_vm<(KV * 3 < 63 ? KV * 3 : 63)>();
    else if (NS > 2 && n == 2) wait_vm<(KV * 2 < 63 ? KV * 2 : 63)>();
    else if (n == 1) wait_vm<KV>();
    else wait_vm<0>();
}

#ifdef RH_PHASE_PROFILE
#define RH_PH_DECL unsigned long long ph_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_last = __builtin_readcyclecounter(); const unsigned long long ph_start = ph_last;
#define RH_PH(i) { const unsigned long long ph_now = __builtin_readcyclecounter(); ph_t[i] += ph_now - ph_last; ph_last = ph_now; }
#else
#define RH_PH_DECL
#define RH_PH(i)
#endif

// =================================================================================================
// k_rlm_fast -- the same pipeline when every source has the same length (the benchmark batch, and
// any mixer fed equal blocks).  Everything that couples lanes and tiles is LINEAR in the lanes'
// zero-state run-end states E_s, and all sources share the filter, so it is done once on the SUM
// over the sources instead of once per source:
//     sum_s scan(E_s) = scan(sum_s E_s),   sum_s carry_s = sum_j B^(L*j) * (sum_s aggregate_s(tile-1-j)).
// Per source the wave only streams the input (LDS-DMA ring), lerps, runs the zero-state biquad over
// its lanes' runs and mixes; Eacc += E_s costs 4 adds.  After the last source: one wave64 scan of
// Eacc, ONE published aggregate per tile, one look-back over the J predecessor tiles (which
// finish at about the same time -- the tiles never wait for each other inside the source loop), one
// homogeneous correction g[r] * (start state).  Ragged batches take k_rlm_wave below instead, where a
// source that ends inside a tile needs its own masked correction.
// =================================================================================================
// RAG: the equal-length kernel as the first half of a ragged batch (rh_rlm_run on sources of different lengths,
// with a filter).  A tile then takes only the sources that stay whole for it and the J tiles after it ("stable":
// nothing about them needs a per-source aggregate, see k_rlm_wave) -- for a mixer's worth of tracks that is almost
// every (tile, source) pair -- and the few pairs in which a source is about to end follow (rag_run_pairs: inside this kernel since
// round 4, Params::rag_merge; k_rlm_resid, launched behind it, in the two-launch form).  Mix order: stable sources first (the filtered pipeline is compared at 1e-5, not bitwise).
// ---- channel count as a template parameter (C = 1: mono, C = 2: stereo) -------------------------------------------------------------
// A frame is C floats.  Everything per channel goes through these few helpers, written so that C = 2 spells out exactly the
// operations the stereo kernels always had (component by component, same order): the stereo code objects do not change.
template <int C>
struct Chan;
template <>
struct Chan<2> {
    typedef v2f V;                    // one frame in registers
    static constexpr uint32_t kFB = 8;   // bytes per frame
    static constexpr uint32_t kVF = 2;   // frames per 16-byte vector
    static __device__ __forceinline__ V zero() { return v2f{0.0f, 0.0f}; }
    static __device__ __forceinline__ V ld_lds(const lds_u8 *q) { return *(const lds_f2 *)q; }
    static __device__ __forceinline__ float get(const V &v, int c) { return c ? v.y : v.x; }
    static __device__ __forceinline__ void set(V &v, int c, float x) {
        if (c) v.y = x;
        else v.x = x;
    }
};
template <>
struct Chan<1> {
    typedef float V;
    static constexpr uint32_t kFB = 4;
    static constexpr uint32_t kVF = 4;
    static __device__ __forceinline__ V zero() { return 0.0f; }
    static __device__ __forceinline__ V ld_lds(const lds_u8 *q) { return *(const RH_LDS float *)q; }
    static __device__ __forceinline__ float get(const V &v, int) { return v; }
    static __device__ __forceinline__ void set(V &v, int, float x) { v = x; }
};
// component-wise helpers (v2f: .x then .y, the order the stereo kernels spell)
__device__ __forceinline__ v2f vfma_s(float s, v2f a, v2f c) { return v2f{fma_(s, a.x, c.x), fma_(s, a.y, c.y)}; }
__device__ __forceinline__ float vfma_s(float s, float a, float c) { return fma_(s, a, c); }
__device__ __forceinline__ v2f vmul_s(v2f a, float s) { return v2f{a.x * s, a.y * s}; }
__device__ __forceinline__ float vmul_s(float a, float s) { return a * s; }
__device__ __forceinline__ v2f vsel(bool c, v2f a, v2f b) { return v2f{c ? a.x : b.x, c ? a.y : b.y}; }
__device__ __forceinline__ float vsel(bool c, float a, float b) { return c ? a : b; }

}  // namespace

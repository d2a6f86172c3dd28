// rh_scan_common.h -- device helpers shared by the time-parallel scan kernels (rh_limit.hip, rh_biquad_scan.hip):
// DPP cross-lane moves, wave reductions, the f32 hand-off words with their 0xFF "not yet" pattern, the swizzled LDS
// tile image and its LDS-DMA fetch.  gfx950 only; include inside an anonymous namespace of a .hip file.
#pragma once
#include <type_traits>

#include "rh_common.h"

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
constexpr int kRowShr = 0x110, kWaveShr1 = 0x138, kBcast15 = 0x142, kBcast31 = 0x143;

// plain sums / maxima over the wave (all values >= 0: zero-fill is neutral)
__device__ __forceinline__ float wave_excl_sum(float v, float &total) {
    v += dpp0<kRowShr + 1, 0xf>(v);
    v += dpp0<kRowShr + 2, 0xf>(v);
    v += dpp0<kRowShr + 4, 0xf>(v);
    v += dpp0<kRowShr + 8, 0xf>(v);
    v += dpp0<kBcast15, 0xa>(v);
    v += dpp0<kBcast31, 0xc>(v);
    total = readlane_f(v, 63);
    return dpp0<kWaveShr1, 0xf>(v);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp0<kRowShr + 1, 0xf>(v));
    v = fmaxf(v, dpp0<kRowShr + 2, 0xf>(v));
    v = fmaxf(v, dpp0<kRowShr + 4, 0xf>(v));
    v = fmaxf(v, dpp0<kRowShr + 8, 0xf>(v));
    v = fmaxf(v, dpp0<kBcast15, 0xa>(v));
    v = fmaxf(v, dpp0<kBcast31, 0xc>(v));
    return readlane_f(v, 63);
}

// ---- two channels per instruction ------------------------------------------------------------------------------------------
// The limiter spends most of its issue slots on vector arithmetic (r02 PMC: 47 instructions per sample), and gfx950 executes the f32
// multiply / add / FMA on PAIRS of registers at full rate (v_pk_mul_f32, v_pk_add_f32, v_pk_fma_f32).  With an even channel
// count the channels of a frame go through the per-sample arithmetic two at a time (T = f2); the operations are the same IEEE
// operations in the same order, so the bits do not change.  max, log2, exp2, compares and DPP moves have no packed form.
typedef float f2 __attribute__((ext_vector_type(2)));
template <int C>
struct Pk {
    static constexpr int W = C % 2 == 0 ? 2 : 1, N = C / W;
    typedef typename std::conditional<W == 2, f2, float>::type T;
};
template <class T>
__device__ __forceinline__ T splat(float s);
template <>
__device__ __forceinline__ float splat<float>(float s) { return s; }
template <>
__device__ __forceinline__ f2 splat<f2>(float s) { return (f2)(s); }
__device__ __forceinline__ float vfma(float a, float b, float c) { return fma_(a, b, c); }
__device__ __forceinline__ f2 vfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ f2 vmax(f2 a, f2 b) {
    f2 r;
    r.x = fmaxf(a.x, b.x), r.y = fmaxf(a.y, b.y);
    return r;
}
__device__ __forceinline__ float comp(float v, int) { return v; }
__device__ __forceinline__ float comp(f2 v, int i) { return i ? v.y : v.x; }
__device__ __forceinline__ float pair_of(const float *p, float) { return p[0]; }
__device__ __forceinline__ f2 pair_of(const float *p, f2) {
    f2 r;
    r.x = p[0], r.y = p[1];
    return r;
}
template <int CTRL, int MASK>
__device__ __forceinline__ float vdpp(float v) { return dpp0<CTRL, MASK>(v); }
template <int CTRL, int MASK>
__device__ __forceinline__ f2 vdpp(f2 v) {
    f2 r;
    r.x = dpp0<CTRL, MASK>(v.x), r.y = dpp0<CTRL, MASK>(v.y);
    return r;
}
__device__ __forceinline__ float vsel(bool c, float a, float b) { return c ? a : b; }
__device__ __forceinline__ f2 vsel(bool c, f2 a, f2 b) { return c ? a : b; }


constexpr uint32_t kNotYet = 0xffffffffu;
__device__ __forceinline__ bool word_ok(float v) { return __float_as_uint(v) != kNotYet; }
__device__ __forceinline__ void word_store(float *p, float v) {
    v = v != v ? __uint_as_float(0x7fc00000u) : v;  // a NaN travels in canonical form, never as the sentinel
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
typedef float v2f_ __attribute__((ext_vector_type(2)));
typedef float v4f_ __attribute__((ext_vector_type(4)));
// N consecutive words with agent-scope (sc1: served by L2, never by this CU's L1) loads of width W.  The loads are inline
// asm -- hipcc does not see them -- so wait_loads() must stand between them and the first use of the values.
template <int N, int W>
__device__ __forceinline__ void load_words(const float *p, float (&out)[N]) {
    static_assert(N % W == 0, "section width");
#pragma unroll
    for (int k = 0; k < N; k += W) {
        if (W == 4) {
            v4f_ r;
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(p + k) : "memory");
            out[k] = r.x, out[k + 1] = r.y, out[k + 2] = r.z, out[k + 3] = r.w;
        } else if (W == 2) {
            v2f_ r;
            asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(r) : "v"(p + k) : "memory");
            out[k] = r.x, out[k + 1] = r.y;
        } else {
            float r;
            asm volatile("global_load_dword %0, %1, off sc1" : "=v"(r) : "v"(p + k) : "memory");
            out[k] = r;
        }
    }
}
template <int N>
__device__ __forceinline__ void wait_loads(float (&v)[N]) {  // the values are operands: nothing that uses them can move above the wait
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("" : "+v"(v[k]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("" : "+v"(v[k]));
}

// ... with at most YOUNGER vector-memory instructions of this wave still in flight: vmcnt retires in order, so loads that were issued in FRONT of
// YOUNGER others have arrived when the count is down to YOUNGER.  (The caller knows that at least that many were issued behind them.)
template <int YOUNGER, int N>
__device__ __forceinline__ void wait_loads_behind(float (&v)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("" : "+v"(v[k]));
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("" : "+v"(v[k]));
}

typedef float v4f __attribute__((ext_vector_type(4)));

// ---- the tile's samples in LDS -------------------------------------------------------------------------------------------
// A wave's share of a tile (64 lanes x V 16-byte vectors, V KiB) sits in LDS as 64*V slots.  Slot o*V + (j ^ f(o)) holds
// vector j of lane o's run; f swizzles the vectors of neighbouring runs so that the 16 lanes one ds_read_b128 / ds_write_b128
// pass serves hit 16 different bank groups although the rows are not padded (V a power of two <= 16; other V: f = 0, V odd is
// conflict-free anyway).  The image is written by LDS-DMA straight from HBM (global_load_lds_dwordx4: no VGPR round trip,
// issued a whole tile ahead) -- the DMA lane that fills slot q simply fetches the vector that belongs there.
template <int V>
__device__ __forceinline__ constexpr uint32_t slot_of(uint32_t o, uint32_t j) {
    return (V > 1 && V <= 16 && (V & (V - 1)) == 0) ? o * V + (j ^ ((o / (16 / V)) & (V - 1))) : o * V + j;
}
template <int V>
__device__ __forceinline__ constexpr uint32_t vec_in_slot(uint32_t q) {  // the inverse: which vector of the share lives in slot q
    const uint32_t o = q / V, jj = q % V;
    return (V > 1 && V <= 16 && (V & (V - 1)) == 0) ? o * V + (jj ^ ((o / (16 / V)) & (V - 1))) : q;
}
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__device__ __forceinline__ const void *uniform_ptr(const void *q) {
    const uint64_t v = (uint64_t)(uintptr_t)q;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const void *)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
// One LDS-DMA instruction: 64 lanes x 16 bytes from sbase + voff (per lane) land at LDS byte address lds_dst + lane*16 (M0 is
// compiler-reserved: saved and restored inside the statement, cdna_hip_programming.md 5.7).  hipcc does not see the load;
// `nt`: every sample is read once (streaming fetch, +10 % on the achievable read rate on this part).
__device__ __forceinline__ void glds16(const void *sbase_, uint32_t voff, uint32_t lds_dst_) {
    const void *sbase = uniform_ptr(sbase_);
    const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
template <int V>
__device__ __forceinline__ void dma_share(const float *src_share, v4f *buf, int lane) {
    asm volatile("" : "+v"(lane));  // not hoisted: see limit_tile
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u8 *)buf;
#pragma unroll
    for (int k = 0; k < V; ++k) glds16(src_share, vec_in_slot<V>(k * 64 + lane) * 16u, lds0 + k * 1024);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


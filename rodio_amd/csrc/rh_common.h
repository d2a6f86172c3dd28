// rh_common.h -- shared host-side helpers of librodio_hip (gfx950 only).
#pragma once
#include <mutex>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../include/rodio_hip.h"

namespace rh {

// Set by rh_init(); every entry point refuses to run before it (no CPU fallback exists).
extern bool g_initialized;
extern int g_device;
extern int g_num_cus;
extern uint32_t *g_async_status;  // device word: sticky failure flag of the state-less scan kernels (rh_async_status)
void set_hip_error(hipError_t e, const char *what);

// Diagnostics / tuning variables.  They are read ONCE, by rh_init() -- never on a call's way to a launch; a
// process that changes one afterwards calls rh_init() again.  knob() returns the value or nullptr.
enum Knob {
    K_AGC_SEQ, K_AGC_VEC, K_BIQUAD_NO_FALLBACK, K_BIQUAD_SEQ, K_BIQUAD_R, K_BIQUAD_NW, K_BIQUAD_WGS, K_LIMIT_SEQ, K_LIMIT_R, K_LIMIT_NW, K_LIMIT_WGS, K_LIMIT_GRID,
    K_LIMIT_SKEW, K_LIMIT_NIO, K_LIMIT_INIT, K_SCAN_DMA_TOP, K_SCAN_SPIN_LIMIT, K_NO_HYBRID, K_NO_TICKET_SHARDS, K_PROF_DUMP, K_HOST_ALLOC, K_NO_MIX_FIRST, K_MIX_U, K_NO_CHUNK, K_CHUNK_HALF, K_AUTOTUNE_LOG, K_RAG_RESIDENT, K_RAG_TWO_KERNELS, K_AGC_SEGMENTS, K_RS_PIPE, K_DASP_I64_VIA_F64, K_MIX_GROUPS, K_CLASSES_SIDE_BY_SIDE, K_AGC_FUSED_R4, K_STREAM_UPLOAD_ALWAYS, K_STREAM_NO_REJOIN, K_NO_SBLK, K_SBLK_KV, K_SBLK_NO_OVERLAP, K_CLASSES_ONE_BY_ONE, K_CLASSES_ONE_WAVE, K_WIDE_GENERAL, K_PCM_NO_TILE, K_PCM_TILE_KB, K_LERP_IEEE_DIV, K_COUNT
};
const char *knob(Knob k);
void load_knobs();

#define RH_HIP_TRY(expr)                                   \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) {                            \
            ::rh::set_hip_error(_e, #expr);                \
            return RH_ERR_HIP;                             \
        }                                                  \
    } while (0)

#define RH_REQUIRE_INIT()                                  \
    do {                                                   \
        if (!::rh::g_initialized) return RH_ERR_NOT_INITIALIZED; \
    } while (0)

#define RH_CHECK_LAUNCH()                                  \
    do {                                                   \
        hipError_t _e = hipGetLastError();                 \
        if (_e != hipSuccess) {                            \
            ::rh::set_hip_error(_e, "kernel launch");      \
            return RH_ERR_HIP;                             \
        }                                                  \
    } while (0)

inline hipStream_t as_stream(rh_stream s) { return reinterpret_cast<hipStream_t>(s); }

// A fill that has HAPPENED when it returns.  hipMemset on device memory is enqueued on the null stream and returns at once;
// the library's streams are hipStreamNonBlocking, so a kernel launched on one of them right afterwards does not wait for
// it -- the fill can then land on state that kernel has already written (tickets taken, aggregates published, filter
// states stored).  Seen as wrong blocks / RH_ERR_TIMEOUT when the null stream was slow (GPU shared with other
// processes).  Every fill that initialises kernel-visible state goes through here.
inline hipError_t fill_now(void *p, int value, size_t bytes) {
    hipError_t e = hipMemset(p, value, bytes);
    return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}

// A fill ON a stream, by a kernel of this library (rh_runtime.hip): whatever initialises state that a later kernel reads is
// ordered the way kernels are (hipMemsetAsync was part of the limiter's wrong-state flake, rh_limit.hip, k_limit_init).
hipError_t fill_async(void *p, int value, size_t bytes, hipStream_t s);

// Scratch memory for one launch on stream `s`: a buffer owned by the library, one per stream, grown on demand (which waits
// for the stream once) and reused by every later launch on that stream -- launches on a stream run one after the other, so
// do their uses of it.  Not hipMallocAsync/hipFreeAsync per call: blocks of the stream-ordered pool were seen to change
// under a launch that was still using them (zeros where the launch had written: 2 of 150 short GpuSource chains on ROCm 7.2
// even with every fill done by our own kernels, 0 of 750 with this -- profiles/r02_limit_flake.md).  Freed by rh_stream_destroy for the
// library's own streams; a foreign stream's buffer (a few hundred KiB) lives until the process ends.
// The caller keeps `hold` (taken here) until its last launch that uses the buffer is enqueued: two host threads that launch on
// the same stream then cannot interleave their initialisation and kernel launches.
// What the last user of a stream's scratch left behind, for a user that can save itself work when IT was the last one (the scan
// kernels: hand-off tables that the launch before has already cleared).  Zeroed when the buffer is (re)allocated and by every call
// that does not ask for it (another user has written over the scratch since).  Read and written under `hold`.
struct ScratchAux {
    uint64_t tag;          // who / what shape (0: nobody)
    uint32_t ticket_base;  // value of the scratch's ticket counter when the next launch starts
    uint32_t parity;       // which of two tables the next launch works on
};
hipError_t stream_scratch(hipStream_t s, size_t bytes, void **out, std::unique_lock<std::mutex> &hold, ScratchAux **aux = nullptr);
// device-to-device copy as a launch on `hs` (hipMemcpyAsync DeviceToDevice makes the calling thread wait for the queue ahead of it)
hipError_t copy_d2d(void *dst, const void *src, size_t bytes, hipStream_t hs);
// The lerp's division `(b - a) * num / T` (math.rs:25) in three instructions instead of the IEEE sequence's ten -- q0 = t * r, rem = fma(-q0, T, t),
// q = fma(rem, r, q0) with r = RN(1 / T) from the host (Markstein) -- is NOT the IEEE quotient for every t: tools/ubench/div_check.hip runs all 2^32
// bit patterns of t (profiles/r06_div_check.txt).  For T = 160 it differs for t = -0 (it returns +0; and t is a zero on every frame that lands on
// a tap), for +-Inf (NaN), and for 1.7 M values below 2^-120 where the residual underflows; whether it is exact on the rest depends on T.
// What the FUSED converter without a filter does (div_lerp below; the only kernel family that needs the short form -- the stand-alone converter
// measured the same with the IEEE division and simply divides): (i) lerp_div_fast_ok(T) CHECKS the T of a plan once per process -- every mantissa,
// both signs, three binades (the quotient scales with t as long as nothing underflows) against the IEEE division on the device -- and caches the
// answer; a T that fails gets r = 0: the IEEE division; (ii) zeros and infinities are put right by v_div_fixup_f32 (forms 1 and 2 of div_check.hip: no
// mismatch left above 2^-119 for any T tried); (iii) nonzero
// |t| < 2^-120 -- 1e-36, the last samples of a tail that decays into subnormals -- may come out one unit of THEIR last place off (1e-45 absolute):
// the one place where the unfiltered fused mix is not the reference's bits (a guard per sample cost the kernel a quarter of its rate, 0.78 -> 0.57
// of the roofline; tests/test_gpu_parity.py::test_lerp_division_*).  RH_LERP_IEEE_DIV=1: the IEEE division everywhere.
bool lerp_div_fast_ok(uint32_t T);
inline float lerp_rcp(uint32_t T) { return lerp_div_fast_ok(T) ? 1.0f / (float)T : 0.0f; }
// rh_wav.hip: ChannelCountConverter straight from sample bytes (PCM or f32 frames), a tile of frames per workgroup; false = not launched
bool pcm_tile_try(float *dst, const uint8_t *data, uint64_t n_samples, uint64_t frames, uint32_t channels, uint32_t to_channels, int fmt, hipStream_t s);
// rh_pipeline_plan.hip: `s` has been synchronised and is about to go -- fused-pipeline handles whose launches went there are idle now
// (they record their idle event lazily, on the stream of their last launch: never on a destroyed one).
void rlm_stream_retired(hipStream_t s);

// Grid for a memory-bound grid-stride kernel whose lanes handle SMALL items (one sample, one frame): enough 256-thread blocks to
// fill 256 CUs x 8, capped so that a workgroup has a few KiB to do.
inline unsigned grid_for(size_t work_items, unsigned block = 256, unsigned max_blocks = 256 * 8) {
    size_t b = (work_items + block - 1) / block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return static_cast<unsigned>(b);
}

// Grid for an elementwise kernel whose lanes handle 16-BYTE VECTORS: one vector per lane -- a workgroup owns 256 consecutive vectors
// (4 KiB) and exits; the dispatcher is the loop.  tools/ubench/write_bw.hip measured the capped grid-stride shape 20-48 % behind
// this one (copy 512 MiB -> 512 MiB: 4.3-5.4 TB/s grid-stride, 6.35 TB/s with a workgroup per 4 KiB; the 1:2 expansion of i16 -> f32:
// 0.170 -> 0.137 ms): what the chip works on at any moment is then ONE compact window of the address range per stream instead of
// 2048 fronts 8 MiB apart.  (Not for 4-byte items: a workgroup with 1 KiB to do loses -- rh_channels_convert 0.72 -> 0.65.)  The
// kernels keep their loops (a batch beyond 2^22 workgroups walks on); they simply run once.
inline unsigned grid_tiles(size_t vectors, unsigned block = 256) { return grid_for(vectors, block, 1u << 22); }

// Streaming (non-temporal) accesses for data touched once per launch: on gfx950 a plain read stream tops out at
// ~6.3 TB/s, the same stream with `nt` at ~7.0 TB/s (tools/ubench/read_bw.hip).
#ifdef __HIPCC__
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_nt(const float4 *p) {
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 ld_nt(const uint4 *p) {
    const nt_u4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float ld_nt(const float *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_nt(float4 *p, float4 v) {
    nt_f4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<nt_f4 *>(p));
}
__device__ __forceinline__ void st_nt(float *p, float v) { __builtin_nontemporal_store(v, p); }

// t / Tf as the fused converter takes it (see lerp_div_fast_ok above); rcpT = rh::lerp_rcp(T), the same in every lane
__device__ __forceinline__ float div_lerp(float t, float Tf, float rcpT) {
    if (rcpT == 0.0f) return t / Tf;
    const float q0 = t * rcpT;
    const float rem = __builtin_fmaf(-q0, Tf, t);
    const float q = __builtin_fmaf(rem, rcpT, q0);
    return __builtin_amdgcn_div_fixupf(q, Tf, t);  // v_div_fixup_f32, the instruction the IEEE sequence itself ends with: -0 for t = -0, Inf for Inf (the residual
                                                   // makes +0 and NaN of them), q otherwise -- the same values as a class test and a select of q0, one instruction less
}
// src[q .. q+3] for a row of n samples, q of any sign and alignment; 0.0 where q + j is outside [0, n).  Inside, two aligned 16-byte loads
// (the second one is the next lane's first: an L1 hit) and a pick by the address' residue -- the same for every lane of a launch.  The
// aligned vectors reach up to 12 bytes outside the row, inside the 16 bytes that hold a sample of it.
__device__ __forceinline__ float4 ld4_at(const float *__restrict__ src, int64_t q, uint64_t n) {
    if (q >= 0 && (uint64_t)q + 4 <= n) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(src + q);
        const uint32_t r = (uint32_t)(a >> 2) & 3u;
        const float4 *b = reinterpret_cast<const float4 *>(a & ~(uintptr_t)15);
        const float4 lo = b[0];
        if (r == 0) return lo;
        const float4 hi = b[1];
        if (r == 1) return make_float4(lo.y, lo.z, lo.w, hi.x);
        if (r == 2) return make_float4(lo.z, lo.w, hi.x, hi.y);
        return make_float4(lo.w, hi.x, hi.y, hi.z);
    }
    float e[4];
    for (int j = 0; j < 4; ++j) e[j] = (q + j >= 0 && (uint64_t)(q + j) < n) ? src[q + j] : 0.0f;
    return make_float4(e[0], e[1], e[2], e[3]);
}
// dst[i] = f(i, src[i]) for i < n, FOUR consecutive samples a lane (one 16-byte load, one 16-byte store), a vector a lane and a grid of
// rh::grid_tiles((n + 3) / 4): the shape tools/bench_rows.py measured at 0.79-0.82 of 8 TB/s (amplify, f32 -> i16) against 0.26-0.61 for a
// sample a lane under a capped grid-stride loop.  vec: bit 0 = src starts on a 16-byte boundary, bit 1 = dst does (rows_vec_bits) -- a src that
// does not is read through ld4_at (aligned vectors around the row), a dst that does not is stored sample by sample.  Works in place.
template <int BLOCK, typename F>
__device__ __forceinline__ void map4(float *__restrict__ dst, const float *__restrict__ src, size_t n, int vec, F f) {
    const size_t nvec = (n + 3) / 4, stride = (size_t)gridDim.x * BLOCK;
    for (size_t v = (size_t)blockIdx.x * BLOCK + threadIdx.x; v < nvec; v += stride) {
        const size_t i = 4 * v;
        if (i + 4 <= n) {
            const float4 x = (vec & 1) ? ld_nt(reinterpret_cast<const float4 *>(src) + v) : ld4_at(src, (int64_t)i, n);
            const float4 y = make_float4(f(i, x.x), f(i + 1, x.y), f(i + 2, x.z), f(i + 3, x.w));
            if (vec & 2) {
                st_nt(reinterpret_cast<float4 *>(dst) + v, y);
            } else {
                dst[i] = y.x, dst[i + 1] = y.y, dst[i + 2] = y.z, dst[i + 3] = y.w;
            }
        } else {
            for (int j = 0; j < 4; ++j)
                if (i + j < n) dst[i + j] = f(i + j, src[i + j]);
        }
    }
}
inline int rows_vec_bits(const void *dst, const void *src) {
    return ((reinterpret_cast<uintptr_t>(src) & 15u) == 0 ? 1 : 0) | ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0 ? 2 : 0);
}
#endif

}  // namespace rh

// What does HBM give a kernel that WRITES as much as it reads?  The ceiling the write-heavy kernels (limiter, biquad scan,
// reverb -> spatial, i16 -> f32) are held against.  The guide quotes a float4 copy at 6.29 TB/s (MI355X_MICROARCH.md:35); round 2's
// version of this file reached 4.4-4.9 TB/s with 2048 x 256 lanes and two 16-byte loads in flight per lane, which is simply
// under-subscribed.  This version sweeps what matters:
//   * U = 16-byte loads in flight per lane (1, 2, 4, 8),
//   * the grid (CUs x k persistent workgroups, grid-stride) against one workgroup per tile (no loop at all),
//   * load / store cache policy (plain, nt),
//   * tile-contiguous (a workgroup owns U consecutive 4 KiB pieces) against grid-strided access,
//   * an LDS-DMA ring copy (global_load_lds_dwordx4 -> ds_read_b128 -> global_store), the shape of the scan kernels' I/O.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/write_bw.hip -o tools/ubench/write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int P>
__device__ __forceinline__ v4f ld(const v4f *p) {
    if (P == 1) return __builtin_nontemporal_load(p);
    return *p;
}
template <int P>
__device__ __forceinline__ void st(v4f *p, v4f v) {
    if (P == 1) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <int SP>
__global__ __launch_bounds__(256) void k_fill(v4f *dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    const v4f v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) st<SP>(dst + i, v);
}

// Grid-stride copy, U loads in flight per lane: lane reads i, i+stride, ... (each wave-instruction = 1 KiB contiguous).
template <int U, int LP, int SP>
__global__ __launch_bounds__(256) void k_copy_gs(v4f *__restrict__ dst, const v4f *__restrict__ src, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<LP>(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) st<SP>(dst + i + u * stride, v[u]);
    }
    for (; i < n; i += stride) st<SP>(dst + i, ld<LP>(src + i));
}

// Tile copy: workgroup b owns vectors [b*256*U, (b+1)*256*U) -- U consecutive 4 KiB pieces -- and exits.  No loop: the dispatcher
// is the loop (n / (256*U) workgroups).
template <int U, int LP, int SP>
__global__ __launch_bounds__(256) void k_copy_tile(v4f *__restrict__ dst, const v4f *__restrict__ src, size_t n) {
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<LP>(src + base + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) st<SP>(dst + base + u * 256, v[u]);
}

// 1:2 expansion (the shape of i16 -> f32: 16 B in, 32 B out per lane-vector), one workgroup per tile of 256 * U input vectors, no loop.
template <int U, int LP, int SP>
__global__ __launch_bounds__(256) void k_expand_tile(v4f *__restrict__ dst, const v4f *__restrict__ src, size_t n) {
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<LP>(src + base + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        st<SP>(dst + 2 * (base + u * 256), v[u]);
        st<SP>(dst + 2 * (base + u * 256) + 1, v[u] * 2.0f);
    }
}
// ... and the grid-stride form the conversion kernels had (2 loads in flight, grid 2048)
template <int SP>
__global__ __launch_bounds__(256) void k_expand_gs(v4f *__restrict__ dst, const v4f *__restrict__ src, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n; i += 2 * stride) {
        const v4f a = ld<1>(src + i), b = ld<1>(src + i + stride);
        st<SP>(dst + 2 * i, a);
        st<SP>(dst + 2 * i + 1, a * 2.0f);
        st<SP>(dst + 2 * (i + stride), b);
        st<SP>(dst + 2 * (i + stride) + 1, b * 2.0f);
    }
    if (i < n) {
        const v4f a = ld<1>(src + i);
        st<SP>(dst + 2 * i, a);
        st<SP>(dst + 2 * i + 1, a * 2.0f);
    }
}

// Persistent tile copy with a software pipeline: the loads of tile k+1 are in flight while tile k is stored.
template <int U, int LP, int SP>
__global__ __launch_bounds__(256) void k_copy_pipe(v4f *__restrict__ dst, const v4f *__restrict__ src, size_t n) {
    const size_t tiles = n / (256 * U);
    size_t t = blockIdx.x;
    if (t >= tiles) return;
    v4f a[U], b[U];
    size_t base = t * (256 * U) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = ld<LP>(src + base + u * 256);
    for (;;) {
        const size_t t2 = t + gridDim.x;
        const size_t base2 = (t2 < tiles ? t2 : t) * (256 * U) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u) b[u] = ld<LP>(src + base2 + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) st<SP>(dst + base + u * 256, a[u]);
        if (t2 >= tiles) break;
        t = t2;
        base = base2;
        const size_t t3 = t + gridDim.x;
        const size_t base3 = (t3 < tiles ? t3 : t) * (256 * U) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = ld<LP>(src + base3 + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) st<SP>(dst + base + u * 256, b[u]);
        if (t3 >= tiles) break;
        t = t3;
        base = base3;
    }
}

// LDS-DMA ring copy: one wave per workgroup (like the fused kernels) or NW waves; each wave owns 8 KiB chunks, pulls chunk k+1
// through global_load_lds_dwordx4 while chunk k is read back from the LDS (ds_read_b128) and stored.
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) v4f lds_f4;
template <int NT>
__device__ __forceinline__ void glds16(const void *sbase, uint32_t voff, uint32_t lds_dst_) {
    // M0 holds the LDS base of the wave-instruction; each lane's 16 bytes land at M0 + lane*16 (the form of rh_pipeline.hip)
    const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_);
    uint32_t keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int NW, int LNT, int SP>
__global__ __launch_bounds__(64 * NW) void k_copy_ring(v4f *__restrict__ dst, const v4f *__restrict__ src, size_t n) {
    constexpr int KV = 8;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NW * 2 * KV * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lds_u8 *const lds = (lds_u8 *)smem + wave * (2 * KV * 1024);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    const size_t chunks = n / (KV * 64);
    const size_t nwaves = (size_t)gridDim.x * NW;
    size_t c = (size_t)blockIdx.x * NW + wave;
    if (c >= chunks) return;
    auto stage_chunk = [&](size_t chunk, int stage) {
        const uint64_t b = (uint64_t)(uintptr_t)(src + chunk * (KV * 64));
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        const void *sb = (const void *)(((uint64_t)hi << 32) | lo);
#pragma unroll
        for (int k = 0; k < KV; ++k) glds16<LNT>(sb, (uint32_t)(k * 1024 + lane * 16), lds0 + stage * (KV * 1024) + k * 1024);
    };
    int stage = 0;
    stage_chunk(c, 0);
    for (;;) {
        const size_t c2 = c + nwaves;
        const bool more = c2 < chunks;
        if (more) {
            stage_chunk(c2, stage ^ 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        v4f v[KV];
#pragma unroll
        for (int k = 0; k < KV; ++k) v[k] = *(const lds_f4 *)(lds + stage * (KV * 1024) + k * 1024 + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        v4f *o = dst + c * (KV * 64) + lane;
#pragma unroll
        for (int k = 0; k < KV; ++k) st<SP>(o + k * 64, v[k]);
        if (!more) break;
        // the stores above count in vmcnt too: the next iteration's wait must see only its own DMA group; drain the stores' count
        // by ordering (stores retire in order with the loads): the KV stores sit BEFORE the next DMA group, so vmcnt(8) there
        // waits for them as well.
        c = c2;
        stage ^= 1;
    }
}

// "I/O waves": NIO waves move whole 64 KiB tiles (8 shares of 8 KiB) through two 64 KiB LDS buffers for a workgroup whose other
// waves would only compute: the shape of a scan kernel whose compute waves never touch vector memory (their polls then travel
// alone in vmcnt).  Each I/O wave owns 8/NIO shares of a tile.
template <int NIO, int LNT, int SP>
__global__ __launch_bounds__(64 * NIO) void k_copy_iow(v4f *__restrict__ dst, const v4f *__restrict__ src, size_t n) {
    constexpr int KV = 8, SH = 8, PER = SH / NIO;  // KiB per share, shares per tile, shares per I/O wave
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * SH * KV * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lds_u8 *const lds = (lds_u8 *)smem;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    const size_t tiles = n / (SH * KV * 64);
    size_t t = blockIdx.x;
    if (t >= tiles) return;
    auto stage_tile = [&](size_t tile, int buf) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int sh = wave * PER + i;
            const uint64_t b = (uint64_t)(uintptr_t)(src + (tile * SH + sh) * (KV * 64));
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
            const void *sb = (const void *)(((uint64_t)hi << 32) | lo);
#pragma unroll
            for (int k = 0; k < KV; ++k) glds16<LNT>(sb, (uint32_t)(k * 1024 + lane * 16), lds0 + (buf * SH + sh) * (KV * 1024) + k * 1024);
        }
    };
    int buf = 0;
    stage_tile(t, 0);
    for (;;) {
        const size_t t2 = t + gridDim.x;
        const bool more = t2 < tiles;
        if (more) {
            stage_tile(t2, buf ^ 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * KV) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();  // (where the compute waves would take over the tile, and hand it back)
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int sh = wave * PER + i;
            v4f v[KV];
#pragma unroll
            for (int k = 0; k < KV; ++k) v[k] = *(const lds_f4 *)(lds + (buf * SH + sh) * (KV * 1024) + k * 1024 + lane * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            v4f *o = dst + (t * SH + sh) * (KV * 64) + lane;
#pragma unroll
            for (int k = 0; k < KV; ++k) st<SP>(o + k * 64, v[k]);
        }
        if (!more) break;
        __syncthreads();
        t = t2;
        buf ^= 1;
    }
}

__global__ void k_pattern(v4f *dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float f = (float)(i & 0xffffff);
        dst[i] = v4f{f, f + 0.25f, f + 0.5f, f + 0.75f};
    }
}
__global__ void k_verify(const v4f *a, const v4f *b, size_t n, unsigned long long *bad) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const v4f x = a[i], y = b[i];
        c += (x.x != y.x) || (x.y != y.y) || (x.z != y.z) || (x.w != y.w);
    }
    if (c) atomicAdd(bad, c);
}

template <class F>
double time_ms(F f, int reps = 20) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms / reps;
}

int main(int argc, char **argv) {
    const size_t MiB = 1ull << 20;
    size_t half = 512 * MiB;  // bytes read = bytes written per launch (the limiter / biquad benches move 537 MB each way)
    if (argc > 1) half = (size_t)atol(argv[1]) * MiB;
    v4f *a, *b;
    CK(hipMalloc(&a, half));
    CK(hipMalloc(&b, half));
    CK(hipMemset(b, 0, half));
    unsigned long long *d_bad;
    CK(hipMalloc(&d_bad, 8));
    CK(hipMemset(d_bad, 0, 8));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t n = half / 16;
    hipLaunchKernelGGL(k_pattern, dim3(cus * 8), dim3(256), 0, 0, a, n);
    CK(hipDeviceSynchronize());
    unsigned long long total_bad = 0;
    printf("# tools/ubench/write_bw on %s (%d CUs): copy %zu MiB -> %zu MiB, 16 B per lane per access, 20 launches each; TB/s = (read + written) / time\n", prop.name, cus, half / MiB, half / MiB);
    auto rep = [&](const char *name, double bytes, double ms) {
        unsigned long long bad = 0;
        if (bytes > (double)half) {  // a copy: every vector of dst against src, then dst is cleared for the next variant
            hipLaunchKernelGGL(k_verify, dim3(cus * 8), dim3(256), 0, 0, a, b, n, d_bad);
            CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
            CK(hipMemset(d_bad, 0, 8));
            CK(hipMemset(b, 0, half));
            total_bad += bad;
        }
        printf("%-84s %.4f ms  %.2f TB/s%s\n", name, ms, bytes / ms / 1e9, bad ? "  WRONG" : "");
        fflush(stdout);
    };
    const double B = 2.0 * (double)half;
    rep("fill, plain stores, grid 2048", half, time_ms([&] { hipLaunchKernelGGL(k_fill<0>, dim3(2048), dim3(256), 0, 0, b, n); }));
    rep("fill, plain stores, grid CUs*8", half, time_ms([&] { hipLaunchKernelGGL(k_fill<0>, dim3(cus * 8), dim3(256), 0, 0, b, n); }));
    rep("fill, nt stores, grid CUs*8", half, time_ms([&] { hipLaunchKernelGGL(k_fill<1>, dim3(cus * 8), dim3(256), 0, 0, b, n); }));
    char name[160];
#define GS(U, LP, SP)                                                                                                                                   \
    for (int k : {2, 4, 8, 16}) {                                                                                                                       \
        snprintf(name, sizeof name, "grid-stride copy, U=%d loads in flight, %s loads, %s stores, grid CUs*%d", U, LP ? "nt" : "plain", SP ? "nt" : "plain", k); \
        rep(name, B, time_ms([&] { hipLaunchKernelGGL((k_copy_gs<U, LP, SP>), dim3(cus * k), dim3(256), 0, 0, b, a, n); }));                             \
    }
    GS(2, 1, 0)
    GS(4, 1, 0)
    GS(8, 1, 0)
    GS(4, 0, 0)
    GS(4, 1, 1)
    GS(8, 1, 1)
#define TL(U, LP, SP)                                                                                                                          \
    {                                                                                                                                          \
        snprintf(name, sizeof name, "tile copy (one workgroup per %d KiB, no loop), %s loads, %s stores", 4 * U, LP ? "nt" : "plain", SP ? "nt" : "plain"); \
        rep(name, B, time_ms([&] { hipLaunchKernelGGL((k_copy_tile<U, LP, SP>), dim3((unsigned)(n / (256 * U))), dim3(256), 0, 0, b, a, n); })); \
    }
    TL(1, 1, 0)
    TL(2, 1, 0)
    TL(4, 1, 0)
    TL(8, 1, 0)
    TL(4, 0, 0)
    TL(4, 1, 1)
    TL(8, 1, 1)
    TL(4, 0, 1)
#define PP(U, LP, SP)                                                                                                                                    \
    for (int k : {2, 4, 8}) {                                                                                                                            \
        snprintf(name, sizeof name, "pipelined persistent tile copy, U=%d (2 sets), %s loads, %s stores, grid CUs*%d", U, LP ? "nt" : "plain", SP ? "nt" : "plain", k); \
        rep(name, B, time_ms([&] { hipLaunchKernelGGL((k_copy_pipe<U, LP, SP>), dim3(cus * k), dim3(256), 0, 0, b, a, n); }));                            \
    }
    PP(4, 1, 0)
    PP(8, 1, 0)
    PP(4, 1, 1)
#define RG(NW, LNT, SP)                                                                                                                                  \
    for (int k : {1, 2, 4}) {                                                                                                                            \
        const int per_cu = NW == 1 ? 4 * k : k;                                                                                                          \
        snprintf(name, sizeof name, "LDS-DMA ring copy, %d wave(s)/workgroup, %s DMA, %s stores, %d workgroups per CU", NW, LNT ? "nt" : "plain", SP ? "nt" : "plain", per_cu); \
        rep(name, B, time_ms([&] { hipLaunchKernelGGL((k_copy_ring<NW, LNT, SP>), dim3(cus * per_cu), dim3(64 * NW), 0, 0, b, a, n); }));                 \
    }
    RG(1, 1, 0)
    RG(1, 1, 1)
    RG(4, 1, 0)
    RG(8, 1, 0)
    RG(8, 0, 0)
    RG(8, 1, 1)
    RG(4, 1, 1)
#define IW(NIO, LNT, SP)                                                                                                                    \
    {                                                                                                                                       \
        snprintf(name, sizeof name, "I/O-wave tile copy (64 KiB tiles, 2 x 64 KiB LDS), %d I/O waves, %s DMA, %s stores, 1 workgroup per CU", NIO, LNT ? "nt" : "plain", SP ? "nt" : "plain"); \
        rep(name, B, time_ms([&] { hipLaunchKernelGGL((k_copy_iow<NIO, LNT, SP>), dim3(cus), dim3(64 * NIO), 0, 0, b, a, n); }));          \
    }

    IW(2, 1, 0)
    IW(4, 1, 0)
    IW(8, 1, 0)
    IW(2, 1, 1)
    IW(4, 1, 1)
    IW(8, 1, 1)
    {  // 1:2 expansion: half/2 bytes in, half bytes out
        const size_t ne = n / 2;
        const double Be = 0.75 * (double)half * 2.0 / 2.0 * 1.0;  // (half/2 read + half written)
        auto rep2 = [&](const char *nm, double ms) { printf("%-84s %.4f ms  %.2f TB/s\n", nm, ms, 0.75 * (double)half / ms / 1e9); fflush(stdout); };
        (void)Be;
        rep2("expand 1:2, grid-stride (grid 2048, 2 loads in flight), plain stores", time_ms([&] { hipLaunchKernelGGL((k_expand_gs<0>), dim3(2048), dim3(256), 0, 0, b, a, ne); }));
#define EX(U, LP, SP)                                                                                                                          \
        {                                                                                                                                      \
            snprintf(name, sizeof name, "expand 1:2, tile (one workgroup per %d KiB in, no loop), %s loads, %s stores", 4 * U, LP ? "nt" : "plain", SP ? "nt" : "plain"); \
            rep2(name, time_ms([&] { hipLaunchKernelGGL((k_expand_tile<U, LP, SP>), dim3((unsigned)(ne / (256 * U))), dim3(256), 0, 0, b, a, ne); })); \
        }
        EX(1, 1, 0)
        EX(2, 1, 0)
        EX(4, 1, 0)
        EX(1, 1, 1)
        EX(2, 1, 1)
        EX(1, 0, 0)
    }
    printf("# check: %llu vectors of dst differed from src over all copy variants\n", total_bad);
    return total_bad != 0;
}

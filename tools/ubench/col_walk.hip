// tools/ubench/col_walk.hip -- what does config 3's ACCESS PATTERN get from HBM, with the arithmetic deleted?
// k_reverb_spatial_cols_p (rh_elementwise.hip) walks every stream in steps of the echo delay D: a wave owns W x 64 16-byte columns and visits
// i = c, c + D, c + 2 D, ... (the direct tap of one step is the echo tap of the next, so every input byte is read once).  Here the same walk
// as a pure copy (dst[i] = src[i]), U steps in flight in two register sets, nt loads and stores -- and the variations that tell what costs:
//   rows      64 rows of 2 Mi floats, row stride 8 MiB (config 3) or 8 MiB + pad
//   D         65536 floats (config 3) or D + 256 (a stride that is no power of two)
//   skew      every wave starts its walk at another step (wraps around): the waves of a moment do not all stand in the same 256 KiB window
//   tile      the same bytes as a plain tile copy (one workgroup per 16 KiB, no loop): the box's copy ceiling of the day
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/col_walk tools/ubench/col_walk.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f ldnt(const v4f *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stnt(v4f *p, v4f v) { __builtin_nontemporal_store(v, p); }

template <int U, int W, bool SKEW>
__global__ __launch_bounds__(256) void k_walk(float *__restrict__ dst, const float *__restrict__ src, size_t n, size_t delay, size_t src_stride, size_t dst_stride, uint32_t rows) {
    const size_t groups = delay / (4 * 64 * W);
    const size_t wv = ((size_t)blockIdx.x * 256 + threadIdx.x) / 64;
    if (wv >= groups * rows) return;
    const uint32_t row = (uint32_t)(wv / groups);
    const size_t c = ((wv - (size_t)row * groups) * W * 64 + (threadIdx.x & 63u)) * 4;
    const float *x = src + (size_t)row * src_stride;
    float *o = dst + (size_t)row * dst_stride;
    const size_t steps = n / delay;  // (n % delay == 0)
    const size_t s0 = SKEW ? (wv * 7) % steps : 0;
    auto at = [&](size_t k) { return c + ((s0 + k) % steps) * delay; };
    v4f cur[U][W], nxt[U][W];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int v = 0; v < W; ++v) cur[u][v] = (size_t)u < steps ? ldnt((const v4f *)(x + at(u) + 256 * v)) : v4f{0, 0, 0, 0};
    for (size_t k0 = 0; k0 < steps; k0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int v = 0; v < W; ++v)
                if (k0 + U + u < steps) nxt[u][v] = ldnt((const v4f *)(x + at(k0 + U + u) + 256 * v));
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int v = 0; v < W; ++v)
                if (k0 + u < steps) stnt((v4f *)(o + at(k0 + u) + 256 * v), cur[u][v]);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int v = 0; v < W; ++v) cur[u][v] = nxt[u][v];
    }
}

template <int U>
__global__ __launch_bounds__(256) void k_tile(float *__restrict__ dst, const float *__restrict__ src, size_t n, size_t src_stride, size_t dst_stride) {
    const size_t per_row = n / (4 * 256 * U);
    const size_t row = blockIdx.x / per_row, t = blockIdx.x - row * per_row;
    const v4f *s = (const v4f *)(src + row * src_stride) + t * 256 * U + threadIdx.x;
    v4f *d = (v4f *)(dst + row * dst_stride) + t * 256 * U + threadIdx.x;
    v4f a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = ldnt(s + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) stnt(d + u * 256, a[u]);
}

template <class F>
static double time_ms(F &&f) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < 20; ++i) f();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / 20;
}

int main() {
    const uint32_t rows = 64;
    const size_t n = 2u << 20;  // floats per row
    float *a, *b;
    const size_t slack = 1 << 20;
    CK(hipMalloc(&a, (rows * (n + 4096) + slack) * 4));
    CK(hipMalloc(&b, (rows * (n + 4096) + slack) * 4));
    CK(hipMemset(a, 1, (rows * (n + 4096) + slack) * 4));
    CK(hipMemset(b, 0, (rows * (n + 4096) + slack) * 4));
    const double bytes = 2.0 * rows * n * 4;
    printf("# col_walk: %u rows x %zu floats, read + written %.0f MB per launch, 20 launches each\n", rows, n, bytes / 1e6);
    auto rep = [&](const char *name, double ms) {
        printf("%-100s %.4f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
        fflush(stdout);
    };
#define WALK(U, W, SKEW, D, PAD, NAME)                                                                                                                         \
    {                                                                                                                                                         \
        const size_t delay = (D), stride = n + (PAD);                                                                                                          \
        const size_t nn = n / delay * delay;                                                                                                                  \
        const size_t waves = delay / (4 * 64 * (W)) * rows;                                                                                                    \
        rep(NAME, time_ms([&] { hipLaunchKernelGGL((k_walk<U, W, SKEW>), dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, 0, b, a, nn, delay, stride, stride, rows); })); \
    }
    rep("tile copy, one workgroup per 16 KiB, no loop", time_ms([&] { hipLaunchKernelGGL(k_tile<4>, dim3((unsigned)(rows * (n / 4096))), dim3(256), 0, 0, b, a, n, n, n); }));
    rep("tile copy, one workgroup per 32 KiB, no loop", time_ms([&] { hipLaunchKernelGGL(k_tile<8>, dim3((unsigned)(rows * (n / 8192))), dim3(256), 0, 0, b, a, n, n, n); }));
    WALK(4, 2, false, 65536, 0, "walk U=4 W=2, D = 65536, rows 8 MiB apart (config 3's pattern)")
    WALK(4, 2, false, 65536, 1024, "walk U=4 W=2, D = 65536, rows 8 MiB + 4 KiB apart")
    WALK(4, 2, false, 65536, 4096, "walk U=4 W=2, D = 65536, rows 8 MiB + 16 KiB apart")
    WALK(4, 2, false, 65536 + 512, 0, "walk U=4 W=2, D = 65536 + 512, rows 8 MiB apart")
    WALK(4, 2, true, 65536, 0, "walk U=4 W=2, D = 65536, every wave starts at another step")
    WALK(4, 2, true, 65536, 1024, "walk U=4 W=2, D = 65536, another step, rows 8 MiB + 4 KiB apart")
    WALK(8, 1, false, 65536, 0, "walk U=8 W=1, D = 65536")
    WALK(2, 4, false, 65536, 0, "walk U=2 W=4, D = 65536")
    WALK(4, 4, false, 65536, 0, "walk U=4 W=4, D = 65536")
    WALK(4, 2, false, 16384, 0, "walk U=4 W=2, D = 16384 (four times the steps, a quarter of the waves)")
    WALK(4, 2, false, 262144, 0, "walk U=4 W=2, D = 262144 (a quarter of the steps, four times the waves)")
    return 0;
}

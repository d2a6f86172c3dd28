// Plain streaming-read ceiling on gfx950: how fast can 2 GiB be pulled through vector loads (no LDS-DMA)?
// hipcc --offload-arch=gfx950 -O3 tools/ubench/read_bw.hip -o tools/ubench/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ src, size_t n_vec, float *out) {
    // each block walks a contiguous chunk; U loads of 16 B per lane in flight
    const size_t per_block = (n_vec + gridDim.x - 1) / gridDim.x;
    const size_t b0 = (size_t)blockIdx.x * per_block;
    const size_t b1 = b0 + per_block < n_vec ? b0 + per_block : n_vec;
    float acc = 0.f;
    for (size_t i = b0 + threadIdx.x; i < b1; i += (size_t)256 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + (size_t)u * 256;
            if (j < b1) { if (NT) { v4f t = __builtin_nontemporal_load((const v4f *)(src + j)); v[u] = make_float4(t.x, t.y, t.z, t.w); } else v[u] = src[j]; }
            else v[u] = make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}
// interleaved: consecutive blocks read consecutive 4 KiB pieces (grid-stride), like a tile-major kernel
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read_gs(const float4 *__restrict__ src, size_t n_vec, float *out) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n_vec; i += stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + (size_t)u * 256;
            if (j < n_vec) { if (NT) { v4f t = __builtin_nontemporal_load((const v4f *)(src + j)); v[u] = make_float4(t.x, t.y, t.z, t.w); } else v[u] = src[j]; }
            else v[u] = make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <typename K>
void run(const char *name, K kern, int grid, const float4 *src, size_t n_vec, float *out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, src, n_vec, out);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, src, n_vec, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-28s grid %6d  %.4f ms  %.0f GB/s\n", name, grid, best, (double)n_vec * 16 / best / 1e6);
}

int main() {
    const size_t bytes = 2ull << 30;
    const size_t n_vec = bytes / 16;
    float4 *src;
    float *out;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(src, 1, bytes));
    for (int grid : {256, 512, 1024, 2048, 4096, 8192}) {
        run("chunk U4", k_read<4, false>, grid, src, n_vec, out);
        run("chunk U8", k_read<8, false>, grid, src, n_vec, out);
        run("chunk U8 nt", k_read<8, true>, grid, src, n_vec, out);
        run("gridstride U4", k_read_gs<4, false>, grid, src, n_vec, out);
        run("gridstride U8", k_read_gs<8, false>, grid, src, n_vec, out);
        run("gridstride U8 nt", k_read_gs<8, true>, grid, src, n_vec, out);
    }
    return 0;
}

// What does HBM give a kernel that mostly WRITES?  Ceilings for the conversion kernels (i16 -> f32 moves 2 B in, 4 B out per
// sample) next to the float4 copy the guide quotes (6.3 TB/s).  16 B per lane per access, grid-stride, plain and nt stores.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/write_bw.hip -o tools/ubench/write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT>
__device__ __forceinline__ void st(v4f *p, v4f v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_fill(v4f *dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    const v4f v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) st<NT>(dst + i, v);
}
template <bool NT>
__global__ __launch_bounds__(256) void k_copy(v4f *dst, const v4f *src, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n; i += 2 * stride) {
        const v4f a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        st<NT>(dst + i, a);
        st<NT>(dst + i + stride, b);
    }
    if (i < n) st<NT>(dst + i, __builtin_nontemporal_load(src + i));
}
// 16 B in, 32 B out per lane (the shape of i16 -> f32)
template <bool NT>
__global__ __launch_bounds__(256) void k_expand(v4f *dst, const v4f *src, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n; i += 2 * stride) {
        const v4f a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        st<NT>(dst + 2 * i, a);
        st<NT>(dst + 2 * i + 1, a * 2.0f);
        st<NT>(dst + 2 * (i + stride), b);
        st<NT>(dst + 2 * (i + stride) + 1, b * 2.0f);
    }
    if (i < n) {
        const v4f a = __builtin_nontemporal_load(src + i);
        st<NT>(dst + 2 * i, a);
        st<NT>(dst + 2 * i + 1, a * 2.0f);
    }
}
template <class F>
double time_ms(F f, int reps = 20) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}
int main() {
    const size_t GiB = 1ull << 30;
    v4f *a, *b;
    CK(hipMalloc(&a, GiB));
    CK(hipMalloc(&b, GiB));
    CK(hipMemset(a, 1, GiB));
    const int grid = 2048;
    auto rep = [&](const char *name, double bytes, double ms) { printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9); };
    rep("fill 1 GiB, plain stores", GiB, time_ms([&] { hipLaunchKernelGGL(k_fill<false>, dim3(grid), dim3(256), 0, 0, b, GiB / 16); }));
    rep("fill 1 GiB, nt stores", GiB, time_ms([&] { hipLaunchKernelGGL(k_fill<true>, dim3(grid), dim3(256), 0, 0, b, GiB / 16); }));
    rep("copy 512 MiB -> 512 MiB (1:1), plain stores", GiB, time_ms([&] { hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, 0, b, a, GiB / 32); }));
    rep("copy 512 MiB -> 512 MiB (1:1), nt stores", GiB, time_ms([&] { hipLaunchKernelGGL(k_copy<true>, dim3(grid), dim3(256), 0, 0, b, a, GiB / 32); }));
    rep("expand 256 MiB -> 512 MiB (1:2), plain stores", 0.75 * GiB, time_ms([&] { hipLaunchKernelGGL(k_expand<false>, dim3(grid), dim3(256), 0, 0, b, a, GiB / 64); }));
    rep("expand 256 MiB -> 512 MiB (1:2), nt stores", 0.75 * GiB, time_ms([&] { hipLaunchKernelGGL(k_expand<true>, dim3(grid), dim3(256), 0, 0, b, a, GiB / 64); }));
    return 0;
}

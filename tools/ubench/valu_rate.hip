// VALU issue-rate microbenchmark for gfx950: how many cycles does one wave64 VALU instruction of each
// flavour occupy a SIMD?  (tools only; not part of the product.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 2048;
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float a, float b) {
    float x[8];
    v2f p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; p[i] = v2f{x[i], x[i] + 1.f}; }
    const v2f pa = {a, a}, pb = {b, b};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
            if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], pa, pb);
            if (MODE == 2) p[i] = p[i] * pa;
            if (MODE == 3) p[i] = p[i] + pb;
            if (MODE == 4) x[i] = x[i] * a;
            if (MODE == 5) x[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[i]), 0x111, 0xf, 0xf, true)) + b;
            if (MODE == 6) x[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[i]), 0x138, 0xf, 0xf, true));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
int run(const char *name, float *d, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double insts_per_simd = (double)ITER * 8 * waves_per_simd;
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (x2.4 GHz = %.2f cyc)\n", name, waves_per_simd, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
    return 0;
}
int main() {
    float *d; CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", d, w); run<1>("v_pk_fma_f32", d, w); run<2>("v_pk_mul_f32", d, w); run<3>("v_pk_add_f32", d, w);
        run<4>("v_mul_f32", d, w); run<5>("v_add_f32 dpp row_shr1", d, w); run<6>("v_mov_b32 dpp wave_shr1", d, w);
    }
    return 0;
}

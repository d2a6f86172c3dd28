// What can ONE CU pull through LDS-DMA rings when a workgroup of W waves shares a short window of many rows (k_rlm_sblk's loop)?
// Every workgroup owns window `blockIdx.x` (KV KiB) of S rows of 8 MiB; wave w walks rows w, w + W, ... through a ring of NS stages.
// Prints GB/s in total and per CU for a number of workgroups (= CUs in use).   (tools only; not part of the product.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
#define LDS __attribute__((address_space(3)))
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int KV, int NS, int W, int PLAIN>
__global__ __launch_bounds__(64 * W) void k(const float *in, float *out, uint32_t S, uint64_t row_f, uint32_t stride_f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS unsigned char *)smem + wave * NS * KV * 1024;
    const float *base = in + (uint64_t)blockIdx.x * stride_f + lane * 4;
    const uint32_t n = (S - wave + W - 1) / W;
    v4f acc[KV];
#pragma unroll
    for (int q = 0; q < KV; ++q) acc[q] = v4f{0.f, 0.f, 0.f, 0.f};
    if (PLAIN) {  // plain vector loads, NS rows in flight per wave
        v4f buf[NS][KV];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int kk = 0; kk < KV; ++kk) buf[s][kk] = s < (int)n ? __builtin_nontemporal_load((const v4f *)(base + (uint64_t)(wave + s * W) * row_f + kk * 256)) : v4f{0, 0, 0, 0};
        for (uint32_t s0 = 0; s0 < n; s0 += NS) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int kk = 0; kk < KV; ++kk) acc[kk] += buf[s][kk];
                const uint32_t nx = s0 + s + NS;
#pragma unroll
                for (int kk = 0; kk < KV; ++kk) buf[s][kk] = nx < n ? __builtin_nontemporal_load((const v4f *)(base + (uint64_t)(wave + nx * W) * row_f + kk * 256)) : v4f{0, 0, 0, 0};
            }
        }
    } else {
        auto issue = [&](uint32_t s) {
            const float *g = base + (uint64_t)(wave + s * W) * row_f;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (s % NS) * KV * 1024);
#pragma unroll
            for (int kk = 0; kk < KV; ++kk) glds16(g + kk * 256, dst + kk * 1024);
        };
        for (uint32_t s = 0; s < NS - 1 && s < n; ++s) issue(s);
        for (uint32_t s = 0; s < n; ++s) {
            if (s + NS - 1 < n) { issue(s + NS - 1); wait_vm<KV *(NS - 1)>(); } else wait_vm<0>();
            const LDS unsigned char *st = (const LDS unsigned char *)smem + wave * NS * KV * 1024 + (s % NS) * KV * 1024;
#pragma unroll
            for (int q = 0; q < KV; ++q) acc[q] += *(const LDS v4f *)(st + (q * 64 + lane) * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    v4f t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < KV; ++q) t += acc[q];
    out[(uint64_t)blockIdx.x * 64 * W + threadIdx.x] = t.x + t.y + t.z + t.w;
}
template <int KV, int NS, int W, int PLAIN>
void run(const float *d_in, float *d_out, uint32_t S, uint32_t blocks, uint64_t row_f) {
    const size_t lds = PLAIN ? 0 : (size_t)W * NS * KV * 1024;
    CHECK(hipFuncSetAttribute((const void *)k<KV, NS, W, PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<KV, NS, W, PLAIN>), dim3(blocks), dim3(64 * W), lds, 0, d_in, d_out, S, row_f, KV * 256);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)S * blocks * KV * 1024;
    printf("%s KV=%d NS=%2d W=%2d blocks=%4u (in flight %3zu KiB per WG): %7.1f us  %6.0f GB/s  %5.1f GB/s per WG\n", PLAIN ? "plain" : "dma  ", KV, NS, W, blocks, (size_t)W * NS * KV, best * 1e3,
           bytes / best / 1e6, bytes / best / 1e6 / blocks);
}
int main() {
    const uint32_t S = 256;
    const uint64_t src_bytes = 8ull << 20;
    float *d_in, *d_out;
    CHECK(hipMalloc(&d_in, S * src_bytes + (1 << 20)));
    CHECK(hipMemset(d_in, 0, S * src_bytes + (1 << 20)));
    CHECK(hipMalloc(&d_out, 64 << 20));
    const uint64_t row_f = src_bytes / 4;
    for (uint32_t b : {1u, 32u, 128u, 256u, 512u}) {
        run<1, 12, 8, 0>(d_in, d_out, S, b, row_f);
        run<2, 6, 8, 0>(d_in, d_out, S, b, row_f);
        run<3, 4, 8, 0>(d_in, d_out, S, b, row_f);
        run<4, 3, 8, 0>(d_in, d_out, S, b, row_f);
        run<2, 4, 16, 0>(d_in, d_out, S, b, row_f);
        run<1, 8, 16, 0>(d_in, d_out, S, b, row_f);
        run<2, 2, 8, 0>(d_in, d_out, S, b, row_f);
        run<2, 3, 8, 0>(d_in, d_out, S, b, row_f);
        run<1, 4, 8, 1>(d_in, d_out, S, b, row_f);
        run<2, 4, 8, 1>(d_in, d_out, S, b, row_f);
        run<2, 4, 16, 1>(d_in, d_out, S, b, row_f);
        run<4, 2, 8, 1>(d_in, d_out, S, b, row_f);
        run<1, 8, 8, 1>(d_in, d_out, S, b, row_f);
    }
    return 0;
}

// HBM -> LDS streaming microbenchmark with the access pattern of the fused pipeline kernel:
// one wave per workgroup owns a time tile; it walks S sources, and for each DMAs KV KiB of that
// source (global_load_lds_dwordx4) into a ring of NS LDS stages, waits with a counted vmcnt and
// reads the stage back with ds_read_b64.  Answers: what fraction of the HBM roofline does this
// data path reach before any DSP work is added?   (tools only; not part of the product.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
#define LDS __attribute__((address_space(3)))

#ifndef RING_NT
#define RING_NT " nt"  // -DRING_NT='""' for the plain fetch
#endif
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" RING_NT "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int KV, int NS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(const float *in, float *out, uint32_t S, uint64_t src_stride_f, int flops) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tile = blockIdx.x * WAVES + wave;
    const uint32_t stage_bytes = KV * 1024;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS unsigned char *)smem + wave * NS * stage_bytes;
    const float *base = in + (uint64_t)tile * (KV * 256) + lane * 4;
    auto issue = [&](uint32_t s) {
        const float *g = base + (uint64_t)s * src_stride_f;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (s % NS) * stage_bytes);
#pragma unroll
        for (int kk = 0; kk < KV; ++kk) glds16(g + kk * 256, dst + kk * 1024);
    };
    for (uint32_t s = 0; s < NS - 1 && s < S; ++s) issue(s);
    v2f acc[KV * 2];
#pragma unroll
    for (int q = 0; q < KV * 2; ++q) acc[q] = v2f{0.f, 0.f};
    for (uint32_t s = 0; s < S; ++s) {
        if (s + NS - 1 < S) { issue(s + NS - 1); wait_vm<KV *(NS - 1)>(); } else wait_vm<0>();
        const LDS unsigned char *st = (const LDS unsigned char *)smem + wave * NS * stage_bytes + (s % NS) * stage_bytes;
#pragma unroll
        for (int q = 0; q < KV * 2; ++q) {
            v2f v = *(const LDS v2f *)(st + (q * 64 + lane) * 8);
            for (int f = 0; f < flops; ++f) v = __builtin_elementwise_fma(v, v2f{0.5f, 0.5f}, v2f{0.25f, 0.25f});
            acc[q] += v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the stage is re-targeted by the next DMA
    }
    v2f t = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < KV * 2; ++q) t += acc[q];
    out[(uint64_t)tile * 64 + lane] = t.x + t.y;
}

// The fused kernel's exact read pattern: tile t of source s reads nvec 16-byte vectors starting at vector
// t*vstride (+ s*src_stride): chunks are 16-byte but not 128-byte aligned, the KV*64 - nvec surplus lanes
// re-fetch the chunk's last vector.
// CL != 0: the instructions cover blocks aligned to A vectors, but lanes outside the needed span [first 128-byte line of
// the span, its last vector] are clamped into it -- aligned instructions without fetching more bytes
template <int KV, int NS, int A = 1, int CL = 0>
__global__ __launch_bounds__(64) void k_mimic(const float *in, float *out, uint32_t S, uint64_t src_stride_f, uint32_t vstride, uint32_t nvec_) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const uint32_t tile = blockIdx.x;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS unsigned char *)smem;
    uint32_t off[KV];
    const uint32_t v0 = tile * vstride, va = v0 & ~(uint32_t)(A - 1);  // span start rounded down to A vectors (A*16 bytes)
    uint32_t nvec = nvec_ + (v0 - va);
    nvec = nvec < (uint32_t)KV * 64 ? nvec : (uint32_t)KV * 64;
#pragma unroll
    for (int kk = 0; kk < KV; ++kk) {
        uint32_t j = lane + kk * 64;
        j = j < nvec ? j : nvec - 1;
        if (CL) {
            const uint32_t lo = (v0 & ~7u) - va;  // first vector of the span's first 128-byte line
            j = j < lo ? lo : j;
        }
        off[kk] = (va + j) * 4;  // floats
    }
    auto issue = [&](uint32_t s) {
        const float *g = in + (uint64_t)s * src_stride_f;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (s % NS) * KV * 1024);
#pragma unroll
        for (int kk = 0; kk < KV; ++kk) glds16(g + off[kk], dst + kk * 1024);
    };
    for (uint32_t s = 0; s < NS - 1 && s < S; ++s) issue(s);
    v2f acc = {0.f, 0.f};
    for (uint32_t s = 0; s < S; ++s) {
        if (s + NS - 1 < S) { issue(s + NS - 1); wait_vm<KV *(NS - 1)>(); } else wait_vm<0>();
        const LDS unsigned char *st = (const LDS unsigned char *)smem + (s % NS) * KV * 1024;
#pragma unroll
        for (int q = 0; q < KV * 2; ++q) acc += *(const LDS v2f *)(st + (q * 64 + lane) * 8);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[(uint64_t)tile * 64 + lane] = acc.x + acc.y;
}

// "Mix first" inside one kernel: tile t owns the ALIGNED 8 KiB chunk t of every source (8 full instructions) and needs one
// 128-byte line on either side of it (the neighbours' taps): a ninth instruction whose lanes 0..7 / 8..15 fetch the line
// before / after the chunk, the other lanes clamped onto the last of them.  HALO = 0: without the ninth instruction.
template <int NS, int HALO, int ROT = 0>
__global__ __launch_bounds__(64) void k_halo(const float *in, float *out, uint32_t S, uint64_t src_stride_f, uint32_t n_chunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KV = 8 + HALO;
    const int lane = threadIdx.x;
    const uint32_t tile = blockIdx.x;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(LDS unsigned char *)smem;
    uint32_t off[KV];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) off[kk] = (tile * 512u + kk * 64u + lane) * 4;
    if (HALO) {
        const uint32_t l = lane < 16 ? lane : 15;
        int64_t v = l < 8 ? (int64_t)tile * 512 - 8 + l : (int64_t)tile * 512 + 512 + (l - 8);
        if (v < 0) v = 0;
        if (v >= (int64_t)n_chunks * 512) v = (int64_t)n_chunks * 512 - 1;
        off[KV - 1] = (uint32_t)v * 4;
    }
    auto issue = [&](uint32_t s) {
        // ROT: every tile starts its walk over the sources somewhere else (ROT * tile sources in): the tiles of a launch then read S
        // different rows at a time instead of all of them the same one
        const uint32_t sr = ROT ? (s + tile * (uint32_t)ROT) % S : s;
        const float *g = in + (uint64_t)sr * src_stride_f;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (s % NS) * KV * 1024);
#pragma unroll
        for (int kk = 0; kk < KV; ++kk) glds16(g + off[kk], dst + kk * 1024);
    };
    for (uint32_t s = 0; s < NS - 1 && s < S; ++s) issue(s);
    v2f acc = {0.f, 0.f};
    for (uint32_t s = 0; s < S; ++s) {
        if (s + NS - 1 < S) { issue(s + NS - 1); wait_vm<KV *(NS - 1)>(); } else wait_vm<0>();
        const LDS unsigned char *st = (const LDS unsigned char *)smem + (s % NS) * KV * 1024;
#pragma unroll
        for (int q = 0; q < KV * 2; ++q) acc += *(const LDS v2f *)(st + (q * 64 + lane) * 8);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[(uint64_t)tile * 64 + lane] = acc.x + acc.y;
}

template <int KV, int NS, int WAVES>
void run(const float *d_in, float *d_out, uint32_t S, uint64_t total_tiles, uint64_t src_stride_f, int flops) {
    const uint32_t blocks = total_tiles / WAVES;
    const size_t lds = (size_t)WAVES * NS * KV * 1024;
    CHECK(hipFuncSetAttribute((const void *)k<KV, NS, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k<KV, NS, WAVES>, 64 * WAVES, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<KV, NS, WAVES>), dim3(blocks), dim3(64 * WAVES), lds, 0, d_in, d_out, S, src_stride_f, flops);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)S * total_tiles * KV * 1024;
    printf("KV=%d NS=%d waves/wg=%d flops=%d tiles=%llu (%.1f waves/CU, occ %d wg/CU, lds %zu) : %.3f ms  %.0f GB/s  (%.1f%% of 8 TB/s)\n", KV, NS, WAVES, flops,
           (unsigned long long)total_tiles, total_tiles / 256.0, occ, lds, best, bytes / best / 1e6, bytes / best / 1e6 / 80.0);
}
int main(int argc, char **argv) {
    const bool cal = argc > 1;  // calibration mode for FETCH_SIZE: one configuration, known byte count
    const uint32_t S = 256;
    const uint64_t src_bytes = 8ull << 20;  // 8 MiB per source, as config 2
    float *d_in, *d_out;
    CHECK(hipMalloc(&d_in, S * src_bytes + (1 << 20)));
    CHECK(hipMemset(d_in, 0, S * src_bytes + (1 << 20)));
    CHECK(hipMalloc(&d_out, 64 << 20));
    const uint64_t stride = src_bytes / 4;
    if (cal && argv[1][0] == 'f') {  // ceiling study: 4 waves per CU, 8 KiB stages, plain VALU work per 8 bytes
        for (int flops : {0, 2, 4, 6, 8, 10, 12}) run<8, 2, 1>(d_in, d_out, S, 1024, stride, flops);
        for (int flops : {0, 4, 8, 10}) run<8, 3, 1>(d_in, d_out, S, 1024, stride, flops);
        for (int flops : {0, 4, 8, 10}) run<4, 2, 1>(d_in, d_out, S, 2048, stride, flops);
        return 0;
    }
    if (cal && argv[1][0] == 't') {  // timing of the kernel's chunk patterns (R = 18: 1152 out frames -> 1058.4 in frames = 529.2 vectors, 533 fetched)
        auto timeit = [&](const char *name, auto kern, uint32_t tiles, size_t lds, uint32_t vstride, uint32_t nvec) {
            CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(kern, dim3(tiles), dim3(64), lds, 0, d_in, d_out, S, stride, vstride, nvec);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            const double bytes = (double)S * ((double)(tiles - 1) * vstride + nvec) * 16.0;
            printf("%-34s tiles %5u vstride %4u nvec %4u lds %6zu : %.3f ms  %.0f GB/s (distinct bytes)\n", name, tiles, vstride, nvec, lds, best, bytes / best / 1e6);
        };
        auto timeh = [&](const char *name, auto kern, size_t lds) {
            CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(kern, dim3(1024), dim3(64), lds, 0, d_in, d_out, S, stride, 1024u);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("%-34s lds %6zu : %.3f ms  %.0f GB/s (2 GiB)\n", name, lds, best, 2147483648.0 / best / 1e6);
        };
        timeh("aligned 8 KiB, no halo, ring 2", k_halo<2, 0>, 2 * 8 * 1024);
        timeh("aligned 8 KiB + halo instr, ring 2", k_halo<2, 1>, 2 * 9 * 1024);
        timeh("aligned 8 KiB + halo, ring 2, 40K", k_halo<2, 1>, 40960);
        timeh("aligned 8 KiB + halo instr, ring 3", k_halo<3, 1>, 3 * 9 * 1024);
        timeit("aligned 8 KiB chunks KV8", k_mimic<8, 2>, 1024, 16384, 512, 512);
        timeit("aligned 9 KiB chunks KV9", k_mimic<9, 2>, 910, 18432, 576, 576);
        timeit("R18 pattern KV9 (clamped tail)", k_mimic<9, 2>, 991, 18432, 529, 533);
        timeit("R18 pattern KV9, lds 40960", k_mimic<9, 2>, 991, 40960, 529, 533);
        timeit("R18 stride, 9 full instr", k_mimic<9, 2>, 991, 40960, 529, 576);
        timeit("R18, starts on 128 B (kernel)", k_mimic<9, 2, 8>, 991, 40960, 529, 533);
        timeit("R18, starts on 256 B", k_mimic<9, 2, 16>, 991, 40960, 529, 533);
        timeit("R18, starts on 512 B, KV9", k_mimic<9, 2, 32>, 991, 40960, 529, 533);
        timeit("R18, starts on 1 KiB, KV10", k_mimic<10, 2, 64>, 991, 40960, 529, 533);
        timeit("R18, starts on 4 KiB, KV10", k_mimic<10, 2, 256>, 991, 40960, 529, 533);
        timeit("R18, 1 KiB instr, clamped, KV10", k_mimic<10, 2, 64, 1>, 991, 40960, 529, 533);
        timeit("R18, 512 B instr, clamped, KV9", k_mimic<9, 2, 32, 1>, 991, 40960, 529, 533);
        timeit("R18, 256 B instr, clamped, KV9", k_mimic<9, 2, 16, 1>, 991, 40960, 529, 533);
        timeit("R9, 1 KiB instr, clamped, KV6", k_mimic<6, 2, 64, 1>, 1982, 20480, 265, 269);
        timeit("R9 pattern KV5", k_mimic<5, 2>, 1982, 20480, 265, 269);
        timeit("R9, starts on 128 B", k_mimic<5, 2, 8>, 1982, 20480, 265, 269);
        timeit("R10 pattern KV5", k_mimic<5, 2>, 1784, 23040, 294, 298);
        return 0;
    }
    if (cal && argv[1][0] == 'o') {  // occupancy study: how fast does ONE wave pull its 256 x 8 KiB when fewer waves share the chip?
        auto timeo = [&](const char *name, auto kern, uint32_t tiles, size_t lds) {
            CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(kern, dim3(tiles), dim3(64), lds, 0, d_in, d_out, S, stride, 1024u);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            const double bytes = (double)S * tiles * 8192.0;
            printf("%-14s waves %5u (%.2f per CU) : %.3f ms  %6.0f GB/s  %5.2f GB/s per wave  %5.0f ns per 8 KiB\n", name, tiles, tiles / 256.0, best, bytes / best / 1e6, bytes / best / 1e6 / tiles,
                   best * 1e6 / S);
        };
        timeo("ring 2, rot 1", k_halo<2, 0, 1>, 1024, 2 * 8 * 1024);
        timeo("ring 2, rot 7", k_halo<2, 0, 7>, 1024, 2 * 8 * 1024);
        timeo("ring 2, rot 37", k_halo<2, 0, 37>, 1024, 2 * 8 * 1024);
        timeo("ring 2, rot 64", k_halo<2, 0, 64>, 1024, 2 * 8 * 1024);
        timeo("ring 3, rot 37", k_halo<3, 0, 37>, 1024, 3 * 8 * 1024);
        for (uint32_t tiles : {1024u, 768u, 512u, 384u, 256u, 128u, 32u, 1u}) {
            timeo("ring of 2", k_halo<2, 0>, tiles, 2 * 8 * 1024);
            timeo("ring of 3", k_halo<3, 0>, tiles, 3 * 8 * 1024);
            timeo("ring of 4", k_halo<4, 0>, tiles, 4 * 8 * 1024);
        }
        return 0;
    }
    if (cal && argv[1][0] == 'm') {  // mimic: R=10 tiles of config 2 (640 out frames -> 588 in frames = 294 vectors; 298 fetched)
        const uint32_t vstride = 294, nvec = 298, tiles = 1784;
        CHECK(hipFuncSetAttribute((const void *)k_mimic<5, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int rep = 0; rep < 4; ++rep) {
            hipLaunchKernelGGL((k_mimic<5, 2>), dim3(tiles), dim3(64), 23040, 0, d_in, d_out, S, stride, vstride, nvec);
            CHECK(hipDeviceSynchronize());
        }
        printf("known_bytes_per_launch %llu\n", (unsigned long long)S * ((unsigned long long)(tiles - 1) * vstride + nvec) * 16ull);
        return 0;
    }
    if (cal) {
        run<4, 2, 1>(d_in, d_out, S, 2048, stride, 0);
        printf("known_bytes_per_launch %llu\n", (unsigned long long)S * 2048ull * 4096ull);
        return 0;
    }
    for (int flops : {0, 8}) {
        // tiles * KV KiB == 8 MiB per source
        run<2, 2, 1>(d_in, d_out, S, 4096, stride, flops);
        run<2, 4, 1>(d_in, d_out, S, 4096, stride, flops);
        run<4, 2, 1>(d_in, d_out, S, 2048, stride, flops);
        run<4, 3, 1>(d_in, d_out, S, 2048, stride, flops);
        run<4, 4, 1>(d_in, d_out, S, 2048, stride, flops);
        run<4, 4, 2>(d_in, d_out, S, 2048, stride, flops);
        run<4, 4, 4>(d_in, d_out, S, 2048, stride, flops);
        run<4, 6, 1>(d_in, d_out, S, 2048, stride, flops);
        run<3, 4, 1>(d_in, d_out, S, 2730, stride, flops);
        run<6, 4, 1>(d_in, d_out, S, 1365, stride, flops);
        run<8, 3, 1>(d_in, d_out, S, 1024, stride, flops);
        run<8, 4, 1>(d_in, d_out, S, 1024, stride, flops);
    }
    return 0;
}

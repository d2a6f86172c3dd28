// Does a kernel launched with hipExtAnyOrderLaunch start while its predecessor ON THE SAME STREAM still runs (gfx950)?
// Kernel A: 256 workgroups that spin 20 us, one straggler that spins 200 us.  Kernel B: 256 workgroups that note the time they start.
// Prints B's earliest start relative to A's straggler's end (negative = overlap).   (tools only; not part of the product.)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void ka(unsigned long long *t, unsigned long long ticks_short, unsigned long long ticks_long) {
    const unsigned long long t0 = wall_clock64();
    const unsigned long long want = blockIdx.x == 0 ? ticks_long : ticks_short;
    while (wall_clock64() - t0 < want) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) t[blockIdx.x] = wall_clock64(), t[256 + blockIdx.x] = t0;
}
__global__ void kb(unsigned long long *t) {
    if (threadIdx.x == 0) t[blockIdx.x] = wall_clock64();
}
int main() {
    int rate_khz = 0;
    CHECK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const double us = 1000.0 / rate_khz;  // microseconds per tick
    unsigned long long *ta, *tb;
    CHECK(hipMalloc(&ta, 512 * 8));
    CHECK(hipMalloc(&tb, 512 * 8));
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const double shorts[3] = {5.0, 20.0, 60.0};
    for (int mode = 0; mode < 5; ++mode) {
        unsigned long long sh = (unsigned long long)(shorts[mode < 3 ? 1 : (mode == 3 ? 0 : 2)] / us), lg = (unsigned long long)(200.0 / us);
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemsetAsync(ta, 0, 512 * 8, s));
            CHECK(hipMemsetAsync(tb, 0, 512 * 8, s));
            CHECK(hipStreamSynchronize(s));
            void *aa[] = {&ta, &sh, &lg};
            void *ab[] = {&tb};
            if (mode == 0) {
                CHECK(hipLaunchKernel((const void *)ka, dim3(256), dim3(64), aa, 0, s));
                CHECK(hipLaunchKernel((const void *)kb, dim3(256), dim3(64), ab, 0, s));
            } else if (mode == 1 || mode >= 3) {
                CHECK(hipExtLaunchKernel((const void *)ka, dim3(256), dim3(64), aa, 0, s, nullptr, nullptr, hipExtAnyOrderLaunch));
                CHECK(hipExtLaunchKernel((const void *)kb, dim3(256), dim3(64), ab, 0, s, nullptr, nullptr, hipExtAnyOrderLaunch));
            } else {
                CHECK(hipExtLaunchKernel((const void *)ka, dim3(256), dim3(64), aa, 0, s, nullptr, nullptr, 0));
                CHECK(hipExtLaunchKernel((const void *)kb, dim3(256), dim3(64), ab, 0, s, nullptr, nullptr, hipExtAnyOrderLaunch));
            }
            CHECK(hipStreamSynchronize(s));
            unsigned long long ha[512], hb[256];
            CHECK(hipMemcpy(ha, ta, sizeof ha, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(hb, tb, sizeof hb, hipMemcpyDeviceToHost));
            unsigned long long a_end = 0, a_short_end = 0, b_first = ~0ull, a_first = ~0ull;
            for (int i = 0; i < 256; ++i) {
                if (ha[256 + i] < a_first) a_first = ha[256 + i];
                if (ha[i] > a_end) a_end = ha[i];
                if (i && ha[i] > a_short_end) a_short_end = ha[i];
                if (hb[i] < b_first) b_first = hb[i];
            }
            printf("%s: B's first workgroup starts %+8.1f us after A's LAST workgroup ends (A's short ones ended %.1f us before its straggler; B's first starts %.1f us after A's first)\n",
                   mode == 0 ? "hipLaunchKernel, hipLaunchKernel       " : mode == 1 ? "AnyOrder, AnyOrder                     " : mode == 2 ? "hipExtLaunchKernel(0), AnyOrder         " : mode == 3 ? "AnyOrder x2, A's short groups 5 us     " : "AnyOrder x2, A's short groups 60 us    ",
                   ((double)b_first - (double)a_end) * us, ((double)a_end - (double)a_short_end) * us, ((double)b_first - (double)a_first) * us);
        }
    }
    return 0;
}

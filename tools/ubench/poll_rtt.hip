// How fast does a consumer wave see a word another workgroup publishes, and does a poll have to travel in vmcnt?
//   method 0: global_load_dword ... sc1 + s_waitcnt vmcnt(0)         (what the scan kernels do today)
//   method 1: s_load_dword ... glc     + s_waitcnt lgkmcnt(0)         (scalar path: not behind LDS-DMA / stores in vmcnt)
//   method 2: s_dcache_inv ; s_load_dword + s_waitcnt lgkmcnt(0)
// Producer: one wave publishes 1, 2, 3, ... every ~3 us (agent-scope store) and notes the 100 MHz clock at each store;
// consumers (one wave per method, other CUs) note the clock when they first see each value.  Also: the round trip of one
// poll of a word that is already there (1000 polls back to back).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/poll_rtt.hip -o tools/ubench/poll_rtt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kN = 200;

__device__ __forceinline__ uint32_t poll_v(const uint32_t *p) {
    uint32_t r;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ uint32_t poll_s_glc(const uint32_t *p) {
    uint32_t r;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    return r;
}
__device__ __forceinline__ uint32_t poll_s_inv(const uint32_t *p) {
    uint32_t r;
    asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    return r;
}
template <int M>
__device__ __forceinline__ uint32_t poll(const uint32_t *p) {
    return M == 0 ? poll_v(p) : (M == 1 ? poll_s_glc(p) : poll_s_inv(p));
}

// block 0: producer; blocks 1..3: consumer with method block-1 (launched with many filler blocks in between so that they sit on other CUs)
__global__ __launch_bounds__(64) void k_handoff(uint32_t *flag, unsigned long long *t_store, unsigned long long *t_seen, uint32_t *gave_up, int stride_blocks) {
    const int b = blockIdx.x;
    if (b == 0) {
        for (int i = 1; i <= kN; ++i) {
            const unsigned long long t0 = wall_clock64();
            while (wall_clock64() - t0 < 300) __builtin_amdgcn_s_sleep(8);  // 3 us at 100 MHz
            if (threadIdx.x == 0) {
                t_store[i] = wall_clock64();
                __hip_atomic_store(flag, (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    if (b % stride_blocks != 0 || b / stride_blocks < 1 || b / stride_blocks > 3) return;
    const int m = b / stride_blocks - 1;
    for (int i = 1; i <= kN; ++i) {
        uint32_t v = 0, spins = 0;
        while (true) {
            v = m == 0 ? poll<0>(flag) : (m == 1 ? poll<1>(flag) : poll<2>(flag));
            if (v >= (uint32_t)i) break;
            if (++spins > 2000000u) {
                if (threadIdx.x == 0) gave_up[m] = (uint32_t)i;
                return;
            }
        }
        if (threadIdx.x == 0) t_seen[m * (kN + 1) + i] = wall_clock64();
    }
}
template <int M>
__global__ __launch_bounds__(64) void k_rtt(const uint32_t *flag, unsigned long long *out, uint32_t *sink) {
    uint32_t acc = 0;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < 1000; ++i) acc += poll<M>(flag);
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[M] = t1 - t0;
        sink[0] = acc;
    }
}

int main() {
    uint32_t *flag, *gave, *sink;
    unsigned long long *ts, *tn, *rt;
    CK(hipMalloc(&flag, 256));
    CK(hipMalloc(&gave, 64));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&ts, sizeof(unsigned long long) * (kN + 1)));
    CK(hipMalloc(&tn, sizeof(unsigned long long) * 3 * (kN + 1)));
    CK(hipMalloc(&rt, 64));
    CK(hipMemset(flag, 0, 256));
    CK(hipMemset(gave, 0, 64));
    CK(hipMemset(tn, 0, sizeof(unsigned long long) * 3 * (kN + 1)));
    CK(hipDeviceSynchronize());
    const int stride = 37;  // consumers in blocks 37, 74, 111: other CUs / XCDs than the producer
    hipLaunchKernelGGL(k_handoff, dim3(4 * stride), dim3(64), 0, 0, flag, ts, tn, gave, stride);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hs(kN + 1), hn(3 * (kN + 1));
    uint32_t hg[3];
    CK(hipMemcpy(hs.data(), ts, sizeof(unsigned long long) * (kN + 1), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hn.data(), tn, sizeof(unsigned long long) * 3 * (kN + 1), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hg, gave, sizeof(hg), hipMemcpyDeviceToHost));
    const char *names[3] = {"global_load sc1 (vmcnt)", "s_load glc (lgkmcnt)", "s_dcache_inv + s_load"};
    for (int m = 0; m < 3; ++m) {
        if (hg[m]) {
            printf("%-28s NEVER saw value %u (stale reads): not coherent\n", names[m], hg[m]);
            continue;
        }
        double sum = 0, mx = 0;
        int n = 0;
        for (int i = 20; i <= kN; ++i) {  // skip the warm-up
            const double d = ((double)hn[m * (kN + 1) + i] - (double)hs[i]) * 10.0;  // ns
            sum += d;
            mx = d > mx ? d : mx;
            ++n;
        }
        printf("%-28s store -> seen: mean %.0f ns, max %.0f ns over %d hand-offs\n", names[m], sum / n, mx, n);
    }
    CK(hipMemset(flag, 0, 4));
    hipLaunchKernelGGL(k_rtt<0>, dim3(1), dim3(64), 0, 0, flag, rt, sink);
    hipLaunchKernelGGL(k_rtt<1>, dim3(1), dim3(64), 0, 0, flag, rt, sink);
    hipLaunchKernelGGL(k_rtt<2>, dim3(1), dim3(64), 0, 0, flag, rt, sink);
    CK(hipDeviceSynchronize());
    unsigned long long hr[3];
    CK(hipMemcpy(hr, rt, sizeof(hr), hipMemcpyDeviceToHost));
    for (int m = 0; m < 3; ++m) printf("%-28s round trip of one poll (idle chip): %.0f ns\n", names[m], (double)hr[m] * 10.0 / 1000.0);
    return 0;
}

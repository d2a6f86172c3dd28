// div_check.hip -- is the three-instruction division of the lerp (q0 = t * rcpT; rem = fma(-q0, T, t); q = fma(rem, rcpT, q0): Markstein, with
// rcpT = RN(1 / T) from the host) the IEEE quotient t / T that math.rs:25 computes?  EXHAUSTIVELY: all 2^32 bit patterns of t, for the
// denominators given on the command line (reduced `to` rates: 160 for 44.1 -> 48 kHz, ...).  Prints the mismatches by exponent of t.
//     hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/ubench/div_check tools/ubench/div_check.hip && tools/ubench/div_check 160 147 6 1 3 48000 44100 7
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

// FORM 0: the bare three instructions; 1: + a class test and a select (zeros and infinities take q0: what rh_common.h's div_lerp ships);
// 2: + v_div_fixup_f32, the instruction the IEEE sequence itself ends with
template <int FORM>
__global__ void k_check(float Tf, float rcpT, unsigned long long *bad_by_exp, unsigned long long *bad_zero_sign) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
        const float t = __uint_as_float((uint32_t)b);
        const float q0 = t * rcpT;
        const float rem = __builtin_fmaf(-q0, Tf, t);
        float q = __builtin_fmaf(rem, rcpT, q0);
        if (FORM == 1) q = __builtin_amdgcn_classf(t, 0x264) ? q0 : q;
        if (FORM == 2) q = __builtin_amdgcn_div_fixupf(q, Tf, t);
        const float ref = t / Tf;
        const bool both_nan = (q != q) && (ref != ref);
        if (!both_nan && __float_as_uint(q) != __float_as_uint(ref)) {
            if (q == ref) atomicAdd(bad_zero_sign, 1ull);  // +0 against -0
            else atomicAdd(&bad_by_exp[((uint32_t)b >> 23) & 0xff], 1ull);
        }
    }
}

int main(int argc, char **argv) {
    unsigned long long *d;
    hipMalloc(&d, 257 * 8);
    for (int a = 1; a < argc; ++a) {
        const uint32_t T = (uint32_t)std::strtoul(argv[a], nullptr, 10);
        const float Tf = (float)T, rcpT = 1.0f / Tf;
      for (int form = 0; form < 3; ++form) {
        hipMemset(d, 0, 257 * 8);
        if (form == 0) hipLaunchKernelGGL(k_check<0>, dim3(256 * 32), dim3(256), 0, 0, Tf, rcpT, d, d + 256);
        else if (form == 1) hipLaunchKernelGGL(k_check<1>, dim3(256 * 32), dim3(256), 0, 0, Tf, rcpT, d, d + 256);
        else hipLaunchKernelGGL(k_check<2>, dim3(256 * 32), dim3(256), 0, 0, Tf, rcpT, d, d + 256);
        unsigned long long h[257];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        unsigned long long total = 0;
        int lo = -1, hi = -1;
        for (int e = 0; e < 256; ++e)
            if (h[e]) {
                total += h[e];
                if (lo < 0) lo = e;
                hi = e;
            }
        std::printf("T = %u, form %d: %llu mismatching values of t (biased exponents %d .. %d), %llu that differ in the sign of a zero\n", T, form, total, lo, hi, h[256]);
        if (total) {
            std::printf("   by biased exponent:");
            for (int e = 0; e < 256; ++e)
                if (h[e]) std::printf(" %d:%llu", e, h[e]);
            std::printf("\n");
        }
      }
    }
    return 0;
}

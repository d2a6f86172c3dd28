// rh_formats.hip -- the rest of cpal's device sample formats, either side of the path:
// egress  f32 -> {u8, I24, U24, u32, i64, u64, f64}   src/stream.rs:538-545,555-568 (`Sample::from_sample`)
// ingress {U24, u32, i64, u64, f64} -> f32            src/microphone.rs:280-291
// (i8/u8/i16/u16/I24/i32 -> f32 and f32 -> i8/i16/u16/i32 live in rh_elementwise.hip.)
// Arithmetic: dasp_sample 0.11.0 conv.rs (Cargo.lock:317-318; un-vendored, restated -- parity unpinned):
// float -> signed is `(s * 2^(bits-1)) as iN` (Rust `as`: truncate, saturate, NaN -> 0), unsigned goes
// THROUGH the signed type (`iN::to_uN`: add 2^(bits-1) with wrap-around), I24/U24 are unchecked i32
// containers, signed -> float is `s as f32 / 2^(bits-1)`.
// Streaming kernels, HBM-bound: one element per lane per grid-stride step (the 8-byte types move
// 512 B per wave-instruction).
#include "rh_common.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int32_t f32_as_i32(float v) {  // Rust `v as i32`
    if (v != v) return 0;
    if (v <= -2147483648.0f) return INT32_MIN;
    if (v >= 2147483648.0f) return INT32_MAX;
    return (int32_t)v;
}
__device__ __forceinline__ int64_t f32_as_i64(float v) {  // Rust `v as i64`
    if (v != v) return 0;
    if (v <= -9223372036854775808.0f) return INT64_MIN;
    if (v >= 9223372036854775808.0f) return INT64_MAX;
    return (int64_t)v;
}
__device__ __forceinline__ int32_t f32_as_i8(float v) {
    if (v != v) return 0;
    if (v <= -128.0f) return -128;
    if (v >= 128.0f) return 127;
    return (int32_t)v;
}

struct F32ToU8 { typedef float In; typedef uint8_t Out; static __device__ __forceinline__ Out cvt(In s) { return (uint8_t)(f32_as_i8(s * 128.0f) + 128); } };
struct F32ToI24 { typedef float In; typedef int32_t Out; static __device__ __forceinline__ Out cvt(In s) { return f32_as_i32(s * 8388608.0f); } };
struct F32ToU24 { typedef float In; typedef int32_t Out; static __device__ __forceinline__ Out cvt(In s) { return (int32_t)((uint32_t)f32_as_i32(s * 8388608.0f) + 8388608u); } };
struct F32ToU32 { typedef float In; typedef uint32_t Out; static __device__ __forceinline__ Out cvt(In s) { return (uint32_t)f32_as_i32(s * 2147483648.0f) + 0x80000000u; } };
struct F32ToI64 { typedef float In; typedef int64_t Out; static __device__ __forceinline__ Out cvt(In s) { return f32_as_i64(s * 9223372036854775808.0f); } };
struct F32ToU64 { typedef float In; typedef uint64_t Out; static __device__ __forceinline__ Out cvt(In s) { return (uint64_t)f32_as_i64(s * 9223372036854775808.0f) + 0x8000000000000000ull; } };
struct F32ToF64 { typedef float In; typedef double Out; static __device__ __forceinline__ Out cvt(In s) { return (double)s; } };
struct U24ToF32 { typedef int32_t In; typedef float Out; static __device__ __forceinline__ Out cvt(In s) { return (float)(s - 8388608) / 8388608.0f; } };
struct U32ToF32 { typedef uint32_t In; typedef float Out; static __device__ __forceinline__ Out cvt(In s) { return (float)(int32_t)(s - 0x80000000u) / 2147483648.0f; } };
struct I64ToF32 { typedef int64_t In; typedef float Out; static __device__ __forceinline__ Out cvt(In s) { return (float)s / 9223372036854775808.0f; } };
struct U64ToF32 { typedef uint64_t In; typedef float Out; static __device__ __forceinline__ Out cvt(In s) { return (float)(int64_t)(s - 0x8000000000000000ull) / 9223372036854775808.0f; } };
struct F64ToF32 { typedef double In; typedef float Out; static __device__ __forceinline__ Out cvt(In s) { return (float)s; } };

template <typename Op>
__global__ __launch_bounds__(kBlock) void k_convert(typename Op::Out *__restrict__ dst, const typename Op::In *__restrict__ src, size_t n) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) dst[i] = Op::cvt(src[i]);
}
template <typename Op>
rh_status launch(typename Op::Out *dst, const typename Op::In *src, size_t n, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (n == 0) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    hipLaunchKernelGGL(k_convert<Op>, dim3(rh::grid_for(n)), dim3(kBlock), 0, rh::as_stream(stream), dst, src, n);
    RH_CHECK_LAUNCH();
    return RH_OK;
}

}  // namespace

extern "C" {
rh_status rh_convert_f32_to_u8(uint8_t *dst, const float *src, size_t n, rh_stream s) { return launch<F32ToU8>(dst, src, n, s); }
rh_status rh_convert_f32_to_i24(int32_t *dst, const float *src, size_t n, rh_stream s) { return launch<F32ToI24>(dst, src, n, s); }
rh_status rh_convert_f32_to_u24(int32_t *dst, const float *src, size_t n, rh_stream s) { return launch<F32ToU24>(dst, src, n, s); }
rh_status rh_convert_f32_to_u32(uint32_t *dst, const float *src, size_t n, rh_stream s) { return launch<F32ToU32>(dst, src, n, s); }
rh_status rh_convert_f32_to_i64(int64_t *dst, const float *src, size_t n, rh_stream s) { return launch<F32ToI64>(dst, src, n, s); }
rh_status rh_convert_f32_to_u64(uint64_t *dst, const float *src, size_t n, rh_stream s) { return launch<F32ToU64>(dst, src, n, s); }
rh_status rh_convert_f32_to_f64(double *dst, const float *src, size_t n, rh_stream s) { return launch<F32ToF64>(dst, src, n, s); }
rh_status rh_convert_u24_to_f32(float *dst, const int32_t *src, size_t n, rh_stream s) { return launch<U24ToF32>(dst, src, n, s); }
rh_status rh_convert_u32_to_f32(float *dst, const uint32_t *src, size_t n, rh_stream s) { return launch<U32ToF32>(dst, src, n, s); }
rh_status rh_convert_i64_to_f32(float *dst, const int64_t *src, size_t n, rh_stream s) { return launch<I64ToF32>(dst, src, n, s); }
rh_status rh_convert_u64_to_f32(float *dst, const uint64_t *src, size_t n, rh_stream s) { return launch<U64ToF32>(dst, src, n, s); }
rh_status rh_convert_f64_to_f32(float *dst, const double *src, size_t n, rh_stream s) { return launch<F64ToF32>(dst, src, n, s); }
}

// rh_common.h -- shared host-side helpers of librodio_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../include/rodio_hip.h"

namespace rh {

// Set by rh_init(); every entry point refuses to run before it (no CPU fallback exists).
extern bool g_initialized;
extern int g_device;
extern int g_num_cus;
void set_hip_error(hipError_t e, const char *what);

#define RH_HIP_TRY(expr)                                   \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) {                            \
            ::rh::set_hip_error(_e, #expr);                \
            return RH_ERR_HIP;                             \
        }                                                  \
    } while (0)

#define RH_REQUIRE_INIT()                                  \
    do {                                                   \
        if (!::rh::g_initialized) return RH_ERR_NOT_INITIALIZED; \
    } while (0)

#define RH_CHECK_LAUNCH()                                  \
    do {                                                   \
        hipError_t _e = hipGetLastError();                 \
        if (_e != hipSuccess) {                            \
            ::rh::set_hip_error(_e, "kernel launch");      \
            return RH_ERR_HIP;                             \
        }                                                  \
    } while (0)

inline hipStream_t as_stream(rh_stream s) { return reinterpret_cast<hipStream_t>(s); }

// Grid for a memory-bound grid-stride kernel: enough 256-thread blocks to fill 256 CUs x 8,
// capped so small inputs stay small (cdna_hip_programming.md G11).
inline unsigned grid_for(size_t work_items, unsigned block = 256, unsigned max_blocks = 256 * 8) {
    size_t b = (work_items + block - 1) / block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return static_cast<unsigned>(b);
}

}  // namespace rh

// rh_runtime.hip -- device bring-up, memory/stream/event helpers of the C ABI.
#include <cstring>
#include <string>

#include "rh_common.h"

namespace rh {
bool g_initialized = false;
int g_device = -1;
int g_num_cus = 256;
static thread_local std::string g_last_error;
void set_hip_error(hipError_t e, const char *what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
}
}  // namespace rh

extern "C" {

int32_t rh_version(void) { return 100; /* 0.1.0 */ }

const char *rh_status_string(rh_status s) {
    switch (s) {
        case RH_OK: return "ok";
        case RH_ERR_INVALID: return "invalid argument";
        case RH_ERR_HIP: return "HIP runtime error";
        case RH_ERR_UNSUPPORTED: return "unsupported configuration";
        case RH_ERR_NOMEM: return "out of memory";
        case RH_ERR_TIMEOUT: return "in-kernel wait timed out";
        case RH_ERR_NOT_INITIALIZED: return "rh_init() has not succeeded (no HIP device: there is no CPU fallback)";
        case RH_ERR_CAPACITY: return "output buffer too small";
        default: return "unknown status";
    }
}

const char *rh_last_hip_error(void) { return rh::g_last_error.c_str(); }

rh_status rh_init(int32_t device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        rh::set_hip_error(e == hipSuccess ? hipErrorNoDevice : e, "hipGetDeviceCount");
        rh::g_initialized = false;
        return RH_ERR_HIP;
    }
    if (device < 0 || device >= count) return RH_ERR_INVALID;
    RH_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    RH_HIP_TRY(hipGetDeviceProperties(&prop, device));
    // This library carries gfx950 code objects only.
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        rh::g_last_error = std::string("device is ") + prop.gcnArchName + ", this build targets gfx950 only";
        return RH_ERR_UNSUPPORTED;
    }
    rh::g_num_cus = prop.multiProcessorCount;
    rh::g_device = device;
    rh::g_initialized = true;
    return RH_OK;
}

rh_status rh_device_name(char *buf, size_t cap) {
    RH_REQUIRE_INIT();
    hipDeviceProp_t prop;
    RH_HIP_TRY(hipGetDeviceProperties(&prop, rh::g_device));
    std::snprintf(buf, cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return RH_OK;
}

rh_status rh_malloc(void **out, size_t bytes) {
    RH_REQUIRE_INIT();
    if (!out) return RH_ERR_INVALID;
    hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e == hipErrorOutOfMemory) return RH_ERR_NOMEM;
    RH_HIP_TRY(e);
    return RH_OK;
}
rh_status rh_free(void *p) {
    RH_REQUIRE_INIT();
    RH_HIP_TRY(hipFree(p));
    return RH_OK;
}
rh_status rh_memset(void *p, int32_t value, size_t bytes, rh_stream stream) {
    RH_REQUIRE_INIT();
    RH_HIP_TRY(hipMemsetAsync(p, value, bytes, rh::as_stream(stream)));
    return RH_OK;
}
rh_status rh_memcpy_h2d(void *dst, const void *src_host, size_t bytes, rh_stream stream) {
    RH_REQUIRE_INIT();
    RH_HIP_TRY(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, rh::as_stream(stream)));
    return RH_OK;
}
rh_status rh_memcpy_d2h(void *dst_host, const void *src, size_t bytes, rh_stream stream) {
    RH_REQUIRE_INIT();
    RH_HIP_TRY(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, rh::as_stream(stream)));
    RH_HIP_TRY(hipStreamSynchronize(rh::as_stream(stream)));
    return RH_OK;
}
rh_status rh_memcpy_d2h_async(void *dst_host, const void *src, size_t bytes, rh_stream stream) {
    RH_REQUIRE_INIT();
    RH_HIP_TRY(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, rh::as_stream(stream)));
    return RH_OK;
}
rh_status rh_memcpy_d2d(void *dst, const void *src, size_t bytes, rh_stream stream) {
    RH_REQUIRE_INIT();
    RH_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, rh::as_stream(stream)));
    return RH_OK;
}
rh_status rh_host_alloc(void **out, size_t bytes) {
    RH_REQUIRE_INIT();
    if (!out) return RH_ERR_INVALID;
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) return RH_ERR_NOMEM;
    RH_HIP_TRY(e);
    return RH_OK;
}
rh_status rh_host_free(void *p) {
    RH_REQUIRE_INIT();
    RH_HIP_TRY(hipHostFree(p));
    return RH_OK;
}
rh_status rh_stream_create(rh_stream *out) {
    RH_REQUIRE_INIT();
    hipStream_t s;
    RH_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = reinterpret_cast<rh_stream>(s);
    return RH_OK;
}
rh_status rh_stream_destroy(rh_stream s) {
    RH_REQUIRE_INIT();
    RH_HIP_TRY(hipStreamDestroy(rh::as_stream(s)));
    return RH_OK;
}
rh_status rh_stream_synchronize(rh_stream s) {
    RH_REQUIRE_INIT();
    RH_HIP_TRY(hipStreamSynchronize(rh::as_stream(s)));
    return RH_OK;
}
rh_status rh_event_create(void **out) {
    RH_REQUIRE_INIT();
    hipEvent_t ev;
    RH_HIP_TRY(hipEventCreate(&ev));
    *out = reinterpret_cast<void *>(ev);
    return RH_OK;
}
rh_status rh_event_destroy(void *ev) {
    RH_REQUIRE_INIT();
    RH_HIP_TRY(hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)));
    return RH_OK;
}
rh_status rh_event_record(void *ev, rh_stream stream) {
    RH_REQUIRE_INIT();
    RH_HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), rh::as_stream(stream)));
    return RH_OK;
}
rh_status rh_event_synchronize(void *ev) {
    RH_REQUIRE_INIT();
    RH_HIP_TRY(hipEventSynchronize(reinterpret_cast<hipEvent_t>(ev)));
    return RH_OK;
}
rh_status rh_event_elapsed_ms(void *start, void *stop, float *ms) {
    RH_REQUIRE_INIT();
    RH_HIP_TRY(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
    RH_HIP_TRY(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return RH_OK;
}

}  // extern "C"

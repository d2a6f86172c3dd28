// rh_runtime.hip -- device bring-up, memory/stream/event helpers of the C ABI.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>

#include "rh_common.h"

namespace rh {
bool g_initialized = false;
int g_device = -1;
int g_num_cus = 256;
uint32_t *g_async_status = nullptr;
static thread_local std::string g_last_error;
void set_hip_error(hipError_t e, const char *what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
}

namespace {
const char *const kKnobNames[K_COUNT] = {"RH_AGC_SEQ", "RH_AGC_VEC", "RH_BIQUAD_NO_FALLBACK", "RH_BIQUAD_SEQ", "RH_BIQUAD_R", "RH_BIQUAD_NW", "RH_BIQUAD_WGS", "RH_LIMIT_SEQ",
                                         "RH_LIMIT_R", "RH_LIMIT_NW", "RH_LIMIT_WGS", "RH_LIMIT_GRID", "RH_LIMIT_SKEW", "RH_LIMIT_NIO", "RH_LIMIT_INIT", "RH_SCAN_DMA_TOP", "RH_SCAN_SPIN_LIMIT", "RH_NO_HYBRID",
                                         "RH_NO_TICKET_SHARDS", "RH_PROF_DUMP", "RH_HOST_ALLOC", "RH_NO_MIX_FIRST", "RH_MIX_U", "RH_NO_CHUNK", "RH_CHUNK_HALF", "RH_AUTOTUNE_LOG", "RH_RAG_RESIDENT", "RH_RAG_TWO_KERNELS", "RH_AGC_SEGMENTS", "RH_RS_PIPE", "RH_DASP_I64_VIA_F64", "RH_MIX_GROUPS", "RH_CLASSES_SIDE_BY_SIDE", "RH_AGC_FUSED_R4", "RH_STREAM_UPLOAD_ALWAYS", "RH_STREAM_NO_REJOIN", "RH_NO_SBLK", "RH_SBLK_KV", "RH_SBLK_NO_OVERLAP", "RH_CLASSES_ONE_BY_ONE", "RH_CLASSES_ONE_WAVE", "RH_WIDE_GENERAL", "RH_PCM_NO_TILE", "RH_PCM_TILE_KB", "RH_LERP_IEEE_DIV"};
std::string g_knob_val[K_COUNT];
bool g_knob_set[K_COUNT];
}  // namespace
void load_knobs() {
    for (int k = 0; k < K_COUNT; ++k) {
        const char *v = getenv(kKnobNames[k]);
        g_knob_set[k] = v != nullptr;
        g_knob_val[k] = v ? v : "";
    }
}
const char *knob(Knob k) { return g_knob_set[k] ? g_knob_val[k].c_str() : nullptr; }
namespace {
// bytes [0, n) of p: the unaligned head and tail byte by byte, the 4-byte aligned body word by word
__global__ void k_fill(unsigned char *p, uint32_t word, size_t n) {
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    const size_t head = ((4u - (reinterpret_cast<uintptr_t>(p) & 3u)) & 3u) < n ? ((4u - (reinterpret_cast<uintptr_t>(p) & 3u)) & 3u) : n;
    const size_t words = (n - head) / 4, tail0 = head + words * 4;
    uint32_t *w = reinterpret_cast<uint32_t *>(p + head);
    for (size_t i = i0; i < words; i += step) w[i] = word;
    if (i0 < head) p[i0] = (unsigned char)word;
    if (i0 < n - tail0) p[tail0 + i0] = (unsigned char)word;
}
}  // namespace
namespace {
// every mantissa, both signs, three binades inside div_exact's short-path domain: the short path against the IEEE division
__global__ void k_lerp_div_check(float Tf, float rcpT, uint32_t *bad) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;  // 2^23 mantissas
    const uint32_t exps[3] = {32u, 127u, 231u};
    uint32_t miss = 0;
    for (int k = 0; k < 3; ++k)
        for (uint32_t sgn = 0; sgn < 2; ++sgn) {
            const float t = __uint_as_float((sgn << 31) | (exps[k] << 23) | m);
            const float q0 = t * rcpT;
            const float rem = __builtin_fmaf(-q0, Tf, t);
            const float q = __builtin_fmaf(rem, rcpT, q0);
            miss |= __float_as_uint(q) != __float_as_uint(t / Tf);
        }
    if (miss) atomicOr(bad, 1u);
}
std::mutex g_div_mu;
std::unordered_map<uint32_t, bool> g_div_ok;
uint32_t *g_div_flag = nullptr;
}  // namespace
bool lerp_div_fast_ok(uint32_t T) {
    if (T == 0 || knob(K_LERP_IEEE_DIV) || !g_initialized) return false;
    std::lock_guard<std::mutex> hold(g_div_mu);
    auto it = g_div_ok.find(T);
    if (it != g_div_ok.end()) return it->second;
    bool ok = false;
    do {  // (any HIP failure: not verified, the IEEE division)
        if (!g_div_flag && hipMalloc(reinterpret_cast<void **>(&g_div_flag), 4) != hipSuccess) break;
        if (fill_now(g_div_flag, 0, 4) != hipSuccess) break;
        const float Tf = (float)T;
        hipLaunchKernelGGL(k_lerp_div_check, dim3((1u << 23) / 256), dim3(256), 0, nullptr, Tf, 1.0f / Tf, g_div_flag);
        uint32_t bad = 1;
        if (hipMemcpy(&bad, g_div_flag, 4, hipMemcpyDeviceToHost) != hipSuccess) break;
        ok = bad == 0;
    } while (false);
    g_div_ok[T] = ok;
    return ok;
}
namespace {
struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
    ScratchAux aux{0, 0, 0};
};
std::mutex g_scratch_mu;
std::unordered_map<hipStream_t, Scratch> g_scratch;
}  // namespace
hipError_t stream_scratch(hipStream_t s, size_t bytes, void **out, std::unique_lock<std::mutex> &hold, ScratchAux **aux) {
    hold = std::unique_lock<std::mutex>(g_scratch_mu);
    Scratch &e = g_scratch[s];
    if (aux) *aux = &e.aux;
    else e.aux = ScratchAux{0, 0, 0};  // this caller writes over whatever the last one left
    if (e.cap < bytes) {
        e.aux = ScratchAux{0, 0, 0};
        if (e.p) {  // launches on `s` may still be using it
            hipError_t w = hipStreamSynchronize(s);
            if (w != hipSuccess) return w;
            (void)hipFree(e.p);
            e.p = nullptr;
            e.cap = 0;
        }
        size_t cap = size_t(1) << 20;
        while (cap < bytes) cap *= 2;
        hipError_t m = hipMalloc(&e.p, cap);
        if (m != hipSuccess) {
            e.p = nullptr;
            return m;
        }
        e.cap = cap;
    }
    *out = e.p;
    return hipSuccess;
}
// A scan kernel reported a lost hand-off: whatever its launch left in the streams' hand-off tables is not to be trusted -- the next launch of
// every stream writes its tables afresh (the "clean" shortcut of rh_limit / rh_biquad mode 1 trusts the launch before it: ADVICE r4).
static void distrust_stream_scratch() {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    for (auto &kv : g_scratch) kv.second.aux = ScratchAux{0, 0, 0};
}
static void drop_stream_scratch(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    auto it = g_scratch.find(s);
    if (it == g_scratch.end()) return;
    if (it->second.p) (void)hipFree(it->second.p);
    g_scratch.erase(it);
}
hipError_t fill_async(void *p, int value, size_t bytes, hipStream_t s) {
    if (!bytes) return hipSuccess;
    const uint32_t b = (uint32_t)value & 0xffu;
    hipLaunchKernelGGL(k_fill, dim3(grid_tiles(bytes / 16 + 1)), dim3(256), 0, s, static_cast<unsigned char *>(p), b * 0x01010101u, bytes);
    return hipGetLastError();
}
}  // namespace rh

namespace {
// A device copy as a kernel: hipMemcpyAsync(DeviceToDevice) makes the calling thread wait for the work queued before it
// (measured: the shim's submit path spent 2.7 ms per block there), a launch does not.  W = the widest word both pointers and the
// length are aligned to.
template <typename W>
__global__ void k_copy(W *__restrict__ dst, const W *__restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
template <typename W>
void launch_copy(void *dst, const void *src, size_t bytes, hipStream_t s) {
    const size_t n = bytes / sizeof(W);
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, (size_t)rh::g_num_cus * 16);
    hipLaunchKernelGGL(k_copy<W>, dim3(blocks ? blocks : 1), dim3(256), 0, s, static_cast<W *>(dst), static_cast<const W *>(src), n);
}
}  // namespace
namespace rh {
hipError_t copy_d2d(void *dst, const void *src, size_t bytes, hipStream_t hs) {
    if (!bytes) return hipSuccess;
    const uintptr_t a = reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | (uintptr_t)bytes;
    if ((a & 15u) == 0) launch_copy<uint4>(dst, src, bytes, hs);
    else if ((a & 3u) == 0) launch_copy<uint32_t>(dst, src, bytes, hs);
    else launch_copy<unsigned char>(dst, src, bytes, hs);
    return hipGetLastError();
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
    rh::load_knobs();
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
    {  // the sticky failure word of the handle-less scan kernels lives on the device it reports about: one per device ever bound
        static uint32_t *per_device[64] = {nullptr};
        if (device >= 64) return RH_ERR_UNSUPPORTED;
        if (!per_device[device]) {
            RH_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&per_device[device]), 128));
            RH_HIP_TRY(rh::fill_now(per_device[device], 0, 128));
        }
        rh::g_async_status = per_device[device];
    }
    rh::g_initialized = true;
    return RH_OK;
}

rh_status rh_async_status(void) {
    RH_REQUIRE_INIT();
    uint32_t v = 0;
    RH_HIP_TRY(hipMemcpy(&v, rh::g_async_status, sizeof(v), hipMemcpyDeviceToHost));
    if (!v) return RH_OK;
    RH_HIP_TRY(rh::fill_now(rh::g_async_status, 0, sizeof(v)));
    rh::distrust_stream_scratch();
    return RH_ERR_TIMEOUT;
}

rh_status rh_bind_thread(void) {
    RH_REQUIRE_INIT();
    RH_HIP_TRY(hipSetDevice(rh::g_device));
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
    if (!bytes) return RH_OK;
    if (!p) return RH_ERR_INVALID;
    RH_HIP_TRY(rh::fill_async(p, value, bytes, rh::as_stream(stream)));
    return RH_OK;
}
rh_status rh_memcpy_h2d(void *dst, const void *src_host, size_t bytes, rh_stream stream) {
    RH_REQUIRE_INIT();
    RH_HIP_TRY(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, rh::as_stream(stream)));
    return RH_OK;
}
rh_status rh_memcpy_h2d_rows(void *dst, const void *src_host, size_t pitch_bytes, size_t width_bytes, size_t rows, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (!rows || !width_bytes) return RH_OK;
    if (!dst || !src_host || width_bytes > pitch_bytes) return RH_ERR_INVALID;
    RH_HIP_TRY(hipMemcpy2DAsync(dst, pitch_bytes, src_host, pitch_bytes, width_bytes, rows, hipMemcpyHostToDevice, rh::as_stream(stream)));
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
    if (!bytes) return RH_OK;
    if (!dst || !src) return RH_ERR_INVALID;
    RH_HIP_TRY(rh::copy_d2d(dst, src, bytes, rh::as_stream(stream)));
    return RH_OK;
}
rh_status rh_host_alloc(void **out, size_t bytes) {
    RH_REQUIRE_INIT();
    if (!out) return RH_ERR_INVALID;
    unsigned flags = hipHostMallocDefault;
    if (const char *k = rh::knob(rh::K_HOST_ALLOC)) flags = (unsigned)std::strtoul(k, nullptr, 0);
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, flags);
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
    RH_HIP_TRY(hipStreamSynchronize(rh::as_stream(s)));
    rh::drop_stream_scratch(rh::as_stream(s));
    rh::rlm_stream_retired(rh::as_stream(s));
    RH_HIP_TRY(hipStreamDestroy(rh::as_stream(s)));
    return RH_OK;
}
rh_status rh_stream_release_scratch(rh_stream s) {
    RH_REQUIRE_INIT();
    RH_HIP_TRY(hipStreamSynchronize(rh::as_stream(s)));
    rh::drop_stream_scratch(rh::as_stream(s));
    rh::rlm_stream_retired(rh::as_stream(s));
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
rh_status rh_stream_wait_event(rh_stream stream, void *ev) {
    RH_REQUIRE_INIT();
    if (!ev) return RH_ERR_INVALID;
    RH_HIP_TRY(hipStreamWaitEvent(rh::as_stream(stream), reinterpret_cast<hipEvent_t>(ev), 0));
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

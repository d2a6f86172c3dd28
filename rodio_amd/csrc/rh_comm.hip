// rh_comm.hip -- the one collective of the path: the mixer sum across the source shards of
// the ranks, `ncclAllReduce(sum, f32)` of the mixed block over RCCL / xGMI, behind plain C entry points so
// that a host without PyTorch (the Rust shim) can run one process per GPU.  rodio itself has no
// collective: streams only meet in mixer.rs:185-198, which is what the all-reduce completes.
// RCCL is loaded lazily (dlopen) the first time a communicator is made: single-GPU users never touch it,
// and inside a PyTorch process the already loaded librccl is the one that gets used.
#include <dlfcn.h>

#include <cstring>

#include "rh_common.h"

namespace {

struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, /* ncclUniqueId by value: 128 bytes */ struct Uid, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Reduce)(const void *, void *, size_t, int, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
struct Uid {
    char internal[128];  // NCCL_UNIQUE_ID_BYTES
};
Rccl g_rccl;
constexpr int kNcclFloat32 = 7, kNcclSum = 0;  // rccl.h: ncclDataType_t / ncclRedOp_t

bool load_rccl() {
    if (g_rccl.h) return true;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
    Rccl r;
    r.h = h;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.Reduce = reinterpret_cast<decltype(r.Reduce)>(dlsym(h, "ncclReduce"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.Reduce || !r.CommDestroy) return false;
    g_rccl = r;
    return true;
}
rh_status nccl_fail(int rc, const char *what) {
    static char msg[256];
    snprintf(msg, sizeof(msg), "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
    rh::set_hip_error(hipErrorUnknown, msg);
    return RH_ERR_HIP;
}

}  // namespace

struct rh_comm {
    void *comm = nullptr;
    int rank = 0, nranks = 1;
};

extern "C" {

rh_status rh_comm_unique_id(uint8_t out128[128]) {
    RH_REQUIRE_INIT();
    if (!out128) return RH_ERR_INVALID;
    if (!load_rccl()) return RH_ERR_UNSUPPORTED;
    Uid id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) return nccl_fail(rc, "ncclGetUniqueId");
    std::memcpy(out128, id.internal, 128);
    return RH_OK;
}

rh_status rh_comm_init(rh_comm **out, int32_t rank, int32_t nranks, const uint8_t uid128[128]) {
    RH_REQUIRE_INIT();
    if (!out || nranks < 1 || rank < 0 || rank >= nranks || !uid128) return RH_ERR_INVALID;
    if (!load_rccl()) return RH_ERR_UNSUPPORTED;
    Uid id;
    std::memcpy(id.internal, uid128, 128);
    rh_comm *c = new rh_comm();
    c->rank = rank;
    c->nranks = nranks;
    const int rc = g_rccl.CommInitRank(&c->comm, nranks, id, rank);  // collective: every rank calls it, on its own device
    if (rc != 0) {
        delete c;
        return nccl_fail(rc, "ncclCommInitRank");
    }
    *out = c;
    return RH_OK;
}

rh_status rh_comm_destroy(rh_comm *c) {
    if (!c) return RH_OK;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return RH_OK;
}

rh_status rh_allreduce_sum_f32(rh_comm *c, float *buf, size_t n, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (!c || (n && !buf)) return RH_ERR_INVALID;
    if (n == 0) return RH_OK;
    const int rc = g_rccl.AllReduce(buf, buf, n, kNcclFloat32, kNcclSum, c->comm, rh::as_stream(stream));
    return rc == 0 ? RH_OK : nccl_fail(rc, "ncclAllReduce");
}

rh_status rh_reduce_sum_f32(rh_comm *c, float *buf, size_t n, int32_t root, rh_stream stream) {
    RH_REQUIRE_INIT();
    if (!c || (n && !buf) || root < 0 || root >= c->nranks) return RH_ERR_INVALID;
    if (n == 0) return RH_OK;
    const int rc = g_rccl.Reduce(buf, buf, n, kNcclFloat32, kNcclSum, root, c->comm, rh::as_stream(stream));
    return rc == 0 ? RH_OK : nccl_fail(rc, "ncclReduce");
}

}  // extern "C"

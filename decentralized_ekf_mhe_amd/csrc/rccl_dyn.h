// rccl_dyn.h — RCCL reached through dlopen so that libdekf.so shares the librccl instance
// already living in the process (torch.distributed's "nccl" backend IS that library on ROCm)
// instead of linking a second copy.  Only what the one exchange step of this path needs:
// an all-gather of the fused base-velocity estimates over xGMI.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <string>

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

inline RcclApi* rccl_api(std::string& err) {
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!api.lib) api.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) {
            api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
            api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
            api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
            api.CommCount = (decltype(api.CommCount))dlsym(api.lib, "ncclCommCount");
            api.CommUserRank = (decltype(api.CommUserRank))dlsym(api.lib, "ncclCommUserRank");
        }
    }
    if (!api.lib || !api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy) {
        err = "librccl.so could not be loaded";
        return nullptr;
    }
    return &api;
}

thread_local std::string g_rccl_err;

inline const char* rccl_fail(RcclApi* a, ncclResult_t r, const char* what) {
    g_rccl_err = std::string(what) + ": " + (a && a->GetErrorString ? a->GetErrorString(r) : "rccl error");
    return g_rccl_err.c_str();
}

static_assert(sizeof(ncclUniqueId) == DEKF_UNIQUE_ID_BYTES, "DEKF_UNIQUE_ID_BYTES must match ncclUniqueId");

inline const char* rccl_unique_id(void* out) {
    RcclApi* a = rccl_api(g_rccl_err);
    if (!a) return g_rccl_err.c_str();
    ncclResult_t r = a->GetUniqueId((ncclUniqueId*)out);
    return r == ncclSuccess ? nullptr : rccl_fail(a, r, "ncclGetUniqueId");
}
inline const char* rccl_init_rank(void** comm, int world, int rank, const void* id) {
    RcclApi* a = rccl_api(g_rccl_err);
    if (!a) return g_rccl_err.c_str();
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t c = nullptr;
    ncclResult_t r = a->CommInitRank(&c, world, uid, rank);
    if (r != ncclSuccess) return rccl_fail(a, r, "ncclCommInitRank");
    *comm = c;
    return nullptr;
}
inline const char* rccl_allgather_f64(void* comm, const double* send, double* recv, size_t count, hipStream_t st) {
    RcclApi* a = rccl_api(g_rccl_err);
    if (!a) return g_rccl_err.c_str();
    ncclResult_t r = a->AllGather(send, recv, count, ncclFloat64, (ncclComm_t)comm, st);
    return r == ncclSuccess ? nullptr : rccl_fail(a, r, "ncclAllGather");
}
inline const char* rccl_comm_info(void* comm, int* count, int* rank) {
    RcclApi* a = rccl_api(g_rccl_err);
    if (!a) return g_rccl_err.c_str();
    if (!a->CommCount || !a->CommUserRank) { g_rccl_err = "librccl.so has no ncclCommCount / ncclCommUserRank"; return g_rccl_err.c_str(); }
    ncclResult_t r = a->CommCount((ncclComm_t)comm, count);
    if (r == ncclSuccess) r = a->CommUserRank((ncclComm_t)comm, rank);
    return r == ncclSuccess ? nullptr : rccl_fail(a, r, "ncclCommCount / ncclCommUserRank");
}
inline void rccl_destroy(void* comm) {
    std::string e;
    RcclApi* a = rccl_api(e);
    if (a && comm) a->CommDestroy((ncclComm_t)comm);
}

}  // namespace

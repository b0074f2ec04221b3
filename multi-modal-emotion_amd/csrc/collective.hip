// The one exchange step of the data-parallel path (SURVEY.md §8b last bullet, §8e): gradient-bucket mean over RCCL / xGMI, as plain C entry
// points for a host that is not Python -- and, since round 4, what ddp.GraphedStep captures INTO the step's hipGraph (a raw RCCL call on the
// caller's stream: no process-group watchdog thread that could invalidate a capture).  RCCL calls are enqueued on the caller's stream like
// every other call of the ABI; communicator creation is the only blocking part.
//
// RCCL is resolved LAZILY (dlopen / dlsym on first use): libtavhip.so itself has no DT_NEEDED entry for it, so single-GPU and C hosts load
// the library without RCCL present, and a process that already holds a librccl.so.1 (PyTorch bundles its own) gets THAT copy -- one RCCL per
// process -- instead of a second one from /opt/rocm.  The header below only supplies types and enum values; tav_comm_rccl_version() reports
// what was loaded so a header / runtime skew is visible (tests record it).
#include <dlfcn.h>
#include <rccl/rccl.h>
#include "common.h"
#include "tavhip_internal.h"

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    bool ok = false;
};
const Rccl& rccl() {
    static const Rccl r = [] {
        Rccl x;
        // a copy the process already mapped first (RTLD_NOLOAD), then the loader's search path, then the ROCm tree
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) if ((x.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!x.handle) for (const char* n : names) if ((x.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!x.handle) return x;
        x.GetVersion = (decltype(x.GetVersion))dlsym(x.handle, "ncclGetVersion");
        x.GetUniqueId = (decltype(x.GetUniqueId))dlsym(x.handle, "ncclGetUniqueId");
        x.CommInitRank = (decltype(x.CommInitRank))dlsym(x.handle, "ncclCommInitRank");
        x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.handle, "ncclCommDestroy");
        x.AllReduce = (decltype(x.AllReduce))dlsym(x.handle, "ncclAllReduce");
        x.ok = x.GetVersion && x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllReduce;
        return x;
    }();
    return r;
}
int nc(ncclResult_t r) { return r == ncclSuccess ? 0 : 1000 + (int)r; }      // 1000 + ncclResult_t: distinct from hipError_t values
}  // namespace

extern "C" int tav_comm_rccl_version(int32_t* version) {
    if (!version) return TAV_ERR_NULL;
    if (!rccl().ok) return TAV_ERR_NO_RCCL;
    int v = 0;
    const int e = nc(rccl().GetVersion(&v));
    *version = v;
    return e;
}
extern "C" int tav_comm_unique_id(void* out128) {
    if (!out128) return TAV_ERR_NULL;
    if (!rccl().ok) return TAV_ERR_NO_RCCL;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    return nc(rccl().GetUniqueId(reinterpret_cast<ncclUniqueId*>(out128)));
}
extern "C" int tav_comm_init_rank(void** comm, int32_t nranks, const void* unique_id128, int32_t rank) {
    if (!comm || !unique_id128) return TAV_ERR_NULL;
    if (nranks <= 0 || rank < 0 || rank >= nranks) return TAV_ERR_SHAPE;
    if (!rccl().ok) return TAV_ERR_NO_RCCL;
    ncclUniqueId id;
    __builtin_memcpy(&id, unique_id128, sizeof(id));
    ncclComm_t c = nullptr;
    const int e = nc(rccl().CommInitRank(&c, nranks, id, rank));
    *comm = (void*)c;
    return e;
}
extern "C" int tav_comm_destroy(void* comm) {
    if (!comm) return TAV_ERR_NULL;
    if (!rccl().ok) return TAV_ERR_NO_RCCL;
    return nc(rccl().CommDestroy((ncclComm_t)comm));
}

extern "C" int tav_allreduce_bucket(void* buf, int64_t nbytes, int32_t dtype, void* comm, void* stream) {
    if (!buf || !comm) return TAV_ERR_NULL;
    if (dtype != TAV_F32 && dtype != TAV_BF16) return TAV_ERR_DTYPE;
    const int es = dtype == TAV_F32 ? 4 : 2;
    if (nbytes <= 0 || nbytes % es) return TAV_ERR_SHAPE;
    if (!rccl().ok) return TAV_ERR_NO_RCCL;
    // in place, mean over the ranks: what DistributedDataParallel does to a gradient bucket
    return nc(rccl().AllReduce(buf, buf, (size_t)(nbytes / es), dtype == TAV_F32 ? ncclFloat32 : ncclBfloat16, ncclAvg, (ncclComm_t)comm, (hipStream_t)stream));
}

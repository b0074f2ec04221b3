// The one exchange step of the data-parallel path (SURVEY.md §8b last bullet, §8e): gradient-bucket mean over RCCL / xGMI, as plain C entry
// points for a host that is not Python.  (The Python host layer issues the same collective through torch.distributed "nccl" = RCCL,
// ddp.py; nothing else of the path communicates.)  RCCL calls are enqueued on the caller's stream like every other call of the ABI;
// communicator creation is the only blocking part.
#include <rccl/rccl.h>
#include "common.h"
#include "tavhip_internal.h"

static int nc(ncclResult_t r) { return r == ncclSuccess ? 0 : 1000 + (int)r; }      // 1000 + ncclResult_t: distinct from hipError_t values

extern "C" int tav_comm_unique_id(void* out128) {
    if (!out128) return TAV_ERR_NULL;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    return nc(ncclGetUniqueId(reinterpret_cast<ncclUniqueId*>(out128)));
}
extern "C" int tav_comm_init_rank(void** comm, int32_t nranks, const void* unique_id128, int32_t rank) {
    if (!comm || !unique_id128) return TAV_ERR_NULL;
    if (nranks <= 0 || rank < 0 || rank >= nranks) return TAV_ERR_SHAPE;
    ncclUniqueId id;
    __builtin_memcpy(&id, unique_id128, sizeof(id));
    ncclComm_t c = nullptr;
    const int e = nc(ncclCommInitRank(&c, nranks, id, rank));
    *comm = (void*)c;
    return e;
}
extern "C" int tav_comm_destroy(void* comm) { return comm ? nc(ncclCommDestroy((ncclComm_t)comm)) : TAV_ERR_NULL; }

extern "C" int tav_allreduce_bucket(void* buf, int64_t nbytes, int32_t dtype, void* comm, void* stream) {
    if (!buf || !comm) return TAV_ERR_NULL;
    if (dtype != TAV_F32 && dtype != TAV_BF16) return TAV_ERR_DTYPE;
    const int es = dtype == TAV_F32 ? 4 : 2;
    if (nbytes <= 0 || nbytes % es) return TAV_ERR_SHAPE;
    // in place, mean over the ranks: what DistributedDataParallel does to a gradient bucket
    return nc(ncclAllReduce(buf, buf, (size_t)(nbytes / es), dtype == TAV_F32 ? ncclFloat32 : ncclBfloat16, ncclAvg, (ncclComm_t)comm, (hipStream_t)stream));
}

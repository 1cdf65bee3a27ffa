// RCCL wrapper of the path's one collective: the all-gather of per-mesh output rows over xGMI, one process
// per GPU. The product's Python layer issues the same collective through torch.distributed (backend "nccl" = RCCL on ROCm,
// morig_amd/dist.py); these exports are the C-ABI form for a host that owns its own communicator (the opaque handle is an
// ncclComm_t created once per process). Nothing here allocates device memory; every call enqueues on the given stream.
#include "common.h"
#include <rccl/rccl.h>
#include <string.h>

#include <dlfcn.h>
#include <mutex>

// librccl is resolved at the FIRST morig_rccl_* call, not at load time: libmorig_hip.so carries no DT_NEEDED on it, so a
// single-GPU host without RCCL still loads the kernels, and a process that already holds an RCCL (PyTorch-ROCm bundles its own)
// keeps using that one copy: the symbols are looked up in the process first, then in librccl.so.1 / librccl.so / /opt/rocm/lib.
namespace morig {
static int g_last_rccl = 0;
static int rccl_status(ncclResult_t r) {
    if (r == ncclSuccess) return MORIG_OK;
    g_last_rccl = (int)r;
    return MORIG_E_HIP;
}
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    bool ok = false;
};
static const RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = RTLD_DEFAULT;
        if (!dlsym(h, "ncclAllGather")) {
            h = nullptr;
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h) break;
            }
            if (!h) return;
        }
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(h, "ncclAllGather"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather;
    });
    return api;
}
}  // namespace morig

using namespace morig;

static_assert(sizeof(ncclUniqueId) == MORIG_RCCL_UNIQUE_ID_BYTES, "MORIG_RCCL_UNIQUE_ID_BYTES must equal sizeof(ncclUniqueId)");

extern "C" int morig_rccl_last_error(void) { return g_last_rccl; }

extern "C" int morig_rccl_unique_id(void* id_out) {
    if (!id_out) return MORIG_E_INVALID;
    if (!rccl().ok) return MORIG_E_UNSUPPORTED;               // no librccl on this host
    ncclUniqueId id;
    const int st = rccl_status(rccl().GetUniqueId(&id));
    if (st == MORIG_OK) memcpy(id_out, &id, sizeof(id));
    return st;
}

extern "C" int morig_rccl_comm_init(int32_t n_ranks, int32_t rank, const void* unique_id, void** comm_out) {
    if (!unique_id || !comm_out || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return MORIG_E_INVALID;
    if (!rccl().ok) return MORIG_E_UNSUPPORTED;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    const int st = rccl_status(rccl().CommInitRank(&comm, n_ranks, id, rank));
    if (st == MORIG_OK) *comm_out = comm;
    return st;
}

extern "C" int morig_rccl_comm_destroy(void* comm) {
    if (!comm) return MORIG_E_INVALID;
    if (!rccl().ok) return MORIG_E_UNSUPPORTED;
    return rccl_status(rccl().CommDestroy(static_cast<ncclComm_t>(comm)));
}

extern "C" int morig_allgather_rows(void* comm, const float* send, float* recv, int64_t rows, int32_t cols, void* stream) {
    if (!comm || !send || !recv || rows < 0 || cols <= 0) return MORIG_E_INVALID;
    if (rows == 0) return MORIG_OK;
    if (!rccl().ok) return MORIG_E_UNSUPPORTED;
    return rccl_status(rccl().AllGather(send, recv, (size_t)rows * (size_t)cols, ncclFloat, static_cast<ncclComm_t>(comm),
                                     reinterpret_cast<hipStream_t>(stream)));
}

extern "C" int morig_allgather_counts(void* comm, const int64_t* send_one, int64_t* recv_n_ranks, void* stream) {
    if (!comm || !send_one || !recv_n_ranks) return MORIG_E_INVALID;
    if (!rccl().ok) return MORIG_E_UNSUPPORTED;
    return rccl_status(rccl().AllGather(send_one, recv_n_ranks, 1, ncclInt64, static_cast<ncclComm_t>(comm), reinterpret_cast<hipStream_t>(stream)));
}

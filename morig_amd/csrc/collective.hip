// RCCL wrapper of the path's one collective: the all-gather of per-mesh output rows over xGMI, one process
// per GPU. The product's Python layer issues the same collective through torch.distributed (backend "nccl" = RCCL on ROCm,
// morig_amd/dist.py); these exports are the C-ABI form for a host that owns its own communicator (the opaque handle is an
// ncclComm_t created once per process). Nothing here allocates device memory; every call enqueues on the given stream.
#include "common.h"
#include <rccl/rccl.h>
#include <string.h>

namespace morig {
static int g_last_rccl = 0;
static int rccl_status(ncclResult_t r) {
    if (r == ncclSuccess) return MORIG_OK;
    g_last_rccl = (int)r;
    return MORIG_E_HIP;
}
}  // namespace morig

using namespace morig;

static_assert(sizeof(ncclUniqueId) == MORIG_RCCL_UNIQUE_ID_BYTES, "MORIG_RCCL_UNIQUE_ID_BYTES must equal sizeof(ncclUniqueId)");

extern "C" int morig_rccl_last_error(void) { return g_last_rccl; }

extern "C" int morig_rccl_unique_id(void* id_out) {
    if (!id_out) return MORIG_E_INVALID;
    ncclUniqueId id;
    const int st = rccl_status(ncclGetUniqueId(&id));
    if (st == MORIG_OK) memcpy(id_out, &id, sizeof(id));
    return st;
}

extern "C" int morig_rccl_comm_init(int32_t n_ranks, int32_t rank, const void* unique_id, void** comm_out) {
    if (!unique_id || !comm_out || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return MORIG_E_INVALID;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    const int st = rccl_status(ncclCommInitRank(&comm, n_ranks, id, rank));
    if (st == MORIG_OK) *comm_out = comm;
    return st;
}

extern "C" int morig_rccl_comm_destroy(void* comm) {
    if (!comm) return MORIG_E_INVALID;
    return rccl_status(ncclCommDestroy(static_cast<ncclComm_t>(comm)));
}

extern "C" int morig_allgather_rows(void* comm, const float* send, float* recv, int64_t rows, int32_t cols, void* stream) {
    if (!comm || !send || !recv || rows < 0 || cols <= 0) return MORIG_E_INVALID;
    if (rows == 0) return MORIG_OK;
    return rccl_status(ncclAllGather(send, recv, (size_t)rows * (size_t)cols, ncclFloat, static_cast<ncclComm_t>(comm),
                                     reinterpret_cast<hipStream_t>(stream)));
}

extern "C" int morig_allgather_counts(void* comm, const int64_t* send_one, int64_t* recv_n_ranks, void* stream) {
    if (!comm || !send_one || !recv_n_ranks) return MORIG_E_INVALID;
    return rccl_status(ncclAllGather(send_one, recv_n_ranks, 1, ncclInt64, static_cast<ncclComm_t>(comm), reinterpret_cast<hipStream_t>(stream)));
}

// comm.hip - the exchange step of the row-partitioned engine: an RCCL all-gather of carry rows on a
// communicator handle.  The reference has no counterpart (NGCF.py is single-device).
//
// libngcf_hip.so does not link RCCL: the entry point binds to the librccl the process already has (the one
// torch.distributed loaded, so the handle and the code that uses it come from the same library) and only falls back to
// loading one by name.  ncclAllGather is asynchronous on the given stream like every other entry point.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {
using allgather_fn = ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
using errstr_fn = const char *(*)(ncclResult_t);
using count_fn = ncclResult_t (*)(const ncclComm_t, int *);

struct Rccl {
    void *handle = nullptr;
    allgather_fn all_gather = nullptr;
    errstr_fn err_string = nullptr;
    count_fn comm_count = nullptr;
};

Rccl *rccl()
{
    static Rccl r = [] {
        Rccl x;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {          // already in the process?
            x.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (x.handle) break;
        }
        if (!x.handle)
            for (const char *name : {"librccl.so.1", "librccl.so"}) {
                x.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (x.handle) break;
            }
        if (x.handle) {
            x.all_gather = (allgather_fn)dlsym(x.handle, "ncclAllGather");
            x.err_string = (errstr_fn)dlsym(x.handle, "ncclGetErrorString");
            x.comm_count = (count_fn)dlsym(x.handle, "ncclCommCount");
        }
        return x;
    }();
    return &r;
}
}  // namespace

extern "C" int ngcf_allgather_rows(void *nccl_comm, const float *send, float *recv, int64_t rows_per_rank, int d, void *stream)
{
    if (!nccl_comm || !send || !recv) return fail(NGCF_ERR_ARG, "allgather_rows: null argument");
    if (rows_per_rank < 0 || d <= 0) return fail(NGCF_ERR_ARG, "allgather_rows: bad sizes");
    Rccl *r = rccl();
    if (!r->handle || !r->all_gather) return fail(NGCF_ERR_HIP, "allgather_rows: librccl is not loadable (%s)", dlerror());
    if (rows_per_rank == 0) return NGCF_OK;
    const ncclResult_t rc = r->all_gather(send, recv, (size_t)rows_per_rank * (size_t)d, ncclFloat32, (ncclComm_t)nccl_comm,
                                          (hipStream_t)stream);
    if (rc != ncclSuccess) return fail(NGCF_ERR_HIP, "ncclAllGather failed: %s", r->err_string ? r->err_string(rc) : "?");
    return NGCF_OK;
}

// ranks of a communicator handle (host call; lets the caller size `recv`)
extern "C" int ngcf_comm_size(void *nccl_comm, int *n_ranks)
{
    if (!nccl_comm || !n_ranks) return fail(NGCF_ERR_ARG, "comm_size: null argument");
    Rccl *r = rccl();
    if (!r->handle || !r->comm_count) return fail(NGCF_ERR_HIP, "comm_size: librccl is not loadable");
    const ncclResult_t rc = r->comm_count((ncclComm_t)nccl_comm, n_ranks);
    if (rc != ncclSuccess) return fail(NGCF_ERR_HIP, "ncclCommCount failed: %s", r->err_string ? r->err_string(rc) : "?");
    return NGCF_OK;
}

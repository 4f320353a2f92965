// sa_comm_*: the collectives of the data-parallel path for a host that does not bring torch.distributed -- thin C-ABI wrappers over RCCL (NCCL API on ROCm),
// resolved with dlopen at the first sa_comm_* call so that the library itself has no link-time dependency on RCCL (a torch host keeps using its own
// process group: runtime/ddp.py; these entry points are what a Go / C++ / Java host would bind instead).
//
// Replaces, for the path of SURVEY section 8(e): DistributedDataParallel's gradient all-reduce (reference run_vqvae.py:71-77, run_transformer.py:95-103) and
// dist.all_reduce(encodings_sum) / dist.all_reduce(dw) of the EMA quantizer (src/networks/vqvae/baseline.py:70-72; here ONE call on the packed [K + K D] buffer).
// One communicator per process (one process per GPU); every call enqueues on the caller's HIP stream and returns -- no synchronisation, like every launcher here.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "sa_common.h"

namespace {

// the slice of the NCCL API that is used (rccl.h: ncclResult_t / ncclDataType_t / ncclRedOp_t are ints; ncclUniqueId is 128 opaque bytes passed BY VALUE)
struct UniqueId { char internal[128]; };
typedef void* Comm;
constexpr int kNcclSum = 0, kNcclFloat32 = 7, kNcclBfloat16 = 9, kNcclFloat16 = 6;

struct Api {
    void* lib = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Api g_api;
std::once_flag g_once;
thread_local char g_comm_error[256] = "";

void load_api() {
    // an RCCL the process already carries (a torch host's) is found by its soname; otherwise the ROCm installation's
    const char* names[] = {getenv("SA_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        g_api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_api.lib) break;
    }
    if (!g_api.lib) return;
    auto sym = [](const char* s) { return dlsym(g_api.lib, s); };
    g_api.GetUniqueId = (decltype(g_api.GetUniqueId))sym("ncclGetUniqueId");
    g_api.CommInitRank = (decltype(g_api.CommInitRank))sym("ncclCommInitRank");
    g_api.CommDestroy = (decltype(g_api.CommDestroy))sym("ncclCommDestroy");
    g_api.AllReduce = (decltype(g_api.AllReduce))sym("ncclAllReduce");
    g_api.ReduceScatter = (decltype(g_api.ReduceScatter))sym("ncclReduceScatter");
    g_api.AllGather = (decltype(g_api.AllGather))sym("ncclAllGather");
    g_api.GetErrorString = (decltype(g_api.GetErrorString))sym("ncclGetErrorString");
    g_api.ok = g_api.GetUniqueId && g_api.CommInitRank && g_api.CommDestroy && g_api.AllReduce && g_api.ReduceScatter && g_api.AllGather;
}

int api_ready() {
    std::call_once(g_once, load_api);
    if (!g_api.ok) {
        snprintf(g_comm_error, sizeof g_comm_error, "RCCL not found (dlopen librccl.so.1 / SA_RCCL_LIB): %s", g_api.lib ? "missing symbols" : dlerror() ? "dlopen failed" : "not loaded");
        return SA_EUNSUPPORTED;
    }
    return 0;
}

int nccl_dtype(int dtype) { return dtype == SA_F32 ? kNcclFloat32 : dtype == SA_BF16 ? kNcclBfloat16 : dtype == SA_F16 ? kNcclFloat16 : -1; }

int check(int rc, const char* what) {
    if (rc == 0) return 0;
    snprintf(g_comm_error, sizeof g_comm_error, "%s: %s (ncclResult %d)", what, g_api.GetErrorString ? g_api.GetErrorString(rc) : "?", rc);
    return SA_ECOMM;
}

}  // namespace

struct sa_comm {
    Comm comm;
    int rank, world;
};

extern "C" const char* sa_comm_last_error(void) { return g_comm_error; }

extern "C" int sa_comm_unique_id(void* id_out) {
    if (!id_out) return SA_EINVAL;
    if (int rc = api_ready()) return rc;
    return check(g_api.GetUniqueId((UniqueId*)id_out), "ncclGetUniqueId");
}

extern "C" int sa_comm_init(sa_comm** out, const void* id, int rank, int world) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return SA_EINVAL;
    if (int rc = api_ready()) return rc;
    UniqueId uid;
    memcpy(&uid, id, sizeof uid);
    Comm c = nullptr;
    if (int rc = check(g_api.CommInitRank(&c, world, uid, rank), "ncclCommInitRank")) return rc;
    *out = new sa_comm{c, rank, world};
    return 0;
}

extern "C" int sa_comm_destroy(sa_comm* c) {
    if (!c) return SA_EINVAL;
    const int rc = check(g_api.CommDestroy(c->comm), "ncclCommDestroy");
    delete c;
    return rc;
}

extern "C" int sa_comm_rank(const sa_comm* c) { return c ? c->rank : SA_EINVAL; }
extern "C" int sa_comm_world(const sa_comm* c) { return c ? c->world : SA_EINVAL; }

// buf[i] <- sum over ranks of buf[i], in place, on `stream`
extern "C" int sa_comm_all_reduce_sum(sa_comm* c, void* buf, int64_t n, int dtype, void* stream) {
    if (!c || !buf || n <= 0 || nccl_dtype(dtype) < 0) return SA_EINVAL;
    return check(g_api.AllReduce(buf, buf, (size_t)n, nccl_dtype(dtype), kNcclSum, c->comm, (hipStream_t)stream), "ncclAllReduce");
}

// recv[0 .. n_per_rank) <- sum over ranks of their send[rank * n_per_rank ..): send holds world * n_per_rank elements
extern "C" int sa_comm_reduce_scatter_sum(sa_comm* c, const void* send, void* recv, int64_t n_per_rank, int dtype, void* stream) {
    if (!c || !send || !recv || n_per_rank <= 0 || nccl_dtype(dtype) < 0) return SA_EINVAL;
    return check(g_api.ReduceScatter(send, recv, (size_t)n_per_rank, nccl_dtype(dtype), kNcclSum, c->comm, (hipStream_t)stream), "ncclReduceScatter");
}

// recv[r * n_per_rank ..) <- rank r's send[0 .. n_per_rank): recv holds world * n_per_rank elements
extern "C" int sa_comm_all_gather(sa_comm* c, const void* send, void* recv, int64_t n_per_rank, int dtype, void* stream) {
    if (!c || !send || !recv || n_per_rank <= 0 || nccl_dtype(dtype) < 0) return SA_EINVAL;
    return check(g_api.AllGather(send, recv, (size_t)n_per_rank, nccl_dtype(dtype), c->comm, (hipStream_t)stream), "ncclAllGather");
}

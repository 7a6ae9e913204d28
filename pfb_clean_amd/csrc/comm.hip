// comm.hip -- the band-shard exchange of the cube PCG, called straight from C on the solver's stream.
//
// The reference sums the CG inner products over bands inside one process (pfb/opt/pcg.py:92-107 on a
// (nband, nx, ny) array; dask threads share memory).  With one process per GPU each rank owns a slice of the
// bands and the same sums need ONE all-reduce of 3..7 doubles per iteration (SURVEY 5.8 / 8b `pfb_comm_init`).
// pfb_pcg_solve takes that exchange as a `pfb_allreduce_fn`; `pfb_comm_allreduce` below IS such a function:
// an RCCL all-reduce enqueued on the solver's own stream -- no host callback into an interpreter, no second
// stream, no event hand-shake (the torch.distributed hook of pfb_clean_amd/dist.py costs 10-15 us of stream
// hand-over per iteration on a 320 us one-band iteration; it stays as the fallback).
//
// RCCL is bound at RUN time (dlopen), first to the copy already mapped into the process -- under PyTorch that is
// torch's own librccl.so, so that one RCCL instance owns the GPU's IPC / xGMI state -- then to the system's
// librccl.so.1.  The library has no link-time dependency on RCCL: a single-GPU user never loads it.
#include "common.hpp"
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <rccl/rccl.h>

namespace pfb {
namespace {

struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;   // optional (failure handling)
    decltype(&ncclCommAbort) CommAbort = nullptr;                   // optional
    char path[512] = {0};
};
Rccl g_rccl;
std::mutex g_mu;

int bind_rccl(const char* path_hint) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_rccl.h) return PFB_OK;
    void* h = nullptr;
    const char* how = "";
    if (path_hint && *path_hint) { h = dlopen(path_hint, RTLD_NOW | RTLD_LOCAL); how = path_hint; }
    static const char* names[] = {"librccl.so", "librccl.so.1"};
    for (int pass = 0; pass < 2 && !h; ++pass)              // pass 0: only a copy that is already loaded
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) { how = n; break; }
        }
    PFB_REQUIRE(h, PFB_ERR_UNSUPPORTED, "comm: librccl not found (%s)", dlerror());
    Rccl r;
    r.h = h;
#define SYM(N) r.N = (decltype(r.N))dlsym(h, "nccl" #N); \
    if (!r.N) { set_error("comm: %s lacks nccl" #N, how); dlclose(h); return PFB_ERR_UNSUPPORTED; }
    SYM(GetVersion) SYM(GetUniqueId) SYM(CommInitRank) SYM(CommDestroy) SYM(AllReduce) SYM(GetErrorString)
#undef SYM
    r.CommGetAsyncError = (decltype(r.CommGetAsyncError))dlsym(h, "ncclCommGetAsyncError");
    r.CommAbort = (decltype(r.CommAbort))dlsym(h, "ncclCommAbort");
    snprintf(r.path, sizeof(r.path), "%s", how);
    g_rccl = r;
    return PFB_OK;
}

#define PFB_NCCL_CHECK(expr)                                                                       \
    do {                                                                                           \
        ncclResult_t _r = (expr);                                                                  \
        if (_r != ncclSuccess) {                                                                   \
            set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r));     \
            return PFB_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

}  // namespace
}  // namespace pfb

struct pfb_comm {
    ncclComm_t comm;
    int rank, nranks, device;
    int dead;                // aborted or failed: nothing may be enqueued on it any more
};

using namespace pfb;

extern "C" {

int pfb_comm_bind(const char* librccl_path) { return bind_rccl(librccl_path); }

int pfb_comm_unique_id(void* id128) {
    PFB_REQUIRE(id128, PFB_ERR_INVALID, "comm_unique_id: null buffer");
    int e = bind_rccl(nullptr);
    if (e != PFB_OK) return e;
    static_assert(sizeof(ncclUniqueId) == PFB_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    PFB_NCCL_CHECK(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return PFB_OK;
}

int pfb_comm_init(int rank, int nranks, const void* id128, pfb_comm** out) {
    PFB_REQUIRE(out && id128, PFB_ERR_INVALID, "comm_init: null argument");
    PFB_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, PFB_ERR_INVALID, "comm_init: rank %d of %d", rank, nranks);
    *out = nullptr;
    int e = bind_rccl(nullptr);
    if (e != PFB_OK) return e;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    int dev = 0;
    PFB_HIP_CHECK(hipGetDevice(&dev));               // the communicator binds to the caller's current device
    ncclComm_t c = nullptr;
    PFB_NCCL_CHECK(g_rccl.CommInitRank(&c, nranks, id, rank));
    pfb_comm* p = new pfb_comm{c, rank, nranks, dev, 0};
    *out = p;
    return PFB_OK;
}

int pfb_comm_destroy(pfb_comm* c) {
    if (!c) return PFB_OK;
    if (c->comm && !c->dead && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);   // (an aborted communicator is already freed)
    delete c;
    return PFB_OK;
}

int pfb_comm_info(const pfb_comm* c, int* rank, int* nranks, int* device, int* rccl_version) {
    PFB_REQUIRE(c, PFB_ERR_INVALID, "comm_info: null communicator");
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    if (device) *device = c->device;
    if (rccl_version) { int v = 0; g_rccl.GetVersion(&v); *rccl_version = v; }
    return PFB_OK;
}

int pfb_comm_check(pfb_comm* c) {
    PFB_REQUIRE(c, PFB_ERR_INVALID, "comm_check: null communicator");
    PFB_REQUIRE(!c->dead, PFB_ERR_COMM, "comm: the communicator of rank %d was aborted", c->rank);
    if (!g_rccl.CommGetAsyncError) return PFB_OK;           // this RCCL cannot tell: the solver's timeout still holds
    ncclResult_t st = ncclSuccess;
    const ncclResult_t q = g_rccl.CommGetAsyncError(c->comm, &st);
    if (q == ncclSuccess && (st == ncclSuccess || st == ncclInProgress)) return PFB_OK;
    set_error("comm: asynchronous error on rank %d of %d: %s", c->rank, c->nranks,
              g_rccl.GetErrorString(q != ncclSuccess ? q : st));
    return PFB_ERR_COMM;
}

int pfb_comm_abort(pfb_comm* c) {
    PFB_REQUIRE(c, PFB_ERR_INVALID, "comm_abort: null communicator");
    if (c->dead) return PFB_OK;
    c->dead = 1;
    if (c->comm && g_rccl.CommAbort) {
        g_rccl.CommAbort(c->comm);           // peers' pending collectives on this communicator now fail instead of waiting
        c->comm = nullptr;
    }
    return PFB_OK;
}

// a pfb_allreduce_fn (ctx = the pfb_comm*): sums `count` doubles in place over the ranks, ordered on `stream`;
// count == 0: probe, count < 0: abort (include/pfb_hip.h, "Failure protocol")
int pfb_comm_allreduce(void* ctx, double* dev_buf, int count, void* stream) {
    pfb_comm* c = static_cast<pfb_comm*>(ctx);
    PFB_REQUIRE(c, PFB_ERR_INVALID, "comm_allreduce: null communicator");
    if (count == 0) return pfb_comm_check(c);
    if (count < 0) return pfb_comm_abort(c);
    PFB_REQUIRE(!c->dead && c->comm, PFB_ERR_COMM, "comm_allreduce: the communicator of rank %d was aborted", c->rank);
    PFB_REQUIRE(dev_buf, PFB_ERR_INVALID, "comm_allreduce: null buffer");
    const ncclResult_t r = g_rccl.AllReduce(dev_buf, dev_buf, (size_t)count, ncclDouble, ncclSum, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) {
        set_error("comm_allreduce: rank %d of %d: %s", c->rank, c->nranks, g_rccl.GetErrorString(r));
        return PFB_ERR_COMM;
    }
    return PFB_OK;
}

}  // extern "C"

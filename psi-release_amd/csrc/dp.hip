// libpsi_hip.so: the data-parallel leg of the fitting loop — an RCCL communicator owned by the library, so that the ONE collective of
// an iteration (the 6-float all-reduce of the loss normalisers between forward and backward; the reference has no
// distributed code: cluster_mpi/htcondor_submission.sub:15 fans whole processes out over a cluster) is issued from C on the engine's
// stream and captured into the iteration's hipGraph together with the kernels (psi_fit_iterate_dp, fit.hip).
//
// RCCL is resolved at the first psi_dp_* call with dlopen: a process that already maps librccl.so.1 (PyTorch-ROCm's torch.distributed
// does) gets THAT copy — one RCCL per process — and a process that never runs data parallel never loads the 500 MB library.
#include "psi_internal.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return 0;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)                  // a copy the process already maps (torch's) wins
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    for (int i = 0; !h && i < 3; i++) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        psi_set_error("data parallel: librccl.so.1 could not be loaded (%s)", dlerror());
        return PSI_EINVAL;
    }
    RcclApi a;
    a.h = h;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    a.GetVersion = (decltype(a.GetVersion))dlsym(h, "ncclGetVersion");
    a.CommCount = (decltype(a.CommCount))dlsym(h, "ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))dlsym(h, "ncclCommUserRank");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString) {
        psi_set_error("data parallel: librccl.so.1 lacks an expected symbol");
        return PSI_EINVAL;
    }
    g_rccl = a;
    return 0;
}

}  // namespace

#define PSI_CHECK_RCCL(expr)                                                                               \
    do {                                                                                                   \
        ncclResult_t _r = (expr);                                                                          \
        if (_r != ncclSuccess) {                                                                           \
            psi_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__);  \
            return 1000 + (int)_r;                                                                         \
        }                                                                                                  \
    } while (0)

struct psi_dp_comm {
    ncclComm_t comm;
    int rank, world, device;
};

extern "C" int psi_dp_unique_id(char *h_id128)
{
    PSI_REQUIRE(h_id128, "null id buffer");
    int rc = rccl_load();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == PSI_DP_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    PSI_CHECK_RCCL(g_rccl.GetUniqueId(&id));
    memcpy(h_id128, &id, sizeof(id));
    return 0;
}

extern "C" int psi_dp_comm_create(psi_dp_comm **out, const char *h_id128, int rank, int world)
{
    PSI_REQUIRE(out && h_id128 && world >= 1 && rank >= 0 && rank < world, "bad arguments");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, h_id128, sizeof(id));
    psi_dp_comm *c = new psi_dp_comm();
    c->rank = rank;
    c->world = world;
    if (hipGetDevice(&c->device) != hipSuccess) c->device = -1;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);       // collective over the `world` ranks (one per GPU)
    if (r != ncclSuccess) {
        psi_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
        delete c;
        return 1000 + (int)r;
    }
    *out = c;
    return 0;
}

extern "C" void psi_dp_comm_destroy(psi_dp_comm *c)
{
    if (!c) return;
    if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

extern "C" int psi_dp_comm_info(const psi_dp_comm *c, int *rank, int *world, int *rccl_version)
{
    PSI_REQUIRE(c, "null communicator");
    // what RCCL itself says about the communicator (not what psi_dp_comm_create was told): a launcher that starts fewer ranks than
    // it announces, or ranks that joined different communicators, show up here
    if (rank) {
        *rank = c->rank;
        if (g_rccl.CommUserRank) PSI_CHECK_RCCL(g_rccl.CommUserRank(c->comm, rank));
    }
    if (world) {
        *world = c->world;
        if (g_rccl.CommCount) PSI_CHECK_RCCL(g_rccl.CommCount(c->comm, world));
    }
    if (rccl_version) {
        *rccl_version = 0;
        if (g_rccl.GetVersion) (void)g_rccl.GetVersion(rccl_version);
    }
    return 0;
}

extern "C" int psi_dp_allreduce_sum(psi_dp_comm *c, float *d_buf, int n, void *stream)
{
    PSI_REQUIRE(c && d_buf && n > 0, "bad arguments");
    PSI_CHECK_RCCL(g_rccl.AllReduce(d_buf, d_buf, (size_t)n, ncclFloat, ncclSum, c->comm, (hipStream_t)stream));
    return 0;
}

int psi_dp_world(const psi_dp_comm *c) { return c ? c->world : 0; }

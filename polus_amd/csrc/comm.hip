// polus_comm_*: the data-parallel collectives of the training step as C-ABI entry points over RCCL
// (xGMI inside a node).  They replace the device-side half of Horovod's surface in the reference:
// hvd.DistributedGradientTape's gradient averaging (polus/training.py:182-185) and hvd.broadcast_variables
// (polus/training.py:210-211).  One communicator per process (= per GPU); every collective is queued on the
// caller's stream and returns at once.
//
// librccl is loaded at the first polus_comm_* call (dlopen, no link-time dependency: the kernel library
// loads on hosts that never go multi-GPU).  PyTorch-ROCm ships its own librccl.so.1 with the same SONAME:
// when the host process has already loaded it, dlopen hands back that copy, so a process never runs two
// RCCL runtimes.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.handle) return POLUS_OK;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) { polus_set_error("polus_comm: cannot load librccl (%s)", dlerror()); return POLUS_ERR_HIP; }
#define POLUS_SYM(field, sym) do { *reinterpret_cast<void**>(&g_rccl.field) = dlsym(h, sym); \
        if (!g_rccl.field) { polus_set_error("polus_comm: librccl lacks %s", sym); dlclose(h); return POLUS_ERR_HIP; } } while (0)
    POLUS_SYM(GetUniqueId, "ncclGetUniqueId");
    POLUS_SYM(CommInitRank, "ncclCommInitRank");
    POLUS_SYM(CommDestroy, "ncclCommDestroy");
    POLUS_SYM(AllReduce, "ncclAllReduce");
    POLUS_SYM(Broadcast, "ncclBroadcast");
    POLUS_SYM(ReduceScatter, "ncclReduceScatter");
    POLUS_SYM(AllGather, "ncclAllGather");
    POLUS_SYM(CommCount, "ncclCommCount");
    POLUS_SYM(CommUserRank, "ncclCommUserRank");
    POLUS_SYM(CommCuDevice, "ncclCommCuDevice");
    POLUS_SYM(GroupStart, "ncclGroupStart");
    POLUS_SYM(GroupEnd, "ncclGroupEnd");
    POLUS_SYM(GetErrorString, "ncclGetErrorString");
#undef POLUS_SYM
    g_rccl.handle = h;
    return POLUS_OK;
}

struct PolusComm { ncclComm_t comm; int rank, world; };

#define POLUS_NCCL(call, what) do { ncclResult_t r__ = (call); \
    if (r__ != ncclSuccess) { polus_set_error("%s: %s", what, g_rccl.GetErrorString(r__)); return POLUS_ERR_HIP; } } while (0)

int nccl_type(int dtype, ncclDataType_t* t) {
    if (dtype == POLUS_F32) { *t = ncclFloat32; return POLUS_OK; }
    if (dtype == POLUS_BF16) { *t = ncclBfloat16; return POLUS_OK; }
    polus_set_error("polus_comm: bad dtype %d", dtype);
    return POLUS_ERR_INVALID;
}

}  // namespace

extern "C" int polus_comm_unique_id(void* out_id128) {
    POLUS_REQUIRE(out_id128, "polus_comm_unique_id: null pointer");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    POLUS_NCCL(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
    static_assert(sizeof(id) == POLUS_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(out_id128, &id, sizeof(id));
    return POLUS_OK;
}

extern "C" int polus_comm_init(void** comm, int rank, int world, const void* unique_id128) {
    POLUS_REQUIRE(comm && unique_id128 && world >= 1 && rank >= 0 && rank < world, "polus_comm_init: bad arguments (rank %d of %d)", rank, world);
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, unique_id128, sizeof(id));
    PolusComm* c = new PolusComm{nullptr, rank, world};
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { polus_set_error("ncclCommInitRank: %s", g_rccl.GetErrorString(r)); delete c; return POLUS_ERR_HIP; }
    *comm = c;
    return POLUS_OK;
}

extern "C" int polus_comm_destroy(void* comm) {
    if (!comm) return POLUS_OK;
    PolusComm* c = static_cast<PolusComm*>(comm);
    ncclResult_t r = g_rccl.CommDestroy(c->comm);
    delete c;
    if (r != ncclSuccess) { polus_set_error("ncclCommDestroy: %s", g_rccl.GetErrorString(r)); return POLUS_ERR_HIP; }
    return POLUS_OK;
}

extern "C" int polus_comm_info(void* comm, int* n_ranks, int* rank, int* device) {
    POLUS_REQUIRE(comm && n_ranks && rank && device, "polus_comm_info: null pointer");
    PolusComm* c = static_cast<PolusComm*>(comm);
    POLUS_NCCL(g_rccl.CommCount(c->comm, n_ranks), "ncclCommCount");
    POLUS_NCCL(g_rccl.CommUserRank(c->comm, rank), "ncclCommUserRank");
    POLUS_NCCL(g_rccl.CommCuDevice(c->comm, device), "ncclCommCuDevice");
    return POLUS_OK;
}

extern "C" int polus_comm_broadcast(void* comm, void* buf, size_t bytes, int root, void* stream) {
    POLUS_REQUIRE(comm && buf, "polus_comm_broadcast: null pointer");
    PolusComm* c = static_cast<PolusComm*>(comm);
    POLUS_REQUIRE(root >= 0 && root < c->world, "polus_comm_broadcast: bad root %d", root);
    POLUS_NCCL(g_rccl.Broadcast(buf, buf, bytes, ncclUint8, root, c->comm, static_cast<hipStream_t>(stream)), "ncclBroadcast");
    return POLUS_OK;
}

extern "C" int polus_comm_allreduce_sum(void* comm, void* buf, size_t count, int dtype, void* stream) {
    POLUS_REQUIRE(comm && buf, "polus_comm_allreduce_sum: null pointer");
    PolusComm* c = static_cast<PolusComm*>(comm);
    ncclDataType_t t;
    int rc = nccl_type(dtype, &t);
    if (rc) return rc;
    POLUS_NCCL(g_rccl.AllReduce(buf, buf, count, t, ncclSum, c->comm, static_cast<hipStream_t>(stream)), "ncclAllReduce");
    return POLUS_OK;
}

extern "C" int polus_comm_reduce_scatter_sum(void* comm, const void* send, void* recv, size_t recv_count, int dtype, void* stream) {
    POLUS_REQUIRE(comm && send && recv, "polus_comm_reduce_scatter_sum: null pointer");
    PolusComm* c = static_cast<PolusComm*>(comm);
    ncclDataType_t t;
    int rc = nccl_type(dtype, &t);
    if (rc) return rc;
    POLUS_NCCL(g_rccl.ReduceScatter(send, recv, recv_count, t, ncclSum, c->comm, static_cast<hipStream_t>(stream)), "ncclReduceScatter");
    return POLUS_OK;
}

extern "C" int polus_comm_all_gather(void* comm, const void* send, void* recv, size_t send_count, int dtype, void* stream) {
    POLUS_REQUIRE(comm && send && recv, "polus_comm_all_gather: null pointer");
    PolusComm* c = static_cast<PolusComm*>(comm);
    ncclDataType_t t;
    int rc = nccl_type(dtype, &t);
    if (rc) return rc;
    POLUS_NCCL(g_rccl.AllGather(send, recv, send_count, t, c->comm, static_cast<hipStream_t>(stream)), "ncclAllGather");
    return POLUS_OK;
}

extern "C" int polus_comm_group_start(void) {
    int rc = rccl_load();
    if (rc) return rc;
    POLUS_NCCL(g_rccl.GroupStart(), "ncclGroupStart");
    return POLUS_OK;
}

extern "C" int polus_comm_group_end(void) {
    int rc = rccl_load();
    if (rc) return rc;
    POLUS_NCCL(g_rccl.GroupEnd(), "ncclGroupEnd");
    return POLUS_OK;
}

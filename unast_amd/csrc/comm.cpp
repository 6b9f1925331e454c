// RCCL behind the C ABI: the gradient all-reduce of the data-parallel train step, issued from C++ on
// ordinary HIP streams -- by unast_amd.ddp in the eager step and by the stream-replay executor (graph_exec.cpp) in the captured one,
// where a collective cannot be a captured node: at capture time unast_allreduce_marker leaves a recognisable kernel node at the
// point of the exchange, and the executor issues ncclAllReduce in its place on that node's stream.
//
// librccl.so is bound at run time (dlopen): libunast_hip.so loads on a machine without RCCL, and a process that already carries
// RCCL (torch.distributed's nccl backend) shares that copy.  The unique id travels through the Python launcher
// (torch.distributed's store / a broadcast), not through this library.
//
// New functionality with respect to the reference (single process, single device: /root/reference/src/utils.py:101-106); the
// ordering it must preserve -- generator update before the discriminator phase's forward -- is /root/reference/src/train.py:628-637.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "../../include/unast_hip.h"

namespace {

typedef struct { char internal[128]; } NcclId;
typedef void* NcclComm;
typedef int (*fn_get_id)(NcclId*);
typedef int (*fn_init_rank)(NcclComm*, int, NcclId, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*fn_destroy)(NcclComm);
typedef const char* (*fn_errstr)(int);

struct Rccl {
    void* lib = nullptr;
    fn_get_id get_id = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_allreduce allreduce = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
};
Rccl g_rccl;

bool rccl_load() {
    if (g_rccl.lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* lib = nullptr;
    // UNAST_COMM_LIB (tests only): another library with the same five entry points -- tests/native/fake_rccl.cpp stages the all-reduce through
    // host shared memory so that several ranks sharing ONE GPU can rehearse this path (RCCL refuses two ranks on one device)
    const char* override_path = getenv("UNAST_COMM_LIB");
    if (override_path && override_path[0]) {
        lib = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
        if (!lib) { unast_set_error(UNAST_ERR_LAUNCH, "unast_comm: UNAST_COMM_LIB=%s could not be loaded (%s)", override_path, dlerror()); return false; }
    }
    if (!lib) for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;       // a copy the process already carries
    if (!lib) for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) { unast_set_error(UNAST_ERR_LAUNCH, "unast_comm: librccl.so not found (%s)", dlerror()); return false; }
    g_rccl.get_id = (fn_get_id)dlsym(lib, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_rank)dlsym(lib, "ncclCommInitRank");
    g_rccl.allreduce = (fn_allreduce)dlsym(lib, "ncclAllReduce");
    g_rccl.destroy = (fn_destroy)dlsym(lib, "ncclCommDestroy");
    g_rccl.errstr = (fn_errstr)dlsym(lib, "ncclGetErrorString");
    if (!g_rccl.get_id || !g_rccl.init_rank || !g_rccl.allreduce || !g_rccl.destroy) {
        unast_set_error(UNAST_ERR_LAUNCH, "unast_comm: librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
        return false;
    }
    g_rccl.lib = lib;
    return true;
}

int rccl_fail(const char* what, int rc) {
    return unast_set_error(UNAST_ERR_LAUNCH, "%s: RCCL error %d (%s)", what, rc, g_rccl.errstr ? g_rccl.errstr(rc) : "?");
}

struct Comm { NcclComm comm; int rank, world; };

}  // namespace

// ncclFloat32 = 7, ncclSum = 0 (rccl.h ncclDataType_t / ncclRedOp_t)
static const int kNcclFloat = 7, kNcclSum = 0;

extern "C" int unast_comm_unique_id(void* out128) {
    UNAST_REQUIRE(out128, "unast_comm_unique_id: null buffer");
    if (!rccl_load()) return UNAST_ERR_LAUNCH;
    NcclId id;
    const int rc = g_rccl.get_id(&id);
    if (rc) return rccl_fail("unast_comm_unique_id", rc);
    memcpy(out128, &id, sizeof(id));
    return UNAST_OK;
}

extern "C" int64_t unast_comm_init(const void* unique_id128, int rank, int world) {
    if (!unique_id128 || world < 1 || rank < 0 || rank >= world) { unast_set_error(UNAST_ERR_ARG, "unast_comm_init: bad arguments (rank %d of %d)", rank, world); return 0; }
    if (!rccl_load()) return 0;
    NcclId id;
    memcpy(&id, unique_id128, sizeof(id));
    Comm* c = new Comm{nullptr, rank, world};
    const int rc = g_rccl.init_rank(&c->comm, world, id, rank);
    if (rc) { rccl_fail("unast_comm_init (ncclCommInitRank)", rc); delete c; return 0; }
    return (int64_t)(intptr_t)c;
}

extern "C" int unast_allreduce(int64_t comm, float* buf, int64_t count, hipStream_t stream) {
    Comm* c = (Comm*)(intptr_t)comm;
    UNAST_REQUIRE(c && buf && count > 0, "unast_allreduce: bad arguments");
    const int rc = g_rccl.allreduce(buf, buf, (size_t)count, kNcclFloat, kNcclSum, c->comm, stream);
    if (rc) return rccl_fail("unast_allreduce (ncclAllReduce)", rc);
    return UNAST_OK;
}

extern "C" int unast_comm_destroy(int64_t comm) {
    Comm* c = (Comm*)(intptr_t)comm;
    if (!c) return UNAST_OK;
    if (g_rccl.destroy && c->comm) g_rccl.destroy(c->comm);
    delete c;
    return UNAST_OK;
}

// ---- the marker a captured step carries where a collective belongs -------------------------------------------------------------
__global__ void unast_allreduce_marker_kernel(float* buf, long long count) {
    // (nothing: hipGraphLaunch of a graph that still holds markers exchanges nothing -- unast_amd.graphed replays distributed steps through
    // the stream executor only)
    (void)buf; (void)count;
}

extern "C" int unast_allreduce_marker(float* buf, int64_t count, hipStream_t stream) {
    UNAST_REQUIRE(buf && count > 0, "unast_allreduce_marker: bad arguments");
    hipLaunchKernelGGL(unast_allreduce_marker_kernel, dim3(1), dim3(1), 0, stream, buf, (long long)count);
    return unast_check_launch("unast_allreduce_marker");
}

// used by graph_exec.cpp
extern "C" const void* unast_allreduce_marker_func(void) { return (const void*)unast_allreduce_marker_kernel; }

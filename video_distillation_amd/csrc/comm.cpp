// vd_comm_*: the exchange steps of the sharded DM / s2d / MTT loops for callers of the C ABI -- a thin layer over RCCL (the
// collectives library of ROCm; xGMI between the GPUs of a node).  The Python trainers issue the same collectives through
// torch.distributed (backend "nccl" == RCCL); a C caller of vd_embed_* / vd_train_* has no torch, so it gets them here.
//
// The reference's only multi-GPU mechanism is nn.DataParallel (reference utils.py:615-623): scatter of the batch, replicated
// forward, gather.  What replaces it on MI355X is one process per GPU and, per step, at most these exchanges:
// the per-class feature sums of split classes (vd_comm_allreduce_f32 on C x 2048 floats), the 327 hallucinator gradients, the
// flat parameter gradient of an MTT student step, and an all-gather of the synthetic clips before evaluation.
//
// librccl.so is resolved with dlopen at the first vd_comm_* call, not linked: libvd_hip.so keeps loading on a box without it,
// and inside a torch process the RCCL torch already mapped is reused instead of a second copy.
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "../../include/vd_hip.h"

namespace {

// the handful of RCCL entry points used, with the ABI of rccl.h (ncclResult_t = int; ncclFloat32 = 7; ncclSum = 0)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void* id) = nullptr;
    int (*CommInitRank)(void** comm, int nranks, VdCommId id, int rank) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*AllReduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, void* stream) = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t sendcount, int dtype, void* comm, void* stream) = nullptr;
    int (*GetVersion)(int* version) = nullptr;        // optional
    int (*CommCount)(void* comm, int* count) = nullptr;      // optional: the communicator's own idea of its size
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);       // the copy the process already has (torch's), if any
        if (g_rccl.lib) break;
    }
    for (int i = 0; !g_rccl.lib && i < 3; ++i) g_rccl.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!g_rccl.lib) return;
    *(void**)&g_rccl.GetUniqueId = dlsym(g_rccl.lib, "ncclGetUniqueId");
    *(void**)&g_rccl.CommInitRank = dlsym(g_rccl.lib, "ncclCommInitRank");
    *(void**)&g_rccl.CommDestroy = dlsym(g_rccl.lib, "ncclCommDestroy");
    *(void**)&g_rccl.AllReduce = dlsym(g_rccl.lib, "ncclAllReduce");
    *(void**)&g_rccl.AllGather = dlsym(g_rccl.lib, "ncclAllGather");
    *(void**)&g_rccl.GetVersion = dlsym(g_rccl.lib, "ncclGetVersion");
    *(void**)&g_rccl.CommCount = dlsym(g_rccl.lib, "ncclCommCount");
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.AllGather;
}

const Rccl* rccl() {
    std::call_once(g_once, load_rccl);
    return g_rccl.ok ? &g_rccl : nullptr;
}

}  // namespace

struct VdComm {
    void* comm;
    int nranks, rank;
};

// Rank 0 creates the 128-byte id and hands it to the other ranks by any host channel (a file, MPI, the launcher's store).
extern "C" int vd_comm_unique_id(VdCommId* id) {
    if (id == nullptr) return -1;
    const Rccl* r = rccl();
    if (r == nullptr) return -10;                 // no RCCL on this box
    return r->GetUniqueId(id) == 0 ? 0 : -11;
}

// One communicator per process / GPU (the current HIP device is the rank's device), nranks >= 1.
extern "C" int vd_comm_create(const VdCommId* id, int nranks, int rank, VdComm** out) {
    if (id == nullptr || out == nullptr || nranks < 1 || rank < 0 || rank >= nranks) return -1;
    const Rccl* r = rccl();
    if (r == nullptr) return -10;
    VdComm* c = static_cast<VdComm*>(calloc(1, sizeof(VdComm)));
    if (c == nullptr) return -5;
    VdCommId copy;
    memcpy(&copy, id, sizeof(copy));
    if (r->CommInitRank(&c->comm, nranks, copy, rank) != 0) { free(c); return -11; }
    c->nranks = nranks; c->rank = rank;
    *out = c;
    return 0;
}

// Number of ranks as the RCCL communicator itself reports it (ncclCommCount), not the value passed to vd_comm_create.
extern "C" int vd_comm_size(const VdComm* c) {
    if (c == nullptr) return -1;
    const Rccl* r = rccl();
    int n = c->nranks;
    if (r != nullptr && r->CommCount != nullptr && c->comm != nullptr && r->CommCount(c->comm, &n) != 0) return -11;
    return n;
}

// RCCL's version code (ncclGetVersion: major * 10000 + minor * 100 + patch for 2.9+).
extern "C" int vd_comm_version(int* version) {
    if (version == nullptr) return -1;
    const Rccl* r = rccl();
    if (r == nullptr || r->GetVersion == nullptr) return -10;
    return r->GetVersion(version) == 0 ? 0 : -11;
}
extern "C" int vd_comm_rank(const VdComm* c) { return c ? c->rank : -1; }

// recv[i] = sum over ranks of send[i]; send == recv is allowed (in place).  Asynchronous on `stream`.
extern "C" int vd_comm_allreduce_f32(VdComm* c, const float* send, float* recv, int64_t n, void* stream) {
    if (c == nullptr || n < 0 || (n > 0 && (send == nullptr || recv == nullptr))) return -1;
    if (n == 0) return 0;
    return rccl()->AllReduce(send, recv, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm, stream) == 0 ? 0 : -11;
}

// recv[r * n .. (r+1) * n) = rank r's send[0 .. n): the synthetic clips of every rank before evaluation / saving.
extern "C" int vd_comm_allgather_f32(VdComm* c, const float* send, float* recv, int64_t n, void* stream) {
    if (c == nullptr || n < 0 || (n > 0 && (send == nullptr || recv == nullptr))) return -1;
    if (n == 0) return 0;
    return rccl()->AllGather(send, recv, (size_t)n, /*ncclFloat32*/ 7, c->comm, stream) == 0 ? 0 : -11;
}

extern "C" void vd_comm_free(VdComm* c) {
    if (c == nullptr) return;
    const Rccl* r = rccl();
    if (r != nullptr && c->comm != nullptr) r->CommDestroy(c->comm);
    free(c);
}

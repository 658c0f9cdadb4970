// The one exchange of the path (SURVEY 8b `grnet_allgather`, 8e): the all-gather that reassembles the per-frame records of a sharded clip before
// anything temporal runs (the reference has no counterpart: demo.py:126-188 / batch_generation.py:289-329 are one process on one device).
//
// RCCL is bound at RUN time: the first grnet_comm_* call looks the process's librccl.so.1 up (the copy PyTorch-ROCm has already loaded when the
// host is the Python mirror -- the loader returns that same copy for the soname -- or /opt/rocm/lib's for a plain C host) and takes five entry
// points from it.  libgrnet_hip.so itself carries no RCCL dependency, so the single-GPU product still loads on a box without the library, and
// include/grnet_hip.h needs none of RCCL's headers: the communicator crosses the ABI as an opaque grnet_comm_t (or, adopted, as a void*).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/grnet_hip.h"

namespace {

// The part of RCCL's published C API (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclAllGather, ncclCommDestroy, ncclGetErrorString) this file binds.
constexpr int kUniqueIdBytes = 128;                 // NCCL_UNIQUE_ID_BYTES
constexpr int kNcclChar = 0;                        // ncclInt8 / ncclChar: the exchange moves bytes
struct UniqueId { char internal[kUniqueIdBytes]; };
using Comm = void*;                                  // ncclComm_t
using GetUniqueIdFn = int (*)(UniqueId*);
using CommInitRankFn = int (*)(Comm*, int, UniqueId, int);
using AllGatherFn = int (*)(const void*, void*, size_t, int, Comm, hipStream_t);
using CommDestroyFn = int (*)(Comm);
using GetErrorStringFn = const char* (*)(int);

struct Rccl {
    void* lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    AllGatherFn all_gather = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GetErrorStringFn error_string = nullptr;
    std::string why;                                 // why binding failed
};

thread_local std::string g_error;

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // GRNET_RCCL_LIB: look THIS library up and nothing else (a deployment's own build of RCCL; the tests point it at a name that does not exist)
        const char* forced = getenv("GRNET_RCCL_LIB");
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        std::string last;
        for (const char* n : names) {
            if (forced) n = forced;
            r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
            const char* e = dlerror();                                       // ONE call: dlerror() clears the message it returns
            last = e ? e : "?";
            if (forced) break;
        }
        if (!r.lib) { r.why = std::string(forced ? forced : "librccl.so.1") + " not found: " + last; return; }
        r.get_unique_id = (GetUniqueIdFn)dlsym(r.lib, "ncclGetUniqueId");
        r.comm_init_rank = (CommInitRankFn)dlsym(r.lib, "ncclCommInitRank");
        r.all_gather = (AllGatherFn)dlsym(r.lib, "ncclAllGather");
        r.comm_destroy = (CommDestroyFn)dlsym(r.lib, "ncclCommDestroy");
        r.error_string = (GetErrorStringFn)dlsym(r.lib, "ncclGetErrorString");
        if (!r.get_unique_id || !r.comm_init_rank || !r.all_gather || !r.comm_destroy || !r.error_string) {
            r.why = "librccl.so.1 lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy / ncclGetErrorString";
            r.lib = nullptr;
        }
    });
    return r;
}

int fail(int code, const std::string& msg) { g_error = msg; return code; }

int nccl_fail(const char* what, int rc) {
    Rccl& r = rccl();
    return fail(GRNET_EHIP, std::string(what) + ": " + (r.error_string ? r.error_string(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
}

}  // namespace

struct grnet_comm {
    Comm comm = nullptr;
    int world = 1, rank = 0, device = 0;
    bool owned = true;                               // false: adopted from the host, never destroyed here
};

extern "C" {

int grnet_comm_probe(void) {
    Rccl& r = rccl();
    return r.lib ? 0 : fail(GRNET_ESTATE, r.why);
}

int grnet_comm_unique_id(void* id_out, int id_size) {
    if (!id_out || id_size < kUniqueIdBytes) return fail(GRNET_EINVAL, "grnet_comm_unique_id: the buffer must hold GRNET_COMM_ID_BYTES (128) bytes");
    Rccl& r = rccl();
    if (!r.lib) return fail(GRNET_ESTATE, r.why);
    UniqueId id;
    if (int rc = r.get_unique_id(&id)) return nccl_fail("ncclGetUniqueId", rc);
    std::memcpy(id_out, id.internal, kUniqueIdBytes);
    return 0;
}

int grnet_comm_create(grnet_comm_t** out, const void* id, int world, int rank, int device_id) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return fail(GRNET_EINVAL, "grnet_comm_create: bad argument (need 0 <= rank < world and the 128-byte id of rank 0)");
    Rccl& r = rccl();
    if (!r.lib) return fail(GRNET_ESTATE, r.why);
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device_id) != hipSuccess) return fail(GRNET_EHIP, "grnet_comm_create: hipSetDevice failed");
    UniqueId uid;
    std::memcpy(uid.internal, id, kUniqueIdBytes);
    auto* c = new grnet_comm;
    c->world = world; c->rank = rank; c->device = device_id;
    const int rc = r.comm_init_rank(&c->comm, world, uid, rank);
    (void)hipSetDevice(prev);
    if (rc) { delete c; return nccl_fail("ncclCommInitRank", rc); }
    *out = c;
    return 0;
}

int grnet_comm_adopt(grnet_comm_t** out, void* nccl_comm, int world, int rank) {
    if (!out || !nccl_comm || world < 1 || rank < 0 || rank >= world) return fail(GRNET_EINVAL, "grnet_comm_adopt: bad argument");
    Rccl& r = rccl();
    if (!r.lib) return fail(GRNET_ESTATE, r.why);
    auto* c = new grnet_comm;
    c->comm = nccl_comm; c->world = world; c->rank = rank; c->owned = false;
    (void)hipGetDevice(&c->device);
    *out = c;
    return 0;
}

int grnet_allgather(grnet_comm_t* c, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream) {
    if (!c || !send_dev || !recv_dev) return fail(GRNET_EINVAL, "grnet_allgather: null argument");
    if (bytes_per_rank == 0) return 0;
    Rccl& r = rccl();
    if (!r.lib) return fail(GRNET_ESTATE, r.why);
    if (int rc = r.all_gather(send_dev, recv_dev, bytes_per_rank, kNcclChar, c->comm, (hipStream_t)stream)) return nccl_fail("ncclAllGather", rc);
    return 0;
}

int grnet_comm_info(grnet_comm_t* c, int* world, int* rank) {
    if (!c) return fail(GRNET_EINVAL, "grnet_comm_info: null communicator");
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    return 0;
}

void grnet_comm_destroy(grnet_comm_t* c) {
    if (!c) return;
    if (c->owned && c->comm && rccl().lib) (void)rccl().comm_destroy(c->comm);
    delete c;
}

const char* grnet_comm_last_error(void) { return g_error.c_str(); }

}  // extern "C"

// RCCL behind mi355_collective_t: the fallback transport of the tensor-parallel step when the IPC peer mapping of
// allreduce.hip is not available.  The reference takes the same route under graph capture -- raw ncclAllReduce /
// ncclAllGather on the capture stream when its custom kernel does not apply (rtp_llm/models_py/distributed/rocm_rccl.py:
// 511-572; communicator set-up from a broadcast unique id: rocm_rccl.py:150-260).
//
// librccl is resolved with dlopen at run time: libmi355_decode.so keeps no link-time dependency on it, and a process that
// already holds torch's copy (torch/lib/librccl.so) shares that one by passing its path.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>
#include <new>
#include "internal.h"

void mi355_set_error(const char* fmt, ...);

namespace {
constexpr size_t kIdBytes = 128;            // sizeof(ncclUniqueId) (NCCL_UNIQUE_ID_BYTES)
struct UniqueId { char internal[kIdBytes]; };
// rccl.h enum values used here
constexpr int kNcclSuccess = 0, kNcclFloat16 = 6, kNcclBfloat16 = 9, kNcclUint8 = 1, kNcclSum = 0;

using get_unique_id_t = int (*)(UniqueId*);
using comm_init_rank_t = int (*)(void** comm, int nranks, UniqueId id, int rank);
using comm_destroy_t = int (*)(void* comm);
using all_reduce_t = int (*)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t st);
using all_gather_t = int (*)(const void* send, void* recv, size_t sendcount, int dtype, void* comm, hipStream_t st);
using get_error_string_t = const char* (*)(int);

struct Api {
    void* lib = nullptr;
    get_unique_id_t get_unique_id = nullptr;
    comm_init_rank_t comm_init_rank = nullptr;
    comm_destroy_t comm_destroy = nullptr;
    all_reduce_t all_reduce = nullptr;
    all_gather_t all_gather = nullptr;
    get_error_string_t error_string = nullptr;
};

bool load_api(const char* path, Api* a) {
    a->lib = dlopen(path && *path ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!a->lib && !(path && *path)) a->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!a->lib) { mi355_set_error("rccl: dlopen(%s): %s", path && *path ? path : "librccl.so", dlerror()); return false; }
    a->get_unique_id = (get_unique_id_t)dlsym(a->lib, "ncclGetUniqueId");
    a->comm_init_rank = (comm_init_rank_t)dlsym(a->lib, "ncclCommInitRank");
    a->comm_destroy = (comm_destroy_t)dlsym(a->lib, "ncclCommDestroy");
    a->all_reduce = (all_reduce_t)dlsym(a->lib, "ncclAllReduce");
    a->all_gather = (all_gather_t)dlsym(a->lib, "ncclAllGather");
    a->error_string = (get_error_string_t)dlsym(a->lib, "ncclGetErrorString");
    if (!a->get_unique_id || !a->comm_init_rank || !a->comm_destroy || !a->all_reduce || !a->all_gather) {
        mi355_set_error("rccl: %s does not export the nccl entry points", path && *path ? path : "librccl.so");
        dlclose(a->lib); a->lib = nullptr;
        return false;
    }
    return true;
}
const char* err_of(const Api& a, int rc) { return a.error_string ? a.error_string(rc) : "?"; }
} // namespace

struct mi355_rccl {
    Api   api;
    void* comm;
    int   rank, world;
};

namespace {
int rccl_all_reduce_f16(void* ctx, void* buf, size_t count, mi355_stream_t stream) {
    auto* r = (mi355_rccl*)ctx;
    const int rc = r->api.all_reduce(buf, buf, count, kNcclFloat16, kNcclSum, r->comm, (hipStream_t)stream);
    if (rc != kNcclSuccess) mi355_set_error("rccl: ncclAllReduce: %s", err_of(r->api, rc));
    return rc;
}
int rccl_all_reduce_bf16(void* ctx, void* buf, size_t count, mi355_stream_t stream) {
    auto* r = (mi355_rccl*)ctx;
    const int rc = r->api.all_reduce(buf, buf, count, kNcclBfloat16, kNcclSum, r->comm, (hipStream_t)stream);
    if (rc != kNcclSuccess) mi355_set_error("rccl: ncclAllReduce(bf16): %s", err_of(r->api, rc));
    return rc;
}
int rccl_all_gather(void* ctx, const void* send, void* recv, size_t bytes_per_rank, mi355_stream_t stream) {
    auto* r = (mi355_rccl*)ctx;
    const int rc = r->api.all_gather(send, recv, bytes_per_rank, kNcclUint8, r->comm, (hipStream_t)stream);
    if (rc != kNcclSuccess) mi355_set_error("rccl: ncclAllGather: %s", err_of(r->api, rc));
    return rc;
}
} // namespace

extern "C" size_t mi355_rccl_unique_id_bytes(void) { return kIdBytes; }

extern "C" int mi355_rccl_unique_id(const char* lib_path, void* id_out) {
    if (!id_out) { mi355_set_error("rccl_unique_id: null output"); return MI355_ERR_ARG; }
    Api a;
    if (!load_api(lib_path, &a)) return MI355_ERR_HIP;
    UniqueId id;
    const int rc = a.get_unique_id(&id);
    if (rc != kNcclSuccess) { mi355_set_error("rccl: ncclGetUniqueId: %s", err_of(a, rc)); return MI355_ERR_HIP; }
    memcpy(id_out, &id, kIdBytes);
    return MI355_OK;   // the library stays loaded: the id's bootstrap listener lives in it
}

extern "C" mi355_rccl_t* mi355_rccl_open(const char* lib_path, const void* unique_id, int32_t rank, int32_t world) {
    if (!unique_id || world <= 0 || rank < 0 || rank >= world) { mi355_set_error("rccl_open: rank=%d world=%d", rank, world); return nullptr; }
    auto* r = new (std::nothrow) mi355_rccl();
    if (!r) return nullptr;
    if (!load_api(lib_path, &r->api)) { delete r; return nullptr; }
    UniqueId id;
    memcpy(&id, unique_id, kIdBytes);
    r->rank = rank; r->world = world; r->comm = nullptr;
    const int rc = r->api.comm_init_rank(&r->comm, world, id, rank);
    if (rc != kNcclSuccess || !r->comm) {
        mi355_set_error("rccl_open: ncclCommInitRank(rank %d of %d): %s", rank, world, err_of(r->api, rc));
        delete r;
        return nullptr;
    }
    return r;
}

extern "C" int mi355_rccl_collective(mi355_rccl_t* r, mi355_collective_t* out) {
    if (!r || !out) { mi355_set_error("rccl_collective: null argument"); return MI355_ERR_ARG; }
    out->ctx = r; out->all_reduce_f16 = rccl_all_reduce_f16; out->all_reduce_bf16 = rccl_all_reduce_bf16; out->all_gather = rccl_all_gather;
    out->rank = r->rank; out->world = r->world;
    return MI355_OK;
}

extern "C" void mi355_rccl_close(mi355_rccl_t* r) {
    if (!r) return;
    if (r->comm) r->api.comm_destroy(r->comm);
    delete r;   // the library handle is kept (torch may share it)
}

// Multi-GPU exchange of the C ABI: one RCCL all-reduce of the V x ldk sufficient statistics per outer
// iteration, usable from a host that has no Python / torch.  RCCL is bound at RUN time
// (dlopen): the library has no link-time dependency on it, and inside a PyTorch process the copy torch
// already loaded is the one that gets used.
#include "../../include/pylda_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "comm.h"

namespace {

// the slice of the RCCL / NCCL API that is used (rccl.h: ncclUniqueId is 128 opaque bytes, passed by value)
struct UniqueId { char internal[PYLDA_COMM_ID_BYTES]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);
constexpr int kNcclFloat64 = 8, kNcclSum = 0;     // ncclDataType_t::ncclFloat64, ncclRedOp_t::ncclSum

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllReduceFn all_reduce = nullptr;
    GetErrorStringFn error_string = nullptr;
    std::string why;
};

Rccl& rccl()
{
    static Rccl r;
    if (r.handle || !r.why.empty()) return r;
    const char* env = getenv("PYLDA_RCCL_PATH");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* name : names) {
        if (!name || !*name) continue;
        r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) {
        r.why = "RCCL not found (tried PYLDA_RCCL_PATH, librccl.so.1, librccl.so, /opt/rocm/lib): ";
        const char* e = dlerror();
        r.why += e ? e : "dlopen failed";
        return r;
    }
    r.get_unique_id = (GetUniqueIdFn)dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (CommInitRankFn)dlsym(r.handle, "ncclCommInitRank");
    r.comm_destroy = (CommDestroyFn)dlsym(r.handle, "ncclCommDestroy");
    r.all_reduce = (AllReduceFn)dlsym(r.handle, "ncclAllReduce");
    r.error_string = (GetErrorStringFn)dlsym(r.handle, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce) {
        r.why = "the RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
        r.handle = nullptr;
    }
    return r;
}

std::string nccl_text(const Rccl& r, int code)
{
    char buf[64];
    snprintf(buf, sizeof buf, "RCCL error %d", code);
    std::string s = buf;
    if (r.error_string) {
        s += ": ";
        s += r.error_string(code);
    }
    return s;
}

}  // namespace

namespace pylda {

int comm_unique_id(void* id_out, std::string* err)
{
    Rccl& r = rccl();
    if (!r.handle) { *err = r.why; return PYLDA_ERR_STATE; }
    UniqueId id;
    const int rc = r.get_unique_id(&id);
    if (rc != 0) { *err = nccl_text(r, rc); return PYLDA_ERR_HIP; }
    memcpy(id_out, id.internal, PYLDA_COMM_ID_BYTES);
    return PYLDA_OK;
}

int comm_init(void** comm, const void* id_bytes, int rank, int world, std::string* err)
{
    Rccl& r = rccl();
    if (!r.handle) { *err = r.why; return PYLDA_ERR_STATE; }
    UniqueId id;
    memcpy(id.internal, id_bytes, PYLDA_COMM_ID_BYTES);
    Comm c = nullptr;
    const int rc = r.comm_init_rank(&c, world, id, rank);
    if (rc != 0) { *err = nccl_text(r, rc); return PYLDA_ERR_HIP; }
    *comm = c;
    return PYLDA_OK;
}

void comm_destroy(void* comm)
{
    Rccl& r = rccl();
    if (r.handle && comm) (void)r.comm_destroy((Comm)comm);
}

int comm_allreduce_sum_f64(void* comm, double* device_buffer, size_t count, hipStream_t stream, std::string* err)
{
    Rccl& r = rccl();
    if (!r.handle) { *err = r.why; return PYLDA_ERR_STATE; }
    const int rc = r.all_reduce(device_buffer, device_buffer, count, kNcclFloat64, kNcclSum, (Comm)comm, stream);
    if (rc != 0) { *err = nccl_text(r, rc); return PYLDA_ERR_HIP; }
    return PYLDA_OK;
}

}  // namespace pylda

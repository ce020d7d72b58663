// ecc_rccl.cpp -- the path's one exchange step as an RCCL all-reduce issued by the library itself (host code only).
//
// ref: the sum over pairs at the end of MetricRadonIntermediate::evaluate (EpipolarConsistencyRadonIntermediate.cpp:216-224);
// sharded over the GPUs of a node it becomes one all-reduce of the 8-byte partial sums.  One process per GPU:
// every rank owns a communicator (ncclCommInitRank on its context's device, the 128-byte id made by rank 0 and handed round by
// whatever the job has -- torch.distributed in bench.py) and queues the all-reduce on its context's stream between the sum
// kernel and the kernel that publishes the scalar to the host: pair kernel -> sum -> ncclAllReduce -> publish_scalar_kernel,
// one stream, no host round trip and no Python or c10d call on the step's path (ecc_metric_evaluate_range_allreduce in
// ecc_evaluate.hip).  RCCL is bound at run time (dlopen "librccl.so.1": the copy a process has loaded already -- PyTorch
// ships one -- or ROCm's), so the library has no link-time dependency on it and a process that never opens a communicator
// never loads it.
#include "ecc_capi_internal.h"

#include <dlfcn.h>

#include <mutex>

#define ECC_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

// the five entry points used, with RCCL's own types spelled out (rccl.h: ncclUniqueId = 128 opaque bytes passed BY VALUE,
// ncclComm_t = opaque pointer, ncclResult_t / ncclDataType_t / ncclRedOp_t = enums: ncclSuccess 0, ncclFloat64 8, ncclSum 0)
struct RcclUniqueId {
    char internal[128];
};
typedef int (*get_unique_id_fn)(RcclUniqueId*);
typedef int (*comm_init_rank_fn)(void**, int, RcclUniqueId, int);
typedef int (*all_reduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*comm_destroy_fn)(void*);
typedef const char* (*error_string_fn)(int);
constexpr int RCCL_FLOAT64 = 8, RCCL_SUM = 0;

struct Rccl {
    void* handle = nullptr;
    get_unique_id_fn get_unique_id = nullptr;
    comm_init_rank_fn comm_init_rank = nullptr;
    all_reduce_fn all_reduce = nullptr;
    comm_destroy_fn comm_destroy = nullptr;
    error_string_fn error_string = nullptr;
    std::string error;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);  // the process's own copy first (PyTorch's)
        if (!r.handle) r.handle = dlopen("librccl.so.1", RTLD_NOW);
        if (!r.handle) r.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW);
        if (!r.handle) {
            const char* e = dlerror();
            r.error = std::string("librccl.so.1 cannot be loaded: ") + (e ? e : "?");
            return;
        }
        r.get_unique_id = (get_unique_id_fn)dlsym(r.handle, "ncclGetUniqueId");
        r.comm_init_rank = (comm_init_rank_fn)dlsym(r.handle, "ncclCommInitRank");
        r.all_reduce = (all_reduce_fn)dlsym(r.handle, "ncclAllReduce");
        r.comm_destroy = (comm_destroy_fn)dlsym(r.handle, "ncclCommDestroy");
        r.error_string = (error_string_fn)dlsym(r.handle, "ncclGetErrorString");
        if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) r.error = "librccl.so.1 lacks an entry point";
    });
    return r;
}

int rccl_fail(const char* what, int code)
{
    Rccl& r = rccl();
    std::string msg = std::string(what) + ": " + (r.error_string ? r.error_string(code) : "RCCL error") + " (" + std::to_string(code) + ")";
    return ecc_set_error(ECC_ERR_HIP, msg.c_str());
}

}  // namespace

struct ecc_comm {
    ecc_ctx* ctx = nullptr;
    void* comm = nullptr;
    int rank = 0, world = 1;
};

// Can this process bind RCCL at all?  No collective, no GPU call: ranks agree on the answer BEFORE any of them enters
// ncclCommInitRank, so that a rank without the library does not leave the others waiting inside it.
ECC_EXPORT int ecc_comm_available(void)
{
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(ECC_ERR_UNSUPPORTED, r.error.c_str());
    return ECC_OK;
}

ECC_EXPORT int ecc_comm_unique_id(void* id128)
{
    if (!id128) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(ECC_ERR_UNSUPPORTED, r.error.c_str());
    RcclUniqueId id;
    const int rc = r.get_unique_id(&id);
    if (rc) return rccl_fail("ncclGetUniqueId", rc);
    std::memcpy(id128, id.internal, sizeof(id.internal));
    return ECC_OK;
}

ECC_EXPORT int ecc_comm_create(ecc_ctx* ctx, const void* id128, int rank, int world, ecc_comm** out)
{
    if (!ctx || !id128 || !out) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(ECC_ERR_INVALID_ARGUMENT, "rank outside [0, world)");
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(ECC_ERR_UNSUPPORTED, r.error.c_str());
    int rc = ecc_internal::set_device(ctx);  // the communicator belongs to the context's device
    if (rc) return rc;
    RcclUniqueId id;
    std::memcpy(id.internal, id128, sizeof(id.internal));
    void* comm = nullptr;
    rc = r.comm_init_rank(&comm, world, id, rank);  // collective over the ranks: returns once all of them have called it
    if (rc) return rccl_fail("ncclCommInitRank", rc);
    ecc_comm* c = new (std::nothrow) ecc_comm;
    if (!c) {
        (void)r.comm_destroy(comm);
        return fail(ECC_ERR_OUT_OF_MEMORY, "out of memory");
    }
    c->ctx = ctx;
    c->comm = comm;
    c->rank = rank;
    c->world = world;
    *out = c;
    return ECC_OK;
}

ECC_EXPORT int ecc_comm_destroy(ecc_comm* c)
{
    if (!c) return ECC_OK;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    int rc = 0;
    if (c->comm) rc = rccl().comm_destroy(c->comm);
    delete c;
    return rc ? rccl_fail("ncclCommDestroy", rc) : ECC_OK;
}

// In-place all-reduce (sum) of one float64 in device memory, queued on the communicator's context's stream.
extern "C" int ecc_comm_allreduce_sum_f64(ecc_comm* c, double* value_d)
{
    if (!c || !value_d) return fail(ECC_ERR_INVALID_ARGUMENT, "null argument");
    const int rc = rccl().all_reduce(value_d, value_d, 1, RCCL_FLOAT64, RCCL_SUM, c->comm, c->ctx->stream);
    return rc ? rccl_fail("ncclAllReduce", rc) : ECC_OK;
}

extern "C" ecc_ctx* ecc_comm_context(ecc_comm* c) { return c ? c->ctx : nullptr; }

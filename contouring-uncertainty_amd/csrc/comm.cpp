// cu_comm_*: the gradient exchange of the data-parallel training path as part of the C ABI.
//
// One communicator per process (one process per GPU); collectives are RCCL's, over xGMI inside a node, issued on the HIP
// stream the caller passes (the Python side gives them a stream of their own and orders it behind the kernel stream with
// an event per bucket, cu_hip/comm.py).  RCCL is bound at first use with dlopen, so the library loads -- and every
// single-GPU path runs -- on machines without it.  The reference is single-device: there is no reference interface
// these replace; they are the N-GPU form of BASELINE.json's north_star.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <string.h>

#include "../../include/contour_hip.h"

void cu_set_error(const char* fmt, ...);

namespace {

typedef struct { char internal[128]; } nccl_uid;
typedef void* nccl_comm;
enum { NCCL_FLOAT32 = 7, NCCL_SUM = 0 };       // ncclDataType_t / ncclRedOp_t values of rccl.h

struct Api {
    int (*get_uid)(nccl_uid*);
    int (*init_rank)(nccl_comm*, int, nccl_uid, int);
    int (*all_reduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t);
    int (*reduce_scatter)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t);
    int (*all_gather)(const void*, void*, size_t, int, nccl_comm, hipStream_t);
    int (*destroy)(nccl_comm);
    const char* (*err)(int);
    bool ok;
};

Api* api() {
    static Api a = [] {
        Api x;
        memset(&x, 0, sizeof(x));
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return x;
        x.get_uid = (int (*)(nccl_uid*))dlsym(h, "ncclGetUniqueId");
        x.init_rank = (int (*)(nccl_comm*, int, nccl_uid, int))dlsym(h, "ncclCommInitRank");
        x.all_reduce = (int (*)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t))dlsym(h, "ncclAllReduce");
        x.reduce_scatter = (int (*)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t))dlsym(h, "ncclReduceScatter");
        x.all_gather = (int (*)(const void*, void*, size_t, int, nccl_comm, hipStream_t))dlsym(h, "ncclAllGather");
        x.destroy = (int (*)(nccl_comm))dlsym(h, "ncclCommDestroy");
        x.err = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
        x.ok = x.get_uid && x.init_rank && x.all_reduce && x.reduce_scatter && x.all_gather && x.destroy;
        return x;
    }();
    return &a;
}

#define CU_COMM_API()                                                                           \
    Api* A = api();                                                                             \
    if (!A->ok) { cu_set_error("cu_comm: librccl.so.1 not found or incomplete"); return -38; /* -ENOSYS */ }
#define CU_COMM_CALL(expr, what)                                                                \
    do {                                                                                        \
        const int rc__ = (expr);                                                                \
        if (rc__ != 0) {                                                                        \
            cu_set_error("cu_comm: %s failed: %s", what, A->err ? A->err(rc__) : "rccl error"); \
            return -5; /* -EIO */                                                               \
        }                                                                                       \
    } while (0)

}  // namespace

struct cu_comm {
    nccl_comm comm;
    int rank, world;
};

extern "C" int cu_comm_unique_id(void* id128) {
    CU_COMM_API();
    if (!id128) { cu_set_error("cu_comm_unique_id: null pointer"); return -22; }
    nccl_uid id;
    CU_COMM_CALL(A->get_uid(&id), "ncclGetUniqueId");
    memcpy(id128, id.internal, sizeof(id.internal));
    return 0;
}

extern "C" int cu_comm_init(int rank, int world, const void* id128, cu_comm_t** out) {
    CU_COMM_API();
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) { cu_set_error("cu_comm_init: bad argument"); return -22; }
    nccl_uid id;
    memcpy(id.internal, id128, sizeof(id.internal));
    cu_comm* c = new cu_comm{nullptr, rank, world};
    const int rc = A->init_rank(&c->comm, world, id, rank);
    if (rc != 0) {
        cu_set_error("cu_comm_init: ncclCommInitRank failed: %s", A->err ? A->err(rc) : "rccl error");
        delete c;
        return -5;
    }
    *out = c;
    return 0;
}

extern "C" int cu_comm_allreduce_bucket(cu_comm_t* c, float* buf, size_t n, void* stream) {
    CU_COMM_API();
    if (!c || !buf || n == 0) { cu_set_error("cu_comm_allreduce_bucket: bad argument"); return -22; }
    CU_COMM_CALL(A->all_reduce(buf, buf, n, NCCL_FLOAT32, NCCL_SUM, c->comm, reinterpret_cast<hipStream_t>(stream)),
                 "ncclAllReduce");
    return 0;
}

extern "C" int cu_comm_reduce_scatter_bucket(cu_comm_t* c, const float* send, float* recv, size_t n_per_rank, void* stream) {
    CU_COMM_API();
    if (!c || !send || !recv || n_per_rank == 0) { cu_set_error("cu_comm_reduce_scatter_bucket: bad argument"); return -22; }
    CU_COMM_CALL(A->reduce_scatter(send, recv, n_per_rank, NCCL_FLOAT32, NCCL_SUM, c->comm,
                                   reinterpret_cast<hipStream_t>(stream)), "ncclReduceScatter");
    return 0;
}

extern "C" int cu_comm_allgather_bucket(cu_comm_t* c, const float* send, float* recv, size_t n_per_rank, void* stream) {
    CU_COMM_API();
    if (!c || !send || !recv || n_per_rank == 0) { cu_set_error("cu_comm_allgather_bucket: bad argument"); return -22; }
    CU_COMM_CALL(A->all_gather(send, recv, n_per_rank, NCCL_FLOAT32, c->comm, reinterpret_cast<hipStream_t>(stream)),
                 "ncclAllGather");
    return 0;
}

extern "C" int cu_comm_destroy(cu_comm_t* c) {
    if (!c) return 0;
    Api* A = api();
    if (A->ok && c->comm) A->destroy(c->comm);
    delete c;
    return 0;
}

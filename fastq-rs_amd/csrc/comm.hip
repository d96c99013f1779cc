// comm.hip — the exchange steps of the byte-range sharded modes under the C ABI: RCCL collectives over xGMI, called
// directly (ncclAllGather of the ranks' carry words, ncclAllReduce(ncclUint64, ncclSum) of counts and histograms) on the
// context's stream.  The reference's counterpart is the gather of the workers' results at the end of
// Parser::parallel_each (src/lib.rs:553-559).  RCCL is bound at run time (dlopen: a process that already holds torch's
// copy of librccl gets that one); a host that never shards never loads it.  The payloads are tiny (7 words per rank,
// ~320 KB of histograms): latency, not link bandwidth, is what these cost.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>

#include "ctx.h"

namespace {
typedef struct ncclComm *comm_t;
typedef struct { char internal[128]; } unique_id;   // NCCL_UNIQUE_ID_BYTES
enum { kUint8 = 1, kUint64 = 5, kSum = 0, kMin = 3 };  // ncclDataType_t / ncclRedOp_t values of rccl.h
struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(unique_id *) = nullptr;
    int (*CommInitRank)(comm_t *, int, unique_id, int) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
bool load_rccl() {
    if (g_rccl.h) return true;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;  // a copy the process already holds (torch's) first
    for (const char *n : names)
        if (!h && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return false;
    Rccl r;
    r.h = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.AllGather) return false;
    g_rccl = r;
    return true;
}
fqh_status rccl_fail(fqh_ctx *ctx, const char *what, int rc) {
    if (ctx) ctx->err = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
    return FQH_E_DEVICE;
}
}  // namespace

extern "C" {

fqh_status fqh_comm_unique_id(uint8_t id[FQH_COMM_ID_BYTES]) {
    if (!id) return FQH_E_ARG;
    if (!load_rccl()) return FQH_E_DEVICE;
    unique_id u;
    if (g_rccl.GetUniqueId(&u) != 0) return FQH_E_DEVICE;
    memcpy(id, u.internal, FQH_COMM_ID_BYTES);
    return FQH_OK;
}

fqh_status fqh_comm_create(fqh_ctx *ctx, int n_ranks, int rank, const uint8_t id[FQH_COMM_ID_BYTES], fqh_comm **out) {
    if (!ctx || !out || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return FQH_E_ARG;
    *out = nullptr;
    if (!load_rccl()) {
        ctx->err = "librccl.so could not be loaded";
        return FQH_E_DEVICE;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    unique_id u;
    memcpy(u.internal, id, FQH_COMM_ID_BYTES);
    comm_t c = nullptr;
    const int rc = g_rccl.CommInitRank(&c, n_ranks, u, rank);
    if (rc != 0) return rccl_fail(ctx, "ncclCommInitRank", rc);
    *out = (fqh_comm *)c;
    return FQH_OK;
}

void fqh_comm_destroy(fqh_comm *comm) {
    if (comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((comm_t)comm);
}

fqh_status fqh_allgather(fqh_ctx *ctx, fqh_comm *comm, const void *d_send, void *d_recv, uint64_t bytes_per_rank) {
    if (!ctx || !comm || !d_send || !d_recv) return FQH_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int rc = g_rccl.AllGather(d_send, d_recv, bytes_per_rank, kUint8, (comm_t)comm, ctx->stream);
    if (rc != 0) return rccl_fail(ctx, "ncclAllGather", rc);
    return FQH_OK;
}

fqh_status fqh_allreduce_u64(fqh_ctx *ctx, fqh_comm *comm, uint64_t *d_buf, uint64_t n) {
    if (!ctx || !comm || (n && !d_buf)) return FQH_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int rc = g_rccl.AllReduce(d_buf, d_buf, n, kUint64, kSum, (comm_t)comm, ctx->stream);
    if (rc != 0) return rccl_fail(ctx, "ncclAllReduce", rc);
    return FQH_OK;
}

fqh_status fqh_allreduce_min_u64(fqh_ctx *ctx, fqh_comm *comm, uint64_t *d_buf, uint64_t n) {
    if (!ctx || !comm || (n && !d_buf)) return FQH_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int rc = g_rccl.AllReduce(d_buf, d_buf, n, kUint64, kMin, (comm_t)comm, ctx->stream);
    if (rc != 0) return rccl_fail(ctx, "ncclAllReduce(min)", rc);
    return FQH_OK;
}

fqh_status fqh_sync(fqh_ctx *ctx) {
    if (!ctx) return FQH_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FQH_OK;
}

}  // extern "C"

// C1 + C2 — RCCL over xGMI, one process per GPU.  librccl is loaded lazily with dlopen so that
// single-GPU users never pay for it.  All traffic runs on the ctx's communication stream;
// a "ticket" (HIP event) lets the compute stream wait for one specific exchange, which is how
// the Pearson row-block schedule overlaps the arrival of shard s+1 with the GEMM on shard s.
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <rccl/rccl.h>

#include "common.hpp"

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_api;

int load_rccl() {
    if (g_api.handle) return SKR_OK;
    // SEEKR_RCCL_LIB: test hook — tests/mock_rccl stands in for RCCL so that several ranks can share the
    // one GPU of a test box (RCCL refuses that); never set in production
    const char* override_path = getenv("SEEKR_RCCL_LIB");
    const char* names[] = {override_path ? override_path : "librccl.so.1", "librccl.so.1", "librccl.so",
                           "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h || override_path) break;  // an override that does not load is an error, not a fallback
    }
    if (!h) return skr_set_error(SKR_ERR_COMM, "cannot load librccl: %s", dlerror());
#define SYM(field, name)                                                          \
    do {                                                                          \
        *(void**)(&g_api.field) = dlsym(h, name);                                 \
        if (!g_api.field) return skr_set_error(SKR_ERR_COMM, "librccl lacks %s", name); \
    } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_api.handle = h;
    return SKR_OK;
}

#define SKR_NCCL(call)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (call);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return skr_set_error(SKR_ERR_COMM, "%s failed: %s", #call, g_api.GetErrorString(r_)); \
    } while (0)

int need_comm(skr_ctx* ctx) {
    SKR_REQUIRE(ctx, "ctx is NULL");
    if (!ctx->comm) return skr_set_error(SKR_ERR_COMM, "communicator not initialised (call skr_comm_init)");
    return skr_activate(ctx);
}

}  // namespace

extern "C" int skr_comm_unique_id(char id[128]) {
    SKR_REQUIRE(id, "id is NULL");
    SKR_TRY(load_rccl());
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId uid;
    SKR_NCCL(g_api.GetUniqueId(&uid));
    memcpy(id, &uid, 128);
    return SKR_OK;
}

extern "C" int skr_comm_init(skr_ctx* ctx, int nranks, int rank, const char id[128]) {
    SKR_REQUIRE(ctx && id, "NULL argument");
    SKR_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
    SKR_REQUIRE(!ctx->comm, "communicator already initialised");
    SKR_TRY(load_rccl());
    SKR_TRY(skr_activate(ctx));
    ncclUniqueId uid;
    memcpy(&uid, id, 128);
    ncclComm_t comm = nullptr;
    SKR_NCCL(g_api.CommInitRank(&comm, nranks, uid, rank));
    ctx->comm = comm;
    ctx->nranks = nranks;
    ctx->rank = rank;
    return SKR_OK;
}

extern "C" int skr_comm_destroy(skr_ctx* ctx) {
    if (!ctx || !ctx->comm) return SKR_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->comm_stream);
    g_api.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->nranks = 1;
    ctx->rank = 0;
    return SKR_OK;
}

extern "C" int skr_comm_sendrecv(skr_ctx* ctx, const skr_mat* src, int64_t srow0, int64_t snrows, int dst_rank,
                                 skr_mat* dst, int64_t drow0, int64_t dnrows, int src_rank, int64_t* ticket) {
    SKR_TRY(need_comm(ctx));
    const bool do_send = dst_rank >= 0 && snrows > 0, do_recv = src_rank >= 0 && dnrows > 0;
    if (do_send) {
        SKR_REQUIRE(src && src->ctx == ctx, "src missing");
        SKR_REQUIRE(srow0 >= 0 && srow0 + snrows <= src->rows, "send rows out of range");
        SKR_REQUIRE(dst_rank < ctx->nranks, "dst_rank out of range");
    }
    if (do_recv) {
        SKR_REQUIRE(dst && dst->ctx == ctx, "dst missing");
        SKR_REQUIRE(drow0 >= 0 && drow0 + dnrows <= dst->rows, "recv rows out of range");
        SKR_REQUIRE(src_rank < ctx->nranks, "src_rank out of range");
    }
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    // the data being sent was produced on the compute stream
    hipEvent_t ready;
    SKR_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    SKR_HIP(hipEventRecord(ready, ctx->stream));
    SKR_HIP(hipStreamWaitEvent(ctx->comm_stream, ready, 0));
    SKR_HIP(hipEventDestroy(ready));
    SKR_NCCL(g_api.GroupStart());
    if (do_send) {
        const size_t rb = (size_t)src->cols * src->elem();
        SKR_NCCL(g_api.Send((const char*)src->data + (size_t)srow0 * rb, (size_t)snrows * rb, ncclUint8, dst_rank, comm,
                            ctx->comm_stream));
    }
    if (do_recv) {
        const size_t rb = (size_t)dst->cols * dst->elem();
        SKR_NCCL(g_api.Recv((char*)dst->data + (size_t)drow0 * rb, (size_t)dnrows * rb, ncclUint8, src_rank, comm,
                            ctx->comm_stream));
    }
    SKR_NCCL(g_api.GroupEnd());
    hipEvent_t done;
    SKR_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    SKR_HIP(hipEventRecord(done, ctx->comm_stream));
    ctx->tickets.push_back(done);
    if (ticket) *ticket = (int64_t)ctx->tickets.size() - 1;
    return SKR_OK;
}

// All-gather of row shards of unequal size: rank g owns rows [bounds[g], bounds[g+1]) of `full`.
// One ncclGroup holds the sends to and the receives from every peer, so all xGMI links of the GPU
// carry traffic at once (the links are point-to-point: P-1 peers = P-1 links); the own shard is a
// device copy on the same stream.
extern "C" int skr_comm_allgather_rows(skr_ctx* ctx, const skr_mat* shard, skr_mat* full, const int64_t* bounds,
                                       int64_t* ticket) {
    SKR_TRY(need_comm(ctx));
    SKR_REQUIRE(shard && full && bounds && shard->ctx == ctx && full->ctx == ctx, "NULL or foreign argument");
    SKR_REQUIRE(shard->cols == full->cols && shard->elem() == full->elem(), "shard and full matrix differ in row layout");
    const int P = ctx->nranks, me = ctx->rank;
    for (int g = 0; g < P; g++) SKR_REQUIRE(bounds[g] <= bounds[g + 1], "bounds must not decrease");
    SKR_REQUIRE(bounds[0] == 0 && bounds[P] <= full->rows, "bounds exceed the full matrix");
    SKR_REQUIRE(shard->rows == bounds[me + 1] - bounds[me], "shard has %lld rows, bounds say %lld", (long long)shard->rows,
                (long long)(bounds[me + 1] - bounds[me]));
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    const size_t rb = (size_t)full->cols * full->elem();
    hipEvent_t ready;
    SKR_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    SKR_HIP(hipEventRecord(ready, ctx->stream));
    SKR_HIP(hipStreamWaitEvent(ctx->comm_stream, ready, 0));
    SKR_HIP(hipEventDestroy(ready));
    if (shard->rows && shard->data != (char*)full->data + (size_t)bounds[me] * rb)
        SKR_HIP(hipMemcpyAsync((char*)full->data + (size_t)bounds[me] * rb, shard->data, (size_t)shard->rows * rb,
                               hipMemcpyDeviceToDevice, ctx->comm_stream));
    SKR_NCCL(g_api.GroupStart());
    for (int s = 1; s < P; s++) {
        const int dst = (me - s + P) % P, src = (me + s) % P;
        if (shard->rows) SKR_NCCL(g_api.Send(shard->data, (size_t)shard->rows * rb, ncclUint8, dst, comm, ctx->comm_stream));
        const int64_t n = bounds[src + 1] - bounds[src];
        if (n)
            SKR_NCCL(g_api.Recv((char*)full->data + (size_t)bounds[src] * rb, (size_t)n * rb, ncclUint8, src, comm,
                                ctx->comm_stream));
    }
    SKR_NCCL(g_api.GroupEnd());
    hipEvent_t done;
    SKR_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    SKR_HIP(hipEventRecord(done, ctx->comm_stream));
    ctx->tickets.push_back(done);
    if (ticket) *ticket = (int64_t)ctx->tickets.size() - 1;
    return SKR_OK;
}

extern "C" int skr_comm_wait(skr_ctx* ctx, int64_t ticket) {
    SKR_TRY(need_comm(ctx));
    SKR_REQUIRE(ticket >= 0 && ticket < (int64_t)ctx->tickets.size() && ctx->tickets[ticket], "unknown ticket");
    SKR_HIP(hipStreamWaitEvent(ctx->stream, ctx->tickets[ticket], 0));
    // a ticket is waited on once: the dependency is now in the compute stream, release the event
    (void)hipEventDestroy(ctx->tickets[ticket]);
    ctx->tickets[ticket] = nullptr;
    return SKR_OK;
}

extern "C" int skr_comm_allreduce_f64(skr_ctx* ctx, double* values, int n, int op) {
    SKR_TRY(need_comm(ctx));
    SKR_REQUIRE(values && n > 0 && n <= 32, "need 1..32 values");
    SKR_REQUIRE(op >= 0 && op <= 2, "op must be 0 (sum), 1 (max) or 2 (min)");
    double* dbuf = reinterpret_cast<double*>(ctx->d_flags + 32);  // 128 bytes of the flag block
    static_assert(sizeof(double) * 16 <= 32 * sizeof(uint32_t), "flag block too small");
    SKR_REQUIRE(n <= 16, "need 1..16 values");
    SKR_HIP(hipStreamSynchronize(ctx->stream));
    SKR_HIP(hipMemcpyAsync(dbuf, values, n * sizeof(double), hipMemcpyHostToDevice, ctx->comm_stream));
    const ncclRedOp_t rop = op == 0 ? ncclSum : (op == 1 ? ncclMax : ncclMin);
    SKR_NCCL(g_api.AllReduce(dbuf, dbuf, (size_t)n, ncclFloat64, rop, (ncclComm_t)ctx->comm, ctx->comm_stream));
    SKR_HIP(hipMemcpyAsync(values, dbuf, n * sizeof(double), hipMemcpyDeviceToHost, ctx->comm_stream));
    SKR_HIP(hipStreamSynchronize(ctx->comm_stream));
    return SKR_OK;
}

extern "C" int skr_comm_barrier(skr_ctx* ctx) {
    double v = 0.0;
    SKR_TRY(skr_ctx_sync(ctx));
    return skr_comm_allreduce_f64(ctx, &v, 1, 0);
}

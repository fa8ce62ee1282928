// C1 + C2 — RCCL over xGMI, one process per GPU.  librccl is loaded lazily with dlopen so that
// single-GPU users never pay for it.  All traffic runs on the ctx's communication stream;
// a "ticket" (HIP event) lets the compute stream wait for one specific exchange, which is how
// the Pearson row-block schedule overlaps the arrival of shard s+1 with the GEMM on shard s.
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
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

std::mutex g_api_lock;  // several host threads, one per GPU, may come here at once (seekr_amd/multi.py)

int load_rccl() {
    std::lock_guard<std::mutex> guard(g_api_lock);
    if (g_api.handle) return SKR_OK;
    // SEEKR_RCCL_LIB: test hook, honoured only under SEEKR_TEST_HOOKS=1 — tests/mock_rccl stands in for RCCL so that
    // several ranks can share the one GPU of a test box (RCCL refuses that); never set in production
    const char* hooks = getenv("SEEKR_TEST_HOOKS");
    const char* override_path = hooks && atoi(hooks) == 1 ? getenv("SEEKR_RCCL_LIB") : nullptr;
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

// completion event of what was just enqueued on comm_stream; *ticket (optional) identifies it for skr_comm_wait.
// With ticket == NULL nothing is recorded: the exchange is ordered on comm_stream only (fire and forget).
int issue_ticket(skr_ctx* ctx, int64_t* ticket, bool vec = false) {
    if (!ticket) return SKR_OK;
    int slot;
    if (!ctx->free_tickets.empty()) {
        slot = ctx->free_tickets.back();
        ctx->free_tickets.pop_back();
    } else {
        SKR_REQUIRE(ctx->tickets.size() < 65536, "more than 65 536 exchanges are waiting to be waited on");
        ctx->tickets.emplace_back();
        slot = (int)ctx->tickets.size() - 1;
    }
    skr_ctx::Ticket& t = ctx->tickets[slot];
    if (!t.ev) SKR_HIP(hipEventCreateWithFlags(&t.ev, hipEventDisableTiming));
    SKR_HIP(hipEventRecord(t.ev, ctx->comm_stream));
    t.live = true;
    t.vec = vec;
    t.gen++;
    *ticket = ((int64_t)t.gen << 16) | slot;
    return SKR_OK;
}

// the data being sent was produced on the compute stream: comm_stream waits for what is enqueued there now
int comm_after_compute(skr_ctx* ctx) {
    hipEvent_t ready;
    SKR_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    SKR_HIP(hipEventRecord(ready, ctx->stream));
    SKR_HIP(hipStreamWaitEvent(ctx->comm_stream, ready, 0));
    SKR_HIP(hipEventDestroy(ready));
    return SKR_OK;
}

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
    SKR_TRY(comm_after_compute(ctx));
    // "comm_xfer": on the communication stream, from the moment the data is ready to the end of the transfer (peer's
    // lateness included); vectors of the column-sum chain (one row) are booked separately as "comm_vec"
    const bool vec = (do_send ? snrows : dnrows) <= 1;
    SkrProfScope prof(ctx, vec ? "comm_vec" : "comm_xfer", ctx->comm_stream);
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
    return issue_ticket(ctx, ticket, vec);
}

// Several exchanges as ONE grouped RCCL operation: xGMI is point-to-point, so transfers to / from different peers
// run on different links at once (the half-ring posts all its shifts this way when it has the receive buffers).
extern "C" int skr_comm_exchange(skr_ctx* ctx, int n, const skr_mat* const* src, const int64_t* srow0, const int64_t* snrows,
                                 const int* dst_rank, skr_mat* const* dst, const int64_t* drow0, const int64_t* dnrows,
                                 const int* src_rank, int64_t* ticket) {
    SKR_TRY(need_comm(ctx));
    SKR_REQUIRE(n >= 0 && n <= 64 && (n == 0 || (src && srow0 && snrows && dst_rank && dst && drow0 && dnrows && src_rank)),
                "bad exchange list");
    for (int i = 0; i < n; i++) {
        if (dst_rank[i] >= 0 && snrows[i] > 0) {
            SKR_REQUIRE(src[i] && src[i]->ctx == ctx && srow0[i] >= 0 && srow0[i] + snrows[i] <= src[i]->rows &&
                            dst_rank[i] < ctx->nranks, "exchange %d: bad send", i);
        }
        if (src_rank[i] >= 0 && dnrows[i] > 0) {
            SKR_REQUIRE(dst[i] && dst[i]->ctx == ctx && drow0[i] >= 0 && drow0[i] + dnrows[i] <= dst[i]->rows &&
                            src_rank[i] < ctx->nranks, "exchange %d: bad receive", i);
        }
    }
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    SKR_TRY(comm_after_compute(ctx));
    SkrProfScope prof(ctx, "comm_xfer", ctx->comm_stream);
    SKR_NCCL(g_api.GroupStart());
    for (int i = 0; i < n; i++) {
        if (dst_rank[i] >= 0 && snrows[i] > 0) {
            const size_t rb = (size_t)src[i]->cols * src[i]->elem();
            SKR_NCCL(g_api.Send((const char*)src[i]->data + (size_t)srow0[i] * rb, (size_t)snrows[i] * rb, ncclUint8, dst_rank[i],
                                comm, ctx->comm_stream));
        }
        if (src_rank[i] >= 0 && dnrows[i] > 0) {
            const size_t rb = (size_t)dst[i]->cols * dst[i]->elem();
            SKR_NCCL(g_api.Recv((char*)dst[i]->data + (size_t)drow0[i] * rb, (size_t)dnrows[i] * rb, ncclUint8, src_rank[i], comm,
                                ctx->comm_stream));
        }
    }
    SKR_NCCL(g_api.GroupEnd());
    return issue_ticket(ctx, ticket);
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
    SKR_TRY(comm_after_compute(ctx));
    SkrProfScope prof(ctx, "comm_xfer", ctx->comm_stream);
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
    return issue_ticket(ctx, ticket);
}

extern "C" int skr_comm_wait(skr_ctx* ctx, int64_t ticket) {
    SKR_TRY(need_comm(ctx));
    const int64_t slot = ticket & 0xFFFF;
    SKR_REQUIRE(ticket >= 0 && slot < (int64_t)ctx->tickets.size() && ctx->tickets[slot].live &&
                    (int64_t)ctx->tickets[slot].gen == (ticket >> 16), "unknown ticket (each ticket is waited on once)");
    {
        // "comm_wait": on the compute stream, from the end of the work enqueued before the wait to the moment the exchange
        // has arrived = the part of a transfer that no kernel hid (0 when the data was there already)
        SkrProfScope prof(ctx, ctx->tickets[slot].vec ? "comm_wait_vec" : "comm_wait");
        SKR_HIP(hipStreamWaitEvent(ctx->stream, ctx->tickets[slot].ev, 0));
    }
    // a ticket is waited on once: the dependency is now in the compute stream; the slot and its event are recycled
    ctx->tickets[slot].live = false;
    ctx->free_tickets.push_back((int)slot);
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


// ---------------------------------------------------------------------------------------
// One host process, several GPUs (seekr_amd/multi.py): rows can move between two ctxs of the SAME process without RCCL —
// a peer copy over xGMI on the receiving ctx's communication stream (SDMA: no CU is taken from the contraction), ordered by
// events that either ctx's streams may wait for.  The host threads hand each other the handles (matrix, event) in memory.
struct skr_event {
    hipEvent_t ev = nullptr;
    int device = 0;
};

// An event at the current end of the ctx's compute (on_comm_stream == 0) or communication stream.
extern "C" int skr_event_record(skr_ctx* ctx, int on_comm_stream, skr_event** out) {
    SKR_REQUIRE(ctx && out, "NULL argument");
    *out = nullptr;
    SKR_TRY(skr_activate(ctx));
    skr_event* e = new skr_event();
    e->device = ctx->device;
    hipError_t rc = hipEventCreateWithFlags(&e->ev, hipEventDisableTiming);
    if (rc == hipSuccess) rc = hipEventRecord(e->ev, on_comm_stream ? ctx->comm_stream : ctx->stream);
    if (rc != hipSuccess) {
        if (e->ev) (void)hipEventDestroy(e->ev);
        delete e;
        return skr_set_error(SKR_ERR_HIP, "recording an event failed: %s", hipGetErrorString(rc));
    }
    *out = e;
    return SKR_OK;
}

// The ctx's compute (0) or communication (1) stream waits for the event — which may have been recorded on another ctx,
// i.e. another GPU of this process.
extern "C" int skr_event_wait(skr_ctx* ctx, int on_comm_stream, const skr_event* ev) {
    SKR_REQUIRE(ctx && ev && ev->ev, "NULL argument");
    SKR_TRY(skr_activate(ctx));
    SKR_HIP(hipStreamWaitEvent(on_comm_stream ? ctx->comm_stream : ctx->stream, ev->ev, 0));
    return SKR_OK;
}

extern "C" int skr_event_free(skr_event* ev) {
    if (!ev) return SKR_OK;
    (void)hipSetDevice(ev->device);
    if (ev->ev) (void)hipEventDestroy(ev->ev);  // an event still being waited for is released when it completes
    delete ev;
    return SKR_OK;
}

// Rows [srow0, srow0 + nrows) of `src` (a matrix of ANOTHER ctx of this process, or of the same one) -> rows drow0.. of
// `dst`, enqueued on dst's ctx's communication stream.  Direct peer access is switched on for the pair on first use where
// the hardware offers it (xGMI); the copy itself works either way.  The caller orders it: skr_event_wait(dst ctx, 1, the
// source's "rows are ready" event) before, skr_event_record(dst ctx, 1) after.
extern "C" int skr_peer_copy_rows(skr_mat* dst, int64_t drow0, const skr_mat* src, int64_t srow0, int64_t nrows) {
    SKR_REQUIRE(dst && src, "NULL argument");
    SKR_REQUIRE((size_t)dst->cols * dst->elem() == (size_t)src->cols * src->elem(), "rows of different byte length");
    SKR_REQUIRE(drow0 >= 0 && srow0 >= 0 && nrows >= 0 && drow0 + nrows <= dst->rows && srow0 + nrows <= src->rows,
                "row range outside the matrices");
    skr_ctx* dctx = dst->ctx;
    SKR_TRY(skr_activate(dctx));
    if (nrows == 0) return SKR_OK;
    const size_t rb = (size_t)dst->cols * dst->elem();
    const int sdev = src->ctx->device, ddev = dctx->device;
    if (sdev != ddev) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, ddev, sdev) == hipSuccess && can) {
            const hipError_t e = hipDeviceEnablePeerAccess(sdev, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();  // staged copies still work
            else (void)hipGetLastError();
        }
    }
    SkrProfScope prof(dctx, nrows <= 1 ? "comm_vec" : "comm_xfer", dctx->comm_stream);
    SKR_HIP(hipMemcpyPeerAsync((char*)dst->data + (size_t)drow0 * rb, ddev, (const char*)src->data + (size_t)srow0 * rb, sdev,
                               (size_t)nrows * rb, dctx->comm_stream));
    return SKR_OK;
}

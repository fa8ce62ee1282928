// A stand-in for librccl used ONLY by tests (SEEKR_RCCL_LIB points the product's dlopen at it): it
// lets several ranks share ONE GPU, which RCCL itself refuses ("duplicate GPU"), so that the N > 1
// path — launch rendezvous, the C-ABI comm layer with its streams and tickets, the HIP engine under
// the sharded orchestration — can run for real on a 1-GPU box.  Transport: files in /dev/shm, one
// per message.  Two modes: by default every call synchronises the stream it is given (stream
// ordering kept trivially); with MOCK_RCCL_ASYNC=1 the operations are ENQUEUED on the stream like
// RCCL's kernels — a pinned staging copy plus a host function that publishes / awaits the file — so
// that a missing event or ticket wait in the product shows up as wrong data.  Timing under this
// mock means nothing.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;

struct MockComm {
    int nranks, rank;
    std::string base;                       // /dev/shm/seekr_mock_<id>
    std::map<int, uint64_t> sent, received;  // per-peer message sequence numbers
    uint64_t reduce_round = 0;
};
typedef MockComm* ncclComm_t;

namespace {
struct Op { bool send; void* buf; size_t bytes; int peer; MockComm* comm; hipStream_t stream; };
thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_ops;

size_t dtype_size(ncclDataType_t t) {
    switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclFloat16: return 2; case ncclInt32: case ncclUint32: case ncclFloat32: return 4; default: return 8; }
}
bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
void wait_for(const std::string& p) {
    for (int i = 0; !exists(p); i++) {
        if (i > 600000) { fprintf(stderr, "mock_rccl: timed out waiting for %s\n", p.c_str()); abort(); }
        usleep(100);
    }
}
void publish(const std::string& path, const void* data, size_t n) {
    const std::string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || (n && fwrite(data, 1, n, f) != n)) { fprintf(stderr, "mock_rccl: cannot write %s\n", tmp.c_str()); abort(); }
    fclose(f);
    rename(tmp.c_str(), path.c_str());
}
std::vector<char> consume(const std::string& path, size_t n) {
    wait_for(path);
    std::vector<char> v(n);
    FILE* f = fopen(path.c_str(), "rb");
    if (!f || (n && fread(v.data(), 1, n, f) != n)) { fprintf(stderr, "mock_rccl: short read of %s (%zu bytes expected)\n", path.c_str(), n); abort(); }
    fclose(f);
    unlink(path.c_str());
    return v;
}
// ---- asynchronous mode: stream-ordered, the host thread does not wait
struct AsyncMsg {
    std::string path;
    void* pinned;
    size_t bytes;
};
void cb_publish(void* p) {
    AsyncMsg* m = static_cast<AsyncMsg*>(p);
    publish(m->path, m->pinned, m->bytes);
}
void cb_consume(void* p) {
    AsyncMsg* m = static_cast<AsyncMsg*>(p);
    const std::vector<char> v = consume(m->path, m->bytes);
    if (m->bytes) memcpy(m->pinned, v.data(), m->bytes);
}
ncclResult_t run_async(const Op& op) {
    MockComm* c = op.comm;
    char name[512];
    AsyncMsg* m = new AsyncMsg();  // lives until the process ends: tests move a few hundred MB at most
    m->bytes = op.bytes;
    if (hipHostMalloc(&m->pinned, op.bytes ? op.bytes : 16, hipHostMallocDefault) != hipSuccess) return ncclUnhandledCudaError;
    if (op.send) {
        snprintf(name, sizeof name, "%s_m_%d_%d_%llu", c->base.c_str(), c->rank, op.peer, (unsigned long long)c->sent[op.peer]++);
        m->path = name;
        if (op.bytes && hipMemcpyAsync(m->pinned, op.buf, op.bytes, hipMemcpyDeviceToHost, op.stream) != hipSuccess) return ncclUnhandledCudaError;
        if (hipLaunchHostFunc(op.stream, cb_publish, m) != hipSuccess) return ncclUnhandledCudaError;
    } else {
        snprintf(name, sizeof name, "%s_m_%d_%d_%llu", c->base.c_str(), op.peer, c->rank, (unsigned long long)c->received[op.peer]++);
        m->path = name;
        if (hipLaunchHostFunc(op.stream, cb_consume, m) != hipSuccess) return ncclUnhandledCudaError;
        if (op.bytes && hipMemcpyAsync(op.buf, m->pinned, op.bytes, hipMemcpyHostToDevice, op.stream) != hipSuccess) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}
bool async_mode() {
    static const bool on = getenv("MOCK_RCCL_ASYNC") && atoi(getenv("MOCK_RCCL_ASYNC")) != 0;
    return on;
}

ncclResult_t run(const Op& op) {
    if (async_mode()) return run_async(op);
    MockComm* c = op.comm;
    if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    char name[512];
    if (op.send) {
        std::vector<char> host(op.bytes);
        if (op.bytes && hipMemcpy(host.data(), op.buf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        snprintf(name, sizeof name, "%s_m_%d_%d_%llu", c->base.c_str(), c->rank, op.peer, (unsigned long long)c->sent[op.peer]++);
        publish(name, host.data(), op.bytes);
    } else {
        snprintf(name, sizeof name, "%s_m_%d_%d_%llu", c->base.c_str(), op.peer, c->rank, (unsigned long long)c->received[op.peer]++);
        const std::vector<char> host = consume(name, op.bytes);
        if (op.bytes && hipMemcpy(op.buf, host.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}
ncclResult_t enqueue(const Op& op) {
    if (g_group_depth > 0) { g_ops.push_back(op); return ncclSuccess; }
    return run(op);
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "%d_%ld", (int)getpid(), (long)random());
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    MockComm* c = new MockComm();
    c->nranks = nranks;
    c->rank = rank;
    c->base = std::string("/dev/shm/seekr_mock_") + id.internal;
    *comm = c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }
ncclResult_t ncclGroupStart() { g_group_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (--g_group_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    for (const Op& op : ops) if (op.send) { ncclResult_t r = run(op); if (r) return r; }   // sends never block
    for (const Op& op : ops) if (!op.send) { ncclResult_t r = run(op); if (r) return r; }
    return ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
    return enqueue(Op{true, const_cast<void*>(buf), count * dtype_size(t), peer, comm, stream});
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
    return enqueue(Op{false, buf, count * dtype_size(t), peer, comm, stream});
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t stream) {
    if (t != ncclFloat64) return ncclInvalidArgument;  // the product reduces a few doubles only
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<double> mine(count), acc(count);
    if (hipMemcpy(mine.data(), send, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    char name[512];
    const unsigned long long round = c->reduce_round++;
    for (int peer = 0; peer < c->nranks; peer++) {  // one copy per reader, so each reader can delete its own
        snprintf(name, sizeof name, "%s_r_%llu_%d_%d", c->base.c_str(), round, c->rank, peer);
        publish(name, mine.data(), count * 8);
    }
    for (int src = 0; src < c->nranks; src++) {
        snprintf(name, sizeof name, "%s_r_%llu_%d_%d", c->base.c_str(), round, src, c->rank);
        const std::vector<char> raw = consume(name, count * 8);
        const double* v = reinterpret_cast<const double*>(raw.data());
        for (size_t i = 0; i < count; i++) {
            if (src == 0) acc[i] = v[i];
            else if (op == ncclSum) acc[i] += v[i];
            else if (op == ncclProd) acc[i] *= v[i];
            else if (op == ncclMax) acc[i] = v[i] > acc[i] ? v[i] : acc[i];
            else acc[i] = v[i] < acc[i] ? v[i] : acc[i];
        }
    }
    if (hipMemcpy(recv, acc.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "mock_rccl error"; }
}

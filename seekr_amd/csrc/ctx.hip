// Context, device matrices, error plumbing and per-kernel HIP-event profiling.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.hpp"

static thread_local char g_err[1024] = "";

int skr_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* skr_last_error(void) { return g_err; }
extern "C" int skr_abi_version(void) { return SKR_ABI_VERSION; }

extern "C" int skr_device_count(int* count) {
    SKR_REQUIRE(count, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return skr_set_error(SKR_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return SKR_OK;
}

// Page-lock a host range the caller owns (hipHostRegister) so that copies to / from it are plain DMA at the link's rate,
// whatever state the runtime's own pinning cache is in; undone with skr_host_unregister before the memory is freed.
extern "C" int skr_host_register(int device, void* ptr, size_t bytes) {
    SKR_REQUIRE(ptr && bytes, "empty range");
    // the runtime registers "on the current device" of the calling thread, and a thread that has never chosen one (a
    // finaliser running on a collector's thread) would bring up device 0's context in every rank process of a node
    // ... and the thread gets its own device back afterwards: a worker of another GPU may be the one that runs the finaliser
    int before = -1;
    if (device >= 0) {
        (void)hipGetDevice(&before);
        SKR_HIP(hipSetDevice(device));
    }
    struct Back {
        int dev;
        ~Back() {
            if (dev >= 0) (void)hipSetDevice(dev);
        }
    } back{before == device ? -1 : before};
    SKR_HIP(hipHostRegister(ptr, bytes, hipHostRegisterPortable));  // every GPU of the node copies at the pinned rate
    return SKR_OK;
}

extern "C" int skr_host_unregister(void* ptr) {
    SKR_REQUIRE(ptr, "NULL pointer");
    SKR_HIP(hipHostUnregister(ptr));
    return SKR_OK;
}

int skr_activate(const skr_ctx* ctx) {
    SKR_REQUIRE(ctx, "ctx is NULL");
    SKR_HIP(hipSetDevice(ctx->device));
    return SKR_OK;
}

// The A/B switches of DESIGN §4 (INTEGRATION.md lists them): read when the ctx is created, and again only when a
// bench tool asks for it after changing its environment (tools/gemm_bench.py, tools/count_bench.py interleave variants
// in one process).  No launch path calls getenv.
extern "C" int skr_ctx_reload_knobs(skr_ctx* c) {
    SKR_REQUIRE(c, "ctx is NULL");
    auto env_int = [](const char* name, int dflt) {
        const char* v = getenv(name);
        if (!v || !*v) return dflt;
        if (!strcmp(v, "spin")) return 1;  // SEEKR_HOST_WAIT's words
        if (!strcmp(v, "yield")) return 2;
        if (!strcmp(v, "block")) return 3;
        return atoi(v);
    };
    SkrKnobs& kn = c->knobs;
    kn.host_wait = env_int("SEEKR_HOST_WAIT", 0);
    kn.gemm_persist = env_int("SEEKR_GEMM_PERSIST", 1) != 0;
    kn.gemm_chunk_tiles = std::max(0, env_int("SEEKR_GEMM_CHUNK_TILES", 0));
    kn.gemm_reserve_cus = env_int("SEEKR_GEMM_RESERVE_CUS", -1);
    kn.gemm_subtile = env_int("SEEKR_GEMM_SUBTILE", 4);
    kn.gemm_wave_tile = env_int("SEEKR_GEMM_WAVE_TILE", 0);
    kn.gemm_epilogue = std::max(0, std::min(3, env_int("SEEKR_GEMM_EPILOGUE", 3)));
    kn.count_percu = std::max(0, env_int("SEEKR_COUNT_PERCU", 0));
    kn.count_persist = env_int("SEEKR_COUNT_PERSIST", 0);
    kn.count_legacy = env_int("SEEKR_COUNT_LEGACY", 0) != 0;
    kn.count_wps = env_int("SEEKR_COUNT_WPS", 0);
    kn.split_max_cols = env_int("SEEKR_SPLIT_MAX_COLS", 262144);
    kn.count_generic_global = env_int("SEEKR_COUNT_GENERIC_GLOBAL", 0) != 0;
    kn.count_generic_wgs = env_int("SEEKR_COUNT_GENERIC_WGS", 0);
    kn.count_k8_global = env_int("SEEKR_COUNT_K8_GLOBAL", 0) != 0;
    kn.count_occ = std::max(0, env_int("SEEKR_COUNT_OCC", 0));
    kn.chain_host_wait = env_int("SEEKR_TEST_HOOKS", 0) == 1 && env_int("SEEKR_CHAIN_HOST_WAIT", 0) != 0;
    return SKR_OK;
}

extern "C" int skr_ctx_create(int device, skr_ctx** out) {
    SKR_REQUIRE(out, "out is NULL");
    *out = nullptr;
    int n = 0;
    SKR_TRY(skr_device_count(&n));
    if (n <= 0) return skr_set_error(SKR_ERR_HIP, "no HIP device visible");
    SKR_REQUIRE(device >= 0 && device < n, "device %d out of range (0..%d)", device, n - 1);
    SKR_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SKR_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return skr_set_error(SKR_ERR_UNSUPPORTED, "device %d is %s; libseekr_hip is built for gfx950 only",
                             device, prop.gcnArchName);
    skr_ctx* c = new skr_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    skr_ctx_reload_knobs(c);
    if (c->knobs.host_wait >= 1 && c->knobs.host_wait <= 3) {
        // How the host waits for the GPU (every flag read of a step is such a wait): an A/B knob, profiles/r6_host_wait.log
        const unsigned f[4] = {hipDeviceScheduleAuto, hipDeviceScheduleSpin, hipDeviceScheduleYield, hipDeviceScheduleBlockingSync};
        (void)hipSetDeviceFlags(f[c->knobs.host_wait]);
        (void)hipGetLastError();
    }
    SKR_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    SKR_HIP(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    SKR_HIP(hipMalloc((void**)&c->d_flags, 64 * sizeof(uint32_t)));
    SKR_HIP(hipHostMalloc((void**)&c->h_flags, 64 * sizeof(uint32_t), hipHostMallocDefault));
    SKR_HIP(hipMemsetAsync(c->d_flags, 0, 64 * sizeof(uint32_t), c->stream));
    // The first host <-> device copies of the ctx happen HERE, on its own stream (a flag word up and down): the runtime
    // hands out its copy engines at first use, and a process whose first copy went through the NULL stream copied at half
    // rate on this stream ever after (pack.hip, upload_seqs: measured).
    SKR_HIP(hipMemcpyAsync(c->d_flags + 32, c->h_flags, 4 * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    SKR_HIP(hipMemcpyAsync(c->h_flags + 32, c->d_flags + 32, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    SKR_HIP(hipStreamSynchronize(c->stream));
    *out = c;
    return SKR_OK;
}

extern "C" int skr_comm_destroy(skr_ctx* ctx);

extern "C" int skr_ctx_destroy(skr_ctx* ctx) {
    if (!ctx) return SKR_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    skr_comm_destroy(ctx);
    for (auto& r : ctx->prof_recs) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    for (auto& p : ctx->event_pool) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    for (auto& t : ctx->tickets)
        if (t.ev) (void)hipEventDestroy(t.ev);
    for (auto& b : ctx->small_blocks) {
        if (b.comm_done) (void)hipEventDestroy(b.comm_done);
        if (b.p) (void)hipFree(b.p);
    }
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->pin_done) (void)hipEventDestroy(ctx->pin_done);
    if (ctx->d_flags) (void)hipFree(ctx->d_flags);
    if (ctx->d_recip) (void)hipFree(ctx->d_recip);
    if (ctx->d_np_plan) (void)hipFree(ctx->d_np_plan);
    if (ctx->h_flags) (void)hipHostFree(ctx->h_flags);
    for (hipEvent_t ev : ctx->marks)
        if (ev) (void)hipEventDestroy(ev);
    for (void* hp : ctx->h_copy)
        if (hp) (void)hipHostFree(hp);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    delete ctx;
    return SKR_OK;
}

extern "C" int skr_ctx_mem_info(skr_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes) {
    SKR_REQUIRE(ctx && free_bytes && total_bytes, "NULL argument");
    SKR_TRY(skr_activate(ctx));
    size_t f = 0, t = 0;
    SKR_HIP(hipMemGetInfo(&f, &t));
    *free_bytes = f;
    *total_bytes = t;
    return SKR_OK;
}

extern "C" int skr_ctx_sync(skr_ctx* ctx) {
    SKR_TRY(skr_activate(ctx));
    SKR_HIP(hipStreamSynchronize(ctx->stream));
    SKR_HIP(hipStreamSynchronize(ctx->comm_stream));
    if (ctx->copy_stream) SKR_HIP(hipStreamSynchronize(ctx->copy_stream));
    return SKR_OK;
}

// ---------------------------------------------------------------- marks + the copy stream
extern "C" int skr_ctx_mark(skr_ctx* ctx, int64_t* mark) {
    SKR_REQUIRE(ctx && mark, "NULL argument");
    SKR_TRY(skr_activate(ctx));
    int slot;
    if (!ctx->free_marks.empty()) {
        slot = ctx->free_marks.back();
        ctx->free_marks.pop_back();
    } else {
        SKR_REQUIRE(ctx->marks.size() < 4096, "more than 4 096 marks are waiting to be used");
        hipEvent_t ev = nullptr;
        SKR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ctx->marks.push_back(ev);
        ctx->mark_live.push_back(0);
        slot = (int)ctx->marks.size() - 1;
    }
    hipError_t e = hipEventRecord(ctx->marks[slot], ctx->stream);
    if (e != hipSuccess) {
        ctx->free_marks.push_back(slot);
        return skr_set_error(SKR_ERR_HIP, "hipEventRecord failed: %s", hipGetErrorString(e));
    }
    ctx->mark_live[slot] = 1;
    *mark = slot;
    return SKR_OK;
}

// A mark that will not be consumed after all (the consumer's arguments were refused, or the caller gave up before using
// it): its slot goes back.  Unknown or already used marks are ignored — releasing is always safe.  (slots were
// recycled by the consumer only, so 4 096 abandoned marks ended skr_ctx_mark for the ctx.)
extern "C" int skr_ctx_mark_release(skr_ctx* ctx, int64_t mark) {
    SKR_REQUIRE(ctx, "ctx is NULL");
    if (mark >= 0 && mark < (int64_t)ctx->marks.size() && ctx->mark_live[mark]) {
        ctx->mark_live[mark] = 0;
        ctx->free_marks.push_back((int)mark);
    }
    return SKR_OK;
}

int skr_copy_stream_after(skr_ctx* ctx, int64_t mark, hipStream_t* out) {
    SKR_TRY(skr_activate(ctx));
    if (!ctx->copy_stream) SKR_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (mark >= 0) {
        SKR_REQUIRE(mark < (int64_t)ctx->marks.size() && ctx->mark_live[mark], "unknown mark (each mark is used once)");
        hipError_t e = hipStreamWaitEvent(ctx->copy_stream, ctx->marks[mark], 0);
        ctx->mark_live[mark] = 0;
        ctx->free_marks.push_back((int)mark);
        if (e != hipSuccess) return skr_set_error(SKR_ERR_HIP, "hipStreamWaitEvent failed: %s", hipGetErrorString(e));
    } else {
        hipEvent_t now;
        SKR_HIP(hipEventCreateWithFlags(&now, hipEventDisableTiming));
        hipError_t e = hipEventRecord(now, ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->copy_stream, now, 0);
        (void)hipEventDestroy(now);
        if (e != hipSuccess) return skr_set_error(SKR_ERR_HIP, "ordering the copy stream failed: %s", hipGetErrorString(e));
    }
    *out = ctx->copy_stream;
    return SKR_OK;
}

extern "C" int skr_ctx_device(const skr_ctx* ctx, int* device) {
    SKR_REQUIRE(ctx && device, "NULL argument");
    *device = ctx->device;
    return SKR_OK;
}

int skr_kernel_lds(skr_ctx* ctx, const void* kern, size_t bytes) {
    auto it = ctx->lds_attr.find(kern);
    if (it != ctx->lds_attr.end() && (size_t)it->second >= bytes) return SKR_OK;
    SKR_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    ctx->lds_attr[kern] = (int)bytes;
    return SKR_OK;
}

int skr_ctx_workspace(skr_ctx* ctx, size_t bytes, void** out) {
    if (bytes > ctx->ws_bytes) {
        SKR_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->ws) SKR_HIP(hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
        size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        SKR_HIP(hipMalloc(&ctx->ws, want));
        ctx->ws_bytes = want;
    }
    *out = ctx->ws;
    return SKR_OK;
}

int skr_ctx_pinned(skr_ctx* ctx, size_t bytes, void** out) {
    if (ctx->pin_done) SKR_HIP(hipEventSynchronize(ctx->pin_done));
    if (bytes > ctx->h_pin_bytes) {
        if (ctx->h_pin) SKR_HIP(hipHostFree(ctx->h_pin));
        ctx->h_pin = nullptr;
        ctx->h_pin_bytes = 0;
        const size_t want = (bytes + 65535) & ~(size_t)65535;
        SKR_HIP(hipHostMalloc(&ctx->h_pin, want, hipHostMallocDefault));
        ctx->h_pin_bytes = want;
    }
    *out = ctx->h_pin;
    return SKR_OK;
}

int skr_ctx_pinned_used(skr_ctx* ctx) {
    if (!ctx->pin_done) SKR_HIP(hipEventCreateWithFlags(&ctx->pin_done, hipEventDisableTiming));
    SKR_HIP(hipEventRecord(ctx->pin_done, ctx->stream));
    return SKR_OK;
}

// ---------------------------------------------------------------- profiling -------------
SkrProfScope::SkrProfScope(skr_ctx* c, const char* name, hipStream_t on) : ctx(c), stream(on ? on : c->stream) {
    if (!c->prof) return;
    skr_ctx::ProfRec rec;
    rec.name = name;
    if (!c->event_pool.empty()) {
        rec.start = c->event_pool.back().first;
        rec.stop = c->event_pool.back().second;
        c->event_pool.pop_back();
    } else {
        if (hipEventCreate(&rec.start) != hipSuccess) return;
        if (hipEventCreate(&rec.stop) != hipSuccess) return;
    }
    (void)hipEventRecord(rec.start, stream);
    c->prof_recs.push_back(rec);
    idx = (int)c->prof_recs.size() - 1;
}

SkrProfScope::~SkrProfScope() {
    if (idx >= 0) (void)hipEventRecord(ctx->prof_recs[idx].stop, stream);
}

extern "C" int skr_prof_enable(skr_ctx* ctx, int on) {
    SKR_REQUIRE(ctx, "ctx is NULL");
    ctx->prof = on != 0;
    return SKR_OK;
}

extern "C" int skr_prof_reset(skr_ctx* ctx) {
    SKR_TRY(skr_activate(ctx));
    SKR_HIP(hipStreamSynchronize(ctx->stream));
    SKR_HIP(hipStreamSynchronize(ctx->comm_stream));
    for (auto& r : ctx->prof_recs) ctx->event_pool.emplace_back(r.start, r.stop);
    ctx->prof_recs.clear();
    return SKR_OK;
}

extern "C" int skr_prof_query(skr_ctx* ctx, const char* name, double* total_ms, int64_t* launches) {
    SKR_REQUIRE(ctx && name && total_ms && launches, "NULL argument");
    SKR_TRY(skr_activate(ctx));
    SKR_HIP(hipStreamSynchronize(ctx->stream));
    SKR_HIP(hipStreamSynchronize(ctx->comm_stream));  // "comm_xfer" scopes are recorded there
    double tot = 0;
    int64_t cnt = 0;
    for (auto& r : ctx->prof_recs) {
        if (r.name != name) continue;  // exact: "colsum_seq" does not collect "colsum_seq_sq"
        float ms = 0;
        SKR_HIP(hipEventElapsedTime(&ms, r.start, r.stop));
        tot += ms;
        cnt++;
    }
    *total_ms = tot;
    *launches = cnt;
    return SKR_OK;
}

extern "C" int skr_prof_names(skr_ctx* ctx, char* buf, int64_t cap) {
    SKR_REQUIRE(ctx && buf && cap > 0, "bad argument");
    std::string all;
    std::vector<std::string> seen;
    for (auto& r : ctx->prof_recs) {
        bool dup = false;
        for (auto& s : seen) dup |= (s == r.name);
        if (dup) continue;
        seen.push_back(r.name);
        if (!all.empty()) all += "\n";
        all += r.name;
    }
    snprintf(buf, (size_t)cap, "%s", all.c_str());
    return SKR_OK;
}

// ---------------------------------------------------------------- matrices --------------
constexpr size_t kSmallBlockBytes = 256 << 10;  // statistic vectors: 16 KiB (k = 6) ... 256 KiB (k = 8, float32)
constexpr size_t kSmallBlockCap = 64;

extern "C" int skr_mat_create(skr_ctx* ctx, int64_t rows, int64_t cols, int dtype, skr_mat** out) {
    SKR_REQUIRE(ctx && out, "NULL argument");
    *out = nullptr;
    SKR_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    SKR_REQUIRE(dtype == SKR_F32 || dtype == SKR_F64 || dtype == SKR_U32, "unknown dtype %d", dtype);
    SKR_TRY(skr_activate(ctx));
    skr_mat* m = new skr_mat();
    m->ctx = ctx;
    m->rows = rows;
    m->cols = cols;
    m->dtype = dtype;
    size_t bytes = m->bytes();
    if (bytes == 0) bytes = 16;
    if (bytes <= kSmallBlockBytes) {  // a recycled block of exactly this size (see skr_mat_free)
        std::lock_guard<std::mutex> guard(ctx->small_lock);
        for (size_t i = 0; i < ctx->small_blocks.size(); i++) {
            if (ctx->small_blocks[i].bytes != bytes) continue;
            skr_ctx::SmallBlock b = ctx->small_blocks[i];
            ctx->small_blocks.erase(ctx->small_blocks.begin() + (long)i);
            // earlier uses on the compute stream are ordered by the stream itself; the communication stream by the event
            hipError_t we = hipStreamWaitEvent(ctx->stream, b.comm_done, 0);
            (void)hipEventDestroy(b.comm_done);
            if (we != hipSuccess) {
                (void)hipFree(b.p);
                break;
            }
            m->data = b.p;
            *out = m;
            return SKR_OK;
        }
    }
    hipError_t e = hipMalloc(&m->data, bytes);
    if (e != hipSuccess) {
        delete m;
        return skr_set_error(SKR_ERR_NOMEM, "hipMalloc(%zu bytes) for a %lld x %lld matrix failed: %s", bytes,
                             (long long)rows, (long long)cols, hipGetErrorString(e));
    }
    *out = m;
    return SKR_OK;
}

extern "C" int skr_mat_free(skr_mat* m) {
    if (!m) return SKR_OK;
    if (m->owner) {  // views own nothing: no synchronisation needed to drop them
        (void)hipSetDevice(m->ctx->device);
        const size_t bytes = m->bytes() ? m->bytes() : 16;
        std::unique_lock<std::mutex> guard(m->ctx->small_lock);
        if (m->data && bytes <= kSmallBlockBytes && m->ctx->small_blocks.size() < kSmallBlockCap) {
            // keep the block for the next matrix of this size: no hipFree, hence no host synchronisation
            skr_ctx::SmallBlock b;
            b.p = m->data;
            b.bytes = bytes;
            if (hipEventCreateWithFlags(&b.comm_done, hipEventDisableTiming) == hipSuccess &&
                hipEventRecord(b.comm_done, m->ctx->comm_stream) == hipSuccess) {
                m->ctx->small_blocks.push_back(b);
                delete m;
                return SKR_OK;
            }
            if (b.comm_done) (void)hipEventDestroy(b.comm_done);
        }
        guard.unlock();
        (void)hipStreamSynchronize(m->ctx->stream);
        (void)hipStreamSynchronize(m->ctx->comm_stream);
        if (m->data) (void)hipFree(m->data);
    }
    delete m;
    return SKR_OK;
}

extern "C" int skr_mat_view(const skr_mat* parent, int64_t row0, int64_t nrows, skr_mat** out) {
    SKR_REQUIRE(parent && out, "NULL argument");
    *out = nullptr;
    SKR_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= parent->rows, "view rows [%lld, %lld) outside 0..%lld",
                (long long)row0, (long long)(row0 + nrows), (long long)parent->rows);
    skr_mat* v = new skr_mat();
    v->ctx = parent->ctx;
    v->rows = nrows;
    v->cols = parent->cols;
    v->dtype = parent->dtype;
    v->owner = false;
    v->data = (char*)parent->data + (size_t)row0 * (size_t)parent->cols * parent->elem();
    *out = v;
    return SKR_OK;
}

extern "C" int skr_mat_shape(const skr_mat* m, int64_t* rows, int64_t* cols, int* dtype) {
    SKR_REQUIRE(m, "matrix is NULL");
    if (rows) *rows = m->rows;
    if (cols) *cols = m->cols;
    if (dtype) *dtype = m->dtype;
    return SKR_OK;
}

extern "C" int skr_mat_device_ptr(const skr_mat* m, void** ptr) {
    SKR_REQUIRE(m && ptr, "NULL argument");
    *ptr = m->data;
    return SKR_OK;
}

static int check_rows(const skr_mat* m, const void* host, int64_t row0, int64_t nrows) {
    SKR_REQUIRE(m, "matrix is NULL");
    SKR_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= m->rows, "row range [%lld, %lld) outside 0..%lld",
                (long long)row0, (long long)(row0 + nrows), (long long)m->rows);
    SKR_REQUIRE(host || nrows * m->cols == 0, "host buffer is NULL");
    return SKR_OK;
}

extern "C" int skr_mat_upload(skr_mat* m, const void* host, int64_t row0, int64_t nrows) {
    SKR_TRY(check_rows(m, host, row0, nrows));
    SKR_TRY(skr_activate(m->ctx));
    size_t rb = (size_t)m->cols * m->elem();
    if (nrows * rb == 0) return SKR_OK;
    SKR_HIP(hipMemcpyAsync((char*)m->data + (size_t)row0 * rb, host, (size_t)nrows * rb, hipMemcpyHostToDevice,
                           m->ctx->stream));
    SKR_HIP(hipStreamSynchronize(m->ctx->stream));
    return SKR_OK;
}

extern "C" int skr_mat_download(const skr_mat* m, void* host, int64_t row0, int64_t nrows) {
    SKR_TRY(check_rows(m, host, row0, nrows));
    SKR_TRY(skr_activate(m->ctx));
    size_t rb = (size_t)m->cols * m->elem();
    if (nrows * rb == 0) return SKR_OK;
    SKR_HIP(hipMemcpyAsync(host, (const char*)m->data + (size_t)row0 * rb, (size_t)nrows * rb,
                           hipMemcpyDeviceToHost, m->ctx->stream));
    SKR_HIP(hipStreamSynchronize(m->ctx->stream));
    return SKR_OK;
}

// skr_mat_download beside the compute stream: the copy waits for `mark` (or, mark < 0, for what is enqueued now), the
// compute stream does not wait for the copy — the caller enqueued the next stripe's contraction before it came here.
extern "C" int skr_mat_download_at(const skr_mat* m, void* host, int64_t row0, int64_t nrows, int64_t mark) {
    if (const int rc = check_rows(m, host, row0, nrows)) {
        if (m && m->ctx) (void)skr_ctx_mark_release(m->ctx, mark);  // a refused call still uses its mark up
        return rc;
    }
    hipStream_t cs = nullptr;
    SKR_TRY(skr_copy_stream_after(m->ctx, mark, &cs));
    size_t rb = (size_t)m->cols * m->elem();
    if (nrows * rb == 0) return SKR_OK;
    SKR_HIP(hipMemcpyAsync(host, (const char*)m->data + (size_t)row0 * rb, (size_t)nrows * rb, hipMemcpyDeviceToHost, cs));
    SKR_HIP(hipStreamSynchronize(cs));
    return SKR_OK;
}

extern "C" int skr_mat_fill_zero(skr_mat* m) {
    SKR_REQUIRE(m, "matrix is NULL");
    SKR_TRY(skr_activate(m->ctx));
    if (m->bytes()) SKR_HIP(hipMemsetAsync(m->data, 0, m->bytes(), m->ctx->stream));
    return SKR_OK;
}

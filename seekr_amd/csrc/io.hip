// Writers for the files the reference produces from the count matrix and from r:
//   np.save(path, a)                                      kmer_counts.py:234, pearson.py:43
//   np.savetxt(path, a, delimiter=",", fmt="%1.6f")       kmer_counts.py:241   (the CLI default)
//   np.savetxt(path, a, delimiter=",")  [fmt "%.18e"]     find_dist.py:292 (a consumer of r)
// byte-identical to numpy's output.  Device matrices are streamed through two pinned buffers
// (the D2H copy of chunk i+1 overlaps formatting / writing chunk i); text is formatted by a pool
// of host threads, each on its own row range, and written in order.  Host code only — no kernel.
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstring>
#include <thread>

#include "common.hpp"

namespace {

struct File {
    FILE* fh = nullptr;
    ~File() {
        if (fh) fclose(fh);
    }
};

int open_out(const char* path, File* f) {
    f->fh = fopen(path, "wb");
    if (!f->fh) return skr_set_error(SKR_ERR_IO, "cannot open %s for writing: %s", path, strerror(errno));
    setvbuf(f->fh, nullptr, _IOFBF, 8 << 20);
    return SKR_OK;
}

int write_all(FILE* fh, const void* p, size_t n, const char* path) {
    if (n && fwrite(p, 1, n, fh) != n) return skr_set_error(SKR_ERR_IO, "short write to %s: %s", path, strerror(errno));
    return SKR_OK;
}

const char* npy_descr(int dtype) { return dtype == SKR_F64 ? "<f8" : (dtype == SKR_U32 ? "<u4" : "<f4"); }

// numpy.lib.format.write_array_header_1_0: magic, version 1.0, little-endian u16 header length,
// the dict literal, padded with spaces and a final '\n' so that the data starts at a multiple of 64.
std::string npy_header(int dtype, int64_t rows, int64_t cols, bool one_dim) {
    char dict[160];
    if (one_dim)
        snprintf(dict, sizeof dict, "{'descr': '%s', 'fortran_order': False, 'shape': (%lld,), }", npy_descr(dtype),
                 (long long)cols);
    else
        snprintf(dict, sizeof dict, "{'descr': '%s', 'fortran_order': False, 'shape': (%lld, %lld), }", npy_descr(dtype),
                 (long long)rows, (long long)cols);
    std::string h(dict);
    const size_t fixed = 6 + 2 + 2;  // magic, version, length field
    const size_t pad = 64 - ((fixed + h.size() + 1) % 64);
    h.append(pad % 64, ' ');
    h.push_back('\n');
    std::string out("\x93NUMPY\x01\x00", 8);
    out.push_back((char)(h.size() & 0xff));
    out.push_back((char)(h.size() >> 8));
    return out + h;
}

// ---- "%1.6f" of a double, exactly as printf rounds it (the exact binary value to 6 decimals, ties to
// even), with integer arithmetic: |x| = m 2^e, R = round(m 2^e 10^6) in 128 bits.
inline char* put_u64(char* p, uint64_t v) {
    char tmp[24];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

inline char* fmt_fixed6(char* p, double x) {
    if (std::isnan(x)) {
        // printf prints the sign of a NaN; Python's '%f' % nan does not
        memcpy(p, "nan", 3);
        return p + 3;
    }
    if (std::signbit(x)) *p++ = '-';
    if (std::isinf(x)) {
        memcpy(p, "inf", 3);
        return p + 3;
    }
    const double a = std::fabs(x);
    if (a >= 9007199254740992.0) return p + snprintf(p, 400, "%1.6f", a);  // >= 2^53: an integer, rare
    int e;
    const double fr = std::frexp(a, &e);               // a = fr 2^e, fr in [0.5, 1)
    const uint64_t m = (uint64_t)std::ldexp(fr, 53);   // 53-bit integer significand (exact)
    e -= 53;                                           // a = m 2^e
    unsigned __int128 r;
    if (a == 0.0) {
        r = 0;
    } else if (e >= 0) {
        r = ((unsigned __int128)m << e) * 1000000u;    // a < 2^53, so m << e < 2^53
    } else {
        const int s = -e;
        const unsigned __int128 prod = (unsigned __int128)m * 1000000u;  // < 2^73
        if (s >= 120) {
            r = 0;  // prod / 2^s < 2^-47
        } else {
            r = prod >> s;
            const unsigned __int128 rem = prod & (((unsigned __int128)1 << s) - 1), half = (unsigned __int128)1 << (s - 1);
            if (rem > half || (rem == half && (r & 1))) r += 1;
        }
    }
    const uint64_t ip = (uint64_t)(r / 1000000u);
    uint64_t fp = (uint64_t)(r % 1000000u);
    p = put_u64(p, ip);
    *p++ = '.';
    for (int i = 5; i >= 0; i--) {
        p[i] = (char)('0' + fp % 10);
        fp /= 10;
    }
    return p + 6;
}

inline char* fmt_sci18(char* p, double x) {
    if (std::isnan(x)) {
        memcpy(p, "nan", 3);
        return p + 3;
    }
    return p + snprintf(p, 64, "%.18e", x);  // glibc prints the exact binary value correctly rounded
}

// The same for a float32 (the count matrix): 24-bit significand x 10^6 < 2^44, so everything below
// 2^39 stays in 64-bit integers (no libm call, divisions by constants); the rest takes the general path.
inline char* fmt_fixed6(char* p, float x) {
    uint32_t bits;
    memcpy(&bits, &x, 4);
    const uint32_t ex = (bits >> 23) & 0xff, frac = bits & 0x7fffffu;
    if (ex == 0xff || ex >= 127 + 39) return fmt_fixed6(p, (double)x);  // nan, inf, >= 2^39
    if (bits >> 31) *p++ = '-';
    const uint64_t m = ex ? (frac | 0x800000u) : frac;
    const int e = (ex ? (int)ex : 1) - 150;  // |x| = m 2^e
    uint64_t r;
    if (e >= 0) {
        r = (m << e) * 1000000u;  // m << e < 2^39
    } else {
        const int s = -e;
        const uint64_t prod = m * 1000000u;
        if (s >= 63) {
            r = 0;  // prod < 2^44
        } else {
            r = prod >> s;
            const uint64_t rem = prod & ((1ull << s) - 1), half = 1ull << (s - 1);
            if (rem > half || (rem == half && (r & 1))) r += 1;
        }
    }
    const uint64_t ip = r / 1000000u;
    uint32_t fp = (uint32_t)(r - ip * 1000000u);
    p = put_u64(p, ip);
    *p++ = '.';
    for (int i = 5; i >= 0; i--) {
        p[i] = (char)('0' + fp % 10);
        fp /= 10;
    }
    return p + 6;
}

template <typename T>
size_t format_rows(const T* data, int64_t rows, int64_t cols, int fmt_mode, char* out) {
    char* p = out;
    for (int64_t i = 0; i < rows; i++) {
        const T* row = data + (size_t)i * cols;
        for (int64_t j = 0; j < cols; j++) {
            if (j) *p++ = ',';
            p = fmt_mode == 0 ? fmt_fixed6(p, row[j]) : fmt_sci18(p, (double)row[j]);
        }
        *p++ = '\n';
    }
    return (size_t)(p - out);
}

// widest field: "%1.6f" of -FLT_MAX is 1 + 39 + 7 = 47 chars, of a double up to 1 + 309 + 7; "%.18e" is 26
size_t field_width(int dtype, int fmt_mode) { return fmt_mode == 1 ? 28 : (dtype == SKR_F64 ? 320 : 48); }

// rows per formatting round: the text buffers are sized for the widest possible field, keep them ~128 MB
int64_t chunk_rows_for(int dtype, int fmt_mode, int64_t cols) {
    const size_t per_row = (size_t)std::max<int64_t>(1, cols) * (field_width(dtype, fmt_mode) + 1) + 1;
    return std::max<int64_t>(1, (int64_t)((128u << 20) / per_row));
}

struct CsvFormatter {
    int dtype, fmt_mode, threads;
    int64_t cols;
    std::vector<std::vector<char>> bufs;
    std::vector<size_t> used;
    CsvFormatter(int dt, int fm, int th, int64_t c) : dtype(dt), fmt_mode(fm), threads(th), cols(c), bufs(th), used(th) {}
    // format `rows` rows starting at `data` and append them to the file, in order
    int run(const void* data, int64_t rows, FILE* fh, const char* path) {
        const int64_t per = (rows + threads - 1) / threads;
        const size_t w = field_width(dtype, fmt_mode) + 1;
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) {
            const int64_t r0 = std::min<int64_t>(rows, t * per), r1 = std::min<int64_t>(rows, r0 + per);
            used[t] = 0;
            if (r1 <= r0) continue;
            bufs[t].resize((size_t)(r1 - r0) * ((size_t)cols * w + 1));
            pool.emplace_back([this, data, r0, r1, t] {
                if (dtype == SKR_F64)
                    used[t] = format_rows((const double*)data + (size_t)r0 * cols, r1 - r0, cols, fmt_mode, bufs[t].data());
                else
                    used[t] = format_rows((const float*)data + (size_t)r0 * cols, r1 - r0, cols, fmt_mode, bufs[t].data());
            });
        }
        for (auto& th : pool) th.join();
        for (int t = 0; t < threads; t++) SKR_TRY(write_all(fh, bufs[t].data(), used[t], path));
        return SKR_OK;
    }
};

int pick_threads(int threads) {
    if (threads > 0) return std::min(threads, 256);
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(hc ? hc : 8u, 32u));
}

int check_host(const void* data, int dtype, int64_t rows, int64_t cols, const char* path) {
    SKR_REQUIRE(path, "path is NULL");
    SKR_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    SKR_REQUIRE(data || rows * cols == 0, "data is NULL");
    SKR_REQUIRE(dtype == SKR_F32 || dtype == SKR_F64 || dtype == SKR_U32, "unknown dtype %d", dtype);
    return SKR_OK;
}

// Streams a device matrix to `sink(host_ptr, row0, nrows)` through two pinned buffers.
template <typename Sink>
int stream_rows(skr_ctx* ctx, const skr_mat* m, int64_t chunk_rows, Sink&& sink) {
    SKR_TRY(skr_activate(ctx));
    const size_t rb = (size_t)m->cols * m->elem();
    if (m->rows == 0 || rb == 0) return SKR_OK;
    chunk_rows = std::max<int64_t>(1, std::min(chunk_rows, m->rows));
    void* pinned[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    int rc = SKR_OK;
    auto cleanup = [&] {
        for (int i = 0; i < 2; i++) {
            if (pinned[i]) (void)hipHostFree(pinned[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
        }
    };
    for (int i = 0; i < 2; i++) {
        if (hipHostMalloc(&pinned[i], (size_t)chunk_rows * rb, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) {
            cleanup();
            return skr_set_error(SKR_ERR_NOMEM, "cannot allocate %zu pinned bytes", (size_t)chunk_rows * rb);
        }
    }
    auto issue = [&](int64_t row0, int slot) -> hipError_t {
        const int64_t n = std::min(chunk_rows, m->rows - row0);
        hipError_t e = hipMemcpyAsync(pinned[slot], (const char*)m->data + (size_t)row0 * rb, (size_t)n * rb,
                                      hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(done[slot], ctx->stream);
        return e;
    };
    hipError_t e = issue(0, 0);
    int slot = 0;
    for (int64_t row0 = 0; row0 < m->rows && e == hipSuccess && rc == SKR_OK; row0 += chunk_rows, slot ^= 1) {
        const int64_t n = std::min(chunk_rows, m->rows - row0);
        if (row0 + chunk_rows < m->rows) e = issue(row0 + chunk_rows, slot ^ 1);
        if (e == hipSuccess) e = hipEventSynchronize(done[slot]);
        if (e == hipSuccess) rc = sink(pinned[slot], row0, n);
    }
    (void)hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) return skr_set_error(SKR_ERR_HIP, "device-to-host streaming failed: %s", hipGetErrorString(e));
    return rc;
}

}  // namespace

extern "C" int skr_host_save_npy(const void* data, int dtype, int64_t rows, int64_t cols, int one_dim, const char* path) {
    SKR_TRY(check_host(data, dtype, rows, cols, path));
    SKR_REQUIRE(!one_dim || rows == 1, "a 1-D array is passed as one row");
    File f;
    SKR_TRY(open_out(path, &f));
    const std::string h = npy_header(dtype, rows, cols, one_dim != 0);
    SKR_TRY(write_all(f.fh, h.data(), h.size(), path));
    return write_all(f.fh, data, (size_t)rows * cols * (dtype == SKR_F64 ? 8 : 4), path);
}

extern "C" int skr_host_save_csv(const void* data, int dtype, int64_t rows, int64_t cols, int fmt_mode, int threads,
                                 const char* path) {
    SKR_TRY(check_host(data, dtype, rows, cols, path));
    SKR_REQUIRE(dtype != SKR_U32, "CSV output is for float matrices");
    SKR_REQUIRE(fmt_mode == 0 || fmt_mode == 1, "fmt_mode must be 0 (%%1.6f) or 1 (%%.18e)");
    File f;
    SKR_TRY(open_out(path, &f));
    CsvFormatter fmt(dtype, fmt_mode, pick_threads(threads), cols);
    const int64_t chunk = chunk_rows_for(dtype, fmt_mode, cols);
    for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
        const int64_t n = std::min(chunk, rows - r0);
        SKR_TRY(fmt.run((const char*)data + (size_t)r0 * cols * (dtype == SKR_F64 ? 8 : 4), n, f.fh, path));
    }
    return SKR_OK;
}

extern "C" int skr_mat_save_npy(skr_ctx* ctx, const skr_mat* m, int one_dim, const char* path) {
    SKR_REQUIRE(ctx && m && m->ctx == ctx && path, "NULL or foreign argument");
    SKR_REQUIRE(!one_dim || m->rows == 1, "a 1-D array is passed as one row");
    File f;
    SKR_TRY(open_out(path, &f));
    const std::string h = npy_header(m->dtype, m->rows, m->cols, one_dim != 0);
    SKR_TRY(write_all(f.fh, h.data(), h.size(), path));
    const size_t rb = std::max<size_t>(1, (size_t)m->cols * m->elem());
    return stream_rows(ctx, m, (int64_t)((64u << 20) / rb) + 1, [&](const void* host, int64_t, int64_t n) {
        return write_all(f.fh, host, (size_t)n * m->cols * m->elem(), path);
    });
}

extern "C" int skr_mat_save_csv(skr_ctx* ctx, const skr_mat* m, int fmt_mode, int threads, const char* path) {
    SKR_REQUIRE(ctx && m && m->ctx == ctx && path, "NULL or foreign argument");
    SKR_REQUIRE(m->dtype != SKR_U32, "CSV output is for float matrices");
    SKR_REQUIRE(fmt_mode == 0 || fmt_mode == 1, "fmt_mode must be 0 (%%1.6f) or 1 (%%.18e)");
    File f;
    SKR_TRY(open_out(path, &f));
    CsvFormatter fmt(m->dtype, fmt_mode, pick_threads(threads), m->cols);
    const int64_t chunk = chunk_rows_for(m->dtype, fmt_mode, m->cols);
    return stream_rows(ctx, m, chunk, [&](const void* host, int64_t, int64_t n) { return fmt.run(host, n, f.fh, path); });
}

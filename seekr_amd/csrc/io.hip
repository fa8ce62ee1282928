// Writers for the files the reference produces from the count matrix and from r:
//   np.save(path, a)                                      kmer_counts.py:234, pearson.py:43
//   np.savetxt(path, a, delimiter=",", fmt="%1.6f")       kmer_counts.py:241   (the CLI default)
//   np.savetxt(path, a, delimiter=",")  [fmt "%.18e"]     find_dist.py:292 (a consumer of r)
// byte-identical to numpy's output.  Device matrices are streamed through two pinned buffers
// (the D2H copy of chunk i+1 overlaps formatting / writing chunk i); text is formatted by a pool
// of host threads, each on its own row range, and written in order.  Host code only — no kernel.
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
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

// ---- numpy's str() of a float scalar, which is what DataFrame.to_csv writes for a float column
// (kmer_counts.py:238-240): the shortest digit string that reads back as the same float32 / float64,
// positional for 1e-4 <= |x| < 1e16 (".0" appended to integers), else d.ddde+XX; NaN is the empty
// field (na_rep), infinities "inf" / "-inf".  The shortest length is found by bisection over
// "%.{p}e" (correctly rounded by glibc) + strtof/strtod round trip: if p digits identify the value,
// so do p+1.
inline bool reads_back(const char* s, float x) { return strtof(s, nullptr) == x; }
inline bool reads_back(const char* s, double x) { return strtod(s, nullptr) == x; }

// Slow, always-right digit generator: shortest digits d[0..nd) and decimal exponent e10 (value =
// d[0].d[1..] x 10^e10) of a > 0.
template <typename T>
inline void shortest_digits_exact(T a, char* digits, int* nd_out, int* e10_out) {
    constexpr int kMaxP = sizeof(T) == 4 ? 8 : 16;  // 9 / 17 significant digits always round-trip
    // A power of two sits at the bottom of its binade: the gap above it is twice the gap below, so a
    // decimal one unit above the nearest p-digit one can still read back as x when the nearest (below)
    // does not — numpy's Dragon4 finds it, so the bisection predicate tries it too.
    int exp2;
    const bool pow2 = std::frexp(a, &exp2) == (T)0.5;
    char buf[48];
    int nd = 0, e10 = 0;
    auto attempt = [&](int p) -> bool {  // leaves the accepted digits / exponent in digits, nd, e10
        snprintf(buf, sizeof buf, "%.*e", p, (double)a);
        const char* q = buf;
        nd = 0;
        for (; *q != 'e'; q++)
            if (*q != '.') digits[nd++] = *q;
        e10 = atoi(q + 1);
        if (reads_back(buf, a)) return true;
        if (!pow2) return false;
        int i = nd - 1;  // digits + 1 unit in the last place
        while (i >= 0 && digits[i] == '9') digits[i--] = '0';
        if (i >= 0) {
            digits[i]++;
        } else {  // 99..9 -> 100..0
            digits[0] = '1';
            e10++;
        }
        char up[48];
        int n = 0;
        up[n++] = digits[0];
        if (nd > 1) {
            up[n++] = '.';
            memcpy(up + n, digits + 1, nd - 1);
            n += nd - 1;
        }
        snprintf(up + n, sizeof up - n, "e%d", e10);
        return reads_back(up, a);
    };
    int lo = 0, hi = kMaxP;  // smallest p in [lo, hi] with a (p+1)-digit decimal that reads back as x
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (attempt(mid)) hi = mid; else lo = mid + 1;
    }
    attempt(lo);
    while (nd > 1 && digits[nd - 1] == '0') nd--;
    *nd_out = nd;
    *e10_out = e10;
}

// Fast digit generator for a float32 in [1e-4, 1e16) (the positional range, i.e. nearly every cell
// of a count matrix).  Everything is done on doubles, which hold a float32, the midpoints to its
// neighbours and 10^k (k <= 22) exactly; the only inexact quantities are a x 10^k and n / 10^k (one
// rounding each, 2^-53 relative), so a candidate n is accepted or rejected only when it clears the
// rounding interval of `a` by a margin 2^20 times wider than that error, and anything closer — or a
// rounding tie — returns false and the caller uses the exact generator.  Same answer, ~25x faster.
inline bool shortest_digits_fast(float af, char* digits, int* nd_out, int* e10_out) {
    static const double p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                   1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const double a = af;
    if (!(a >= 1e-4 && a < 1e16)) return false;
    uint32_t bits;
    memcpy(&bits, &af, 4);
    if (((bits >> 23) & 0xff) < 2) return false;  // subnormal neighbourhood: not in this range anyway
    const bool even = (bits & 1) == 0, pow2 = (bits & 0x7fffffu) == 0;
    uint32_t ub = bits + 1, lb = bits - 1;
    float upf, lof;
    memcpy(&upf, &ub, 4);
    memcpy(&lof, &lb, 4);
    const double hi = 0.5 * (a + (double)upf), lo = 0.5 * (a + (double)lof);  // exact midpoints
    const double margin = a * 1e-10;  // >> 2^-52 a (double rounding), << 2^-25 a (half a float32 gap)
    int E = -4;  // 10^E <= a < 10^(E+1)
    while (E < 15 && a >= (E + 1 >= 0 ? p10[E + 1] : 1.0 / p10[-(E + 1)])) E++;
    // a within a few double ulps of a power of ten that is not itself a float32: E could be misjudged
    {
        const double pe = E >= 0 ? p10[E] : 1.0 / p10[-E], pn = E + 1 >= 0 ? p10[E + 1] : 1.0 / p10[-(E + 1)];
        if ((a != pe && std::fabs(a - pe) < margin) || std::fabs(a - pn) < margin) return false;
    }
    // 0 = rejected, 1 = accepted (n_out set), -1 = too close to call
    auto attempt = [&](int p, uint64_t* n_out, int* shift_out) -> int {
        const int k = p - E;  // candidate = n x 10^-k with p+1 digits
        const double scaled = k >= 0 ? a * p10[k] : a / p10[-k];
        const double n0 = std::nearbyint(scaled);
        if (std::fabs(std::fabs(scaled - n0) - 0.5) < 1e-5) return -1;  // which decimal is nearest is in doubt
        for (int up = 0; up <= (pow2 ? 1 : 0); up++) {
            const double n = n0 + up;
            const double v = k >= 0 ? n / p10[k] : n * p10[-k];
            const bool in_lo = even ? v >= lo + margin : v > lo + margin, in_hi = even ? v <= hi - margin : v < hi - margin;
            if (in_lo && in_hi) {
                *n_out = (uint64_t)n;
                *shift_out = k;
                return 1;
            }
            const bool out = v < lo - margin || v > hi + margin;
            if (!out) return -1;
        }
        return 0;
    };
    int plo = 0, phi = 8;
    uint64_t n = 0;
    int k = 0;
    while (plo < phi) {
        const int mid = (plo + phi) / 2;
        const int r = attempt(mid, &n, &k);
        if (r < 0) return false;
        if (r) phi = mid; else plo = mid + 1;
    }
    if (attempt(plo, &n, &k) != 1) return false;
    // n has plo+1 digits unless it rounded up to 10^(plo+1)
    int e10 = E;
    char tmp[24];
    int len = 0;
    for (uint64_t t = n; t; t /= 10) tmp[len++] = (char)('0' + t % 10);
    if (len == plo + 2) e10++;  // 99..9 -> 100..0
    else if (len != plo + 1) return false;
    int nd = 0;
    for (int i = len - 1; i >= 0; i--) digits[nd++] = tmp[i];
    while (nd > 1 && digits[nd - 1] == '0') nd--;
    *nd_out = nd;
    *e10_out = e10;
    return true;
}
inline bool shortest_digits_fast(double, char*, int*, int*) { return false; }  // 17 digits do not fit a double product

template <typename T>
inline char* fmt_repr(char* p, T x) {
    if (std::isnan(x)) return p;
    if (std::signbit(x)) *p++ = '-';
    if (std::isinf(x)) {
        memcpy(p, "inf", 3);
        return p + 3;
    }
    const T a = std::fabs(x);
    if (a == 0) {
        memcpy(p, "0.0", 3);
        return p + 3;
    }
    char digits[24];
    int nd = 0, e10 = 0;
    if (!shortest_digits_fast(a, digits, &nd, &e10)) shortest_digits_exact(a, digits, &nd, &e10);
    if ((double)a >= 1e16 || (double)a < 1e-4) {
        *p++ = digits[0];
        if (nd > 1) {
            *p++ = '.';
            memcpy(p, digits + 1, nd - 1);
            p += nd - 1;
        }
        *p++ = 'e';
        *p++ = e10 < 0 ? '-' : '+';
        const int ae = e10 < 0 ? -e10 : e10;
        if (ae < 10) *p++ = '0';
        return put_u64(p, (uint64_t)ae);
    }
    if (e10 >= 0) {
        for (int i = 0; i <= e10; i++) *p++ = i < nd ? digits[i] : '0';
        *p++ = '.';
        if (nd > e10 + 1) {
            memcpy(p, digits + e10 + 1, nd - e10 - 1);
            p += nd - e10 - 1;
        } else {
            *p++ = '0';
        }
    } else {
        *p++ = '0';
        *p++ = '.';
        for (int i = 0; i < -e10 - 1; i++) *p++ = '0';
        memcpy(p, digits, nd);
        p += nd;
    }
    return p;
}

// pandas' csv.QUOTE_MINIMAL: quote a label that holds the delimiter, a quote or a line break
inline char* put_label(char* p, const std::string& s) {
    if (s.find_first_of(",\"\r\n") == std::string::npos) {
        memcpy(p, s.data(), s.size());
        return p + s.size();
    }
    *p++ = '"';
    for (char c : s) {
        if (c == '"') *p++ = '"';
        *p++ = c;
    }
    *p++ = '"';
    return p;
}

template <typename T>
size_t format_rows(const T* data, int64_t rows, int64_t cols, int fmt_mode, const std::string* labels, char* out) {
    char* p = out;
    for (int64_t i = 0; i < rows; i++) {
        const T* row = data + (size_t)i * cols;
        if (labels) {
            p = put_label(p, labels[i]);
            *p++ = ',';
        }
        for (int64_t j = 0; j < cols; j++) {
            if (j) *p++ = ',';
            p = fmt_mode == 0 ? fmt_fixed6(p, row[j]) : (fmt_mode == 1 ? fmt_sci18(p, (double)row[j]) : fmt_repr(p, row[j]));
        }
        *p++ = '\n';
    }
    return (size_t)(p - out);
}

// widest field: "%1.6f" of -FLT_MAX is 1 + 39 + 7 = 47 chars, of a double up to 1 + 309 + 7; "%.18e" is 26
size_t field_width(int dtype, int fmt_mode) {
    if (fmt_mode == 0) return dtype == SKR_F64 ? 320 : 48;
    return 28;  // "%.18e" is 26; the shortest repr at most 1 + 17 + 16 zeros... positional tops out at 1e16: 20
}

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
    const std::vector<std::string>* labels = nullptr;  // row labels (labelled CSV), indexed by global row
    size_t max_label = 0;
    // format `rows` rows starting at `data` (global row `row0`) and append them to the file, in order
    int run(const void* data, int64_t rows, FILE* fh, const char* path, int64_t row0 = 0) {
        const int64_t per = (rows + threads - 1) / threads;
        const size_t w = field_width(dtype, fmt_mode) + 1;
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) {
            const int64_t r0 = std::min<int64_t>(rows, t * per), r1 = std::min<int64_t>(rows, r0 + per);
            used[t] = 0;
            if (r1 <= r0) continue;
            bufs[t].resize((size_t)(r1 - r0) * ((size_t)cols * w + 2 + 2 * max_label + 3));
            pool.emplace_back([this, data, r0, r1, t, row0] {
                const std::string* lab = labels ? labels->data() + row0 + r0 : nullptr;
                if (dtype == SKR_F64)
                    used[t] = format_rows((const double*)data + (size_t)r0 * cols, r1 - r0, cols, fmt_mode, lab, bufs[t].data());
                else
                    used[t] = format_rows((const float*)data + (size_t)r0 * cols, r1 - r0, cols, fmt_mode, lab, bufs[t].data());
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
    return (int)std::max(1u, std::min(hc ? hc : 8u, 64u));
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
    SKR_REQUIRE(fmt_mode >= 0 && fmt_mode <= 2, "fmt_mode must be 0 (%%1.6f), 1 (%%.18e) or 2 (shortest repr)");
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

namespace {
std::vector<std::string> split_lines(const char* joined, int64_t expect, bool* ok) {
    std::vector<std::string> out;
    const char* s = joined ? joined : "";
    if (expect > 0) {
        for (const char* q = s;; q++) {
            if (*q == '\n' || *q == 0) {
                out.emplace_back(s, q - s);
                s = q + 1;
                if (*q == 0) break;
            }
        }
    }
    *ok = (int64_t)out.size() == expect;
    return out;
}

int write_header_line(FILE* fh, const std::vector<std::string>& cols, const char* path) {
    std::string line;
    for (const auto& c : cols) {
        line.push_back(',');
        const size_t at = line.size();
        line.resize(at + 2 * c.size() + 2);
        char* end = put_label(&line[at], c);
        line.resize(end - line.data());
    }
    line.push_back('\n');
    return write_all(fh, line.data(), line.size(), path);
}
}  // namespace

// DataFrame(data, index=row_labels, columns=col_labels).to_csv(path) (kmer_counts.py:236-240): labels are
// '\n'-joined (labels themselves cannot hold a line break); a header line ",c0,c1,..." then one line per row.
extern "C" int skr_host_save_csv_labelled(const void* data, int dtype, int64_t rows, int64_t cols, const char* row_labels,
                                          const char* col_labels, int threads, const char* path) {
    SKR_TRY(check_host(data, dtype, rows, cols, path));
    SKR_REQUIRE(dtype != SKR_U32, "CSV output is for float matrices");
    bool ok_r = false, ok_c = false;
    const std::vector<std::string> rl = split_lines(row_labels, rows, &ok_r), cl = split_lines(col_labels, cols, &ok_c);
    SKR_REQUIRE(ok_r && ok_c, "need %lld row labels and %lld column labels", (long long)rows, (long long)cols);
    File f;
    SKR_TRY(open_out(path, &f));
    SKR_TRY(write_header_line(f.fh, cl, path));
    CsvFormatter fmt(dtype, 2, pick_threads(threads), cols);
    fmt.labels = &rl;
    for (const auto& l : rl) fmt.max_label = std::max(fmt.max_label, l.size());
    const int64_t chunk = chunk_rows_for(dtype, 2, cols);
    for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
        const int64_t n = std::min(chunk, rows - r0);
        SKR_TRY(fmt.run((const char*)data + (size_t)r0 * cols * (dtype == SKR_F64 ? 8 : 4), n, f.fh, path, r0));
    }
    return SKR_OK;
}

extern "C" int skr_mat_save_csv_labelled(skr_ctx* ctx, const skr_mat* m, const char* row_labels, const char* col_labels,
                                         int threads, const char* path) {
    SKR_REQUIRE(ctx && m && m->ctx == ctx && path, "NULL or foreign argument");
    SKR_REQUIRE(m->dtype != SKR_U32, "CSV output is for float matrices");
    bool ok_r = false, ok_c = false;
    const std::vector<std::string> rl = split_lines(row_labels, m->rows, &ok_r), cl = split_lines(col_labels, m->cols, &ok_c);
    SKR_REQUIRE(ok_r && ok_c, "need %lld row labels and %lld column labels", (long long)m->rows, (long long)m->cols);
    File f;
    SKR_TRY(open_out(path, &f));
    SKR_TRY(write_header_line(f.fh, cl, path));
    CsvFormatter fmt(m->dtype, 2, pick_threads(threads), m->cols);
    fmt.labels = &rl;
    for (const auto& l : rl) fmt.max_label = std::max(fmt.max_label, l.size());
    return stream_rows(ctx, m, chunk_rows_for(m->dtype, 2, m->cols),
                       [&](const void* host, int64_t row0, int64_t n) { return fmt.run(host, n, f.fh, path, row0); });
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

// ---- np.save of a matrix that comes into being one row stripe at a time, possibly on several GPUs at once (pearson.py:43
// for a result larger than the HBM, or than one GPU's share): the file is created with numpy's header and its final size,
// every stripe is then written at its own offset, in any order, by whoever produced it.
extern "C" int skr_npy_create(const char* path, int dtype, int64_t rows, int64_t cols, int64_t* data_offset) {
    SKR_REQUIRE(path && data_offset, "NULL argument");
    SKR_REQUIRE(dtype == SKR_F32 || dtype == SKR_F64 || dtype == SKR_U32, "unknown dtype %d", dtype);
    SKR_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    const std::string h = npy_header(dtype, rows, cols, false);
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return skr_set_error(SKR_ERR_IO, "cannot open %s for writing: %s", path, strerror(errno));
    const size_t total = h.size() + (size_t)rows * (size_t)cols * (dtype == SKR_F64 ? 8 : 4);
    bool ok = pwrite(fd, h.data(), h.size(), 0) == (ssize_t)h.size() && ftruncate(fd, (off_t)total) == 0;
    const int err = errno;
    ok = close(fd) == 0 && ok;
    if (!ok) return skr_set_error(SKR_ERR_IO, "cannot lay out %s (%zu bytes): %s", path, total, strerror(err));
    *data_offset = (int64_t)h.size();
    return SKR_OK;
}

// Rows [row0, row0 + nrows) of a device matrix -> the file at byte offset file_offset.  Runs on the ctx's copy stream
// behind `mark` (skr_ctx_mark; < 0: behind everything enqueued so far): two pinned buffers, the pwrite of one chunk under
// the copy of the next; the compute stream is not held up.
extern "C" int skr_mat_write_rows_at(const skr_mat* m, int64_t row0, int64_t nrows, const char* path, int64_t file_offset,
                                     int64_t mark) {
    SKR_REQUIRE(m && path, "NULL argument");
    if (!(row0 >= 0 && nrows >= 0 && row0 + nrows <= m->rows) || file_offset < 0) {
        (void)skr_ctx_mark_release(m->ctx, mark);  // a refused call still uses its mark up
        SKR_REQUIRE(file_offset >= 0, "negative file offset");
        return skr_set_error(SKR_ERR_INVALID, "row range [%lld, %lld) outside 0..%lld", (long long)row0, (long long)(row0 + nrows),
                             (long long)m->rows);
    }
    skr_ctx* ctx = m->ctx;
    hipStream_t cs = nullptr;
    SKR_TRY(skr_copy_stream_after(ctx, mark, &cs));
    const size_t rb = (size_t)m->cols * m->elem();
    if (nrows == 0 || rb == 0) return SKR_OK;
    const size_t chunk_bytes = std::max<size_t>(rb, std::min<size_t>((size_t)nrows * rb, (size_t)32 << 20));
    const int64_t chunk_rows = (int64_t)(chunk_bytes / rb);
    if (ctx->h_copy_bytes < (size_t)chunk_rows * rb) {
        for (void*& hp : ctx->h_copy) {
            if (hp) (void)hipHostFree(hp);
            hp = nullptr;
        }
        ctx->h_copy_bytes = 0;
        for (void*& hp : ctx->h_copy) SKR_HIP(hipHostMalloc(&hp, (size_t)chunk_rows * rb, hipHostMallocDefault));
        ctx->h_copy_bytes = (size_t)chunk_rows * rb;
    }
    const int fd = open(path, O_WRONLY);
    if (fd < 0) return skr_set_error(SKR_ERR_IO, "cannot open %s for writing: %s", path, strerror(errno));
    hipEvent_t done[2] = {nullptr, nullptr};
    hipError_t e = hipEventCreateWithFlags(&done[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&done[1], hipEventDisableTiming);
    auto issue = [&](int64_t r, int slot) -> hipError_t {
        const int64_t n = std::min(chunk_rows, nrows - r);
        hipError_t ee = hipMemcpyAsync(ctx->h_copy[slot], (const char*)m->data + (size_t)(row0 + r) * rb, (size_t)n * rb,
                                       hipMemcpyDeviceToHost, cs);
        if (ee == hipSuccess) ee = hipEventRecord(done[slot], cs);
        return ee;
    };
    int rc = SKR_OK;
    if (e == hipSuccess) e = issue(0, 0);
    int slot = 0;
    for (int64_t r = 0; r < nrows && e == hipSuccess && rc == SKR_OK; r += chunk_rows, slot ^= 1) {
        const int64_t n = std::min(chunk_rows, nrows - r);
        if (r + chunk_rows < nrows) e = issue(r + chunk_rows, slot ^ 1);
        if (e == hipSuccess) e = hipEventSynchronize(done[slot]);
        if (e != hipSuccess) break;
        const char* src = (const char*)ctx->h_copy[slot];
        size_t left = (size_t)n * rb;
        off_t at = (off_t)file_offset + (off_t)((size_t)r * rb);
        while (left) {
            const ssize_t w = pwrite(fd, src, left, at);
            if (w < 0 && errno == EINTR) continue;
            if (w <= 0) {
                rc = skr_set_error(SKR_ERR_IO, "short write to %s: %s", path, strerror(errno));
                break;
            }
            src += w;
            at += w;
            left -= (size_t)w;
        }
    }
    (void)hipStreamSynchronize(cs);
    for (hipEvent_t ev : done)
        if (ev) (void)hipEventDestroy(ev);
    if (close(fd) != 0 && rc == SKR_OK && e == hipSuccess) rc = skr_set_error(SKR_ERR_IO, "closing %s failed: %s", path, strerror(errno));
    if (e != hipSuccess) return skr_set_error(SKR_ERR_HIP, "device-to-file streaming failed: %s", hipGetErrorString(e));
    return rc;
}

extern "C" int skr_mat_save_csv(skr_ctx* ctx, const skr_mat* m, int fmt_mode, int threads, const char* path) {
    SKR_REQUIRE(ctx && m && m->ctx == ctx && path, "NULL or foreign argument");
    SKR_REQUIRE(m->dtype != SKR_U32, "CSV output is for float matrices");
    SKR_REQUIRE(fmt_mode >= 0 && fmt_mode <= 2, "fmt_mode must be 0 (%%1.6f), 1 (%%.18e) or 2 (shortest repr)");
    File f;
    SKR_TRY(open_out(path, &f));
    CsvFormatter fmt(m->dtype, fmt_mode, pick_threads(threads), m->cols);
    const int64_t chunk = chunk_rows_for(m->dtype, fmt_mode, m->cols);
    return stream_rows(ctx, m, chunk, [&](const void* host, int64_t, int64_t n) { return fmt.run(host, n, f.fh, path); });
}

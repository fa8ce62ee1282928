// Reader for the labelled count CSVs that seekr_pearson takes by default:
//     pd.read_csv(path, index_col=0)                       console_scripts.py:626-631
// -> float64 values [rows, cols], row labels (the index) and column labels.  Host code only.
//
// Scope is deliberately the files seekr_kmer_counts writes (DataFrame.to_csv of a float matrix:
// header line ",c0,c1,...", then label,v0,v1,...).  pandas' default float converter is NOT correctly
// rounded in general (measured here: 30 % of 17-digit fields and 6 % of 12-digit fields with a large
// decimal exponent come back 1 ulp off): it accumulates the digits in a double and applies ONE
// multiplication or division by a tabulated power of ten.  That is exact — and equal to strtod —
// whenever the digits fit 2^53 (<= 15 significant digits here), the field has at most 17 digit
// characters (it drops the rest) and the power of ten is itself exact (|exponent| <= 22), which
// covers every float32 value printed positionally and the "%1.6f" files.  Exactly those fields are parsed here, by threads on disjoint line ranges;
// anything else (longer digit strings, big exponents, unbalanced quotes, ragged lines, text cells)
// makes the call return SKR_ERR_UNSUPPORTED and the host falls back to pandas, so a result is
// always bit-identical to the reference's.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <thread>

#include "common.hpp"

struct skr_csv {
    int64_t rows = 0, cols = 0;
    std::vector<double> values;
    std::vector<std::string> row_labels, col_labels;
};

namespace {

const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                           1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// pandas' default na_values (io/parsers: STR_NA_VALUES)
bool is_na(const char* s, size_t n) {
    static const char* const na[] = {"",     "#N/A", "#N/A N/A", "#NA",  "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN",
                                     "<NA>", "N/A",  "NA",       "NULL", "NaN",     "None",     "n/a",  "nan",  "null"};
    for (const char* t : na)
        if (strlen(t) == n && memcmp(t, s, n) == 0) return true;
    return false;
}

// 0 = parsed, 1 = not in the exactly-reproducible subset
int parse_field(const char* s, size_t n, double* out) {
    if (is_na(s, n)) {
        *out = std::numeric_limits<double>::quiet_NaN();
        return 0;
    }
    const char* p = s;
    const char* end = s + n;
    bool neg = false;
    if (*p == '-' || *p == '+') neg = *p++ == '-';
    if (end - p == 3 && (memcmp(p, "inf", 3) == 0 || memcmp(p, "Inf", 3) == 0)) {
        *out = neg ? -std::numeric_limits<double>::infinity() : std::numeric_limits<double>::infinity();
        return 0;
    }
    uint64_t mant = 0;
    int sig = 0, e10 = 0, digit_chars = 0;
    bool any = false, seen_dot = false;
    for (; p < end; p++) {
        const char c = *p;
        if (c >= '0' && c <= '9') {
            any = true;
            // pandas' converter keeps the first 17 digit CHARACTERS (leading zeros included) and drops
            // the rest: longer fields are truncated there, not rounded
            if (++digit_chars > 17) return 1;
            if (mant == 0 && c == '0') {  // leading zeros carry no digits
                if (seen_dot) e10--;
                continue;
            }
            if (++sig > 15) return 1;
            mant = mant * 10 + (uint64_t)(c - '0');
            if (seen_dot) e10--;
        } else if (c == '.' && !seen_dot) {
            seen_dot = true;
        } else {
            break;
        }
    }
    if (!any) return 1;
    // "-0", "-000": in a column of nothing but integers pandas parses int64 and the sign of the zero is gone,
    // in any other column it is -0.0 — the field alone does not say which
    if (neg && mant == 0 && !seen_dot && p == end) return 1;
    if (p < end) {
        if (*p != 'e' && *p != 'E') return 1;
        p++;
        bool eneg = false;
        if (p < end && (*p == '-' || *p == '+')) eneg = *p++ == '-';
        if (p == end) return 1;
        int ex = 0;
        for (; p < end; p++) {
            if (*p < '0' || *p > '9') return 1;
            ex = ex * 10 + (*p - '0');
            if (ex > 400) return 1;
        }
        e10 += eneg ? -ex : ex;
    }
    double v = (double)mant;  // < 10^15 < 2^53: exact
    if (mant != 0) {
        if (e10 > 22 || e10 < -22) return 1;
        v = e10 >= 0 ? v * kPow10[e10] : v / kPow10[-e10];  // two exact doubles, one rounding
    }
    *out = neg ? -v : v;
    return 0;
}

// Next field of a line: [*p, end).  Handles "quoted, ""fields""" (label columns).  Returns false on
// an unbalanced quote.  `text` receives the unquoted content.
bool next_field(const char*& p, const char* end, std::string* text, const char** raw, size_t* raw_n) {
    if (p < end && *p == '"') {
        std::string t;
        const char* q = p + 1;
        for (;;) {
            if (q >= end) return false;
            if (*q == '"') {
                if (q + 1 < end && q[1] == '"') {
                    t.push_back('"');
                    q += 2;
                    continue;
                }
                q++;
                break;
            }
            t.push_back(*q++);
        }
        if (q < end && *q != ',') return false;
        if (text) *text = std::move(t);
        *raw = nullptr;
        *raw_n = 0;
        p = q < end ? q + 1 : end + 1;
        return true;
    }
    const char* q = (const char*)memchr(p, ',', end - p);
    if (!q) q = end;
    if (text) text->assign(p, q - p);
    *raw = p;
    *raw_n = q - p;
    p = q < end ? q + 1 : end + 1;  // end + 1: no more fields
    return true;
}

struct Mapped {
    const char* data = nullptr;
    size_t size = 0;
    int fd = -1;
    ~Mapped() {
        if (data && size) munmap((void*)data, size);
        if (fd >= 0) close(fd);
    }
};

}  // namespace

extern "C" int skr_csv_read(const char* path, int threads, skr_csv** out) {
    SKR_REQUIRE(path && out, "NULL argument");
    *out = nullptr;
    Mapped f;
    f.fd = open(path, O_RDONLY);
    if (f.fd < 0) return skr_set_error(SKR_ERR_IO, "cannot open %s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(f.fd, &st) != 0) return skr_set_error(SKR_ERR_IO, "cannot stat %s: %s", path, strerror(errno));
    f.size = (size_t)st.st_size;
    if (f.size == 0) return skr_set_error(SKR_ERR_UNSUPPORTED, "%s is empty", path);
    void* m = mmap(nullptr, f.size, PROT_READ, MAP_PRIVATE, f.fd, 0);
    if (m == MAP_FAILED) {
        f.size = 0;
        return skr_set_error(SKR_ERR_IO, "cannot map %s: %s", path, strerror(errno));
    }
    f.data = (const char*)m;
    // ---- line table (blank lines are skipped, as pandas does; "\r\n" tolerated)
    std::vector<std::pair<const char*, const char*>> lines;
    for (const char* p = f.data; p < f.data + f.size;) {
        const char* nl = (const char*)memchr(p, '\n', f.data + f.size - p);
        const char* e = nl ? nl : f.data + f.size;
        const char* le = e;
        if (le > p && le[-1] == '\r') le--;
        if (le > p) lines.emplace_back(p, le);
        p = e + 1;
    }
    if (lines.size() < 1) return skr_set_error(SKR_ERR_UNSUPPORTED, "%s has no header line", path);
    skr_csv* csv = new skr_csv();
    auto fail = [&](const char* why, int64_t line) {
        delete csv;
        return skr_set_error(SKR_ERR_UNSUPPORTED, "%s, line %lld: %s (outside the natively parsed subset)", path,
                             (long long)line + 1, why);
    };
    {  // header: first cell names the index (empty in DataFrame.to_csv output), the rest are the columns
        const char* p = lines[0].first;
        const char* end = lines[0].second;
        const char* raw;
        size_t raw_n;
        std::string cell;
        bool first = true;
        while (p <= end) {
            if (!next_field(p, end, &cell, &raw, &raw_n)) return fail("unbalanced quote", 0);
            if (first) first = false; else csv->col_labels.push_back(cell);
        }
    }
    csv->cols = (int64_t)csv->col_labels.size();
    csv->rows = (int64_t)lines.size() - 1;
    if (csv->cols == 0) return fail("no data columns", 0);
    csv->values.resize((size_t)csv->rows * csv->cols);
    csv->row_labels.resize(csv->rows);
    int nthreads = threads > 0 ? std::min(threads, 256) : (int)std::max(1u, std::min(std::thread::hardware_concurrency(), 64u));
    nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, csv->rows));
    std::atomic<int64_t> bad_line{-1};
    std::atomic<int> bad_kind{0};
    auto work = [&](int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1 && bad_line.load(std::memory_order_relaxed) < 0; r++) {
            const char* p = lines[r + 1].first;
            const char* end = lines[r + 1].second;
            const char* raw;
            size_t raw_n;
            int kind = 0;
            if (!next_field(p, end, &csv->row_labels[r], &raw, &raw_n)) kind = 1;
            double* row = csv->values.data() + (size_t)r * csv->cols;
            int64_t c = 0;
            while (!kind && p <= end) {
                const char* q = (const char*)memchr(p, ',', end - p);
                if (!q) q = end;
                if (c >= csv->cols) kind = 2;
                else if (parse_field(p, q - p, &row[c])) kind = 3;
                c++;
                p = q + 1;
            }
            if (!kind && c != csv->cols) kind = 2;
            if (kind) {
                int64_t expect = -1;
                if (bad_line.compare_exchange_strong(expect, r + 1)) bad_kind = kind;
                return;
            }
        }
    };
    std::vector<std::thread> pool;
    const int64_t per = (csv->rows + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; t++) {
        const int64_t r0 = std::min<int64_t>(csv->rows, t * per), r1 = std::min<int64_t>(csv->rows, r0 + per);
        if (r1 > r0) pool.emplace_back(work, r0, r1);
    }
    for (auto& th : pool) th.join();
    if (bad_line >= 0) {
        static const char* const why[] = {"", "unbalanced quote", "ragged line", "a cell pandas' converter may round differently or text"};
        return fail(why[bad_kind.load()], bad_line.load());
    }
    *out = csv;
    return SKR_OK;
}

extern "C" int skr_csv_shape(const skr_csv* csv, int64_t* rows, int64_t* cols) {
    SKR_REQUIRE(csv && rows && cols, "NULL argument");
    *rows = csv->rows;
    *cols = csv->cols;
    return SKR_OK;
}

extern "C" int skr_csv_values(const skr_csv* csv, double* out) {
    SKR_REQUIRE(csv && (out || csv->values.empty()), "NULL argument");
    if (!csv->values.empty()) memcpy(out, csv->values.data(), csv->values.size() * sizeof(double));
    return SKR_OK;
}

extern "C" int skr_csv_labels(const skr_csv* csv, int which, char* buf, int64_t cap, int64_t* needed) {
    SKR_REQUIRE(csv && needed, "NULL argument");
    const std::vector<std::string>& v = which == 0 ? csv->row_labels : csv->col_labels;
    int64_t total = 0;
    for (const auto& s : v) total += (int64_t)s.size() + 1;
    *needed = total;
    if (!buf || cap < total) return SKR_OK;  // caller sizes the buffer from *needed and calls again
    char* p = buf;
    for (const auto& s : v) {
        memcpy(p, s.data(), s.size());
        p += s.size();
        *p++ = '\n';
    }
    return SKR_OK;
}

extern "C" int skr_csv_free(skr_csv* csv) {
    delete csv;
    return SKR_OK;
}

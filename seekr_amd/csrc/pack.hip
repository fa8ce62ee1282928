// K1 — host packer: ASCII bases -> 2 bits/base words + validity bits, and the FASTA reader.
//
// Layout in HBM (what count.hip consumes):
//   packed   : uint32 words, 16 bases per word, base i of a sequence in word i/16 at bits
//              [30 - 2*(i%16), 31 - 2*(i%16)]  (first base in the top bits, so that a window's
//              bits read as an integer ARE its column index: first base most significant,
//              kmer_counts.py:121-122).  Every sequence starts on a word boundary and is
//              followed by one zero pad word (the kernel reads word w+1 for the window halo).
//   word_off : int64 [n+1], first word of each sequence
//   len      : int64 [n], length in characters (every character counts, kmer_counts.py:143)
//   mask     : uint32 words, 1 bit per base (bit i%32 of word i/32, 1 = not in the alphabet),
//              stored only for sequences that contain such a base; mask_off[i] = -1 otherwise.
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <thread>

#include "common.hpp"

namespace {

struct Lut {
    uint8_t code[256];
};

int make_lut(const char alphabet[4], bool fold_upper, Lut* lut) {
    memset(lut->code, 0xFF, sizeof(lut->code));
    for (int c = 0; c < 4; c++) {
        unsigned char ch = (unsigned char)alphabet[c];
        if (lut->code[ch] != 0xFF)
            return skr_set_error(SKR_ERR_UNSUPPORTED,
                                 "alphabet must hold 4 distinct characters on the MI355X path (got a repeated '%c')", ch);
        lut->code[ch] = (uint8_t)c;
    }
    if (fold_upper) {
        // the reader upper-cases sequences (fasta_reader.py:55,62): 'a' counts as 'A'
        for (int ch = 'a'; ch <= 'z'; ch++) lut->code[ch] = lut->code[ch - 'a' + 'A'];
    }
    return SKR_OK;
}

struct SeqView {
    const char* p;
    int64_t len;
};

// pack sequences [first, last) into the host staging arrays
void pack_range(const std::vector<SeqView>& seqs, int64_t first, int64_t last, const Lut& lut,
                const std::vector<int64_t>& word_off, const std::vector<int64_t>& mask_off, uint32_t* packed,
                uint32_t* mask) {
    for (int64_t s = first; s < last; s++) {
        const unsigned char* src = (const unsigned char*)seqs[s].p;
        const int64_t len = seqs[s].len;
        uint32_t* dst = packed + word_off[s];
        uint32_t* mdst = mask_off[s] >= 0 ? mask + mask_off[s] : nullptr;
        const int64_t nw = word_off[s + 1] - word_off[s];
        for (int64_t w = 0; w < nw; w++) {
            uint32_t word = 0;
            const int64_t base0 = w * 16;
            const int64_t lim = std::min<int64_t>(16, len - base0);
            for (int64_t j = 0; j < lim; j++) {
                uint8_t c = lut.code[src[base0 + j]];
                word |= (uint32_t)(c & 3u) << (30 - 2 * j);
            }
            dst[w] = word;
        }
        if (mdst) {
            const int64_t nm = (len + 31) / 32 + 1;
            for (int64_t w = 0; w < nm; w++) {
                uint32_t word = 0;
                const int64_t base0 = w * 32;
                const int64_t lim = std::min<int64_t>(32, len - base0);
                for (int64_t j = 0; j < lim; j++)
                    if (lut.code[src[base0 + j]] == 0xFF) word |= 1u << j;
                mdst[w] = word;
            }
        }
    }
}

int upload_seqs(skr_ctx* ctx, const std::vector<SeqView>& seqs, const Lut& lut, std::string&& headers,
                skr_seqs** out) {
    const int64_t n = (int64_t)seqs.size();
    std::vector<int64_t> word_off(n + 1, 0), mask_off(n, -1), len(n, 0);
    int64_t total = 0, max_len = 0, mask_words = 0;
    const int nthreads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    // pass 1: which sequences hold a non-alphabet byte (parallel), then prefix sums (serial)
    std::vector<uint8_t> dirty(n, 0);
    {
        auto scan = [&](int64_t a, int64_t b) {
            for (int64_t s = a; s < b; s++) {
                const unsigned char* p = (const unsigned char*)seqs[s].p;
                uint8_t bad = 0;
                for (int64_t i = 0; i < seqs[s].len; i++) bad |= (lut.code[p[i]] == 0xFF);
                dirty[s] = bad;
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back(scan, n * t / nthreads, n * (t + 1) / nthreads);
        for (auto& t : th) t.join();
    }
    for (int64_t s = 0; s < n; s++) {
        len[s] = seqs[s].len;
        total += len[s];
        max_len = std::max(max_len, len[s]);
        word_off[s + 1] = word_off[s] + (len[s] + 15) / 16 + 1;
        if (dirty[s]) {
            mask_off[s] = mask_words;
            mask_words += (len[s] + 31) / 32 + 1;
        }
    }
    // +2: the counting kernel prefetches word w+1 / mask word mw+1 unconditionally
    std::vector<uint32_t> packed((size_t)word_off[n] + 2, 0), mask((size_t)mask_words + 2, 0);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++)
            th.emplace_back(pack_range, std::cref(seqs), n * t / nthreads, n * (t + 1) / nthreads, std::cref(lut),
                            std::cref(word_off), std::cref(mask_off), packed.data(), mask.data());
        for (auto& t : th) t.join();
    }
    SKR_TRY(skr_activate(ctx));
    skr_seqs* s = new skr_seqs();
    s->ctx = ctx;
    s->n = n;
    s->total_bases = total;
    s->max_len = max_len;
    s->n_words = word_off[n];
    s->n_mask_words = mask_words;
    s->h_len = len;
    s->headers = std::move(headers);
    auto fail = [&](int rc) {
        skr_seqs_free(s);
        return rc;
    };
#define UP(dptr, hvec, T)                                                                       \
    do {                                                                                        \
        size_t b_ = std::max<size_t>(hvec.size(), 1) * sizeof(T);                               \
        hipError_t e_ = hipMalloc((void**)&dptr, b_);                                           \
        if (e_ != hipSuccess) return fail(skr_set_error(SKR_ERR_NOMEM, "hipMalloc(%zu): %s", b_, hipGetErrorString(e_))); \
        if (!hvec.empty()) {                                                                    \
            e_ = hipMemcpy(dptr, hvec.data(), hvec.size() * sizeof(T), hipMemcpyHostToDevice);  \
            if (e_ != hipSuccess) return fail(skr_set_error(SKR_ERR_HIP, "hipMemcpy H2D: %s", hipGetErrorString(e_))); \
        }                                                                                       \
    } while (0)
    UP(s->d_packed, packed, uint32_t);
    UP(s->d_word_off, word_off, int64_t);
    UP(s->d_len, len, int64_t);
    UP(s->d_mask, mask, uint32_t);
    UP(s->d_mask_off, mask_off, int64_t);
#undef UP
    *out = s;
    return SKR_OK;
}

inline bool py_space(unsigned char c) {
    // what str.strip() removes among single bytes
    return c == ' ' || (c >= 0x09 && c <= 0x0d) || (c >= 0x1c && c <= 0x1f);
}

}  // namespace

extern "C" int skr_seqs_pack(skr_ctx* ctx, const char* bases, const int64_t* offsets, int64_t n,
                             const char alphabet[4], skr_seqs** out) {
    SKR_REQUIRE(ctx && out && alphabet, "NULL argument");
    *out = nullptr;
    SKR_REQUIRE(n >= 0, "negative sequence count");
    SKR_REQUIRE(n == 0 || (offsets && (bases || offsets[n] == offsets[0])), "bases/offsets is NULL");
    Lut lut;
    SKR_TRY(make_lut(alphabet, /*fold_upper=*/false, &lut));
    std::vector<SeqView> seqs((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        SKR_REQUIRE(offsets[i + 1] >= offsets[i], "offsets must be non-decreasing (at %lld)", (long long)i);
        seqs[i] = {bases + offsets[i], offsets[i + 1] - offsets[i]};
    }
    return upload_seqs(ctx, seqs, lut, std::string(), out);
}

extern "C" int skr_seqs_from_fasta(skr_ctx* ctx, const char* path, const char alphabet[4], skr_seqs** out) {
    SKR_REQUIRE(ctx && out && alphabet && path, "NULL argument");
    *out = nullptr;
    Lut lut;
    SKR_TRY(make_lut(alphabet, /*fold_upper=*/true, &lut));
    int fd = open(path, O_RDONLY);
    if (fd < 0) return skr_set_error(SKR_ERR_IO, "cannot open '%s': %s", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        return skr_set_error(SKR_ERR_IO, "cannot stat '%s'", path);
    }
    const size_t fsize = (size_t)st.st_size;
    const char* data = nullptr;
    if (fsize > 0) {
        data = (const char*)mmap(nullptr, fsize, PROT_READ, MAP_PRIVATE, fd, 0);
        if (data == MAP_FAILED) {
            close(fd);
            return skr_set_error(SKR_ERR_IO, "cannot mmap '%s'", path);
        }
    }
    close(fd);
    // Pass over the lines (universal newlines: \n, \r\n, \r).  A sequence spread over several
    // lines is not contiguous in the file, so joined sequences are materialised in `joined`.
    std::string joined;
    joined.reserve(fsize);
    std::vector<std::pair<int64_t, int64_t>> spans;  // (start, len) into `joined`
    std::string headers;
    int rc = SKR_OK;
    int64_t lineno = 0;
    bool have_pending = false;   // a sequence (possibly empty) is being accumulated
    int64_t pending_start = 0;
    bool first_line = true;
    size_t pos = 0;
    while (pos < fsize && rc == SKR_OK) {
        size_t eol = pos;
        while (eol < fsize && data[eol] != '\n' && data[eol] != '\r') eol++;
        size_t next = eol;
        if (next < fsize) next += (data[next] == '\r' && next + 1 < fsize && data[next + 1] == '\n') ? 2 : 1;
        size_t a = pos, b = eol;
        while (a < b && py_space((unsigned char)data[a])) a++;
        while (b > a && py_space((unsigned char)data[b - 1])) b--;
        if (a == b) {
            rc = skr_set_error(SKR_ERR_FASTA_BLANK, "string index out of range");  // fasta_reader.py:53
            break;
        }
        if (data[a] == '>') {
            if (have_pending && (int64_t)joined.size() > pending_start) {
                spans.emplace_back(pending_start, (int64_t)joined.size() - pending_start);
            } else if (lineno != 0) {
                rc = skr_set_error(SKR_ERR_FASTA_HEADER, "There may be a header without a sequence at line %lld.",
                                   (long long)lineno);  // fasta_reader.py:58
                break;
            }
            if (!headers.empty()) headers += '\n';
            headers.append(data + a, b - a);
            have_pending = true;
            pending_start = (int64_t)joined.size();
        } else {
            if (first_line) {
                rc = skr_set_error(SKR_ERR_INVALID, "'%s' does not start with a '>' header line", path);
                break;
            }
            joined.append(data + a, b - a);
        }
        first_line = false;
        lineno++;
        pos = next;
    }
    if (rc == SKR_OK && have_pending) spans.emplace_back(pending_start, (int64_t)joined.size() - pending_start);
    if (data) munmap((void*)data, fsize);
    if (rc != SKR_OK) return rc;
    std::vector<SeqView> seqs(spans.size());
    for (size_t i = 0; i < spans.size(); i++) seqs[i] = {joined.data() + spans[i].first, spans[i].second};
    return upload_seqs(ctx, seqs, lut, std::move(headers), out);
}

extern "C" int skr_seqs_free(skr_seqs* s) {
    if (!s) return SKR_OK;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->d_packed) (void)hipFree(s->d_packed);
    if (s->d_word_off) (void)hipFree(s->d_word_off);
    if (s->d_len) (void)hipFree(s->d_len);
    if (s->d_mask) (void)hipFree(s->d_mask);
    if (s->d_mask_off) (void)hipFree(s->d_mask_off);
    delete s;
    return SKR_OK;
}

extern "C" int skr_seqs_info(const skr_seqs* s, int64_t* n, int64_t* total_bases, int64_t* max_len) {
    SKR_REQUIRE(s, "seqs is NULL");
    if (n) *n = s->n;
    if (total_bases) *total_bases = s->total_bases;
    if (max_len) *max_len = s->max_len;
    return SKR_OK;
}

extern "C" int skr_seqs_lengths(const skr_seqs* s, int64_t* lengths) {
    SKR_REQUIRE(s && (lengths || s->n == 0), "NULL argument");
    if (s->n) memcpy(lengths, s->h_len.data(), (size_t)s->n * sizeof(int64_t));
    return SKR_OK;
}

extern "C" int skr_seqs_headers(const skr_seqs* s, char* buf, int64_t cap, int64_t* needed) {
    SKR_REQUIRE(s, "seqs is NULL");
    if (needed) *needed = (int64_t)s->headers.size() + 1;
    if (buf && cap > 0) {
        size_t ncopy = std::min<size_t>((size_t)cap - 1, s->headers.size());
        memcpy(buf, s->headers.data(), ncopy);
        buf[ncopy] = 0;
    }
    return SKR_OK;
}

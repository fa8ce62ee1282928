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
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
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
void pack_range(const SeqView* seqs, int64_t first, int64_t last, const Lut& lut,
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

// Host-side hooks (read where a file is opened or a set is packed, never on a launch path)
int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
// Threads of the reader and the packer: SEEKR_HOST_THREADS (default 16), at most the cores there are
int host_threads() {
    const unsigned cap = (unsigned)std::max(1, env_int("SEEKR_HOST_THREADS", 16));
    return (int)std::max(1u, std::min(cap, std::thread::hardware_concurrency()));
}

int upload_seqs(skr_ctx* ctx, const std::vector<SeqView>& all_seqs, const Lut& lut, std::string&& headers,
                skr_seqs** out, int64_t first = 0, int64_t count = -1) {
    // sequences [first, first + count) of `all_seqs` (count < 0: all of them)
    const SeqView* seqs = all_seqs.data() + first;
    const int64_t n = count < 0 ? (int64_t)all_seqs.size() - first : count;
    std::vector<int64_t> word_off(n + 1, 0), mask_off(n, -1), len(n, 0);
    int64_t total = 0, max_len = 0, mask_words = 0;
    const int nthreads = host_threads();
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
            th.emplace_back(pack_range, seqs, n * t / nthreads, n * (t + 1) / nthreads, std::cref(lut),
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
            e_ = hipMemcpyAsync(dptr, hvec.data(), hvec.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream);  \
            if (e_ != hipSuccess) return fail(skr_set_error(SKR_ERR_HIP, "hipMemcpy H2D: %s", hipGetErrorString(e_))); \
        }                                                                                       \
    } while (0)
    UP(s->d_packed, packed, uint32_t);
    UP(s->d_word_off, word_off, int64_t);
    UP(s->d_len, len, int64_t);
    UP(s->d_mask, mask, uint32_t);
    UP(s->d_mask_off, mask_off, int64_t);
#undef UP
    // The copies ride the ctx's own stream (the host vectors live until this sync).  They used to be synchronous
    // hipMemcpy calls — the NULL stream — and when those were the first copies of a process (BasicCounter(infasta) before
    // anything else) every later host <-> device copy on the ctx's stream ran at HALF rate for the life of the process:
    // 190 MB up in 6.6 instead of 3.5 ms, 549 MB down in 19 instead of 10 ms (tools/e2e_probe3.py; the runtime hands out
    // its copy engines at first use).  Nothing in this library touches the NULL stream any more.
    {
        hipError_t e_ = hipStreamSynchronize(ctx->stream);
        if (e_ != hipSuccess) return fail(skr_set_error(SKR_ERR_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e_)));
    }
    *out = s;
    return SKR_OK;
}

inline bool py_space(unsigned char c) {
    // what str.strip() removes among single bytes
    return c == ' ' || (c >= 0x09 && c <= 0x0d) || (c >= 0x1c && c <= 0x1f);
}

// Any byte >= 0x80 in p[0, n)?  Eight bytes a step (the compiler widens it further): far below the cost of the parse.
inline bool has_high_byte(const char* p, size_t n) {
    uint64_t acc = 0;
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        memcpy(w, p + i, 32);
        acc |= (w[0] | w[1]) | (w[2] | w[3]);
    }
    for (; i < n; i++) acc |= (uint64_t)(unsigned char)p[i];
    return (acc & 0x8080808080808080ull) != 0;
}

// What one thread learns from the lines of data[begin, end) (begin is a line start).
struct FastaPiece {
    std::string joined;                // sequence characters of the piece, stripped lines back to back
    std::vector<int64_t> seq_start;    // per header of the piece: offset in `joined` where its sequence begins
    std::vector<int64_t> header_line;  // per header: line index inside the piece
    std::string headers;               // header texts, '\n'-joined
    int64_t n_lines = 0;
    int64_t blank_line = -1;           // first line that is empty after strip()
    bool first_is_header = false;
    bool high_byte = false;            // a byte >= 0x80: the piece was not parsed (see skr_fasta_open)
};

void parse_piece(const char* data, size_t begin, size_t end, FastaPiece* out) {
    if (has_high_byte(data + begin, end - begin)) {
        out->high_byte = true;
        return;
    }
    out->joined.reserve(end - begin);
    size_t pos = begin;
    while (pos < end) {
        // end of line: the first '\n' or '\r' (memchr: the loop is bound by memory, not by byte tests)
        const char* nl = (const char*)memchr(data + pos, '\n', end - pos);
        const size_t lim = nl ? (size_t)(nl - data) : end;
        const char* cr = (const char*)memchr(data + pos, '\r', lim - pos);
        const size_t eol = cr ? (size_t)(cr - data) : lim;
        size_t next = eol;
        if (next < end) next += (data[next] == '\r' && next + 1 < end && data[next + 1] == '\n') ? 2 : 1;
        size_t a = pos, b = eol;
        while (a < b && py_space((unsigned char)data[a])) a++;
        while (b > a && py_space((unsigned char)data[b - 1])) b--;
        if (a == b) {
            if (out->blank_line < 0) out->blank_line = out->n_lines;
        } else if (data[a] == '>') {
            if (out->n_lines == 0) out->first_is_header = true;
            if (!out->seq_start.empty()) out->headers += '\n';
            out->headers.append(data + a, b - a);
            out->seq_start.push_back((int64_t)out->joined.size());
            out->header_line.push_back(out->n_lines);
        } else {
            out->joined.append(data + a, b - a);
        }
        out->n_lines++;
        pos = next;
    }
}

}  // namespace

// A FASTA file parsed into host memory (skr_fasta_open): the reader's work done once, the sequences then packed and
// uploaded whole (skr_seqs_from_fasta) or range by range, each range by the GPU that counts it (skr_fasta_pack).
struct skr_fasta {
    std::vector<FastaPiece> pieces;     // own the sequence bytes
    std::vector<std::string> bridges;   // sequences that ran over a piece boundary, copied together
    std::vector<SeqView> seqs;          // views into the two above
    std::vector<size_t> header_at;      // per sequence: where its header starts in `headers`; one more = the end + 1
    std::string headers;                // '\n'-joined
    int64_t total_bases = 0;
};

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

extern "C" int skr_fasta_open(const char* path, skr_fasta** out) {
    SKR_REQUIRE(out && path, "NULL argument");
    *out = nullptr;
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
    // Lines (universal newlines: \n, \r\n, \r) are parsed by up to 16 threads, each on a piece of the
    // file cut at line starts; a piece yields its sequence bytes, where each of its headers' sequences
    // begins in them, and its first blank line.  The pieces are then stitched in file order: sequence
    // lines at the head of a piece continue the last sequence of the piece before, and the checks that
    // need the neighbouring lines (a header that follows a header) run on the stitched list.  The
    // error reported is the one the reference's sequential loop would have reached first.
    const int want_threads = host_threads();
    const size_t piece_min = (size_t)std::max(1, env_int("SEEKR_FASTA_PIECE_BYTES", 1 << 20));  // bytes below which another thread is not worth starting (tests: force stitching)
    const int n_pieces = (int)std::max<size_t>(1, std::min<size_t>((size_t)want_threads, fsize / piece_min));
    std::vector<size_t> cut((size_t)n_pieces + 1, fsize);
    cut[0] = 0;
    auto line_start = [&](size_t p) {
        return p == 0 || data[p - 1] == '\n' || (data[p - 1] == '\r' && data[p] != '\n');
    };
    for (int i = 1; i < n_pieces; i++) {
        size_t p = std::max(cut[i - 1], fsize / (size_t)n_pieces * (size_t)i);
        while (p < fsize && !line_start(p)) p++;
        cut[i] = p;
    }
    std::unique_ptr<skr_fasta> fa(new skr_fasta());
    fa->pieces.resize((size_t)n_pieces);
    std::vector<FastaPiece>& pieces = fa->pieces;
    {
        std::vector<std::thread> th;
        for (int i = 1; i < n_pieces; i++) th.emplace_back(parse_piece, data, cut[i], cut[i + 1], &pieces[i]);
        parse_piece(data, cut[0], cut[1], &pieces[0]);
        for (auto& t : th) t.join();
    }
    // The reference opens the file in text mode (fasta_reader.py:44): bytes are decoded before strip() / upper() / len()
    // see them, so a multi-byte character counts once in len(seq) (kmer_counts.py:143-144), Unicode white space (NBSP,
    // NEL, U+2028 ...) is stripped, upper() may change the length, and an undecodable byte raises.  None of that is a
    // byte-level rule; a file with any byte >= 0x80 is declined here and read by the text-mode Reader of the package.
    for (int i = 0; i < n_pieces; i++)
        if (pieces[i].high_byte) {
            if (data) munmap((void*)data, fsize);
            return skr_set_error(SKR_ERR_FASTA_TEXT, "'%s' holds bytes >= 0x80: it is read in text mode, as the reference does", path);
        }
    // ---- stitch
    int rc = SKR_OK;
    int64_t err_line = INT64_MAX;  // line of the first error in file order
    std::vector<int64_t> joined_off((size_t)n_pieces + 1, 0), line_off((size_t)n_pieces + 1, 0);
    size_t n_headers = 0, header_bytes = 0;
    for (int i = 0; i < n_pieces; i++) {
        joined_off[i + 1] = joined_off[i] + (int64_t)pieces[i].joined.size();
        line_off[i + 1] = line_off[i] + pieces[i].n_lines;
        n_headers += pieces[i].seq_start.size();
        header_bytes += pieces[i].headers.size() + 1;
    }
    const int64_t total_joined = joined_off[n_pieces];
    if (line_off[n_pieces] > 0 && !pieces[0].first_is_header && pieces[0].blank_line != 0) {
        rc = skr_set_error(SKR_ERR_INVALID, "'%s' does not start with a '>' header line", path);
        err_line = 0;
    }
    for (int i = 0; i < n_pieces; i++)
        if (pieces[i].blank_line >= 0 && line_off[i] + pieces[i].blank_line < err_line) {
            err_line = line_off[i] + pieces[i].blank_line;
            rc = skr_set_error(SKR_ERR_FASTA_BLANK, "string index out of range");  // fasta_reader.py:53
            break;  // later pieces only hold later lines
        }
    std::vector<int64_t> starts, hdr_lines;  // per sequence: first byte in the stitched buffer, line of its header
    starts.reserve(n_headers + 1);
    hdr_lines.reserve(n_headers);
    std::string headers;
    headers.reserve(header_bytes);
    for (int i = 0; i < n_pieces; i++) {
        for (size_t h = 0; h < pieces[i].seq_start.size(); h++) {
            starts.push_back(joined_off[i] + pieces[i].seq_start[h]);
            hdr_lines.push_back(line_off[i] + pieces[i].header_line[h]);
        }
        if (!pieces[i].headers.empty()) {
            if (!headers.empty()) headers += '\n';
            headers += pieces[i].headers;
        }
    }
    starts.push_back(total_joined);
    for (size_t j = 1; j < hdr_lines.size(); j++)
        if (starts[j] == starts[j - 1]) {  // header j follows header j-1 without sequence bytes in between
            if (hdr_lines[j] < err_line) {
                err_line = hdr_lines[j];
                rc = skr_set_error(SKR_ERR_FASTA_HEADER, "There may be a header without a sequence at line %lld.",
                                   (long long)hdr_lines[j]);  // fasta_reader.py:58
            }
            break;
        }
    if (rc != SKR_OK) {
        if (data) munmap((void*)data, fsize);
        return rc;
    }
    if (data) munmap((void*)data, fsize);
    // Sequences stay where the pieces put them; only one that runs over a piece boundary (at most one
    // per boundary) is copied together.
    std::vector<SeqView>& seqs = fa->seqs;
    seqs.resize(hdr_lines.size());
    std::vector<std::string>& bridges = fa->bridges;
    bridges.reserve((size_t)n_pieces);  // never reallocated: the views point into its strings
    int pc = 0;
    for (size_t j = 0; j < seqs.size(); j++) {
        const int64_t lo = starts[j], hi = starts[j + 1];
        while (pc + 1 < n_pieces && lo >= joined_off[pc + 1]) pc++;
        if (hi <= joined_off[pc + 1]) {
            seqs[j] = {pieces[pc].joined.data() + (lo - joined_off[pc]), hi - lo};
        } else {
            std::string whole;
            whole.reserve((size_t)(hi - lo));
            for (int q = pc; q < n_pieces && joined_off[q] < hi; q++) {
                const int64_t a0 = std::max(lo, joined_off[q]), a1 = std::min(hi, joined_off[q + 1]);
                if (a1 > a0) whole.append(pieces[q].joined.data() + (a0 - joined_off[q]), (size_t)(a1 - a0));
            }
            bridges.push_back(std::move(whole));
            seqs[j] = {bridges.back().data(), hi - lo};
        }
    }
    fa->total_bases = total_joined;
    fa->header_at.reserve(seqs.size() + 1);
    fa->header_at.push_back(0);
    for (size_t i = 0; i < headers.size(); i++)
        if (headers[i] == '\n') fa->header_at.push_back(i + 1);
    fa->header_at.push_back(headers.size() + 1);
    fa->headers = std::move(headers);
    *out = fa.release();
    return SKR_OK;
}

extern "C" int skr_fasta_free(skr_fasta* fa) {
    delete fa;
    return SKR_OK;
}

extern "C" int skr_fasta_info(const skr_fasta* fa, int64_t* n, int64_t* total_bases) {
    SKR_REQUIRE(fa, "fasta is NULL");
    if (n) *n = (int64_t)fa->seqs.size();
    if (total_bases) *total_bases = fa->total_bases;
    return SKR_OK;
}

extern "C" int skr_fasta_lengths(const skr_fasta* fa, int64_t* lengths) {
    SKR_REQUIRE(fa && (lengths || fa->seqs.empty()), "NULL argument");
    for (size_t i = 0; i < fa->seqs.size(); i++) lengths[i] = fa->seqs[i].len;
    return SKR_OK;
}

extern "C" int skr_fasta_headers(const skr_fasta* fa, char* buf, int64_t cap, int64_t* needed) {
    SKR_REQUIRE(fa, "fasta is NULL");
    if (needed) *needed = (int64_t)fa->headers.size() + 1;
    if (buf && cap > 0) {
        size_t ncopy = std::min<size_t>((size_t)cap - 1, fa->headers.size());
        memcpy(buf, fa->headers.data(), ncopy);
        buf[ncopy] = 0;
    }
    return SKR_OK;
}

// Sequences [first, first + count) packed onto ctx's GPU.  Only reads `fa`: the GPUs of a node pack their ranges at once.
extern "C" int skr_fasta_pack(skr_ctx* ctx, const skr_fasta* fa, int64_t first, int64_t count, const char alphabet[4],
                              skr_seqs** out) {
    SKR_REQUIRE(ctx && fa && out && alphabet, "NULL argument");
    *out = nullptr;
    const int64_t n = (int64_t)fa->seqs.size();
    SKR_REQUIRE(first >= 0 && count >= 0 && first + count <= n, "sequences [%lld, %lld) outside 0..%lld", (long long)first,
                (long long)(first + count), (long long)n);
    Lut lut;
    SKR_TRY(make_lut(alphabet, /*fold_upper=*/true, &lut));
    std::string headers;
    if (count > 0 && fa->header_at.size() == (size_t)n + 1) {
        const size_t h0 = fa->header_at[(size_t)first], h1 = fa->header_at[(size_t)(first + count)] - 1;
        headers.assign(fa->headers, h0, h1 - h0);
    }
    return upload_seqs(ctx, fa->seqs, lut, std::move(headers), out, first, count);
}

extern "C" int skr_seqs_from_fasta(skr_ctx* ctx, const char* path, const char alphabet[4], skr_seqs** out) {
    SKR_REQUIRE(ctx && out && alphabet && path, "NULL argument");
    *out = nullptr;
    Lut lut;
    SKR_TRY(make_lut(alphabet, /*fold_upper=*/true, &lut));  // a bad alphabet is reported before the file is read
    skr_fasta* fa = nullptr;
    SKR_TRY(skr_fasta_open(path, &fa));
    const int rc = skr_fasta_pack(ctx, fa, 0, (int64_t)fa->seqs.size(), alphabet, out);
    skr_fasta_free(fa);
    return rc;
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

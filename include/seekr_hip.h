/* seekr_hip.h — C-ABI of libseekr_hip.so, the MI355X (gfx950) implementation of SEEKR's
 * k-mer counting + column normalisation + all-pairs Pearson hot path.
 *
 * The reference (CalabreseLab/seekr v2.0.2) is pure Python and has no FFI of its own
 * (SURVEY.md §8b); the boundary it exposes for this path is its Python API.  Every entry
 * point below therefore cites the reference *Python* lines whose result it must reproduce;
 * `seekr_amd/` is the ctypes host that re-creates the reference classes on top of these
 * symbols, and INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; no exceptions, no callbacks, no torch types
 *   - every function returns an int status (SKR_OK == 0, negative == error) and leaves a
 *     human-readable message retrievable with skr_last_error() (thread-local)
 *   - the caller owns every host buffer; the library owns device memory behind the opaque
 *     handles (skr_ctx / skr_seqs / skr_mat) until the matching *_free / *_destroy
 *   - one skr_ctx == one GPU + one HIP stream; work submitted through a ctx is ordered on
 *     that stream; functions that fill host memory synchronise before returning
 *   - matrices are dense row-major (C order), float32 unless stated otherwise
 */
#ifndef SEEKR_HIP_H
#define SEEKR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKR_ABI_VERSION 1

/* status codes */
enum {
    SKR_OK = 0,
    SKR_ERR_INVALID = -1,     /* bad argument (NULL, shape/dtype mismatch, k out of range …)      */
    SKR_ERR_HIP = -2,         /* a HIP runtime call failed (message carries hipGetErrorString)    */
    SKR_ERR_NOMEM = -3,       /* host or device allocation failed                                 */
    SKR_ERR_UNSUPPORTED = -4, /* valid request the device path does not implement                 */
    SKR_ERR_ZERODIV = -5,     /* a sequence has len == k-1 (kmer_counts.py:144 ZeroDivisionError) */
    SKR_ERR_COMM = -6,        /* RCCL failure / communicator not initialised                      */
    SKR_ERR_IO = -7,          /* file could not be read or written                                */
    SKR_ERR_FASTA_BLANK = -8, /* blank line in FASTA (fasta_reader.py:53 IndexError)              */
    SKR_ERR_FASTA_HEADER = -9,/* header without sequence (fasta_reader.py:58 AssertionError)      */
    SKR_ERR_FASTA_TEXT = -10  /* the file holds a byte >= 0x80: the reference decodes it in text mode
                               * (fasta_reader.py:44) before strip / upper / len; the byte parser
                               * declines it and the caller reads it through a text-mode reader    */
};

/* element types of a skr_mat */
enum { SKR_F32 = 0, SKR_F64 = 1, SKR_U32 = 2 };

/* log2 handling of BasicCounter (kmer_counts.py:201-209) */
enum { SKR_LOG2_NONE = 0, SKR_LOG2_PRE = 1, SKR_LOG2_POST = 2 };

/* arithmetic of the Pearson contraction (pearson.py:41) */
enum {
    SKR_PREC_FP32 = 0,   /* v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate           */
    SKR_PREC_BF16X3 = 1, /* split-bf16: hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16       */
    SKR_PREC_F64 = 2,    /* v_mfma_f64_16x16x4_f64 (float64 inputs: CSV / integer count files)   */
    SKR_PREC_BF16X4 = 3, /* RETIRED in round 5 (split-bf16 with the lo*lo term: 25 % slower than BF16X3 with the same
                          * 16-bit residual, no use left): the value stays reserved, every entry point answers SKR_ERR_INVALID */
    SKR_PREC_F16F8 = 5,  /* opt-in (round 4): hi*hi on v_mfma_f32_16x16x32_f16, the two cross terms as ONE block-scaled
                          * fp8 product (v_mfma_scale_f32_16x16x128_f8f6f4): 2 product-units per k instead of 3; for
                          * row-standardised rows of 4 096 / 16 384 columns (k = 6, 7); any other shape and rows flagged as
                          * few-valued fall back to SKR_PREC_F16X3 by themselves                       */
    SKR_PREC_F16X3 = 4   /* split-fp16 (11-bit halves: float32-grade operands), 3 products on    */
                         /* v_mfma_f32_16x16x32_f16.  The host API's default for row-standardised */
                         /* rows; values must fit fp16 (|z| <= sqrt(K) always does)              */
};

typedef struct skr_ctx skr_ctx;   /* one GPU + one stream + scratch + optional RCCL communicator */
typedef struct skr_seqs skr_seqs; /* a set of sequences packed 2 bits/base, resident in HBM      */
typedef struct skr_mat skr_mat;   /* a row-major device matrix                                   */

/* ---------------------------------------------------------------- library / context ---- */
const char* skr_last_error(void);
int skr_abi_version(void);
int skr_device_count(int* count);
/* Page-lock / release a host range the caller owns (hipHostRegister / hipHostUnregister): copies to and from a registered
 * range are plain DMA at the link's rate.  The package registers the result arrays it keeps between calls.  `device`: the
 * GPU the calling thread is switched to first (the runtime registers on the thread's current device; the range is
 * portable to every GPU either way; the thread's own device is restored before returning), -1 = leave it alone.         */
int skr_host_register(int device, void* ptr, size_t bytes);
int skr_host_unregister(void* ptr);
int skr_ctx_create(int device, skr_ctx** out);
int skr_ctx_destroy(skr_ctx* ctx);
int skr_ctx_sync(skr_ctx* ctx);
/* Free and total device memory of the ctx's GPU in bytes (hipMemGetInfo): callers size stripes and tests against it. */
int skr_ctx_mem_info(skr_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes);
int skr_ctx_device(const skr_ctx* ctx, int* device);
/* The SEEKR_GEMM_* / SEEKR_COUNT_* A/B switches are read from the environment when the ctx is created;
 * this reads them again (bench tools that interleave variants in one process).  No launch calls getenv.            */
int skr_ctx_reload_knobs(skr_ctx* ctx);
/* device-side timing of every kernel launched through the ctx (HIP events on the ctx stream) */
int skr_prof_enable(skr_ctx* ctx, int on);
int skr_prof_reset(skr_ctx* ctx);
/* total milliseconds and launch count recorded under exactly `name` (skr_prof_names lists the names seen).  Besides
 * the kernels: "comm_xfer" / "comm_vec" = transfers on the communication stream (row blocks / statistic vectors),
 * "comm_wait" / "comm_wait_vec" = how long the compute stream stood still waiting for one (the exposed part).   */
int skr_prof_query(skr_ctx* ctx, const char* name, double* total_ms, int64_t* launches);
/* names of all recorded kernels, '\n'-separated, into buf (truncated to cap-1 chars) */
int skr_prof_names(skr_ctx* ctx, char* buf, int64_t cap);

/* ---------------------------------------------------------------- device matrices ------ */
int skr_mat_create(skr_ctx* ctx, int64_t rows, int64_t cols, int dtype, skr_mat** out);
int skr_mat_free(skr_mat* m);
int skr_mat_shape(const skr_mat* m, int64_t* rows, int64_t* cols, int* dtype);
/* copy `nrows` rows starting at `row0` between a dense host buffer and the matrix */
int skr_mat_upload(skr_mat* m, const void* host, int64_t row0, int64_t nrows);
int skr_mat_download(const skr_mat* m, void* host, int64_t row0, int64_t nrows);
/* A download that runs BESIDE the compute stream.  skr_ctx_mark notes the point the compute stream has reached (the work
 * enqueued so far); skr_mat_download_at copies rows to the host on the ctx's copy stream as soon as that point is passed —
 * not waiting for work enqueued after the mark — and returns when the rows are in `host`.  A caller that produces r one
 * row stripe at a time enqueues the contraction of stripe s + 1, then downloads stripe s: the two overlap.  A mark is
 * used once; mark < 0 = "everything enqueued so far".                                                                */
int skr_ctx_mark(skr_ctx* ctx, int64_t* mark);
/* A mark is used up by the one call it is passed to — also when that call refuses its other arguments.  A mark that will
 * never be passed on is handed back with skr_ctx_mark_release (unknown / used marks are ignored).                        */
int skr_ctx_mark_release(skr_ctx* ctx, int64_t mark);
int skr_mat_download_at(const skr_mat* m, void* host, int64_t row0, int64_t nrows, int64_t mark);
int skr_mat_fill_zero(skr_mat* m);
/* non-owning view of rows [row0, row0+nrows) of `parent`; free it with skr_mat_free before the parent */
int skr_mat_view(const skr_mat* parent, int64_t row0, int64_t nrows, skr_mat** out);
/* raw device pointer (for RCCL / interop); valid until skr_mat_free */
int skr_mat_device_ptr(const skr_mat* m, void** ptr);

/* ---------------------------------------------------------------- sequences (K1) -------- */
/* Pack `n` sequences given as one concatenated ASCII buffer + n+1 byte offsets.
 * alphabet[c] is the character whose code is c (kmer_counts.py:121-122: code = position in
 * `alphabet`, default "AGTC"); any other byte marks every window that covers it as skipped
 * (kmer_counts.py:149).  No case folding is applied here (the reader upper-cases,
 * fasta_reader.py:55,62; a caller assigning `seqs` directly gets lower case skipped).      */
int skr_seqs_pack(skr_ctx* ctx, const char* bases, const int64_t* offsets, int64_t n,
                  const char alphabet[4], skr_seqs** out);
/* The reader's half of skr_seqs_from_fasta as a step of its own: the file parsed into host memory once (same semantics,
 * same errors), then packed onto one GPU or, range by range, onto several — each GPU packs and uploads the sequences it
 * will count (skr_fasta_pack only reads the parsed file: the ranges may be packed from several host threads at once).
 * lengths: int64 [n]; headers as skr_seqs_headers.
 * A file with any byte >= 0x80 returns SKR_ERR_FASTA_TEXT before any other check: the reference's text-mode open
 * (fasta_reader.py:44) decodes first, so its strip / upper / len see characters, not bytes; the caller hands such a file
 * to a text-mode reader (the package: seekr_amd.fasta_reader.Reader) and packs the strings with skr_seqs_pack.      */
typedef struct skr_fasta skr_fasta;
int skr_fasta_open(const char* path, skr_fasta** out);
int skr_fasta_free(skr_fasta* fa);
int skr_fasta_info(const skr_fasta* fa, int64_t* n, int64_t* total_bases);
int skr_fasta_lengths(const skr_fasta* fa, int64_t* lengths);
int skr_fasta_headers(const skr_fasta* fa, char* buf, int64_t cap, int64_t* needed);
int skr_fasta_pack(skr_ctx* ctx, const skr_fasta* fa, int64_t first, int64_t count, const char alphabet[4],
                   skr_seqs** out);
/* Read + pack a FASTA file with the reference reader's semantics (fasta_reader.py:41-63):
 * lines stripped, '>' first char == header, other lines concatenated and upper-cased.
 * Headers are returned through skr_seqs_headers.                                            */
int skr_seqs_from_fasta(skr_ctx* ctx, const char* path, const char alphabet[4], skr_seqs** out);
int skr_seqs_free(skr_seqs* s);
int skr_seqs_info(const skr_seqs* s, int64_t* n, int64_t* total_bases, int64_t* max_len);
/* lengths[n] (int64) of the packed sequences */
int skr_seqs_lengths(const skr_seqs* s, int64_t* lengths);
/* FASTA headers ('\n'-joined, including the leading '>'); *needed receives the byte count   */
int skr_seqs_headers(const skr_seqs* s, char* buf, int64_t cap, int64_t* needed);

/* ---------------------------------------------------------------- counting (K2+K3) ------ */
/* Integer surface: out[i, j] = number of windows of sequence i equal to k-mer j
 * (kmer_counts.py:142-150 before scaling).  `out` is a SKR_U32 matrix [n, 4^k].            */
int skr_count_u32(skr_ctx* ctx, const skr_seqs* s, int k, skr_mat* out);
/* Per-kb matrix exactly as BasicCounter.get_counts fills it (kmer_counts.py:194-200):
 * out[i, j] = float32(sum of n[i,j] float64 additions of 1000/(L_i-k+1)).  `out` is
 * SKR_F32 or SKR_F64 [n, 4^k].  log2_pre != 0 additionally applies log2(x + 1)
 * (kmer_counts.py:201-202, 189-192) in the same pass (SKR_F32 only).
 * Returns SKR_ERR_ZERODIV if any sequence has length k-1.                                   */
int skr_count_per_kb(skr_ctx* ctx, const skr_seqs* s, int k, int log2_pre, skr_mat* out);

/* Any alphabet (kmer_counts.py:120-122 takes any string): `alen` letters give alen^k columns,
 * column = sum code(c_p) * alen^(k-1-p), code = position in `alphabet` — the LAST position for a
 * repeated letter, as the reference's dict {kmer: index} resolves it.  Sequences come as one ASCII
 * buffer + n+1 offsets (no case folding, like skr_seqs_pack); `out` is [n, alen^k] SKR_F32 /
 * SKR_F64 (per-kb values as skr_count_per_kb) or SKR_U32 (raw counts).  The 2-bit kernels behind
 * skr_count_per_kb are the fast path for 4 distinct letters; this one serves everything else.   */
int skr_count_generic(skr_ctx* ctx, const char* bases, const int64_t* offsets, int64_t n, const char* alphabet,
                      int alen, int k, int log2_pre, skr_mat* out);

/* The same from ASCII sequences RESIDENT on the device (`skr_aseqs`: one upload, any number of counting calls — what a
 * pipeline that re-counts, and bench.py's timed region, use).  Up to 16 384 columns (5 letters to k = 6, 20 to k = 3)
 * the histogram lives in the LDS and the row write is the only device traffic; wider rows count into a uint32 scratch
 * histogram in HBM.  skr_count_generic above = skr_aseqs_create + skr_count_generic_dev + skr_aseqs_free.             */
typedef struct skr_aseqs skr_aseqs;
int skr_aseqs_create(skr_ctx* ctx, const char* bases, const int64_t* offsets, int64_t n, skr_aseqs** out);
int skr_aseqs_free(skr_aseqs* a);
int skr_count_generic_dev(skr_ctx* ctx, const skr_aseqs* a, const char* alphabet, int alen, int k, int log2_pre,
                          skr_mat* out);

/* ---------------------------------------------------------------- normalisation (K4+K5) - */
/* Sequential float32 column sums in row order, continuing from `acc` (1 x cols, SKR_F32):
 *     acc[j] = fl32(acc[j] + t(x[i, j]))   for i = 0 .. rows-1
 * t(x) = x                                   (center == NULL, square == 0)
 * t(x) = fl32(x - center[j])                 (center given; f64 center: round once from f64)
 * t(x) = fl32(d*d), d = fl32(t_prev(x) - center2[j])   (square != 0; center2 may be NULL)
 * This is the order numpy's axis-0 reduce uses in the reference (kmer_counts.py:168,174;
 * SURVEY Appendix A.4) and is what makes mean/std bit-identical.  A multi-GPU run passes
 * `acc` from rank to rank.                                                                   */
int skr_colsum_seq(skr_ctx* ctx, const skr_mat* x, const skr_mat* center, const skr_mat* center2,
                   int square, skr_mat* acc);
/* skr_colsum_seq's first pass (no centre) that also returns the column minima of the raw matrix: colmin is
 * float32 [4, cols], the minimum over its four rows is the column's (NaN where the column holds one).  Float32
 * rounding is monotone, so min_i z[i,j] = z(min_i x[i,j]) for scale >= 0 and skr_min_nan over `colmin` gives the
 * Log2.post shift of kmer_counts.py:208 without a pass over the matrix.  Any column count since round 5 (a matrix
 * without rows: SKR_ERR_UNSUPPORTED — use skr_colsum_seq + skr_min_nan on x).                                       */
int skr_colsum_seq_colmin(skr_ctx* ctx, const skr_mat* x, skr_mat* acc, skr_mat* colmin);
/* The same chain across the GPUs of a node WITHOUT a transfer between two kernels (one process per GPU).  Every rank
 * creates a chain (a mailbox in uncached device memory for up to cols_cap columns), exports it as a 64-byte HIP IPC
 * handle, and connects with the handles of all ranks (exchanged by the caller, e.g. through skr_comm_allgather_rows of a
 * [nranks, 16] float32 matrix).  skr_colsum_seq_chain then runs one pass on this rank: its kernel waits — inside the
 * kernel, its loads already in flight — for rank - 1's running sums, continues them over this rank's rows in row order
 * and stores them into rank + 1's mailbox (a peer store over xGMI); the last rank stores the finished sums into every
 * rank's result box, from where they are copied into `acc`.  On return (stream order) acc holds the sums over all
 * ranks' rows: bit-identical to one GPU walking all rows.  colmin as in skr_colsum_seq_colmin, or NULL.  All ranks make
 * the same sequence of calls.  skr_chain_check reports (and clears) whether a wait gave up because a peer never
 * delivered (about half a minute).  skr_chain_connect_local connects chains that live in ONE process (emulation, tests). */
typedef struct skr_chain skr_chain;
int skr_chain_create(skr_ctx* ctx, int64_t cols_cap, skr_chain** out);
int skr_chain_export(skr_chain* chain, char handle[64]);
int skr_chain_connect(skr_chain* chain, int nranks, int rank, const char* handles /* [nranks][64] */);
int skr_chain_connect_local(skr_chain* chain, int nranks, int rank, skr_chain* const* all);
int skr_colsum_seq_chain(skr_chain* chain, const skr_mat* x, const skr_mat* center, const skr_mat* center2, int square,
                         skr_mat* acc, skr_mat* colmin, int defer_result /* 0; 1: skr_chain_result comes later (emulation on one stream) */);
int skr_chain_result(skr_chain* chain, skr_mat* acc);
int skr_chain_check(skr_chain* chain, int* timed_out);
int skr_chain_free(skr_chain* chain);
/* v[j] = fl32(v[j] / fl32(n));  if take_sqrt: v[j] = sqrt_rn(that)  (np.mean / np.std tail) */
int skr_vec_finish(skr_ctx* ctx, skr_mat* v, int64_t n, int take_sqrt);
/* NaN-propagating minimum (np.min, kmer_counts.py:208) of z = (x - center) / scale over the
 * whole matrix, plus whether any z is NaN (kmer_counts.py:176).  center/scale may be NULL.  */
int skr_min_nan(skr_ctx* ctx, const skr_mat* x, const skr_mat* center, const skr_mat* scale,
                float* min_out, int* has_nan);
/* y = post(scale(center(pre(x)))) elementwise, y may alias x:
 *   pre    : log2(x + 1)                                  (kmer_counts.py:189-192)
 *   center : x - center[j]                                (:169)
 *   scale  : x / scale[j]   (IEEE division)               (:175)
 *   post   : log2(fl32(fl32(x + shift) + 1))              (:208-209, shift = |min|)
 * has_nan (optional) reports NaN after the scale step (the warning condition, :176).        */
int skr_apply(skr_ctx* ctx, const skr_mat* x, int pre, const skr_mat* center, const skr_mat* scale,
              int post, float shift, skr_mat* y, int* has_nan);
/* Whole single-GPU pipeline of get_counts after counting (kmer_counts.py:201-209), in place.
 * mean_mode / std_mode: 0 = skip, 1 = compute (written to mean_out / std_out, 1 x cols F32),
 * 2 = use the supplied vector (mean_vec / std_vec, 1 x cols, F32 or F64).                   */
int skr_normalize(skr_ctx* ctx, skr_mat* x, int log2_mode, int mean_mode, const skr_mat* mean_vec,
                  int std_mode, const skr_mat* std_vec, skr_mat* mean_out, skr_mat* std_out,
                  int* has_nan);

/* BasicCounter.center() / standardize() / log2_norm() on a hand-assigned HOST count matrix whose dtype is not float32
 * (kmer_counts.py:165-192 act on whatever `self.counts` holds — test_kmer_counts.py:44-90 assigns it by hand; SURVEY §8(b):
 * "those methods must work on an arbitrary host array").  x: C-contiguous [rows, cols] of element type np_type; the
 * arithmetic is numpy's for that type, step for step (seekr_amd/csrc/normalize_any.hip lists it): float64 in float64,
 * integers converted to float64 for the statistics, float16 with float32 sums for the mean and half-rounded steps for
 * the std.  Upload, one kernel, download: the drop-in surface, not the hot path (SKR_NP_F32 is refused: skr_colsum_seq /
 * skr_apply are the float32 path).                                                                                   */
enum {
    SKR_NP_F16 = 0, SKR_NP_F32 = 1, SKR_NP_F64 = 2, SKR_NP_I8 = 3, SKR_NP_I16 = 4, SKR_NP_I32 = 5, SKR_NP_I64 = 6,
    SKR_NP_U8 = 7, SKR_NP_U16 = 8, SKR_NP_U32 = 9, SKR_NP_U64 = 10, SKR_NP_BOOL = 11
};
/* what = 0: np.mean(x, axis=0) (:168), 1: np.std(x, axis=0) (:174).  out: [cols] float16 for a float16 matrix, float64
 * for every other type (numpy's result types).                                                                       */
int skr_host_colstat(skr_ctx* ctx, const void* x, int64_t rows, int64_t cols, int np_type, int what, void* out);
/* The same for a COLUMN-MAJOR matrix (cell (i, j) at j * rows + i: a Fortran-ordered array such as `DataFrame.values`, or a
 * single column): there numpy reduces column by column in the PAIRWISE order of its float loops — for 50 000 rows up to
 * 1e-5 relative away from the row-after-row order in float32.  SKR_NP_F16 / SKR_NP_F32 / SKR_NP_F64 (float16: numpy's
 * half loops, float32 accumulators within a buffer piece; integer matrices are passed as the float64 values numpy casts
 * them to); out: [cols] of the same type.                                                                                */
int skr_host_colstat_colmajor(skr_ctx* ctx, const void* x, int64_t rows, int64_t cols, int np_type, int what, void* out);
/* In place on x.  op 0: x -= vec[col] (:169), op 1: x /= vec[col] (:175) — float matrices; vec is float64 (vec_is_f64) or
 * float32 [cols], the type numpy's promotion evaluates the operation in (a float64 matrix: always float64; a float16
 * matrix: float32 for float16 / float32 / 8- and 16-bit integer vectors, float64 otherwise), the result rounded once to
 * the matrix's type; has_nan = a NaN was stored (:176).  op 2: integer x -= vec[col], vec int64 [cols], two's-complement
 * wrap like numpy's cast.  op 3: x += 1 in the matrix's type (integers wrap), then y = log2(x) (:191-192); y_np_type must
 * be numpy's result type: F64 for F64 and 32/64-bit integers, F16 for F16 and 8-bit integers, F32 for 16-bit integers.
 * The cases numpy itself refuses (float statistics into an integer matrix, `+= 1` on bool) are the caller's to raise.  */
int skr_host_apply(skr_ctx* ctx, void* x, int64_t rows, int64_t cols, int np_type, int op, const void* vec, int vec_is_f64,
                   void* y, int y_np_type, int* has_nan);

/* ---------------------------------------------------------------- Pearson (K6+K7) ------- */
/* z = row-standardised x (pearson.py:35-38): per row, subtract the mean, divide by the
 * population std of the centred row.  F32 -> F32 or F64 -> F64, same shape.                 */
int skr_row_standardize(skr_ctx* ctx, const skr_mat* x, skr_mat* z);
/* r[row0 + i, col0 + j] = <a_i, b_j> / K   (pearson.py:41) for all rows of a and b.
 * a: [M, K], b: [N, K], r: at least [row0+M, col0+N]; dtype F32 (precision FP32 or BF16X3)
 * or F64 (precision F64).  symmetric != 0 promises a and b hold the same rows and
 * row0 == col0, letting the kernel compute one triangle and mirror it.                      */
int skr_pearson_gemm(skr_ctx* ctx, const skr_mat* a, const skr_mat* b, int precision, int symmetric,
                     skr_mat* r, int64_t row0, int64_t col0);
/* Prepared operands.  A skr_operand holds rows in the layout the chosen contraction consumes
 * (split-interleaved 16-bit halves for the split precisions — fp16 halves from 64 columns,
 * bf16 halves from 1024, both up to 16 384 — zero-padded float32 otherwise), so that standardised rows are produced once, exchanged between GPUs as
 * they are, and multiplied any number of times.                                              */
typedef struct skr_operand skr_operand;
int skr_operand_create(skr_ctx* ctx, int64_t rows, int64_t cols, int precision, skr_operand** out);
int skr_operand_free(skr_operand* op);
/* non-owning view of rows [row0, row0+nrows) */
int skr_operand_view(const skr_operand* parent, int64_t row0, int64_t nrows, skr_operand** out);
/* the operand's storage as a float32-typed matrix view [rows, 32*ceil(cols/32)] (for
 * skr_comm_sendrecv); free the view with skr_mat_free                                        */
int skr_operand_as_mat(skr_operand* op, skr_mat** view);
/* One fused pass over the float32 matrix x [rows, cols]:
 *   y  = post(scale(center(x)))   — the elementwise tail of the normalisation, as skr_apply
 *                                   (kmer_counts.py:169,175,208-209); written to `y` if non-NULL
 *                                   (y may alias x); skipped when center, scale are NULL, post 0
 *   z  = row-standardised y (pearson.py:35-38) if row_standardize, else y
 *   op = z in the operand layout.
 * has_nan (optional) as in skr_apply.                                                        */
int skr_operand_fill(skr_ctx* ctx, const skr_mat* x, const skr_mat* center, const skr_mat* scale, int post,
                     float shift, skr_mat* y, int row_standardize, skr_operand* op, int* has_nan);
/* Storage kind an operand ended up with: 0 = zero-padded float32 (fp32 kernel), 1 = bf16 halves,
 * 2 = fp16 halves.  skr_operand_fill falls back from a split kind to 0 when a row is dominated by
 * so few columns that one float32 accumulator per cell would drop the others (e.g. the raw counts
 * of a homopolymer: one non-zero k-mer): the MFMA adds each product aligned to the accumulator, so
 * products below ~2^-24 of a huge one vanish; the fp32 kernel accumulates in blocks.  Both operands
 * of a contraction must have the same kind; skr_operand_adopt_layout re-tags a buffer of the same
 * width (a receive buffer) with the kind and scale of `like`.                                    */
int skr_operand_kind(const skr_operand* op, int* kind);
int skr_operand_adopt_layout(skr_operand* op, const skr_operand* like);
/* get (set = 0) or set (set != 0, from *value) the "rows are mostly one repeated value, or lie on two tight levels"
 * flag that makes the split contraction restart its accumulators every 1 024 columns (near-copies of such rows feed
 * the MFMA's truncating accumulate sums that are all cut the same way: DESIGN.md §2); skr_operand_fill computes it
 * from the rows it sees, a multi-GPU caller all-reduces it so that every shard of a set carries the same value.   */
int skr_operand_coherent(skr_operand* op, int set, int* value);

/* SKR_PREC_F16F8 only.  The fill of an operand in the H / X layout keeps three numbers with it — the largest |row mean| of
 * hi - 128 h8, of lo and of lo - l8 / 16 over its rows — because the cross term of two rows is off by the products of
 * those means when the fp8 roundings do not average out over a row ("Two product-units").  Within one operand
 * skr_operand_fill applies the rule itself and routes the rows back to SKR_PREC_F16X3; two operands that were filled
 * separately are checked pairwise by skr_pearson_gemm_op / skr_pearson_gemm_edges (SKR_ERR_UNSUPPORTED when they do not go
 * together; skr_pearson and skr_pearson_gemm refill both instead).  Shards of ONE matrix on several GPUs: all-reduce (max)
 * the three values, apply the rule to the result, set it on every shard; receive buffers take it over in
 * skr_operand_adopt_layout.  set = 0: read into v[3]; set = 1: store v[3].  Zeros for every other layout.
 * (New in round 4; the reference has no counterpart — numpy's float32 inner product, seekr/pearson.py:41.) */
int skr_operand_x8_stats(skr_operand* op, int set, float* v);
/* The rule on those maxima as the library applies it: *bound = the error of a cell of r that the row means of a's and b's
 * rounding residues allow (b == a: the rows of one matrix against each other), *ok = whether skr_operand_fill's limit
 * admits it.  0 / ok for operands in any other layout.  A multi-GPU caller sets the all-reduced maxima and asks here. */
int skr_operand_x8_pair_bound(const skr_operand* a, const skr_operand* b, double* bound, int* ok);
/* r[row0 + i, col0 + j] = <a_i, b_j> / K on prepared operands (same meaning as skr_pearson_gemm).
 * symmetric = 2: a plain block whose cells carry the bits of the MIRROR of the swapped call — r[i, j] = what
 * skr_pearson_gemm_op(b, a) leaves at [j, i].  The split contractions add a cell's three products in an order that
 * names A's halves first, so a_i . b_j and b_j . a_i differ in the last bits; a self-comparison keeps, for both cells of
 * a pair, the value computed with the row of the smaller index as A.  Whoever computes a block BELOW the diagonal of a
 * self-comparison from its own rows (a row stripe, a GPU's row block) asks for it with 2 and gets those bits.       */
int skr_pearson_gemm_op(skr_ctx* ctx, const skr_operand* a, const skr_operand* b, int symmetric, skr_mat* r,
                        int64_t row0, int64_t col0);
/* A row stripe of a self-comparison (np.inner(z, z) / K of pearson.py:41 when r is produced stripe by stripe — larger
 * than the HBM — or row block by row block on several GPUs): r[row0 + i, j] for the rows i of `a` and all rows j of
 * `full`, bit for bit the rows a_row0 .. a_row0 + a.rows of skr_pearson_gemm_op(full, full, 1, ...).  `a` holds those
 * rows: a view of `full`, or the rank's own shard, of which `full` holds the all-gathered copy.  r: float32, at least
 * [row0 + a.rows, full.rows].                                                                                       */
int skr_pearson_gemm_op_rows(skr_ctx* ctx, const skr_operand* a, const skr_operand* full, int64_t a_row0, skr_mat* r,
                             int64_t row0);
/* The float64 contraction of skr_pearson as a step of its own: r[row0 + i, col0 + j] = <a_i, b_j> / K for float64 rows
 * that are standardised already (skr_row_standardize) and may be zero-padded to a->cols >= K columns.  symmetric != 0:
 * a and b are the same rows (one triangle computed and mirrored).  Float64 products round once, so stripes of a
 * self-comparison need no swapped form: any tiling gives the same bits.                                             */
int skr_pearson_gemm_f64(skr_ctx* ctx, const skr_mat* a, const skr_mat* b, int64_t K, int symmetric, skr_mat* r,
                         int64_t row0, int64_t col0);
/* As skr_pearson_gemm_op without the symmetric shortcut, and additionally the transposed block
 * rt[trow0 + j, tcol0 + i] = r[row0 + i, col0 + j].  r(b, a) = r(a, b)^T (np.inner is
 * symmetric in its arguments, pearson.py:41), so a rank that multiplied shard a by shard b
 * produces both blocks from one contraction.  The two blocks must not overlap.              */
int skr_pearson_gemm_op_mirror(skr_ctx* ctx, const skr_operand* a, const skr_operand* b, skr_mat* r, int64_t row0,
                               int64_t col0, skr_mat* rt, int64_t trow0, int64_t tcol0);
/* pearson(counts1, counts2, row_standardize) end to end on device matrices                 */
int skr_pearson(skr_ctx* ctx, const skr_mat* counts1, const skr_mat* counts2, int row_standardize,
                int precision, skr_mat* r);

/* ---------------------------------------------------------------- consumers of r -------- */
/* In place: r[r < cutoff] = 0 (NaN stays), then the diagonal r[i, i + diag_col0] = 0
 * (kmer_leiden.py:94-96; diag_col0 places the diagonal inside a row block).                 */
int skr_threshold_zero_diag(skr_ctx* ctx, skr_mat* r, float cutoff, int64_t diag_col0);
/* out = r[np.triu_indices(n, k)] for a square float32 r (find_dist.py:163 uses k = 1);
 * out holds (n-k)(n-k+1)/2 values.                                                           */
int skr_triu_flatten(skr_ctx* ctx, const skr_mat* r, int64_t k, skr_mat* out);
/* out_host[i] = src.flat[idx_host[i]] — the device side of np.random.choice(values, size,
 * replace=False) (find_dist.py:169): the host draws the indices with numpy's generator.     */
int skr_gather_f32(skr_ctx* ctx, const skr_mat* src, const int64_t* idx_host, int64_t n, float* out_host);
/* p[i,j] = float32(count(bg > r[i,j]) / total_len) (find_pval.py:158-164); sorted_bg: the
 * background values ascending with NaNs removed, total_len: len(fitres) including NaNs.    */
int skr_empirical_pvalues(skr_ctx* ctx, const skr_mat* r, const skr_mat* sorted_bg, int64_t total_len, skr_mat* p);
/* p[i,j] = float32(1 - dist(*params).cdf(r[i,j])), find_pval.py:118-133 with the scipy.stats distribution that
 * find_dist fitted: dist_name one of cauchy, chi2, expon, exponpow, gamma, lognorm, norm, pareto, rayleigh,
 * uniform (find_dist.py:96-98); params = shape parameter (where the distribution has one), loc, scale — the
 * tuple find_dist returns.  Evaluated in float64 like scipy; other names: SKR_ERR_UNSUPPORTED.                */
int skr_parametric_pvalues(skr_ctx* ctx, const skr_mat* r, const char* dist_name, const double* params, int n_params,
                           skr_mat* p);

/* The non-zero cells that kmer_leiden.py:94-96 leaves in a block of r, as an edge list and
 * without writing the zeros: cells of r[0:nrows, col_begin:col_end] with !(v < cutoff) (NaN
 * stays, like numpy's mask), v != 0 and global row != global column, where global row =
 * row_global0 + i and global column = col_global0 + j (j = r's own column index);
 * upper_only != 0 keeps global column > global row only.  *count receives the number of edges.
 * With out_rows / out_cols (SKR_U32) and out_vals (SKR_F32) non-NULL (each at least *count
 * cells) the triplets are written in row-major order — the order of np.nonzero on the
 * thresholded matrix; with all three NULL only the count is made.  Blocks of r produced one
 * row stripe at a time never need the N x N matrix.                      */
int skr_edges(skr_ctx* ctx, const skr_mat* r, int64_t nrows, int64_t col_begin, int64_t col_end,
              int64_t row_global0, int64_t col_global0, float cutoff, int upper_only, skr_mat* out_rows,
              skr_mat* out_cols, skr_mat* out_vals, int64_t* count);

/* The edge list of the block r = a b^T / K WITHOUT writing the block: the contraction's epilogue applies
 * kmer_leiden.py:94-96 (kept iff !(v < cutoff), v != 0, off the global diagonal; upper_only: column > row) and
 * appends the survivors; they are then sorted by (row, column) — np.nonzero order — into out_rows / out_cols /
 * out_vals (U32, U32, F32; global indices = row_global0 + i, col_global0 + j).  Same values, bit for bit, as
 * skr_pearson_gemm_op followed by skr_edges.  *count = cells found; if it exceeds the outputs' capacity nothing is
 * written to them and the caller calls again with larger outputs.  `scratch` (float32, at least [a.rows, b.rows]) is
 * needed only for rows of more than 4 096 columns (1 024 for operands flagged by skr_operand_coherent): the earlier
 * k chunks leave their partial sums there.  Split-precision operands only (SKR_ERR_UNSUPPORTED otherwise).          */
int skr_pearson_gemm_edges(skr_ctx* ctx, const skr_operand* a, const skr_operand* b, skr_mat* scratch, int64_t row_global0,
                           int64_t col_global0, float cutoff, int upper_only, skr_mat* out_rows, skr_mat* out_cols,
                           skr_mat* out_vals, int64_t* count);

/* *needs = 1 when skr_pearson_gemm_edges(a, b, ...) must be given a scratch block, i.e. when the rows span more than
 * one accumulator restart of the contraction — the library's own rule (4 096 columns; 1 024 for operands flagged by
 * skr_operand_coherent; the A/B knob SEEKR_GEMM_CHUNK_TILES as the ctx read it), so that a caller never re-derives it.   */
int skr_pearson_gemm_edges_needs_scratch(skr_ctx* ctx, const skr_operand* a, const skr_operand* b, int* needs);
/* Per-row top-k of the block r[0:nrows, col_begin:col_end]: out_idx[i, t] / out_val[i, t] = global
 * column and value of the t-th largest cell of row i (descending, ties to the smaller column,
 * NaN last: np.argsort(-row, kind="stable")[:k]), the row's own diagonal cell (global column ==
 * global row, as in skr_edges) excluded.  Rows with fewer than k candidates are padded with
 * index 0xFFFFFFFF / NaN.  out_idx: SKR_U32, out_val: SKR_F32, each at least nrows * k cells.  */
int skr_topk_rows(skr_ctx* ctx, const skr_mat* r, int64_t nrows, int64_t col_begin, int64_t col_end,
                  int64_t row_global0, int64_t col_global0, int k, skr_mat* out_idx, skr_mat* out_val);

/* ---------------------------------------------------------------- one-call host forms --- */
/* The two reference calls of the hot path over caller-owned host buffers, for bindings that do
 * not want to manage device handles (both run the same kernels as the handle-based functions).
 *
 * BasicCounter(...).get_counts() (kmer_counts.py:194-209) on packed sequences:
 *   counts_out  float32 [n, 4^k], row-major
 *   mean_mode / std_mode   0 = skip (mean=False), 1 = compute (mean=True; the vector lands in
 *                          mean_out / std_out, float32 [4^k], when that pointer is not NULL),
 *                          2 = use mean_vec / std_vec (4^k entries of mean_dtype / std_dtype,
 *                          SKR_F32 or SKR_F64 — a float64 vector is applied in float64 and
 *                          rounded once, as numpy's in-place `counts -= mean` does)
 *   has_nan     optional; set when standardisation produced NaN (the warning of :176-187)
 * Errors as the reference raises them: one sequence with std_mode 1 -> SKR_ERR_INVALID (:124-130),
 * unknown log2_mode -> SKR_ERR_INVALID (:134-135), len(seq) == k-1 -> SKR_ERR_ZERODIV (:144).   */
int skr_host_get_counts(skr_ctx* ctx, const skr_seqs* s, int k, int log2_mode, int mean_mode,
                        const void* mean_vec, int mean_dtype, int std_mode, const void* std_vec,
                        int std_dtype, float* counts_out, float* mean_out, float* std_out, int* has_nan);
/* pearson(counts1, counts2, row_standardize) (pearson.py:32-41): a [m, K] and b [n, K] of `dtype`
 * (SKR_F32 or SKR_F64; pass the same pointer twice for a self-comparison) -> out [m, n] of the
 * same dtype.  float64 always runs the f64 MFMA; for float32 `precision` picks the arithmetic
 * (SKR_PREC_F16X3 is the host package's default and falls back to SKR_PREC_FP32 without row
 * standardisation).                                                                            */
int skr_host_pearson(skr_ctx* ctx, const void* a, int64_t m, const void* b, int64_t n, int64_t K, int dtype,
                     int row_standardize, int precision, void* out);

/* ---------------------------------------------------------------- writers --------------- */
/* The files the reference writes from the count matrix and from r, byte-identical to numpy's:
 *   skr_*_save_npy                   np.save(path, a)              kmer_counts.py:234, pearson.py:43
 *   skr_*_save_csv, fmt_mode 0       np.savetxt(path, a, delimiter=",", fmt="%1.6f")   kmer_counts.py:241
 *                   fmt_mode 1       np.savetxt(path, a, delimiter=",")  ("%.18e")     find_dist.py:292
 * (the caller appends ".npy" where np.save would).  one_dim != 0 writes shape (cols,) for a
 * 1-row matrix (the mean / std vectors of seekr_norm_vectors).  Device matrices are streamed
 * through pinned buffers; text is formatted by `threads` host threads (0 = pick).  The host
 * variants take a C-contiguous array in host memory and need no device.                      */
int skr_mat_save_npy(skr_ctx* ctx, const skr_mat* m, int one_dim, const char* path);
/* np.save of a matrix that comes into being one row stripe at a time, on one GPU or on several at once (pearson.py:43
 * when r is larger than the HBM, or each GPU holds a row block): skr_npy_create writes numpy's header for [rows, cols]
 * and gives the file its final size (*data_offset = where row 0 starts); skr_mat_write_rows_at then streams rows of a
 * device matrix to byte offset file_offset (pwrite; through pinned buffers on the copy stream, behind `mark` as
 * skr_mat_download_at) — stripes in any order, from any number of threads.                                           */
int skr_npy_create(const char* path, int dtype, int64_t rows, int64_t cols, int64_t* data_offset);
int skr_mat_write_rows_at(const skr_mat* m, int64_t row0, int64_t nrows, const char* path, int64_t file_offset,
                          int64_t mark);
int skr_mat_save_csv(skr_ctx* ctx, const skr_mat* m, int fmt_mode, int threads, const char* path);
int skr_host_save_npy(const void* data, int dtype, int64_t rows, int64_t cols, int one_dim, const char* path);
int skr_host_save_csv(const void* data, int dtype, int64_t rows, int64_t cols, int fmt_mode, int threads,
                      const char* path);
/* fmt_mode 2 of the two functions above writes numpy's str() of each value: the shortest digits
 * that read back as the same float32 / float64 (positional for 1e-4 <= |x| < 1e16), NaN as an
 * empty field — the cell format of DataFrame.to_csv.  The labelled variants write the file of
 *   DataFrame(a, index=row_labels, columns=col_labels).to_csv(path)      kmer_counts.py:236-240
 * (the reference CLI's default output): header line ",c0,c1,…", then label,values per row, labels
 * quoted as csv.QUOTE_MINIMAL does.  row_labels / col_labels: '\n'-joined, exactly rows / cols of them. */
int skr_mat_save_csv_labelled(skr_ctx* ctx, const skr_mat* m, const char* row_labels, const char* col_labels,
                              int threads, const char* path);
int skr_host_save_csv_labelled(const void* data, int dtype, int64_t rows, int64_t cols, const char* row_labels,
                               const char* col_labels, int threads, const char* path);

/* ---------------------------------------------------------------- CSV input ------------- */
/* pd.read_csv(path, index_col=0) (console_scripts.py:626-631) for the labelled count files
 * seekr_kmer_counts writes: float64 values [rows, cols] + row and column labels.  Only cells
 * that pandas' own converter turns into the correctly rounded double are parsed (<= 15
 * significant digits, <= 17 digit characters, decimal exponent within +-22, NA strings, inf);
 * any other content returns SKR_ERR_UNSUPPORTED so that the caller can hand the file to pandas
 * — a successful read is bit-identical to the reference's.  Labels come back '\n'-terminated:
 * call skr_csv_labels with buf NULL to learn the size.  which: 0 = rows, 1 = columns.        */
typedef struct skr_csv skr_csv;
int skr_csv_read(const char* path, int threads, skr_csv** out);
int skr_csv_shape(const skr_csv* csv, int64_t* rows, int64_t* cols);
int skr_csv_values(const skr_csv* csv, double* out);
int skr_csv_labels(const skr_csv* csv, int which, char* buf, int64_t cap, int64_t* needed);
int skr_csv_free(skr_csv* csv);

/* ---------------------------------------------------------------- multi-GPU (C1, C2) ---- */
/* One process per GPU.  Rank 0 creates an id and distributes the 128 bytes out of band.    */
int skr_comm_unique_id(char id[128]);
int skr_comm_init(skr_ctx* ctx, int nranks, int rank, const char id[128]);
int skr_comm_destroy(skr_ctx* ctx);
int skr_comm_barrier(skr_ctx* ctx);
/* send rows [srow0, srow0+snrows) of `src` to `dst_rank` and receive rows into `dst` from
 * `src_rank` as one grouped RCCL operation on the ctx's communication stream; either side
 * may be skipped with rank < 0.  Completion is tracked per call: the returned ticket can be
 * waited on (once) by the compute stream with skr_comm_wait.  ticket == NULL: fire and forget —
 * the exchange is ordered against later exchanges (same stream), no event is kept for it.    */
int skr_comm_sendrecv(skr_ctx* ctx, const skr_mat* src, int64_t srow0, int64_t snrows, int dst_rank,
                      skr_mat* dst, int64_t drow0, int64_t dnrows, int src_rank, int64_t* ticket);
/* all-gather of row shards of unequal size: rank g contributes `shard` (bounds[g+1]-bounds[g]
 * rows) and every rank ends with all rows of `full` in global order.  One grouped RCCL
 * operation: the sends to and receives from all P-1 peers are in flight together (one xGMI
 * link per peer).  bounds: nranks+1 row offsets, bounds[0] == 0.                              */
int skr_comm_allgather_rows(skr_ctx* ctx, const skr_mat* shard, skr_mat* full, const int64_t* bounds,
                            int64_t* ticket);
/* n exchanges (arguments as skr_comm_sendrecv, one entry per exchange) as ONE grouped RCCL
 * operation: transfers with different peers run on their own xGMI links at once; one ticket.   */
int skr_comm_exchange(skr_ctx* ctx, int n, const skr_mat* const* src, const int64_t* srow0, const int64_t* snrows,
                      const int* dst_rank, skr_mat* const* dst, const int64_t* drow0, const int64_t* dnrows,
                      const int* src_rank, int64_t* ticket);
int skr_comm_wait(skr_ctx* ctx, int64_t ticket);
/* all-reduce of a few host doubles (op: 0 = sum, 1 = max, 2 = min)                          */
int skr_comm_allreduce_f64(skr_ctx* ctx, double* values, int n, int op);


/* One host process, several GPUs (SEEKR_DEVICES with SEEKR_TRANSPORT=peer, or when RCCL cannot be set up): rows move
 * between two ctxs of the SAME process by a peer copy over xGMI on the receiving ctx's communication stream — no RCCL, no
 * CU taken from the contraction.  Ordering is by events either ctx's streams may wait for; the host threads hand each other
 * the handles in memory.  skr_event_record: an event at the current end of the ctx's compute (0) or communication (1)
 * stream; skr_event_wait: that stream of `ctx` waits for an event of ANY ctx of this process; skr_peer_copy_rows: enqueued
 * on dst's ctx's communication stream (direct peer access is switched on for the pair where the hardware offers it).
 * The receiver's sequence: skr_event_wait(dst ctx, 1, source's "rows are ready" event); skr_peer_copy_rows;
 * skr_event_record(dst ctx, 1) -> what the receiver's compute stream (and, before it overwrites the rows, the source) waits for. */
typedef struct skr_event skr_event;
int skr_event_record(skr_ctx* ctx, int on_comm_stream, skr_event** out);
int skr_event_wait(skr_ctx* ctx, int on_comm_stream, const skr_event* ev);
int skr_event_free(skr_event* ev);
int skr_peer_copy_rows(skr_mat* dst, int64_t drow0, const skr_mat* src, int64_t srow0, int64_t nrows);

/* ---------------------------------------------------------------- diagnostics ------------- */
/* NOT part of libseekr_hip.so.  A second library, libseekr_hip_diag.so (python -m seekr_amd.build --diag: the same
 * sources with -DSEEKR_DIAG), additionally holds a build of the split-fp16 contraction whose workgroups stamp
 * s_memtime / s_memrealtime around the k loop and the epilogue of every tile (in-kernel clock and phase anatomy;
 * MI355X_MICROARCH.md "DVFS give-back" item 6; tools/gemm_diag.py).  skr_gemm_diag_mode selects it for the ctx
 * (0 = production kernels, 1 = stamps, 2-4 = timing experiments whose r is meaningless); skr_gemm_diag_read copies
 * the records of the last such launch to the host: out[max_records][8] uint64.                                     */
#ifdef SEEKR_DIAG
int skr_gemm_diag_mode(skr_ctx* ctx, int mode);
int skr_gemm_diag_read(skr_ctx* ctx, unsigned long long* out, int64_t max_records, int64_t* n_records);
#endif

#ifdef __cplusplus
}
#endif
#endif /* SEEKR_HIP_H */

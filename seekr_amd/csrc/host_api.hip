// One-call forms of the hot path over caller-owned host buffers: what a binding from a language
// without object lifetimes (R's .C, Julia's ccall, a C program) would call.  They compose the
// handle-based entry points — the work is the same kernels; nothing here computes on the host.
#include <memory>

#include "common.hpp"

namespace {

struct MatDeleter {
    void operator()(skr_mat* m) const { (void)skr_mat_free(m); }
};
using MatPtr = std::unique_ptr<skr_mat, MatDeleter>;

int make_mat(skr_ctx* ctx, int64_t rows, int64_t cols, int dtype, MatPtr* out) {
    skr_mat* m = nullptr;
    SKR_TRY(skr_mat_create(ctx, rows, cols, dtype, &m));
    out->reset(m);
    return SKR_OK;
}

// user vector (float32 or float64, `cols` entries) -> 1 x cols device vector of the same dtype
int upload_vec(skr_ctx* ctx, const void* vec, int dtype, int64_t cols, const char* what, MatPtr* out) {
    SKR_REQUIRE(vec, "%s_mode is 2 (use the supplied vector) but %s_vec is NULL", what, what);
    SKR_REQUIRE(dtype == SKR_F32 || dtype == SKR_F64, "%s_vec must be float32 or float64", what);
    SKR_TRY(make_mat(ctx, 1, cols, dtype, out));
    return skr_mat_upload(out->get(), vec, 0, 1);
}

}  // namespace

extern "C" int skr_host_get_counts(skr_ctx* ctx, const skr_seqs* s, int k, int log2_mode, int mean_mode,
                                   const void* mean_vec, int mean_dtype, int std_mode, const void* std_vec,
                                   int std_dtype, float* counts_out, float* mean_out, float* std_out, int* has_nan) {
    SKR_REQUIRE(ctx && s && counts_out, "NULL argument");
    SKR_REQUIRE(log2_mode == SKR_LOG2_NONE || log2_mode == SKR_LOG2_PRE || log2_mode == SKR_LOG2_POST,
                "log2 must be one of ['Log2.pre', 'Log2.post', 'Log2.none']");  // kmer_counts.py:134-135
    SKR_REQUIRE(mean_mode >= 0 && mean_mode <= 2 && std_mode >= 0 && std_mode <= 2, "mean_mode / std_mode are 0, 1 or 2");
    int64_t n = 0;
    SKR_TRY(skr_seqs_info(s, &n, nullptr, nullptr));
    // kmer_counts.py:124-130
    SKR_REQUIRE(!(n == 1 && std_mode == 1), "You cannot standardize a single sequence. Please pass the path to an std. "
                                           "dev. array, or use raw counts by setting std=False.");
    SKR_REQUIRE(k >= 1 && k <= 12, "k = %d outside 1..12", k);
    const int64_t cols = (int64_t)1 << (2 * k);
    if (has_nan) *has_nan = 0;
    MatPtr x, mv, sv, mo, so;
    SKR_TRY(make_mat(ctx, n, cols, SKR_F32, &x));
    SKR_TRY(skr_count_per_kb(ctx, s, k, log2_mode == SKR_LOG2_PRE ? 1 : 0, x.get()));
    if (mean_mode == 2) SKR_TRY(upload_vec(ctx, mean_vec, mean_dtype, cols, "mean", &mv));
    if (std_mode == 2) SKR_TRY(upload_vec(ctx, std_vec, std_dtype, cols, "std", &sv));
    if (mean_mode == 1) SKR_TRY(make_mat(ctx, 1, cols, SKR_F32, &mo));
    if (std_mode == 1) SKR_TRY(make_mat(ctx, 1, cols, SKR_F32, &so));
    // Log2.pre went into the counting flush; the normaliser only has the post step left to do
    SKR_TRY(skr_normalize(ctx, x.get(), log2_mode == SKR_LOG2_POST ? SKR_LOG2_POST : SKR_LOG2_NONE, mean_mode, mv.get(),
                          std_mode, sv.get(), mo.get(), so.get(), has_nan));
    if (n > 0) SKR_TRY(skr_mat_download(x.get(), counts_out, 0, n));
    if (mo && mean_out) SKR_TRY(skr_mat_download(mo.get(), mean_out, 0, 1));
    if (so && std_out) SKR_TRY(skr_mat_download(so.get(), std_out, 0, 1));
    return SKR_OK;
}

extern "C" int skr_host_pearson(skr_ctx* ctx, const void* a, int64_t m, const void* b, int64_t n, int64_t K, int dtype,
                                int row_standardize, int precision, void* out) {
    SKR_REQUIRE(ctx && a && b && out, "NULL argument");
    SKR_REQUIRE(m >= 0 && n >= 0 && K > 0, "bad shape [%lld, %lld] x [%lld, %lld]", (long long)m, (long long)K, (long long)n,
                (long long)K);
    SKR_REQUIRE(dtype == SKR_F32 || dtype == SKR_F64, "counts must be float32 or float64");
    if (dtype == SKR_F64) precision = SKR_PREC_F64;
    SKR_REQUIRE(dtype == SKR_F64 || precision != SKR_PREC_F64, "SKR_PREC_F64 needs float64 inputs");
    // fp16 halves only hold row-standardised rows (pearson.py of the host package makes the same choice)
    if (precision == SKR_PREC_F16X3 && !row_standardize) precision = SKR_PREC_FP32;
    if (m == 0 || n == 0) return SKR_OK;
    MatPtr da, db, r;
    SKR_TRY(make_mat(ctx, m, K, dtype, &da));
    SKR_TRY(skr_mat_upload(da.get(), a, 0, m));
    const bool same = a == b && m == n;
    if (!same) {
        SKR_TRY(make_mat(ctx, n, K, dtype, &db));
        SKR_TRY(skr_mat_upload(db.get(), b, 0, n));
    }
    SKR_TRY(make_mat(ctx, m, n, dtype, &r));
    SKR_TRY(skr_pearson(ctx, da.get(), same ? da.get() : db.get(), row_standardize, precision, r.get()));
    return skr_mat_download(r.get(), out, 0, m);
}

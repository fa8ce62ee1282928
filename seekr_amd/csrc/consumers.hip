// Consumers of the Pearson matrix that the reference runs right after `pearson`:
//   * kmer_leiden.py:94-96   ld_sim[ld_sim < cutoff] = 0; np.fill_diagonal(ld_sim, 0)
//   * find_dist.py:163       sim_counts[np.triu_indices(N, k=1)]   (+ :169 random subsample)
//   * find_pval.py:158-164   p[i,j] = np.sum(fitres > sim[i,j]) / len(fitres)   (O(M N |fitres|) Python loop)
// All three are streaming passes over r in HBM (HBM-bound); keeping them on the device avoids
// moving an N x N matrix over PCIe just to reduce it.
#include <algorithm>

#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void threshold_zero_diag_kernel(float* __restrict__ r, int64_t rows, int64_t cols,
                                                                  float cutoff, int64_t diag_col0) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cols, col = i % cols;
        float v = r[i];
        if (v < cutoff) v = 0.f;  // NaN < cutoff is false: NaN stays, as with numpy's boolean mask
        if (col == row + diag_col0) v = 0.f;
        r[i] = v;
    }
}

// out[start(i) + (j - i - k)] = r[i, j] for j >= i + k, start(i) = number of selected cells in rows < i:
// exactly the order of np.triu_indices(n, k).  One workgroup per row segment, coalesced copies.
__global__ __launch_bounds__(256) void triu_flatten_kernel(const float* __restrict__ r, int64_t n, int64_t k,
                                                           float* __restrict__ out) {
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const int64_t first = i + k;
        if (first >= n) continue;
        // cells selected in rows 0..i-1: sum_{t<i} max(0, n - t - k)
        const int64_t full = std::min<int64_t>(i, std::max<int64_t>(0, n - k));
        const int64_t start = full * (n - k) - full * (full - 1) / 2;
        const float* src = r + (size_t)i * n + first;
        float* dst = out + start;
        for (int64_t c = threadIdx.x; c < n - first; c += blockDim.x) dst[c] = src[c];
    }
}

__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                     int64_t n, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = src[idx[i]];
}

// p = float32( #(bg > v) / total ), bg ascending without NaN; #(bg > v) = n_bg - upper_bound(v)
__global__ __launch_bounds__(256) void empirical_p_kernel(const float* __restrict__ r, int64_t total_cells,
                                                          const float* __restrict__ bg, int64_t n_bg, double total_len,
                                                          float* __restrict__ p) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_cells; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = r[i];
        int64_t lo = 0, hi = n_bg;  // first index with bg[idx] > v
        if (v != v) {
            lo = n_bg;  // nothing compares greater than NaN
        } else {
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (bg[mid] > v) hi = mid; else lo = mid + 1;
            }
        }
        p[i] = (float)((double)(n_bg - lo) / total_len);
    }
}

unsigned grid_for(const skr_ctx* ctx, int64_t items) {
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>((items + 255) / 256, (int64_t)ctx->num_cu * 8));
}

}  // namespace

extern "C" int skr_threshold_zero_diag(skr_ctx* ctx, skr_mat* r, float cutoff, int64_t diag_col0) {
    SKR_REQUIRE(ctx && r && r->ctx == ctx && r->dtype == SKR_F32, "need a float32 matrix of this ctx");
    SKR_TRY(skr_activate(ctx));
    if (r->rows * r->cols == 0) return SKR_OK;
    SkrProfScope prof(ctx, "threshold_zero_diag");
    hipLaunchKernelGGL(threshold_zero_diag_kernel, dim3(grid_for(ctx, r->rows * r->cols)), dim3(256), 0, ctx->stream,
                       (float*)r->data, r->rows, r->cols, cutoff, diag_col0);
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

extern "C" int skr_triu_flatten(skr_ctx* ctx, const skr_mat* r, int64_t k, skr_mat* out) {
    SKR_REQUIRE(ctx && r && out && r->ctx == ctx && out->ctx == ctx, "NULL argument or foreign ctx");
    SKR_REQUIRE(r->dtype == SKR_F32 && out->dtype == SKR_F32, "float32 only");
    SKR_REQUIRE(r->rows == r->cols, "triu needs a square matrix");
    SKR_REQUIRE(k >= 0, "k must be >= 0");
    const int64_t n = r->rows, m = std::max<int64_t>(0, n - k);
    SKR_REQUIRE(out->rows * out->cols == m * (m + 1) / 2, "out must hold %lld values", (long long)(m * (m + 1) / 2));
    SKR_TRY(skr_activate(ctx));
    if (m == 0) return SKR_OK;
    SkrProfScope prof(ctx, "triu_flatten");
    hipLaunchKernelGGL(triu_flatten_kernel, dim3((unsigned)std::min<int64_t>(n, (int64_t)ctx->num_cu * 16)), dim3(256), 0,
                       ctx->stream, (const float*)r->data, n, k, (float*)out->data);
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

extern "C" int skr_gather_f32(skr_ctx* ctx, const skr_mat* src, const int64_t* idx_host, int64_t n, float* out_host) {
    SKR_REQUIRE(ctx && src && src->ctx == ctx && src->dtype == SKR_F32, "need a float32 matrix of this ctx");
    SKR_REQUIRE(n >= 0 && (n == 0 || (idx_host && out_host)), "NULL argument");
    SKR_TRY(skr_activate(ctx));
    if (n == 0) return SKR_OK;
    const int64_t limit = src->rows * src->cols;
    for (int64_t i = 0; i < n; i++)
        SKR_REQUIRE(idx_host[i] >= 0 && idx_host[i] < limit, "index %lld out of range", (long long)idx_host[i]);
    void* ws = nullptr;
    SKR_TRY(skr_ctx_workspace(ctx, (size_t)n * 12 + 64, &ws));
    int64_t* d_idx = (int64_t*)ws;
    float* d_out = (float*)((char*)ws + (size_t)n * 8);
    SKR_HIP(hipMemcpyAsync(d_idx, idx_host, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(gather_kernel, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, (const float*)src->data, d_idx, n,
                       d_out);
    SKR_HIP(hipGetLastError());
    SKR_HIP(hipMemcpyAsync(out_host, d_out, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    SKR_HIP(hipStreamSynchronize(ctx->stream));
    return SKR_OK;
}

extern "C" int skr_empirical_pvalues(skr_ctx* ctx, const skr_mat* r, const skr_mat* sorted_bg, int64_t total_len,
                                     skr_mat* p) {
    SKR_REQUIRE(ctx && r && sorted_bg && p, "NULL argument");
    SKR_REQUIRE(r->ctx == ctx && sorted_bg->ctx == ctx && p->ctx == ctx, "foreign ctx");
    SKR_REQUIRE(r->dtype == SKR_F32 && sorted_bg->dtype == SKR_F32 && p->dtype == SKR_F32, "float32 only");
    SKR_REQUIRE(p->rows == r->rows && p->cols == r->cols, "p must have r's shape");
    SKR_REQUIRE(total_len > 0, "total_len must be positive");
    SKR_TRY(skr_activate(ctx));
    const int64_t cells = r->rows * r->cols;
    if (cells == 0) return SKR_OK;
    SkrProfScope prof(ctx, "empirical_pvalues");
    hipLaunchKernelGGL(empirical_p_kernel, dim3(grid_for(ctx, cells)), dim3(256), 0, ctx->stream, (const float*)r->data,
                       cells, (const float*)sorted_bg->data, sorted_bg->rows * sorted_bg->cols, (double)total_len,
                       (float*)p->data);
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

// Consumers of the Pearson matrix that the reference runs right after `pearson`:
//   * kmer_leiden.py:94-96   ld_sim[ld_sim < cutoff] = 0; np.fill_diagonal(ld_sim, 0)
//   * find_dist.py:163       sim_counts[np.triu_indices(N, k=1)]   (+ :169 random subsample)
//   * find_pval.py:158-164   p[i,j] = np.sum(fitres > sim[i,j]) / len(fitres)   (O(M N |fitres|) Python loop)
// All three are streaming passes over r in HBM (HBM-bound); keeping them on the device avoids
// moving an N x N matrix over PCIe just to reduce it.
#include <algorithm>
#include <cstring>

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

// p = float32( #(bg > v) / total ), bg ascending without NaN; #(bg > v) = n_bg - upper_bound(v).
// Two-level search: every `stride`-th background value sits in LDS (4096 entries), so 12 of the ~20
// steps of a 1 M-entry search are LDS reads and only the last ~8 walk one 256-entry bucket in the L2.
constexpr int kPvalTable = 4096;
__global__ __launch_bounds__(256) void empirical_p_kernel(const float* __restrict__ r, int64_t total_cells,
                                                          const float* __restrict__ bg, int64_t n_bg, double total_len,
                                                          float* __restrict__ p) {
    __shared__ float table[kPvalTable];
    const int64_t stride = (n_bg + kPvalTable - 1) / kPvalTable;          // table[t] = bg[min(n_bg-1, (t+1)*stride-1)]
    const int64_t n_table = (n_bg + stride - 1) / stride;                 // buckets actually used
    for (int64_t t = threadIdx.x; t < n_table; t += 256) table[t] = bg[std::min<int64_t>(n_bg - 1, (t + 1) * stride - 1)];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_cells; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = r[i];
        int64_t lo;  // first index with bg[idx] > v
        if (v != v) {
            lo = n_bg;  // nothing compares greater than NaN
        } else {
            int64_t tl = 0, th = n_table;  // first bucket whose last value is > v
            while (tl < th) {
                const int64_t mid = (tl + th) >> 1;
                if (table[mid] > v) th = mid; else tl = mid + 1;
            }
            lo = tl * stride;
            int64_t hi = std::min<int64_t>(n_bg, lo + stride);
            if (tl == n_table) lo = hi = n_bg;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (bg[mid] > v) hi = mid; else lo = mid + 1;
            }
        }
        p[i] = (float)((double)(n_bg - lo) / total_len);
    }
}

// ---- threshold -> edge list (kmer_leiden.py:94-96 without ever writing the zeros) -------------
// A cell survives `ld_sim[ld_sim < cutoff] = 0; np.fill_diagonal(ld_sim, 0)` as a non-zero iff
// !(v < cutoff) (NaN stays), v != 0 and it is off the diagonal.  upper: keep column > row only.
struct EdgeArgs {
    const float* r;
    int64_t ld, rows, col_begin, col_end, row_global0, col_global0;
    float cutoff;
    int upper;
};

__device__ __forceinline__ bool edge_kept(const EdgeArgs& a, int64_t grow, int64_t gcol, float v) {
    if (v < a.cutoff || v == 0.f) return false;
    return a.upper ? gcol > grow : gcol != grow;
}

__global__ __launch_bounds__(256) void edges_count_kernel(EdgeArgs a, unsigned long long* __restrict__ counts) {
    __shared__ unsigned red[4];
    for (int64_t i = blockIdx.x; i < a.rows; i += gridDim.x) {
        const float* row = a.r + (size_t)i * a.ld;
        const int64_t grow = a.row_global0 + i;
        unsigned n = 0;
        for (int64_t c = a.col_begin + threadIdx.x; c < a.col_end; c += 256) n += edge_kept(a, grow, a.col_global0 + c, row[c]);
        for (int off = 32; off; off >>= 1) n += __shfl_down(n, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n;
        __syncthreads();
        if (threadIdx.x == 0) counts[i] = (unsigned long long)red[0] + red[1] + red[2] + red[3];
        __syncthreads();
    }
}

// exclusive prefix sum of the per-row counts, in place; one workgroup (rows <= a few 10^5 per call)
__global__ __launch_bounds__(1024) void edges_scan_kernel(unsigned long long* __restrict__ counts, int64_t n,
                                                          unsigned long long* __restrict__ total) {
    __shared__ unsigned long long part[1024];
    const int64_t per = (n + 1023) / 1024, lo = std::min<int64_t>(n, threadIdx.x * per), hi = std::min<int64_t>(n, lo + per);
    unsigned long long s = 0;
    for (int64_t i = lo; i < hi; i++) s += counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int t = 0; t < 1024; t++) {
            const unsigned long long v = part[t];
            part[t] = run;
            run += v;
        }
        *total = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (int64_t i = lo; i < hi; i++) {
        const unsigned long long v = counts[i];
        counts[i] = run;
        run += v;
    }
}

// row-major order (== np.nonzero of the thresholded matrix): a workgroup walks its row 256 columns at
// a time; position inside the chunk from wave ballots + a 4-entry scan
__global__ __launch_bounds__(256) void edges_fill_kernel(EdgeArgs a, const unsigned long long* __restrict__ offsets,
                                                         uint32_t* __restrict__ out_row, uint32_t* __restrict__ out_col,
                                                         float* __restrict__ out_val) {
    __shared__ unsigned wave_n[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t i = blockIdx.x; i < a.rows; i += gridDim.x) {
        const float* row = a.r + (size_t)i * a.ld;
        const int64_t grow = a.row_global0 + i;
        unsigned long long base = offsets[i];
        for (int64_t c0 = a.col_begin; c0 < a.col_end; c0 += 256) {
            const int64_t c = c0 + threadIdx.x;
            const float v = c < a.col_end ? row[c] : 0.f;
            const bool keep = c < a.col_end && edge_kept(a, grow, a.col_global0 + c, v);
            const unsigned long long mask = __ballot(keep);
            if (lane == 0) wave_n[wave] = (unsigned)__popcll(mask);
            __syncthreads();
            unsigned before = 0, all = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                if (w < wave) before += wave_n[w];
                all += wave_n[w];
            }
            if (keep) {
                const unsigned long long pos = base + before + __popcll(mask & ((1ull << lane) - 1));
                out_row[pos] = (uint32_t)grow;
                out_col[pos] = (uint32_t)(a.col_global0 + c);
                out_val[pos] = v;
            }
            base += all;
            __syncthreads();
        }
    }
}

// ---- per-row top-k: the k largest cells of each row, descending, ties to the
// smaller column — np.argsort(-row, kind="stable")[:k] with NaN last and the row's own (global)
// diagonal cell excluded.  One workgroup per row; k selection passes over the row (it stays in the
// L2): pass t finds the largest key below the t-1-th winner.  Keys order (value desc, column asc).
__device__ __forceinline__ unsigned long long topk_key(float v, uint32_t col) {
    // monotone map of the float to uint32 (larger float -> larger key), NaN to the smallest key;
    // the low word prefers the smaller column on equal values
    uint32_t b = __float_as_uint(v == 0.f ? 0.f : v);  // -0 and +0 compare equal
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    if (v != v) b = 0u;
    return ((unsigned long long)b << 32) | (0xFFFFFFFFu - col);
}

__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ r, int64_t ld, int64_t rows,
                                                        int64_t col_begin, int64_t col_end, int64_t row_global0,
                                                        int64_t col_global0, int k, uint32_t* __restrict__ out_idx,
                                                        float* __restrict__ out_val) {
    __shared__ unsigned long long red[4];
    __shared__ unsigned long long s_best;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t i = blockIdx.x; i < rows; i += gridDim.x) {
        const float* row = r + (size_t)i * ld;
        const int64_t skip = row_global0 + i - col_global0;  // local column of the diagonal cell, if any
        unsigned long long bound = ~0ull;                     // keys must be strictly below this
        for (int t = 0; t < k; t++) {
            unsigned long long best = 0ull;
            for (int64_t c = col_begin + threadIdx.x; c < col_end; c += 256) {
                if (c == skip) continue;
                const unsigned long long key = topk_key(row[c], (uint32_t)(c - col_begin));
                if (key < bound && key > best) best = key;
            }
            for (int off = 32; off; off >>= 1) {
                const unsigned long long o = __shfl_down(best, off, 64);
                if (o > best) best = o;
            }
            if (lane == 0) red[wave] = best;
            __syncthreads();
            if (threadIdx.x == 0) {
                unsigned long long b = red[0];
                for (int w = 1; w < 4; w++) b = red[w] > b ? red[w] : b;
                s_best = b;
            }
            __syncthreads();
            best = s_best;
            if (threadIdx.x == 0) {
                if (best == 0ull) {  // fewer than k candidates in the row
                    out_idx[(size_t)i * k + t] = 0xFFFFFFFFu;
                    out_val[(size_t)i * k + t] = __uint_as_float(0x7FC00000u);
                } else {
                    const uint32_t c = 0xFFFFFFFFu - (uint32_t)best;
                    out_idx[(size_t)i * k + t] = (uint32_t)(col_global0 + col_begin + c);
                    out_val[(size_t)i * k + t] = row[col_begin + c];
                }
            }
            bound = best;
            if (best == 0ull) {  // nothing left: fill the remaining slots the same way
                for (int u = t + 1; u < k; u++)
                    if (threadIdx.x == 0) {
                        out_idx[(size_t)i * k + u] = 0xFFFFFFFFu;
                        out_val[(size_t)i * k + u] = __uint_as_float(0x7FC00000u);
                    }
                break;
            }
        }
        __syncthreads();
    }
}

unsigned grid_for(const skr_ctx* ctx, int64_t items) {
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>((items + 255) / 256, (int64_t)ctx->num_cu * 8));
}


// ---------------------------------------------------------------------------------------
// Parametric p-values (find_pval.py:118-133): p[i,j] = 1 - dist(*params).cdf(sim[i,j]) for the scipy.stats
// distribution find_dist fitted (find_dist.py:96-98, the `common10` list).  scipy evaluates the cdf of a float32
// argument in float64, `1 - cdf` in float64, and the assignment into np.zeros_like(sim) rounds to float32: the
// kernel does the same, one cell per thread.  params = shape parameters..., loc, scale as scipy orders them.
// ---------------------------------------------------------------------------------------
enum { DIST_CAUCHY = 0, DIST_CHI2, DIST_EXPON, DIST_EXPONPOW, DIST_GAMMA, DIST_LOGNORM, DIST_NORM, DIST_PARETO, DIST_RAYLEIGH,
       DIST_UNIFORM, DIST_COUNT };

// regularised lower incomplete gamma P(a, x), float64: series for x < a + 1, Lentz continued fraction for Q otherwise
// (Abramowitz & Stegun 6.5.29 / 6.5.31; both converge to ~1e-16 within a few hundred terms for the shapes a fit returns)
__device__ double gamma_p(double a, double x) {
    if (!(x > 0.0)) return 0.0;
    if (isinf(x)) return 1.0;
    const double lg = lgamma(a);
    if (x < a + 1.0) {
        double ap = a, del = 1.0 / a, sum = del;
        for (int n = 0; n < 2000; n++) {
            ap += 1.0;
            del *= x / ap;
            sum += del;
            if (fabs(del) < fabs(sum) * 1e-17) break;
        }
        return sum * exp(-x + a * log(x) - lg);
    }
    const double tiny = 1e-300;
    double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
    for (int i = 1; i < 2000; i++) {
        const double an = -(double)i * ((double)i - a);
        b += 2.0;
        d = an * d + b;
        if (fabs(d) < tiny) d = tiny;
        c = b + an / c;
        if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < 1e-16) break;
    }
    return 1.0 - exp(-x + a * log(x) - lg) * h;
}

struct DistParams {
    int dist;
    double shape, loc, scale;
};

__device__ __forceinline__ double dist_cdf(const DistParams& d, double x) {
    const double z = (x - d.loc) / d.scale;
    switch (d.dist) {
        case DIST_CAUCHY: return 0.5 + atan(z) / 3.14159265358979323846;
        case DIST_CHI2: return z > 0.0 ? gamma_p(0.5 * d.shape, 0.5 * z) : 0.0;
        case DIST_EXPON: return z > 0.0 ? -expm1(-z) : 0.0;
        case DIST_EXPONPOW: return z > 0.0 ? -expm1(-expm1(pow(z, d.shape))) : 0.0;
        case DIST_GAMMA: return z > 0.0 ? gamma_p(d.shape, z) : 0.0;
        case DIST_LOGNORM: return z > 0.0 ? 0.5 * erfc(-(log(z) / d.shape) * 0.70710678118654752440) : 0.0;
        case DIST_NORM: return 0.5 * erfc(-z * 0.70710678118654752440);
        case DIST_PARETO: return z >= 1.0 ? 1.0 - pow(z, -d.shape) : 0.0;
        case DIST_RAYLEIGH: return z > 0.0 ? -expm1(-0.5 * z * z) : 0.0;
        default: return z <= 0.0 ? 0.0 : (z >= 1.0 ? 1.0 : z);  // uniform
    }
}

__global__ __launch_bounds__(256) void parametric_p_kernel(const float* __restrict__ r, int64_t cells, DistParams d,
                                                           float* __restrict__ p) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = (double)r[i];
        p[i] = x != x ? NAN : (float)(1.0 - dist_cdf(d, x));  // cdf(nan) is nan in scipy
    }
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

extern "C" int skr_edges(skr_ctx* ctx, const skr_mat* r, int64_t nrows, int64_t col_begin, int64_t col_end,
                         int64_t row_global0, int64_t col_global0, float cutoff, int upper_only, skr_mat* out_rows,
                         skr_mat* out_cols, skr_mat* out_vals, int64_t* count) {
    SKR_REQUIRE(ctx && r && count && r->ctx == ctx && r->dtype == SKR_F32, "need a float32 matrix of this ctx and a count");
    SKR_REQUIRE(nrows >= 0 && nrows <= r->rows, "nrows out of range");
    SKR_REQUIRE(col_begin >= 0 && col_begin <= col_end && col_end <= r->cols, "column range out of the matrix");
    SKR_REQUIRE(row_global0 >= 0 && col_global0 >= 0 && row_global0 + nrows <= 0xffffffffLL &&
                    col_global0 + col_end <= 0xffffffffLL, "global indices must fit 32 bits");
    const bool fill = out_rows || out_cols || out_vals;
    if (fill) {
        SKR_REQUIRE(out_rows && out_cols && out_vals, "pass all three outputs or none");
        SKR_REQUIRE(out_rows->ctx == ctx && out_cols->ctx == ctx && out_vals->ctx == ctx, "foreign ctx");
        SKR_REQUIRE(out_rows->dtype == SKR_U32 && out_cols->dtype == SKR_U32 && out_vals->dtype == SKR_F32,
                    "outputs are U32, U32, F32");
    }
    SKR_TRY(skr_activate(ctx));
    *count = 0;
    if (nrows == 0 || col_begin == col_end) return SKR_OK;
    void* ws = nullptr;
    SKR_TRY(skr_ctx_workspace(ctx, (size_t)(nrows + 2) * 8, &ws));
    unsigned long long* counts = (unsigned long long*)ws;
    unsigned long long* total = counts + nrows;
    EdgeArgs a{(const float*)r->data, r->cols, nrows, col_begin, col_end, row_global0, col_global0, cutoff, upper_only != 0};
    const unsigned grid = (unsigned)std::min<int64_t>(nrows, (int64_t)ctx->num_cu * 16);
    {
        SkrProfScope prof(ctx, "edges_count");
        hipLaunchKernelGGL(edges_count_kernel, dim3(grid), dim3(256), 0, ctx->stream, a, counts);
        SKR_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(edges_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, counts, nrows, total);
    SKR_HIP(hipGetLastError());
    unsigned long long h_total = 0;
    SKR_HIP(hipMemcpyAsync(&h_total, total, 8, hipMemcpyDeviceToHost, ctx->stream));
    SKR_HIP(hipStreamSynchronize(ctx->stream));
    *count = (int64_t)h_total;
    if (!fill || h_total == 0) return SKR_OK;
    const int64_t cap = std::min(out_rows->rows * out_rows->cols, std::min(out_cols->rows * out_cols->cols,
                                                                           out_vals->rows * out_vals->cols));
    SKR_REQUIRE((int64_t)h_total <= cap, "%lld edges do not fit outputs of %lld cells", (long long)h_total, (long long)cap);
    SkrProfScope prof(ctx, "edges_fill");
    hipLaunchKernelGGL(edges_fill_kernel, dim3(grid), dim3(256), 0, ctx->stream, a, counts, (uint32_t*)out_rows->data,
                       (uint32_t*)out_cols->data, (float*)out_vals->data);
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

extern "C" int skr_topk_rows(skr_ctx* ctx, const skr_mat* r, int64_t nrows, int64_t col_begin, int64_t col_end,
                             int64_t row_global0, int64_t col_global0, int k, skr_mat* out_idx, skr_mat* out_val) {
    SKR_REQUIRE(ctx && r && out_idx && out_val, "NULL argument");
    SKR_REQUIRE(r->ctx == ctx && out_idx->ctx == ctx && out_val->ctx == ctx, "foreign ctx");
    SKR_REQUIRE(r->dtype == SKR_F32 && out_idx->dtype == SKR_U32 && out_val->dtype == SKR_F32, "r F32, out_idx U32, out_val F32");
    SKR_REQUIRE(nrows >= 0 && nrows <= r->rows, "nrows out of range");
    SKR_REQUIRE(col_begin >= 0 && col_begin <= col_end && col_end <= r->cols, "column range out of the matrix");
    SKR_REQUIRE(k >= 1 && k <= 4096, "k must be in 1..4096");
    SKR_REQUIRE(row_global0 >= 0 && col_global0 >= 0 && col_global0 + col_end <= 0xffffffffLL, "global indices must fit 32 bits");
    SKR_REQUIRE(out_idx->rows * out_idx->cols >= nrows * k && out_val->rows * out_val->cols >= nrows * k,
                "outputs must hold %lld cells", (long long)(nrows * k));
    SKR_TRY(skr_activate(ctx));
    if (nrows == 0) return SKR_OK;
    SkrProfScope prof(ctx, "topk_rows");
    hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)std::min<int64_t>(nrows, (int64_t)ctx->num_cu * 16)), dim3(256), 0,
                       ctx->stream, (const float*)r->data, r->cols, nrows, col_begin, col_end, row_global0, col_global0, k,
                       (uint32_t*)out_idx->data, (float*)out_val->data);
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

extern "C" int skr_parametric_pvalues(skr_ctx* ctx, const skr_mat* r, const char* dist_name, const double* params, int n_params,
                                      skr_mat* p) {
    SKR_REQUIRE(ctx && r && dist_name && p, "NULL argument");
    SKR_REQUIRE(r->ctx == ctx && p->ctx == ctx, "foreign ctx");
    SKR_REQUIRE(r->dtype == SKR_F32 && p->dtype == SKR_F32, "float32 only");
    SKR_REQUIRE(p->rows == r->rows && p->cols == r->cols, "p must have r's shape");
    static const struct { const char* name; int code; int shapes; } kDists[] = {
        {"cauchy", DIST_CAUCHY, 0}, {"chi2", DIST_CHI2, 1}, {"expon", DIST_EXPON, 0}, {"exponpow", DIST_EXPONPOW, 1},
        {"gamma", DIST_GAMMA, 1}, {"lognorm", DIST_LOGNORM, 1}, {"norm", DIST_NORM, 0}, {"pareto", DIST_PARETO, 1},
        {"rayleigh", DIST_RAYLEIGH, 0}, {"uniform", DIST_UNIFORM, 0}};
    DistParams d{-1, 0.0, 0.0, 1.0};
    for (const auto& k : kDists) {
        if (strcmp(k.name, dist_name) != 0) continue;
        // scipy: dist(*shapes, loc=0, scale=1); find_dist's fit returns all of them
        SKR_REQUIRE(n_params >= k.shapes && n_params <= k.shapes + 2 && (n_params == 0 || params),
                    "%s takes %d shape parameter(s) plus loc and scale, got %d values", dist_name, k.shapes, n_params);
        d.dist = k.code;
        if (k.shapes) d.shape = params[0];
        if (n_params > k.shapes) d.loc = params[k.shapes];
        if (n_params > k.shapes + 1) d.scale = params[k.shapes + 1];
    }
    if (d.dist < 0)
        return skr_set_error(SKR_ERR_UNSUPPORTED,
                             "distribution '%s' has no device cdf (available: cauchy, chi2, expon, exponpow, gamma, lognorm, "
                             "norm, pareto, rayleigh, uniform — find_dist's common10 list)", dist_name);
    SKR_REQUIRE(d.scale > 0.0, "scale must be positive");
    SKR_TRY(skr_activate(ctx));
    const int64_t cells = r->rows * r->cols;
    if (cells == 0) return SKR_OK;
    SkrProfScope prof(ctx, "parametric_pvalues");
    hipLaunchKernelGGL(parametric_p_kernel, dim3(grid_for(ctx, cells)), dim3(256), 0, ctx->stream, (const float*)r->data, cells, d,
                       (float*)p->data);
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

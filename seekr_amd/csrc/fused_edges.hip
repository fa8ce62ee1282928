// Threshold fused into the contraction (kmer_leiden.py:91-96): the edge list of a block of r
// without ever writing the block.  The EDGES mode of the split contraction (pearson_bf16.hip) appends the surviving
// cells — unordered — to a list in the ctx workspace; here the list is sorted by (row, column) with a device radix
// sort (hipCUB / rocPRIM), which is np.nonzero's order, and split into the three output arrays.  Against the two-step
// path (skr_pearson_gemm_op into a stripe buffer, then skr_edges) this saves the write of the stripe and the two
// reads of skr_edges' count and fill passes; the values are the same bits (same kernel arithmetic).
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "common.hpp"

namespace {
__global__ __launch_bounds__(256) void split_keys_kernel(const unsigned long long* __restrict__ keys, int64_t n,
                                                         uint32_t* __restrict__ rows, uint32_t* __restrict__ cols) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long k = keys[i];
        rows[i] = (uint32_t)(k >> 32);
        cols[i] = (uint32_t)k;
    }
}
}  // namespace

extern "C" int skr_pearson_gemm_edges(skr_ctx* ctx, const skr_operand* a, const skr_operand* b, skr_mat* scratch,
                                      int64_t row_global0, int64_t col_global0, float cutoff, int upper_only,
                                      skr_mat* out_rows, skr_mat* out_cols, skr_mat* out_vals, int64_t* count) {
    SKR_REQUIRE(ctx && a && b && out_rows && out_cols && out_vals && count, "NULL argument");
    SKR_REQUIRE(a->ctx == ctx && b->ctx == ctx && out_rows->ctx == ctx && out_cols->ctx == ctx && out_vals->ctx == ctx,
                "handle belongs to a different ctx");
    SKR_REQUIRE(a->cols == b->cols && a->kind == b->kind && a->precision == b->precision,
                "operands were prepared for different shapes or precisions");
    if (a->kind == 0)
        return skr_set_error(SKR_ERR_UNSUPPORTED, "float32-layout operands take the two-step path (skr_pearson_gemm_op + skr_edges)");
    SKR_REQUIRE(out_rows->dtype == SKR_U32 && out_cols->dtype == SKR_U32 && out_vals->dtype == SKR_F32, "outputs are U32, U32, F32");
    const int64_t M = a->rows, N = b->rows, K = a->cols;
    SKR_REQUIRE(row_global0 >= 0 && col_global0 >= 0 && row_global0 + M <= 0xffffffffLL && col_global0 + N <= 0xffffffffLL,
                "global indices must fit 32 bits");
    const bool coherent = a->coherent || b->coherent;
    const int64_t chunk = skr_gemm_chunk_tiles(ctx, coherent);  // the rule the contraction itself applies (knob included)
    const bool multi_chunk = a->kt > chunk;
    if (multi_chunk)
        SKR_REQUIRE(scratch && scratch->ctx == ctx && scratch->dtype == SKR_F32 && scratch->rows >= M && scratch->cols >= N,
                    "rows of more than %lld columns need a float32 scratch block of at least [%lld, %lld]", (long long)(chunk * 32), (long long)M,
                    (long long)N);
    SKR_TRY(skr_activate(ctx));
    *count = 0;
    if (M == 0 || N == 0) return SKR_OK;
    const int64_t cap = std::min(out_rows->rows * out_rows->cols, std::min(out_cols->rows * out_cols->cols,
                                                                           out_vals->rows * out_vals->cols));
    // workspace: count | keys_in | keys_out | vals_in | sort temp
    size_t temp_bytes = 0;
    if (cap > 0) {
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, (const unsigned long long*)nullptr,
                                                          (unsigned long long*)nullptr, (const float*)nullptr, (float*)nullptr,
                                                          (int)std::min<int64_t>(cap, 0x7fffffff), 0, 64, ctx->stream);
        if (e != hipSuccess) return skr_set_error(SKR_ERR_HIP, "radix sort sizing failed: %s", hipGetErrorString(e));
    }
    SKR_REQUIRE(cap <= 0x7fffffff, "at most 2^31 - 1 edges per call");
    const size_t capu = (size_t)cap;
    const size_t off_keys_in = 256, off_keys_out = off_keys_in + capu * 8, off_vals_in = off_keys_out + capu * 8;
    const size_t off_temp = (off_vals_in + capu * 4 + 255) & ~(size_t)255;
    void* ws = nullptr;
    SKR_TRY(skr_ctx_workspace(ctx, off_temp + temp_bytes + 256, &ws));
    char* base = (char*)ws;
    SkrEdgeSink sink;
    sink.count = (unsigned long long*)base;
    sink.keys = (unsigned long long*)(base + off_keys_in);
    sink.vals = (float*)(base + off_vals_in);
    sink.cap = (unsigned long long)cap;
    sink.row_global0 = row_global0;
    sink.col_global0 = col_global0;
    sink.cutoff = cutoff;
    sink.upper = upper_only != 0;
    SKR_HIP(hipMemsetAsync(sink.count, 0, 8, ctx->stream));
    float* C = multi_chunk ? (float*)scratch->data : nullptr;
    const int64_t ldc = multi_chunk ? scratch->cols : 0;
    SKR_TRY(skr_launch_gemm_edges(ctx, a->precision, a->data, b->data, C, M, N, a->kt, ldc, (float)K * a->scale * b->scale, coherent, sink));
    unsigned long long found = 0;
    SKR_HIP(hipMemcpyAsync(&found, sink.count, 8, hipMemcpyDeviceToHost, ctx->stream));
    SKR_HIP(hipStreamSynchronize(ctx->stream));
    *count = (int64_t)found;
    if (found == 0 || (int64_t)found > cap) return SKR_OK;  // too many for the outputs: the caller retries with larger ones
    SkrProfScope prof(ctx, "edges_sort");
    // rows and columns are below 2^32 each; sort only the bits that can be set
    int end_bit = 64;
    {
        const unsigned long long top = (unsigned long long)(row_global0 + M);
        end_bit = 32;
        while (end_bit < 64 && (top >> (end_bit - 32)) != 0) end_bit++;
    }
    unsigned long long* keys_out = (unsigned long long*)(base + off_keys_out);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(base + off_temp, temp_bytes, sink.keys, keys_out, sink.vals,
                                                      (float*)out_vals->data, (int)found, 0, end_bit, ctx->stream);
    if (e != hipSuccess) return skr_set_error(SKR_ERR_HIP, "radix sort failed: %s", hipGetErrorString(e));
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(((int64_t)found + 255) / 256, (int64_t)ctx->num_cu * 8));
    hipLaunchKernelGGL(split_keys_kernel, dim3(grid), dim3(256), 0, ctx->stream, keys_out, (int64_t)found, (uint32_t*)out_rows->data,
                       (uint32_t*)out_cols->data);
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

// Threshold fused into the contraction (kmer_leiden.py:91-96): the edge list of a block of r
// without ever writing the block.  The EDGES mode of the split contraction (pearson_bf16.hip) appends the surviving
// cells — unordered — to a list in the ctx workspace; here the list is put into (row, column) order, which is
// np.nonzero's order, by a hand-written least-significant-digit radix sort (below: no library sort), whose last pass
// writes the three output arrays directly.  Against the two-step path (skr_pearson_gemm_op into a stripe buffer, then
// skr_edges) this saves the write of the stripe and the two reads of skr_edges' count and fill passes; the values are
// the same bits (same kernel arithmetic).
#include <algorithm>

#include "common.hpp"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// LSD radix sort of (key = row << 32 | column, value) pairs, 8 bits per pass, only over the bits that can be set:
// ceil(bits(N) / 8) passes over the column field, then ceil(bits(M) / 8) over the row field (rows taken relative to
// the block's first row).  One pass = digit histogram per chunk -> exclusive scan of the [256][chunks] table ->
// stable scatter.  A chunk belongs to ONE wave, which walks it slice by slice (64 keys): the lanes holding the same
// digit find each other with eight ballots (one per digit bit), a lane's place is the chunk's running count of its
// digit plus the number of lower lanes in its group — stable by construction, no atomics, no cross-wave ranking.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kDigits = 256;

struct RadixArgs {
    const unsigned long long* keys_in;
    const float* vals_in;
    unsigned long long* keys_out;  // middle passes
    float* vals_out;               // every pass (the last one: the caller's value array)
    uint32_t* rows_out;            // last pass only
    uint32_t* cols_out;
    int64_t n;
    int64_t chunk;  // keys per wave, a multiple of 64
    int64_t n_chunks;
    uint32_t* table;  // [256][n_chunks] counts, then (scanned in place) start offsets
    uint32_t row0;    // subtracted from the row field before a row digit is taken
    int shift;        // within the field
    int row_field;    // 0: digit from the column field, 1: from the row field
};

__device__ __forceinline__ uint32_t digit_of(unsigned long long key, const RadixArgs& a) {
    const uint32_t f = a.row_field ? (uint32_t)(key >> 32) - a.row0 : (uint32_t)key;
    return (f >> a.shift) & (kDigits - 1);
}

// lanes of the wave that hold the same 8-bit digit as this one (all 64 lanes take part; `live` lanes only match live ones)
__device__ __forceinline__ unsigned long long same_digit(uint32_t d, bool live) {
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const unsigned long long ones = __ballot(live && ((d >> b) & 1u));
        peers &= ((d >> b) & 1u) ? ones : ~ones;
    }
    return peers;
}

__global__ __launch_bounds__(64) void radix_count_kernel(const RadixArgs a) {
    __shared__ uint32_t cnt[kDigits];
    const int lane = threadIdx.x;
    const int64_t c = blockIdx.x;
    for (int d = lane; d < kDigits; d += 64) cnt[d] = 0;
    __syncthreads();
    const int64_t begin = c * a.chunk, end = std::min(a.n, begin + a.chunk);
    for (int64_t i = begin + lane; i - lane < end; i += 64) {
        const bool live = i < end;
        const uint32_t d = live ? digit_of(a.keys_in[i], a) : 0u;
        const unsigned long long peers = same_digit(d, live);
        // the highest lane of each group books the whole group: one LDS add per distinct digit and slice
        if (live && (peers >> lane) == 1ull) cnt[d] += (uint32_t)__popcll(peers);
        __syncthreads();
    }
    for (int d = lane; d < kDigits; d += 64) a.table[(size_t)d * a.n_chunks + c] = cnt[d];
}

template <bool LAST>
__global__ __launch_bounds__(64) void radix_scatter_kernel(const RadixArgs a) {
    __shared__ uint32_t base[kDigits];
    const int lane = threadIdx.x;
    const int64_t c = blockIdx.x;
    for (int d = lane; d < kDigits; d += 64) base[d] = a.table[(size_t)d * a.n_chunks + c];
    __syncthreads();
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const int64_t begin = c * a.chunk, end = std::min(a.n, begin + a.chunk);
    for (int64_t i = begin + lane; i - lane < end; i += 64) {
        const bool live = i < end;
        unsigned long long key = 0;
        float val = 0.f;
        if (live) {
            key = a.keys_in[i];
            val = a.vals_in[i];
        }
        const uint32_t d = live ? digit_of(key, a) : 0u;
        const unsigned long long peers = same_digit(d, live);
        uint32_t pos = 0;
        if (live) pos = base[d] + (uint32_t)__popcll(peers & below);
        __syncthreads();  // every lane has read its digit's running start before any group leader moves it
        if (live) {
            if ((peers >> lane) == 1ull) base[d] += (uint32_t)__popcll(peers);
            if (LAST) {
                a.rows_out[pos] = (uint32_t)(key >> 32);
                a.cols_out[pos] = (uint32_t)key;
            } else {
                a.keys_out[pos] = key;
            }
            a.vals_out[pos] = val;
        }
        __syncthreads();
    }
}

// ---- exclusive prefix sum of a uint32 array in place (any length): block totals -> recursive scan of the totals ->
// local scan with the block's offset.  4 096 elements per 256-thread block.
constexpr int kScanBlock = 256, kScanPer = 16, kScanTile = kScanBlock * kScanPer;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds, uint32_t* total) {
    // wave scan, then the four wave totals
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kScanBlock / 64; w++) {
        const uint32_t t = lds[w];
        if (w < wave) before += t;
        all += t;
    }
    __syncthreads();
    *total = all;
    return before + incl - v;
}

__global__ __launch_bounds__(kScanBlock) void scan_totals_kernel(const uint32_t* __restrict__ x, int64_t n, uint32_t* __restrict__ totals) {
    __shared__ uint32_t lds[kScanBlock / 64];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanPer;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; j++)
        if (base + j < n) s += x[base + j];
    uint32_t total;
    (void)block_exclusive_scan(s, lds, &total);
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

// offsets == nullptr: a single block scans the whole (short) array
__global__ __launch_bounds__(kScanBlock) void scan_local_kernel(uint32_t* __restrict__ x, int64_t n, const uint32_t* __restrict__ offsets) {
    __shared__ uint32_t lds[kScanBlock / 64];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanPer;
    uint32_t v[kScanPer], s = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; j++) {
        v[j] = base + j < n ? x[base + j] : 0u;
        s += v[j];
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan(s, lds, &total) + (offsets ? offsets[blockIdx.x] : 0u);
#pragma unroll
    for (int j = 0; j < kScanPer; j++) {
        if (base + j < n) x[base + j] = run;
        run += v[j];
    }
}

// scratch: at least scan_scratch_words(n) uint32 behind the array's own storage
size_t scan_scratch_words(int64_t n) {
    size_t words = 0;
    while (n > kScanTile) {
        n = (n + kScanTile - 1) / kScanTile;
        words += (size_t)n;
    }
    return words;
}

int exclusive_scan(skr_ctx* ctx, uint32_t* x, int64_t n, uint32_t* scratch) {
    if (n <= kScanTile) {
        hipLaunchKernelGGL(scan_local_kernel, dim3(1), dim3(kScanBlock), 0, ctx->stream, x, n, (const uint32_t*)nullptr);
        SKR_HIP(hipGetLastError());
        return SKR_OK;
    }
    const int64_t blocks = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(scan_totals_kernel, dim3((unsigned)blocks), dim3(kScanBlock), 0, ctx->stream, x, n, scratch);
    SKR_HIP(hipGetLastError());
    SKR_TRY(exclusive_scan(ctx, scratch, blocks, scratch + blocks));
    hipLaunchKernelGGL(scan_local_kernel, dim3((unsigned)blocks), dim3(kScanBlock), 0, ctx->stream, x, n, (const uint32_t*)scratch);
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

int bit_length(uint64_t v) {
    int b = 0;
    while (v) {
        b++;
        v >>= 1;
    }
    return b;
}

struct SortPlan {
    int64_t chunk = 0, n_chunks = 0;
    size_t table_words = 0, scratch_words = 0;
};

// keys per wave: enough waves to fill the chip on short lists, at most 8 192 keys each on long ones (the [256][chunks]
// table stays at n / 8 bytes)
SortPlan plan_sort(const skr_ctx* ctx, int64_t n) {
    SortPlan p;
    const int64_t want_waves = (int64_t)ctx->num_cu * 16;
    int64_t chunk = (n + want_waves - 1) / want_waves;
    chunk = std::min<int64_t>(8192, std::max<int64_t>(512, chunk));
    p.chunk = (chunk + 63) / 64 * 64;
    p.n_chunks = std::max<int64_t>(1, (n + p.chunk - 1) / p.chunk);
    p.table_words = (size_t)kDigits * (size_t)p.n_chunks;
    p.scratch_words = scan_scratch_words((int64_t)p.table_words);
    return p;
}

}  // namespace

extern "C" int skr_pearson_gemm_edges_needs_scratch(skr_ctx* ctx, const skr_operand* a, const skr_operand* b, int* needs) {
    SKR_REQUIRE(ctx && a && b && needs, "NULL argument");
    *needs = a->kind != 0 && a->kt > skr_gemm_chunk_tiles(ctx, a->coherent || b->coherent);
    return SKR_OK;
}

extern "C" int skr_pearson_gemm_edges(skr_ctx* ctx, const skr_operand* a, const skr_operand* b, skr_mat* scratch,
                                      int64_t row_global0, int64_t col_global0, float cutoff, int upper_only,
                                      skr_mat* out_rows, skr_mat* out_cols, skr_mat* out_vals, int64_t* count) {
    SKR_REQUIRE(ctx && a && b && out_rows && out_cols && out_vals && count, "NULL argument");
    SKR_REQUIRE(a->ctx == ctx && b->ctx == ctx && out_rows->ctx == ctx && out_cols->ctx == ctx && out_vals->ctx == ctx,
                "handle belongs to a different ctx");
    SKR_REQUIRE(a->cols == b->cols && a->kind == b->kind && a->precision == b->precision,
                "operands were prepared for different shapes or precisions");
    SKR_TRY(skr_x8_pair_check(a, b));
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
                    "rows of more than %lld columns need a float32 scratch block of at least [%lld, %lld]", (long long)(chunk * 32),
                    (long long)M, (long long)N);
    SKR_TRY(skr_activate(ctx));
    *count = 0;
    if (M == 0 || N == 0) return SKR_OK;
    const int64_t cap = std::min(out_rows->rows * out_rows->cols, std::min(out_cols->rows * out_cols->cols,
                                                                           out_vals->rows * out_vals->cols));
    SKR_REQUIRE(cap <= 0x7fffffff, "at most 2^31 - 1 edges per call");
    const size_t capu = (size_t)cap;
    // workspace: count | keys A | keys B | vals A | vals B | digit table + scan scratch (sized for a full list)
    const SortPlan worst = plan_sort(ctx, std::max<int64_t>(cap, 1));
    const size_t off_keys_a = 256, off_keys_b = off_keys_a + capu * 8, off_vals_a = off_keys_b + capu * 8;
    const size_t off_vals_b = off_vals_a + capu * 4;
    const size_t off_table = (off_vals_b + capu * 4 + 255) & ~(size_t)255;
    // chunk count of any list of up to cap entries: short lists use short chunks (at most ~16 waves per CU of them),
    // long ones 8 192-key chunks
    const int64_t want_waves = (int64_t)ctx->num_cu * 16;
    const int64_t max_chunks = std::min<int64_t>(std::max<int64_t>(want_waves, cap / 8192 + 1) + 1, (cap + 511) / 512 + 1);
    const size_t table_words = std::max(worst.table_words, (size_t)kDigits * (size_t)max_chunks);
    const size_t scan_words = scan_scratch_words((int64_t)table_words) + 16;
    void* ws = nullptr;
    SKR_TRY(skr_ctx_workspace(ctx, off_table + (table_words + scan_words) * 4 + 256, &ws));
    char* base = (char*)ws;
    SkrEdgeSink sink;
    sink.count = (unsigned long long*)base;
    sink.keys = (unsigned long long*)(base + off_keys_a);
    sink.vals = (float*)(base + off_vals_a);
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
    const SortPlan plan = plan_sort(ctx, (int64_t)found);
    SKR_REQUIRE(plan.table_words <= table_words, "internal: digit table larger than planned");
    // columns are below col_global0 + N, rows (relative to the block) below M: sort only the bits that can be set
    const int col_bits = std::max(1, bit_length((uint64_t)(col_global0 + N - 1)));
    const int row_bits = bit_length((uint64_t)(M - 1));
    const int col_passes = (col_bits + 7) / 8, row_passes = (row_bits + 7) / 8;
    RadixArgs ra;
    ra.n = (int64_t)found;
    ra.chunk = plan.chunk;
    ra.n_chunks = plan.n_chunks;
    ra.table = (uint32_t*)(base + off_table);
    ra.row0 = (uint32_t)row_global0;
    ra.rows_out = (uint32_t*)out_rows->data;
    ra.cols_out = (uint32_t*)out_cols->data;
    uint32_t* scan_scratch = ra.table + table_words;
    unsigned long long* kbuf[2] = {(unsigned long long*)(base + off_keys_a), (unsigned long long*)(base + off_keys_b)};
    float* vbuf[2] = {(float*)(base + off_vals_a), (float*)(base + off_vals_b)};
    int cur = 0;
    for (int p = 0; p < col_passes + row_passes; p++) {
        const bool last = p + 1 == col_passes + row_passes;
        ra.row_field = p >= col_passes;
        ra.shift = 8 * (ra.row_field ? p - col_passes : p);
        ra.keys_in = kbuf[cur];
        ra.vals_in = vbuf[cur];
        ra.keys_out = kbuf[cur ^ 1];
        ra.vals_out = last ? (float*)out_vals->data : vbuf[cur ^ 1];
        hipLaunchKernelGGL(radix_count_kernel, dim3((unsigned)ra.n_chunks), dim3(64), 0, ctx->stream, ra);
        SKR_HIP(hipGetLastError());
        SKR_TRY(exclusive_scan(ctx, ra.table, (int64_t)plan.table_words, scan_scratch));
        if (last)
            hipLaunchKernelGGL(radix_scatter_kernel<true>, dim3((unsigned)ra.n_chunks), dim3(64), 0, ctx->stream, ra);
        else
            hipLaunchKernelGGL(radix_scatter_kernel<false>, dim3((unsigned)ra.n_chunks), dim3(64), 0, ctx->stream, ra);
        SKR_HIP(hipGetLastError());
        cur ^= 1;
    }
    return SKR_OK;
}

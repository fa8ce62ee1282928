// K2 + K3 — k-mer counting: sliding-window 4^k indexer, per-sequence LDS histogram, and the
// per-kb scaling fused into the histogram flush (kmer_counts.py:140-151, 194-202).
//
// One 256-thread workgroup owns one sequence at a time (persistent grid-stride loop over the
// sequences, 8 workgroups per CU at k=6).  Two threads share a packed word: thread t of a sweep
// takes 8 of the 16 windows that start inside word w = sweep*128 + t/2; it holds words w and w+1
// (consecutive lanes -> consecutive words, coalesced; prefetched one sequence ahead), and every
// window's column index is a bit-field of that 64-bit pair because the packer stores the first
// base in the top bits.  Counts go to a 4^k-bin uint32 histogram in LDS (16 KiB at k=6, 64 KiB
// at k=7) with ds_add_u32; runs of equal indices inside a thread (homopolymers) are merged
// before the atomic.  The flush converts bins to the reference's float32 per-kb values (a
// 16-entry per-sequence table covers almost every bin), zeroes them for the next sequence, and
// streams the dense row to HBM as 16-byte stores — the row write (4*4^k bytes per sequence) is
// the algorithmic traffic that bounds this kernel.  Two barriers per sequence.
//
// k >= 8 (4^k bins no longer fit the LDS): the same kernel with GLOBAL = true counts straight into
// the sequence's output row, used as a uint32 histogram in HBM (zeroed by a memset first, L2
// atomics), and the flush converts the row in place.
#include <cstring>
#include <type_traits>
#include <vector>

#include "common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kTabSize = 16;  // counts below this are looked up per sequence instead of recomputed

// float32( n sequential float64 additions of `inc` ) — what kmer_counts.py:144-150 stores.
// n*inc (one rounding) equals the sequential sum unless the product sits within the
// accumulated rounding slack of a float32 rounding boundary; only then replay the additions.
__device__ __forceinline__ float per_kb_value(uint32_t n, double inc) {
    if (n == 0) return 0.0f;
    const double p = (double)n * inc;
    const float f = (float)p;
    if (n <= 3) return f;  // 1*inc, inc+inc and fl(2inc+inc) are single roundings of n*inc
    const double slack = p * ((double)(n + 4) * 0x1.0p-53);
    if ((float)(p - slack) == f && (float)(p + slack) == f) return f;
    double s = 0.0;
    for (uint32_t i = 0; i < n; i++) s += inc;
    return (float)s;
}

__device__ __forceinline__ double per_kb_value_f64(uint32_t n, double inc) {
    double s = 0.0;
    for (uint32_t i = 0; i < n; i++) s += inc;  // exact replay; f64 output is a small-input path
    return s;
}

enum OutKind { OUT_F32 = 0, OUT_F32_LOG2 = 1, OUT_U32 = 2, OUT_F64 = 3 };

template <int OUT, bool GLOBAL>
__global__ __launch_bounds__(kThreads) void count_kmers_kernel(
    const uint32_t* __restrict__ packed, const int64_t* __restrict__ word_off, const int64_t* __restrict__ len,
    const uint32_t* __restrict__ mask, const int64_t* __restrict__ mask_off, int64_t n_seqs, int k, void* __restrict__ out,
    uint32_t* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_hist[];
    __shared__ float tab[kTabSize];  // 64 bytes: keeps the dynamic region 16-byte aligned
    uint32_t* hist = lds_hist;       // GLOBAL: re-pointed at the output row of each sequence
    const int tid = threadIdx.x;
    const uint32_t nbins = 1u << (2 * k);
    const uint32_t idx_mask = nbins - 1u;
    const uint32_t win_mask = (1u << k) - 1u;  // k consecutive validity bits

    // 8 windows that start in one half of a packed word (windows j0 .. j0+7 of the word): bins of a
    // 64-bit pair, runs of equal bins merged.  Two threads share a word, so all 256 threads of the
    // workgroup count a 2 kb sequence in one sweep.
    auto count_word = [&](uint32_t hi, uint32_t lo, uint32_t invalid, int j0, int lim) {
        const unsigned long long pair = ((unsigned long long)hi << 32) | lo;
        uint32_t run_idx = 0xFFFFFFFFu, run_len = 0;
#pragma unroll
        for (int jj = 0; jj < 8; jj++) {
            const int j = j0 + jj;
            if (j < lim && ((invalid >> j) & win_mask) == 0) {
                const uint32_t idx = (uint32_t)(pair >> (64 - 2 * j - 2 * k)) & idx_mask;
                if (idx == run_idx) {
                    run_len++;
                } else {
                    if (run_len) atomicAdd(&hist[run_idx], run_len);
                    run_idx = idx;
                    run_len = 1;
                }
            }
        }
        if (run_len) atomicAdd(&hist[run_idx], run_len);
    };

    // The words of the NEXT sequence (first sweep: 4096 bases) are fetched while the current one
    // is flushed, so the HBM latency of the tiny input never sits on a workgroup's critical path.
    // The loads are unconditional (indices clamped into the padded arrays): a branch around them
    // would make hipcc drain vmcnt(0) at the join.
    int64_t nL = 0, n_woff = 0, n_moff = -1;
    uint32_t n_hi = 0, n_lo = 0, n_m0 = 0, n_m1 = 0;
    auto prefetch = [&](int64_t s) {
        s = s < n_seqs ? s : n_seqs - 1;
        nL = len[s];
        n_woff = word_off[s];
        n_moff = mask_off[s];
        const int64_t W = nL - k + 1;
        const int64_t nww = W > 0 ? (W + 15) >> 4 : 0;
        const int64_t w = (tid >> 1) < nww ? (tid >> 1) : nww;
        n_hi = packed[n_woff + w];
        n_lo = packed[n_woff + w + 1];
        const int64_t mb = n_moff >= 0 ? n_moff + (w >> 1) : 0;
        n_m0 = mask[mb];
        n_m1 = mask[mb + 1];
    };

    if (!GLOBAL)
        for (uint32_t b = tid * 4; b < nbins; b += kThreads * 4) *reinterpret_cast<uint4*>(&hist[b]) = make_uint4(0, 0, 0, 0);
    prefetch(blockIdx.x);
    __syncthreads();

    for (int64_t seq = blockIdx.x; seq < n_seqs; seq += gridDim.x) {
        const int64_t L = nL, woff = n_woff, moff = n_moff;
        const uint32_t c_hi = n_hi, c_lo = n_lo, c_m0 = n_m0, c_m1 = n_m1;
        const int64_t W = L - k + 1;  // windows, counting every character (kmer_counts.py:143-144)
        if (W == 0 && tid == 0) atomicOr(&flags[2], 1u);  // ZeroDivisionError in the reference
        if (GLOBAL) hist = reinterpret_cast<uint32_t*>(out) + (size_t)seq * nbins;  // 4-byte cells in every OUT handled here
        const int64_t n_win_words = W > 0 ? (W + 15) >> 4 : 0;
        // the sequence's output value for every small count (almost all bins): 16 threads do the
        // float64 work once, the flush just looks it up (visible after the barrier below)
        const double inc = W > 0 ? 1000.0 / (double)W : 0.0;
        if (OUT != OUT_U32 && OUT != OUT_F64 && tid < kTabSize) {
            float t = per_kb_value((uint32_t)tid, inc);
            if (OUT == OUT_F32_LOG2) t = skr_log2_cr(t + 1.0f);  // kmer_counts.py:189-192: counts += 1; log2
            tab[tid] = t;
        }
        const int j0 = (tid & 1) * 8;  // which half of the word's 16 windows this thread takes
        if ((tid >> 1) < n_win_words) {  // first sweep (2048 bases) from the prefetched registers
            const int64_t w = tid >> 1;
            uint32_t invalid = 0;  // bit j: base 16w+j is not in the alphabet
            if (moff >= 0)
                invalid = (uint32_t)(((unsigned long long)c_m0 | ((unsigned long long)c_m1 << 32)) >> ((w & 1) * 16));
            const int64_t left = W - (w << 4);
            count_word(c_hi, c_lo, invalid, j0, (int)(left < 16 ? left : 16));
        }
        for (int64_t w = (tid + kThreads) >> 1; w < n_win_words; w += kThreads / 2) {  // longer sequences
            const uint32_t* words = packed + woff;
            uint32_t invalid = 0;
            if (moff >= 0) {
                const uint32_t* mwords = mask + moff;
                const int64_t mw = w >> 1;
                invalid = (uint32_t)(((unsigned long long)mwords[mw] | ((unsigned long long)mwords[mw + 1] << 32)) >>
                                     ((w & 1) * 16));
            }
            const int64_t left = W - (w << 4);
            count_word(words[w], words[w + 1], invalid, j0, (int)(left < 16 ? left : 16));
        }
        prefetch(seq + gridDim.x);  // in flight across the barrier and the flush
        __syncthreads();

        // ---- flush: bins -> per-kb values, dense row to HBM; the bins are zeroed on the way out
        auto value_of = [&](uint32_t n) -> float {
            if (n < (uint32_t)kTabSize) return tab[n];
            float t = per_kb_value(n, inc);
            if (OUT == OUT_F32_LOG2) t = skr_log2_cr(t + 1.0f);
            return t;
        };
        if (GLOBAL) {
            // the row holds this sequence's counts (written by L2 atomics: read them past the L1);
            // uint32 output is already in place, float output is converted where it stands
            if (OUT != OUT_U32) {
                __threadfence();
                for (uint32_t b = tid; b < nbins; b += kThreads) {
                    const uint32_t c = __hip_atomic_load(&hist[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    reinterpret_cast<float*>(hist)[b] = value_of(c);
                }
            }
        } else if (OUT == OUT_F64) {
            double* row = reinterpret_cast<double*>(out) + (size_t)seq * nbins;
            for (uint32_t b = tid; b < nbins; b += kThreads) {
                row[b] = per_kb_value_f64(hist[b], inc);
                hist[b] = 0;
            }
        } else {
            for (uint32_t b = tid * 4; b < nbins; b += kThreads * 4) {
                const uint4 c = *reinterpret_cast<const uint4*>(&hist[b]);
                *reinterpret_cast<uint4*>(&hist[b]) = make_uint4(0, 0, 0, 0);
                if (OUT == OUT_U32) {
                    *reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(out) + (size_t)seq * nbins + b) = c;
                } else {
                    float4 v;
                    v.x = value_of(c.x);
                    v.y = value_of(c.y);
                    v.z = value_of(c.z);
                    v.w = value_of(c.w);
                    // the row is written once and not read again by this kernel: keep it out of the L2
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(f4{v.x, v.y, v.z, v.w},
                                                reinterpret_cast<f4*>(reinterpret_cast<float*>(out) + (size_t)seq * nbins + b));
                }
            }
        }
        __syncthreads();  // zeroed bins visible before the next sequence is counted
    }
}

template <int OUT>
int launch_count(skr_ctx* ctx, const skr_seqs* s, int k, void* out, const char* name) {
    if (k > 7) {  // histogram in the output row itself
        if (s->n < 1) return SKR_OK;
        SKR_HIP(hipMemsetAsync(out, 0, (size_t)s->n * ((size_t)4 << (2 * k)), ctx->stream));
        const int64_t grid = std::min<int64_t>(s->n, (int64_t)ctx->num_cu * 8);
        SkrProfScope prof(ctx, name);
        hipLaunchKernelGGL((count_kmers_kernel<OUT, true>), dim3((unsigned)grid), dim3(kThreads), 0, ctx->stream, s->d_packed,
                           s->d_word_off, s->d_len, s->d_mask, s->d_mask_off, s->n, k, out, ctx->d_flags);
        SKR_HIP(hipGetLastError());
        return SKR_OK;
    }
    const size_t lds = (size_t)4 << (2 * k);
    SKR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(count_kmers_kernel<OUT, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // as many resident workgroups as LDS allows, capped by the wave limit (8 x 256 threads / CU)
    int per_cu = (int)std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1));
    if (per_cu < 1) per_cu = 1;
    int64_t grid = std::min<int64_t>(s->n, (int64_t)ctx->num_cu * per_cu);
    if (grid < 1) return SKR_OK;
    SkrProfScope prof(ctx, name);
    hipLaunchKernelGGL((count_kmers_kernel<OUT, false>), dim3((unsigned)grid), dim3(kThreads), lds, ctx->stream, s->d_packed,
                       s->d_word_off, s->d_len, s->d_mask, s->d_mask_off, s->n, k, out, ctx->d_flags);
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

int check_count_args(skr_ctx* ctx, const skr_seqs* s, int k, const skr_mat* out) {
    SKR_REQUIRE(ctx && s && out, "NULL argument");
    SKR_REQUIRE(s->ctx == ctx && out->ctx == ctx, "handles belong to a different ctx");
    SKR_REQUIRE(k >= 1, "k must be >= 1 (got %d)", k);
    if (k > 12) return skr_set_error(SKR_ERR_UNSUPPORTED, "k=%d: rows of 4^k columns are supported up to k = 12", k);
    SKR_REQUIRE(out->rows == s->n && out->cols == ((int64_t)1 << (2 * k)),
                "output must be [%lld, %lld], got [%lld, %lld]", (long long)s->n, (long long)1 << (2 * k),
                (long long)out->rows, (long long)out->cols);
    return SKR_OK;
}

// ---------------------------------------------------------------------------------------
// Any alphabet (kmer_counts.py:120-122 takes any string: A = len(alphabet) letters, A^k columns,
// column = sum code(c_p) * A^(k-1-p)).  Nothing here is bit-packed: sequences stay ASCII on the device,
// a window's column is computed from its k characters, counts go to a uint32 scratch histogram
// (one row per sequence of the batch, L2 atomics) and a second kernel turns the batch into the
// output dtype with the same per-kb arithmetic as the 4-letter path.  A correctness path for the rare
// caller; the 4-letter kernels above are the tuned ones.
// ---------------------------------------------------------------------------------------
struct GenericLut {
    int8_t code[256];  // -1: not in the alphabet
};

__global__ __launch_bounds__(kThreads) void count_generic_kernel(const unsigned char* __restrict__ bases,
                                                                 const int64_t* __restrict__ offsets, int64_t seq0,
                                                                 int64_t n_seqs, int k, int alen, int64_t nbins,
                                                                 GenericLut lut, uint32_t* __restrict__ hist) {
    for (int64_t s = blockIdx.x; s < n_seqs; s += gridDim.x) {
        const unsigned char* seq = bases + offsets[seq0 + s];
        const int64_t len = offsets[seq0 + s + 1] - offsets[seq0 + s];
        uint32_t* row = hist + (size_t)s * nbins;
        for (int64_t w = threadIdx.x; w + k <= len; w += kThreads) {
            int64_t idx = 0;
            bool ok = true;
            for (int p = 0; p < k; p++) {
                const int c = lut.code[seq[w + p]];
                ok &= c >= 0;
                idx = idx * alen + (c >= 0 ? c : 0);
            }
            if (ok) atomicAdd(&row[idx], 1u);
        }
    }
}

template <typename OutT, bool LOG2>
__global__ __launch_bounds__(kThreads) void convert_generic_kernel(const uint32_t* __restrict__ hist,
                                                                   const int64_t* __restrict__ offsets, int64_t seq0,
                                                                   int64_t n_seqs, int k, int64_t nbins, OutT* __restrict__ out) {
    for (int64_t s = blockIdx.y; s < n_seqs; s += gridDim.y) {
        const int64_t len = offsets[seq0 + s + 1] - offsets[seq0 + s];
        const double inc = 1000.0 / (double)(len - k + 1);  // len == k-1 was refused on the host
        for (int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x; b < nbins; b += (int64_t)gridDim.x * kThreads) {
            const uint32_t n = hist[(size_t)s * nbins + b];
            OutT v;
            if (sizeof(OutT) == 8) {
                v = (OutT)per_kb_value_f64(n, inc);
            } else if (std::is_same<OutT, uint32_t>::value) {
                v = (OutT)n;
            } else {
                float t = per_kb_value(n, inc);
                if (LOG2) t = skr_log2_cr(t + 1.0f);
                v = (OutT)t;
            }
            out[(size_t)(seq0 + s) * nbins + b] = v;
        }
    }
}

}  // namespace

extern "C" int skr_count_generic(skr_ctx* ctx, const char* bases, const int64_t* offsets, int64_t n, const char* alphabet,
                                 int alen, int k, int log2_pre, skr_mat* out) {
    SKR_REQUIRE(ctx && alphabet && out && out->ctx == ctx, "NULL or foreign argument");
    SKR_REQUIRE(n >= 0 && (n == 0 || offsets), "bad sequence count / offsets");
    SKR_REQUIRE(alen >= 1 && alen <= 127 && k >= 1, "alphabet of 1..127 letters and k >= 1 expected");
    double bins_d = 1.0;
    for (int p = 0; p < k; p++) bins_d *= alen;
    if (bins_d > (double)(1 << 26))
        return skr_set_error(SKR_ERR_UNSUPPORTED, "%d^%d columns: rows are supported up to 2^26 columns", alen, k);
    const int64_t nbins = (int64_t)bins_d;
    SKR_REQUIRE(out->rows == n && out->cols == nbins, "output must be [%lld, %lld], got [%lld, %lld]", (long long)n,
                (long long)nbins, (long long)out->rows, (long long)out->cols);
    SKR_REQUIRE(!(log2_pre && out->dtype != SKR_F32), "log2_pre is implemented for SKR_F32 output only");
    if (n == 0) return SKR_OK;
    SKR_REQUIRE(bases || offsets[n] == offsets[0], "bases is NULL");
    for (int64_t i = 0; i < n; i++) {
        SKR_REQUIRE(offsets[i + 1] >= offsets[i], "offsets must be non-decreasing (at %lld)", (long long)i);
        if (out->dtype != SKR_U32 && offsets[i + 1] - offsets[i] == k - 1)  // kmer_counts.py:144
            return skr_set_error(SKR_ERR_ZERODIV, "division by zero (a sequence has length k-1 = %d)", k - 1);
    }
    GenericLut lut;
    memset(lut.code, -1, sizeof(lut.code));
    // a repeated letter keeps its LAST position, as the reference's dict {kmer: index} does (:122)
    for (int c = 0; c < alen; c++) lut.code[(unsigned char)alphabet[c]] = (int8_t)c;
    SKR_TRY(skr_activate(ctx));
    const size_t total = (size_t)(offsets[n] - offsets[0]);
    unsigned char* d_bases = nullptr;
    int64_t* d_off = nullptr;
    uint32_t* d_hist = nullptr;
    const int64_t batch = std::max<int64_t>(1, std::min<int64_t>(n, ((int64_t)256 << 20) / (nbins * 4)));
    auto cleanup = [&] {
        (void)hipStreamSynchronize(ctx->stream);
        if (d_bases) (void)hipFree(d_bases);
        if (d_off) (void)hipFree(d_off);
        if (d_hist) (void)hipFree(d_hist);
    };
    hipError_t e = hipMalloc((void**)&d_bases, std::max<size_t>(total, 1));
    if (e == hipSuccess) e = hipMalloc((void**)&d_off, (size_t)(n + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc((void**)&d_hist, (size_t)batch * nbins * 4);
    std::vector<int64_t> rel((size_t)n + 1);
    for (int64_t i = 0; i <= n; i++) rel[i] = offsets[i] - offsets[0];
    if (e == hipSuccess && total) e = hipMemcpyAsync(d_bases, bases + offsets[0], total, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, rel.data(), (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
    for (int64_t s0 = 0; s0 < n && e == hipSuccess; s0 += batch) {
        const int64_t ns = std::min(batch, n - s0);
        e = hipMemsetAsync(d_hist, 0, (size_t)ns * nbins * 4, ctx->stream);
        if (e != hipSuccess) break;
        SkrProfScope prof(ctx, "count_generic");
        hipLaunchKernelGGL(count_generic_kernel, dim3((unsigned)std::min<int64_t>(ns, (int64_t)ctx->num_cu * 8)), dim3(kThreads),
                           0, ctx->stream, d_bases, d_off, s0, ns, k, alen, nbins, lut, d_hist);
        const dim3 cgrid((unsigned)std::min<int64_t>((nbins + kThreads - 1) / kThreads, 64), (unsigned)std::min<int64_t>(ns, 4096));
        if (out->dtype == SKR_U32)
            hipLaunchKernelGGL((convert_generic_kernel<uint32_t, false>), cgrid, dim3(kThreads), 0, ctx->stream, d_hist, d_off, s0,
                               ns, k, nbins, (uint32_t*)out->data);
        else if (out->dtype == SKR_F64)
            hipLaunchKernelGGL((convert_generic_kernel<double, false>), cgrid, dim3(kThreads), 0, ctx->stream, d_hist, d_off, s0, ns,
                               k, nbins, (double*)out->data);
        else if (log2_pre)
            hipLaunchKernelGGL((convert_generic_kernel<float, true>), cgrid, dim3(kThreads), 0, ctx->stream, d_hist, d_off, s0, ns,
                               k, nbins, (float*)out->data);
        else
            hipLaunchKernelGGL((convert_generic_kernel<float, false>), cgrid, dim3(kThreads), 0, ctx->stream, d_hist, d_off, s0, ns,
                               k, nbins, (float*)out->data);
        e = hipGetLastError();
    }
    cleanup();
    if (e != hipSuccess) return skr_set_error(SKR_ERR_HIP, "generic-alphabet counting failed: %s", hipGetErrorString(e));
    return SKR_OK;
}

extern "C" int skr_count_u32(skr_ctx* ctx, const skr_seqs* s, int k, skr_mat* out) {
    SKR_TRY(check_count_args(ctx, s, k, out));
    SKR_REQUIRE(out->dtype == SKR_U32, "skr_count_u32 needs a SKR_U32 matrix");
    SKR_TRY(skr_activate(ctx));
    return launch_count<OUT_U32>(ctx, s, k, out->data, "count_kmers_u32");
}

extern "C" int skr_count_per_kb(skr_ctx* ctx, const skr_seqs* s, int k, int log2_pre, skr_mat* out) {
    SKR_TRY(check_count_args(ctx, s, k, out));
    SKR_REQUIRE(out->dtype == SKR_F32 || out->dtype == SKR_F64, "skr_count_per_kb needs a float matrix");
    SKR_REQUIRE(!(log2_pre && out->dtype == SKR_F64), "log2_pre is implemented for SKR_F32 output only");
    SKR_TRY(skr_activate(ctx));
    // len == k-1 is an error in the reference (ZeroDivisionError, kmer_counts.py:144): detect on the host
    for (int64_t L : s->h_len)
        if (L == k - 1)
            return skr_set_error(SKR_ERR_ZERODIV, "division by zero (a sequence has length k-1 = %d)", k - 1);
    if (out->dtype == SKR_F64) {
        if (k > 7) return skr_set_error(SKR_ERR_UNSUPPORTED, "float64 count output is implemented for k <= 7");
        return launch_count<OUT_F64>(ctx, s, k, out->data, "count_kmers_f64");
    }
    if (log2_pre) return launch_count<OUT_F32_LOG2>(ctx, s, k, out->data, "count_kmers_f32_log2");
    return launch_count<OUT_F32>(ctx, s, k, out->data, "count_kmers_f32");
}

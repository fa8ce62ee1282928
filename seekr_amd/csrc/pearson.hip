// K6 + K7 — Pearson: row standardisation (pearson.py:35-38) and the all-pairs contraction
// r = Z1 . Z2^T / K (pearson.py:41) on the MFMA matrix cores.
//
// Both operands are K-contiguous row-major ("NT" GEMM), which is exactly the layout the MFMA
// fragments want: a lane's A/B fragment is a run of consecutive k for one row.
#include <algorithm>
#include <cmath>

#include "common.hpp"

namespace {

// =======================================================================================
// Row standardisation.  One 256-thread workgroup per row; the row (16 KiB at k=6) is read
// three times from L1/L2, written once.  Mirrors the reference step by step: mean, centre,
// then np.std of the centred row (its own mean m2, squared deviations), divide.
// =======================================================================================
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    T tot = scratch[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) tot += scratch[w];
    return tot;
}

template <typename T>
__global__ __launch_bounds__(256) void row_standardize_kernel(const T* __restrict__ x, int64_t rows, int64_t cols,
                                                              T* __restrict__ z, int64_t ldz) {
    __shared__ T scratch[4];
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const T* xr = x + (size_t)r * cols;
        T* zr = z + (size_t)r * ldz;
        const T kf = (T)cols;
        T s = 0;
        for (int64_t c = threadIdx.x; c < cols; c += blockDim.x) s += xr[c];
        const T mean = block_sum(s, scratch) / kf;
        s = 0;
        for (int64_t c = threadIdx.x; c < cols; c += blockDim.x) s += xr[c] - mean;
        const T m2 = block_sum(s, scratch) / kf;
        s = 0;
        for (int64_t c = threadIdx.x; c < cols; c += blockDim.x) {
            const T d = (xr[c] - mean) - m2;
            s += d * d;
        }
        const T sd = sqrt(block_sum(s, scratch) / kf);
        for (int64_t c = threadIdx.x; c < ldz; c += blockDim.x) zr[c] = c < cols ? (xr[c] - mean) / sd : (T)0;
    }
}

// =======================================================================================
// fp32 MFMA contraction: v_mfma_f32_32x32x2_f32 (exact f32 products, f32 accumulate).
//   block tile 128 x 128, BK = 32, 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA tiles
//   (64 accumulator VGPRs).  Operand tiles are staged global -> LDS with 16-byte LDS-DMA
//   (global_load_lds_dwordx4), double buffered; the LDS image is lane-linear, so the
//   bank-conflict swizzle is applied to the per-lane SOURCE address and undone on the read:
//   16-byte chunk c of row r lives at chunk position c ^ ((r >> 1) & 7) of the 128-byte row.
//   A lane's ds_read_b128 then delivers 4 consecutive k of its row; lanes 0-31 take k-chunk
//   2*kk, lanes 32-63 chunk 2*kk+1, and MFMA j consumes element j of both operands — the k
//   order inside a BK tile is permuted identically for A and B, which a dot product allows.
// =======================================================================================
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int kStageBytes = (BM + BN) * BK * 4;  // 32 KiB
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

// XCD-aware, grouped tile order: the 8 XCDs (blocks b, b+8, … share one) each walk a
// contiguous range of the virtual order, and inside it 64 consecutive ids form an 8 x 8
// super-tile so co-resident workgroups of an XCD share A/B panels in its L2.
__device__ __forceinline__ void tile_of_block(int64_t bid, int64_t nblk, int64_t tiles_m, int64_t tiles_n,
                                              int64_t* tm, int64_t* tn) {
    const int64_t q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
    const int64_t v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    constexpr int64_t G = 8;
    const int64_t per_group = G * tiles_n;
    const int64_t g = v / per_group;
    const int64_t first_m = g * G;
    const int64_t gsize = std::min<int64_t>(G, tiles_m - first_m);
    const int64_t in = v % per_group;
    *tm = first_m + in % gsize;
    *tn = in / gsize;
}

// SYM (self-comparison, square result block): tiles below the diagonal are skipped and written as mirrors.
template <bool SYM>
__global__ __launch_bounds__(256, 2) void pearson_gemm_f32_kernel(const float* __restrict__ A,
                                                                   const float* __restrict__ B, float* __restrict__ C,
                                                                   int64_t M, int64_t N, int64_t K, int64_t lda,
                                                                   int64_t ldb, int64_t ldc, float kdiv,
                                                                   int64_t tiles_m, int64_t tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int64_t tm, tn;
    if (SYM) {
        // plain row-major order: the grouped order hands each XCD a contiguous range of tile rows, and with the
        // lower triangle skipped the first XCD would keep all of its tiles while the last kept none
        tm = blockIdx.x / tiles_n;
        tn = blockIdx.x % tiles_n;
        if (tn < tm) return;
    } else {
        tile_of_block(blockIdx.x, (int64_t)gridDim.x, tiles_m, tiles_n, &tm, &tn);
    }
    const int64_t row_base = tm * BM, col_base = tn * BN;

    // ---- staging addresses: wave w moves pieces 4w..4w+3 (rows 32w..32w+31) of A and of B
    const float* a_src[4];
    const float* b_src[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int row = wave * 32 + p * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int64_t ra = std::min<int64_t>(row_base + row, M - 1);
        const int64_t rb = std::min<int64_t>(col_base + row, N - 1);
        a_src[p] = A + (size_t)ra * lda + chunk * 4;
        b_src[p] = B + (size_t)rb * ldb + chunk * 4;
    }
    auto stage = [&](int buf, int64_t k0) {
        char* abase = smem + buf * kStageBytes;
        char* bbase = abase + BM * BK * 4;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            lds_dma16(a_src[p] + k0, abase + (wave * 4 + p) * 1024);
            lds_dma16(b_src[p] + k0, bbase + (wave * 4 + p) * 1024);
        }
    };

    // ---- fragment read offsets (bytes inside a stage), swizzle folded in
    const int h = lane >> 5;
    int a_off[2], b_off[2], a_swz[2], b_swz[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int ra = wm * 64 + t * 32 + (lane & 31);
        const int rb = wn * 64 + t * 32 + (lane & 31);
        a_off[t] = ra * 128;
        a_swz[t] = (ra >> 1) & 7;
        b_off[t] = BM * BK * 4 + rb * 128;
        b_swz[t] = (rb >> 1) & 7;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    auto compute = [&](int buf) {
        const char* base = smem + buf * kStageBytes;
#pragma unroll
        for (int kk = 0; kk < BK / 8; kk++) {
            f32x4 af[2], bf[2];
            const int chunk = 2 * kk + h;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                af[t] = *reinterpret_cast<const f32x4*>(base + a_off[t] + ((chunk ^ a_swz[t]) << 4));
                bf[t] = *reinterpret_cast<const f32x4*>(base + b_off[t] + ((chunk ^ b_swz[t]) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int nt = 0; nt < 2; nt++)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][j], bf[nt][j], acc[mt][nt], 0, 0, 0);
        }
    };

    // A single f32 accumulator over K = 4096 strictly sequential FMAs drifts by ~2e-6 on r ~ 1
    // (error grows like sqrt(K)); BLAS in the reference sums in blocks.  Every kFlush k-tiles
    // the running tile sum is folded into `total`, which cuts the drift by the block count.
    // The MFMA's accumulate step truncates instead of rounding, which matters on rows dominated by one
    // column (a homopolymer's one-hot profile): once the big product is in the accumulator every further
    // add loses up to an ulp of it, one-sidedly — 2e-5 on r = 1 after 500 of them.  Partial sums of 64 k
    // keep that below 4e-6; the fold is 64 VALU adds per 8 192 MFMA cycles.
    constexpr int kFlush = 2;  // 2 * BK = 64 k per partial sum
    f32x16 total[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) total[i][j][e] = 0.f;
    auto flush = [&]() {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    total[i][j][e] += acc[i][j][e];
                    acc[i][j][e] = 0.f;
                }
    };

    const int64_t nk = K / BK;
    stage(0, 0);
    __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and publishes the tile
    int cur = 0;
    for (int64_t t = 0; t + 1 < nk; t++) {
        stage(cur ^ 1, (t + 1) * BK);
        compute(cur);
        if ((t + 1) % kFlush == 0) flush();
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);
    flush();

    // ---- epilogue: r = acc / K, C/D layout col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 h
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const int64_t n = col_base + wn * 64 + nt * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int64_t m = row_base + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < M && n < N) C[(size_t)m * ldc + n] = __fdiv_rn(total[mt][nt][e], kdiv);
            }
            if (SYM && tm != tn && n < N) {
                // mirror: this lane's 16 values are 4 runs of 4 consecutive rows m — 16-byte stores into row n
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int64_t m0 = row_base + wm * 64 + mt * 32 + 8 * q + 4 * h;
                    float* dst = C + (size_t)n * ldc + m0;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = __fdiv_rn(total[mt][nt][4 * q + e], kdiv);
                    if (m0 + 3 < M && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
                        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (m0 + e < M) dst[e] = v[e];
                    }
                }
            }
        }
}

// =======================================================================================
// float64 contraction (CSV / integer inputs promote to float64 in the reference):
// v_mfma_f64_16x16x4_f64, one wave per 32 x 32 output tile, operands straight from L1/L2.
// Lane l holds A[row l&15][k l>>4] and B[k l>>4][col l&15]; D: col = l&15, row = (l>>4) + 4 e.
// =======================================================================================
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void pearson_gemm_f64_kernel(const double* __restrict__ A,
                                                               const double* __restrict__ B, double* __restrict__ C,
                                                               int64_t M, int64_t N, int64_t K, int64_t lda,
                                                               int64_t ldb, int64_t ldc, double kdiv, int64_t tiles_n) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t tm = wave_global / tiles_n, tn = wave_global % tiles_n;
    if (tm * 32 >= M) return;
    const int r = lane & 15, kq = lane >> 4;
    f64x4 acc[2][2];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int e = 0; e < 4; e++) acc[i][j][e] = 0.0;
    const double* ap[2];
    const double* bp[2];
    for (int t = 0; t < 2; t++) {
        ap[t] = A + (size_t)std::min<int64_t>(tm * 32 + t * 16 + r, M - 1) * lda;
        bp[t] = B + (size_t)std::min<int64_t>(tn * 32 + t * 16 + r, N - 1) * ldb;
    }
    for (int64_t k0 = 0; k0 < K; k0 += 4) {
        const int64_t k = k0 + kq;
        double a[2], b[2];
        for (int t = 0; t < 2; t++) {
            a[t] = k < K ? ap[t][k] : 0.0;
            b[t] = k < K ? bp[t][k] : 0.0;
        }
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int e = 0; e < 4; e++) {
                const int64_t m = tm * 32 + i * 16 + kq + 4 * e;
                const int64_t n = tn * 32 + j * 16 + r;
                if (m < M && n < N) C[(size_t)m * ldc + n] = acc[i][j][e] / kdiv;
            }
}


// The same contraction, LDS-tiled: 128 x 128 block tile, 16 k per stage, 4 waves as 2 x 2, each wave
// 64 x 64 = 4 x 4 MFMA tiles (64 float64 accumulators per lane).  The one-wave-per-32x32 kernel above
// streams 4 flop per byte through the L2 (17 TFLOP/s, L2-bandwidth-bound); a 128 x 128 tile moves 16
// flop per byte, and the operand tiles reach the LDS by 16-byte LDS-DMA, double buffered, one barrier
// per stage.  The LDS image of an LDS-DMA is lane-linear, and 16 rows at a 128-byte pitch share two
// bank groups: read as they lie, every ds_read_b64 cost 128 conflict cycles and the LDS (80 % busy)
// held the matrix cores at 40 %.  So, as in the 16-bit kernel, the swizzle goes on the per-lane SOURCE
// address — 16-byte chunk c of row r lives at chunk position c ^ ((r >> 1) & 7) — and is undone on
// the read: two lanes per bank, the minimum for 512 bytes.
// SYM (self-comparison): tiles below the diagonal are skipped and written as mirrors of the ones above.
constexpr int DT = 128, DK = 16;
constexpr int kRowBytes64 = DK * 8;                 // 128
constexpr int kStageBytes64 = 2 * DT * kRowBytes64;  // A tile + B tile: 32 KiB

template <bool SYM>
__global__ __launch_bounds__(256, 2) void pearson_gemm_f64_tiled_kernel(
    const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, int64_t M, int64_t N, int64_t K,
    int64_t lda, int64_t ldb, int64_t ldc, double kdiv, int64_t tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    if (SYM && tn < tm) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t row_base = tm * DT, col_base = tn * DT;

    // staging: one LDS-DMA wave-instruction moves 8 rows x 128 bytes; wave w moves pieces 4w .. 4w+3 of A and of B
    const char* a_src[4];
    const char* b_src[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int row = (wave * 4 + p) * 8 + (lane >> 3);
        const int64_t ra = std::min<int64_t>(row_base + row, M - 1);  // rows past the end re-read the last one
        const int64_t rb = std::min<int64_t>(col_base + row, N - 1);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        a_src[p] = reinterpret_cast<const char*>(A + (size_t)ra * lda) + chunk * 16;
        b_src[p] = reinterpret_cast<const char*>(B + (size_t)rb * ldb) + chunk * 16;
    }
    auto stage = [&](int buf, int64_t k0) {
        char* abase = smem + buf * kStageBytes64;
        char* bbase = abase + DT * kRowBytes64;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            lds_dma16(a_src[p] + k0 * 8, abase + (wave * 4 + p) * 1024);
            lds_dma16(b_src[p] + k0 * 8, bbase + (wave * 4 + p) * 1024);
        }
    };
    const int r16 = lane & 15, kq = lane >> 4;
    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[i][j][e] = 0.0;

    int cur = 0;
    stage(0, 0);
    __syncthreads();
    for (int64_t k0 = 0; k0 < K; k0 += DK) {
        if (k0 + DK < K) stage(cur ^ 1, k0 + DK);
        const char* abase = smem + cur * kStageBytes64 + (wm * 64 + r16) * kRowBytes64 + (kq & 1) * 8;
        const char* bbase = smem + cur * kStageBytes64 + DT * kRowBytes64 + (wn * 64 + r16) * kRowBytes64 + (kq & 1) * 8;
#pragma unroll
        for (int kk = 0; kk < DK / 4; kk++) {
            double a[4], b[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                // rows of one fragment differ by multiples of 16, so (row >> 1) & 7 is (r16 >> 1) for A and B alike
                const int pos = ((kk * 2 + (kq >> 1)) ^ (r16 >> 1)) << 4;
                a[t] = *reinterpret_cast<const double*>(abase + t * 16 * kRowBytes64 + pos);
                b[t] = *reinterpret_cast<const double*>(bbase + t * 16 * kRowBytes64 + pos);
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        cur ^= 1;
    }
    const bool mirror = SYM && tm != tn;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int64_t m = row_base + wm * 64 + i * 16 + kq + 4 * e;
                const int64_t n = col_base + wn * 64 + j * 16 + r16;
                if (m < M && n < N) {
                    const double v = acc[i][j][e] / kdiv;
                    C[(size_t)m * ldc + n] = v;
                    if (mirror) C[(size_t)n * ldc + m] = v;
                }
            }
}

}  // namespace

extern "C" int skr_row_standardize(skr_ctx* ctx, const skr_mat* x, skr_mat* z) {
    SKR_REQUIRE(ctx && x && z, "NULL argument");
    SKR_REQUIRE(x->ctx == ctx && z->ctx == ctx, "matrix belongs to a different ctx");
    SKR_REQUIRE(x->dtype == z->dtype && (x->dtype == SKR_F32 || x->dtype == SKR_F64), "dtype mismatch");
    SKR_REQUIRE(x->rows == z->rows && z->cols >= x->cols, "z must be [rows, >= cols]");
    SKR_TRY(skr_activate(ctx));
    if (x->rows == 0 || z->cols == 0) return SKR_OK;
    const unsigned grid = (unsigned)std::min<int64_t>(x->rows, (int64_t)ctx->num_cu * 8);
    SkrProfScope prof(ctx, "row_standardize");
    if (x->dtype == SKR_F32)
        hipLaunchKernelGGL(row_standardize_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)x->data,
                           x->rows, x->cols, (float*)z->data, z->cols);
    else
        hipLaunchKernelGGL(row_standardize_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream,
                           (const double*)x->data, x->rows, x->cols, (double*)z->data, z->cols);
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

// ---- launchers used by operand.hip -----------------------------------------------------------
int skr_launch_gemm_f32(skr_ctx* ctx, const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t Kp,
                        int64_t lda, int64_t ldb, int64_t ldc, int64_t K, int symmetric) {
    const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const bool sym = symmetric && A == B && M == N && lda == ldb;  // mirrors land inside the same square block
    SkrProfScope prof(ctx, "pearson_gemm_f32");
    if (sym) {
        SKR_TRY(skr_kernel_lds(ctx, reinterpret_cast<const void*>(pearson_gemm_f32_kernel<true>), 2 * kStageBytes));
        hipLaunchKernelGGL(pearson_gemm_f32_kernel<true>, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 2 * kStageBytes,
                           ctx->stream, A, B, C, M, N, Kp, lda, ldb, ldc, (float)K, tiles_m, tiles_n);
    } else {
        SKR_TRY(skr_kernel_lds(ctx, reinterpret_cast<const void*>(pearson_gemm_f32_kernel<false>), 2 * kStageBytes));
        hipLaunchKernelGGL(pearson_gemm_f32_kernel<false>, dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 2 * kStageBytes,
                           ctx->stream, A, B, C, M, N, Kp, lda, ldb, ldc, (float)K, tiles_m, tiles_n);
    }
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

int skr_launch_gemm_f64(skr_ctx* ctx, const double* A, const double* B, double* C, int64_t M, int64_t N, int64_t K,
                        int64_t lda, int64_t ldb, int64_t ldc, double kdiv, int symmetric) {
    SkrProfScope prof(ctx, "pearson_gemm_f64");
    if (K % DK == 0 && lda % 2 == 0 && ldb % 2 == 0) {  // whole stages, 16-byte aligned rows: the tiled kernel
        const int64_t tiles_m = (M + DT - 1) / DT, tiles_n = (N + DT - 1) / DT;
        const bool sym = symmetric && A == B && M == N && lda == ldb;  // mirrors land inside the same square block
        if (sym) {
            SKR_TRY(skr_kernel_lds(ctx, reinterpret_cast<const void*>(pearson_gemm_f64_tiled_kernel<true>), 2 * kStageBytes64));
            hipLaunchKernelGGL(pearson_gemm_f64_tiled_kernel<true>, dim3((unsigned)(tiles_m * tiles_n)), dim3(256),
                               2 * kStageBytes64, ctx->stream, A, B, C, M, N, K, lda, ldb, ldc, kdiv, tiles_n);
        } else {
            SKR_TRY(skr_kernel_lds(ctx, reinterpret_cast<const void*>(pearson_gemm_f64_tiled_kernel<false>), 2 * kStageBytes64));
            hipLaunchKernelGGL(pearson_gemm_f64_tiled_kernel<false>, dim3((unsigned)(tiles_m * tiles_n)), dim3(256),
                               2 * kStageBytes64, ctx->stream, A, B, C, M, N, K, lda, ldb, ldc, kdiv, tiles_n);
        }
    } else {
        const int64_t tiles_m = (M + 31) / 32, tiles_n = (N + 31) / 32;
        const unsigned grid = (unsigned)((tiles_m * tiles_n + 3) / 4);
        hipLaunchKernelGGL(pearson_gemm_f64_kernel, dim3(grid), dim3(256), 0, ctx->stream, A, B, C, M, N, K, lda, ldb, ldc,
                           kdiv, tiles_n);
    }
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

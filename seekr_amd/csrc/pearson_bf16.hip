// K7 (fast path) — Pearson contraction on the bf16 matrix cores with split operands.
//
// Every float32 z is split as z = hi + lo + O(2^-17 |z|) with hi = bf16(z), lo = bf16(z - hi).
// r = sum_k z1 z2 is then accumulated in float32 from NPROD bf16 products per k:
//     NPROD = 3 : hi*hi + hi*lo + lo*hi            (|error| <= ~3e-6 on r ~ 1, ~2e-7 rms elsewhere)
//     NPROD = 4 : ... + lo*lo                      (error ~4e-7 = what float32 BLAS gives)
// on v_mfma_f32_32x32x16_bf16, which runs 16x the f32-input MFMA rate; a single bf16 product
// (2.7e-4) is nowhere near the 1e-5 parity bar.
//
// Operand layout ("split-interleaved", produced by split_bf16_kernel): for every row and every
// 32-wide k tile, 32 hi values followed by 32 lo values — one 128-byte line — so that staging a
// k tile of a row is one full cache line and a 1-KiB LDS-DMA piece covers 8 rows.
//
// Geometry: block tile 256 x 256, BK = 32, LDS 2 stages x (256+256) rows x 128 B = 128 KiB, one
// workgroup per CU; 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 (4 x 2 MFMA tiles, 128
// accumulator registers, 184 VGPRs), two waves per SIMD cover each other's LDS waits.  Per 16-k
// step a wave issues 12 ds_read_b128 and 8*NPROD MFMAs.
// The LDS image is lane-linear (LDS-DMA), so the bank swizzle chunk ^= (row >> 1) & 7 is applied
// on the source address and again on the read (same involution), as in the fp32 kernel.
#include <algorithm>

#include "common.hpp"

namespace {

constexpr int TM = 256, TN = 256;
constexpr int kRowBytes = 128;                       // one k tile of one row: 32 hi + 32 lo bf16
constexpr int kStageBytes = (TM + TN) * kRowBytes;   // 64 KiB
template <typename T>
using vec8 = T __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// T = __bf16 (8-bit significand halves) or _Float16 (11-bit halves: hi + lo carry 22 bits, so the
// three-product sum is float32-grade; |z| <= sqrt(K) << 65504 and fp16 subnormals are kept)
__device__ __forceinline__ f32x16 mfma16(vec8<__bf16> a, vec8<__bf16> b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma16(vec8<_Float16> a, vec8<_Float16> b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

// z (f32, [rows, cols]) -> split-interleaved bf16 [rows, kt, {hi,lo}, 32]; k >= cols padded with 0
template <typename T>
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ z, int64_t rows, int64_t cols,
                                                    int64_t kt, T* __restrict__ out) {
    const int64_t groups_per_row = kt * 4;  // 8 k per thread
    const int64_t total = rows * groups_per_row;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = g / groups_per_row, q = g % groups_per_row;
        const int64_t tile = q >> 2, sub = q & 3;
        const int64_t k0 = tile * 32 + sub * 8;
        float v[8];
        const float* src = z + (size_t)row * cols + k0;
        if (k0 + 8 <= cols && (cols % 4) == 0) {
            const float4 a = *reinterpret_cast<const float4*>(src);
            const float4 b = *reinterpret_cast<const float4*>(src + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
            v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = (k0 + j < cols) ? src[j] : 0.f;
        }
        vec8<T> hi, lo;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const T h = (T)v[j];            // hardware convert: RNE, NaN stays NaN
            hi[j] = h;
            lo[j] = (T)(v[j] - (float)h);   // exact difference, then RNE
        }
        T* dst = out + ((size_t)row * kt + tile) * 64 + sub * 8;
        *reinterpret_cast<vec8<T>*>(dst) = hi;
        *reinterpret_cast<vec8<T>*>(dst + 32) = lo;
    }
}

// Tile order.  Blocks are dealt round-robin to the 8 XCDs (b % 8 labels the XCD group); the i-th
// block of an XCD is placed so that 32 consecutive ones form an 8 x 4 sub-tile (sharing 8 A and
// 4 B panels in that XCD's L2) and the 8 sub-tiles of the 8 XCDs form one 16 x 16 super-tile
// whose 32 panels (~134 MB at K=4096) fit the 256 MB Infinity Cache.  Speed only, never
// correctness: blocks that fall outside the matrix (or below the diagonal when SYM) exit.
__device__ __forceinline__ bool tile_of_block(int64_t bid, int64_t super_n, int64_t tiles_m, int64_t tiles_n,
                                              int64_t* tm, int64_t* tn) {
    const int64_t xcd = bid & 7, i = bid >> 3;
    const int64_t s = i >> 5, j = i & 31;
    const int64_t sr = s / super_n, sc = s % super_n;
    *tm = sr * 16 + (xcd & 1) * 8 + (j & 7);
    *tn = sc * 16 + (xcd >> 1) * 4 + (j >> 3);
    return *tm < tiles_m && *tn < tiles_n;
}

template <typename T, int NPROD, int MT, int NT>
__device__ __forceinline__ void mma_step(f32x16 (&acc)[MT][NT], const vec8<T> (&ahi)[MT], const vec8<T> (&alo)[MT],
                                         const vec8<T> (&bhi)[NT], const vec8<T> (&blo)[NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            if (NPROD >= 4) acc[mt][nt] = mfma16(alo[mt], blo[nt], acc[mt][nt]);
            acc[mt][nt] = mfma16(alo[mt], bhi[nt], acc[mt][nt]);
            acc[mt][nt] = mfma16(ahi[mt], blo[nt], acc[mt][nt]);
            acc[mt][nt] = mfma16(ahi[mt], bhi[nt], acc[mt][nt]);
        }
}

// WM x WN waves; each wave owns (256/WM) x (256/WN) of the block tile.
template <typename T, int NPROD, bool SYM, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN) / 4) void pearson_gemm_bf16s_kernel(
    const T* __restrict__ A, const T* __restrict__ B, float* __restrict__ C, float* __restrict__ Ct,
    int64_t M, int64_t N, int64_t kt, int64_t ldc, float kdiv, int64_t tiles_m, int64_t tiles_n, int64_t super_n) {
    constexpr int NW = WM * WN;
    constexpr int MT = TM / WM / 32, NT = TN / WN / 32;
    constexpr int PP = 32 / NW;  // 1-KiB pieces per wave per operand per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int64_t tm, tn;
    if (!tile_of_block(blockIdx.x, super_n, tiles_m, tiles_n, &tm, &tn)) return;
    if (SYM && tn < tm) return;  // the mirror block writes this tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int64_t row_base = tm * TM, col_base = tn * TN;
    const int64_t pitch = kt * 64;  // bf16 elements per row

    // ---- staging: wave w moves pieces PP*w .. PP*w+PP-1 (8 rows each) of the A tile and of the B tile
    const T* a_src[PP];
    const T* b_src[PP];
#pragma unroll
    for (int p = 0; p < PP; p++) {
        const int row = (wave * PP + p) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int64_t ra = std::min<int64_t>(row_base + row, M - 1);
        const int64_t rb = std::min<int64_t>(col_base + row, N - 1);
        a_src[p] = A + (size_t)ra * pitch + chunk * 8;
        b_src[p] = B + (size_t)rb * pitch + chunk * 8;
    }
    auto stage = [&](int buf, int64_t tile) {
        char* abase = smem + buf * kStageBytes;
        char* bbase = abase + TM * kRowBytes;
#pragma unroll
        for (int p = 0; p < PP; p++) {
            lds_dma16(a_src[p] + tile * 64, abase + (wave * PP + p) * 1024);
            lds_dma16(b_src[p] + tile * 64, bbase + (wave * PP + p) * 1024);
        }
    };

    // ---- fragment addresses
    const int h = lane >> 5;
    int a_off[MT], a_swz[MT], b_off[NT], b_swz[NT];
#pragma unroll
    for (int t = 0; t < MT; t++) {
        const int ra = wm * (TM / WM) + t * 32 + (lane & 31);
        a_off[t] = ra * kRowBytes;
        a_swz[t] = (ra >> 1) & 7;
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int rb = wn * (TN / WN) + t * 32 + (lane & 31);
        b_off[t] = TM * kRowBytes + rb * kRowBytes;
        b_swz[t] = (rb >> 1) & 7;
    }
    auto load_frags = [&](int buf, int s, vec8<T> (&ahi)[MT], vec8<T> (&alo)[MT], vec8<T> (&bhi)[NT], vec8<T> (&blo)[NT]) {
        const char* base = smem + buf * kStageBytes;
        const int c_hi = 2 * s + h, c_lo = 4 + 2 * s + h;  // 16-byte chunk of the 128-byte row
#pragma unroll
        for (int t = 0; t < MT; t++) {
            ahi[t] = *reinterpret_cast<const vec8<T>*>(base + a_off[t] + ((c_hi ^ a_swz[t]) << 4));
            alo[t] = *reinterpret_cast<const vec8<T>*>(base + a_off[t] + ((c_lo ^ a_swz[t]) << 4));
        }
#pragma unroll
        for (int t = 0; t < NT; t++) {
            bhi[t] = *reinterpret_cast<const vec8<T>*>(base + b_off[t] + ((c_hi ^ b_swz[t]) << 4));
            blo[t] = *reinterpret_cast<const vec8<T>*>(base + b_off[t] + ((c_lo ^ b_swz[t]) << 4));
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    // Two LDS stages: tile t+1 streams in by LDS-DMA while tile t is consumed; one barrier per
    // k tile (its vmcnt(0) drain costs nothing: the fill was issued a whole iteration earlier).
    // hipcc sinks the last 16 MFMAs of an iteration below the barrier, which covers the first
    // fragment reads of the next tile.  (Measured alternatives that did not pay on gfx950: an
    // explicit two-deep fragment pipeline at 248 VGPRs, -1 %; 4 waves x 128x128 with 256
    // accumulator registers: hipcc spills the accumulators across the loop back-edge, 8x slower.)
    int cur = 0;
    stage(0, 0);
    __syncthreads();  // vmcnt(0) drain of the LDS-DMA + barrier
    for (int64_t t = 0; t < kt; t++) {
        if (t + 1 < kt) stage(cur ^ 1, t + 1);
#pragma unroll
        for (int s = 0; s < 2; s++) {
            vec8<T> ahi[MT], alo[MT], bhi[NT], blo[NT];
            load_frags(cur, s, ahi, alo, bhi, blo);
            mma_step<T, NPROD, MT, NT>(acc, ahi, alo, bhi, blo);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue.  C/D layout: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 h
    const bool mirror = SYM && tm != tn;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int64_t n = col_base + wn * (TN / WN) + nt * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int64_t m0 = row_base + wm * (TM / WM) + mt * 32 + 8 * g + 4 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = acc[mt][nt][4 * g + e] / kdiv;
                if (SYM && tm == tn) {
                    // diagonal tile: hi*lo and lo*hi enter the accumulator in a different order for
                    // (i,j) and (j,i); keep the upper element and mirror it so r is exactly symmetric
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int64_t m = m0 + e;
                        if (n < N && m < M && n >= m) {
                            C[(size_t)m * ldc + n] = v[e];
                            if (n > m) Ct[(size_t)n * ldc + m] = v[e];
                        }
                    }
                } else if (n < N) {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (m0 + e < M) C[(size_t)(m0 + e) * ldc + n] = v[e];
                    if (mirror) {  // r[n, m0..m0+3] = r[m0..m0+3, n]: 16 contiguous bytes per lane
                        float* dst = Ct + (size_t)n * ldc + m0;
                        if (m0 + 3 < M && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
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
}

template <typename T, int NPROD, bool SYM, int WM, int WN>
int launch(skr_ctx* ctx, const T* A, const T* B, float* C, int64_t M, int64_t N, int64_t kt, int64_t ldc,
           int64_t K, const char* name) {
    const int64_t tiles_m = (M + TM - 1) / TM, tiles_n = (N + TN - 1) / TN;
    const int64_t super_m = (tiles_m + 15) / 16, super_n = (tiles_n + 15) / 16;
    const int64_t grid = super_m * super_n * 256;
    auto kern = pearson_gemm_bf16s_kernel<T, NPROD, SYM, WM, WN>;
    SKR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * kStageBytes));
    SkrProfScope prof(ctx, name);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WM * WN * 64), 2 * kStageBytes, ctx->stream, A, B, C, C, M, N,
                       kt, ldc, (float)K, tiles_m, tiles_n, super_n);
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

template <typename T>
int gemm_split(skr_ctx* ctx, const skr_mat* a, const skr_mat* b, int nprod, int symmetric, skr_mat* r, int64_t row0,
               int64_t col0, const char* name) {
    const int64_t M = a->rows, N = b->rows, K = a->cols;
    const int64_t kt = (K + 31) / 32;
    const bool same = a->data == b->data && M == N;
    const bool sym = symmetric && same;  // mirror is relative to the block's own base pointer
    // split-interleaved operands in the ctx workspace
    const size_t a_bytes = (size_t)M * kt * kRowBytes, b_bytes = same ? 0 : (size_t)N * kt * kRowBytes;
    void* ws = nullptr;
    SKR_TRY(skr_ctx_workspace(ctx, a_bytes + b_bytes + 256, &ws));
    T* As = (T*)ws;
    T* Bs = same ? As : (T*)((char*)ws + ((a_bytes + 255) & ~(size_t)255));
    auto split = [&](const skr_mat* m, T* dst) -> int {
        const int64_t total = m->rows * kt * 4;
        const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, (int64_t)ctx->num_cu * 16));
        SkrProfScope prof(ctx, "split_halves");
        hipLaunchKernelGGL(split_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const float*)m->data,
                           m->rows, m->cols, kt, dst);
        SKR_HIP(hipGetLastError());
        return SKR_OK;
    };
    SKR_TRY(split(a, As));
    if (!same) SKR_TRY(split(b, Bs));
    float* C = (float*)r->data + (size_t)row0 * r->cols + col0;
    if (nprod == 3) {
        if (sym) return launch<T, 3, true, 2, 4>(ctx, As, Bs, C, M, N, kt, r->cols, K, name);
        return launch<T, 3, false, 2, 4>(ctx, As, Bs, C, M, N, kt, r->cols, K, name);
    }
    if (sym) return launch<T, 4, true, 2, 4>(ctx, As, Bs, C, M, N, kt, r->cols, K, name);
    return launch<T, 4, false, 2, 4>(ctx, As, Bs, C, M, N, kt, r->cols, K, name);
}

}  // namespace

int skr_pearson_gemm_split(skr_ctx* ctx, const skr_mat* a, const skr_mat* b, int precision, int symmetric, skr_mat* r,
                           int64_t row0, int64_t col0) {
    switch (precision) {
        case SKR_PREC_BF16X3: return gemm_split<__bf16>(ctx, a, b, 3, symmetric, r, row0, col0, "pearson_gemm_bf16x3");
        case SKR_PREC_BF16X4: return gemm_split<__bf16>(ctx, a, b, 4, symmetric, r, row0, col0, "pearson_gemm_bf16x4");
        case SKR_PREC_F16X3: return gemm_split<_Float16>(ctx, a, b, 3, symmetric, r, row0, col0, "pearson_gemm_f16x3");
        default: return skr_set_error(SKR_ERR_INVALID, "not a split precision: %d", precision);
    }
}

// Pearson operands: one fused pass turns (raw or normalised) count rows into what the matrix
// cores consume — optional elementwise normalisation tail (kmer_counts.py:169,175,208-209), row
// standardisation (pearson.py:35-38) and the split into 16-bit halves — so the float32 z matrix
// never makes a round trip through HBM, and a shard received from another GPU is used as is.
#include <algorithm>
#include <cmath>

#include <cstdlib>
#include <type_traits>

#include "common.hpp"

namespace {

// hi half of the two-halves split.  Rounded to nearest, every copy of a repeated value gets the same residual,
// and a count profile that is mostly zeros becomes thousands of identical (hi, lo) pairs whose tiny hi*lo
// products the MFMA's truncating accumulate cuts all in the same direction: 2e-5 on r = 0.96 (found by the
// fuzzer; alphabet of 7 letters, 90 % of the columns structurally zero).  So fp16 halves are rounded down or up
// by a hash of the COLUMN: the residuals of a repeated value come with both signs, the truncation errors cancel
// like noise, and hi + lo still carries the value to 2^-21 relative.  (bf16 halves keep round-to-nearest.)
//
// Directed rounding without the device library's two software conversions (round-down AND round-up evaluated for
// every cell, ~30 instructions; the fill is bound by its VALU instructions, not by HBM): RU(z) = -RD(-z), so the sign
// of z is flipped where the hash says "up", the value converted to nearest, stepped one bit pattern towards -inf if
// that overshot, and the sign flipped back — the same fp16 value as __ocml_cvtrtp / __ocml_cvtrtn bit for bit.
__device__ __forceinline__ uint32_t split_flip(int64_t col) {  // 0x80000000 where the hi half is rounded UP
    return ((uint32_t)col * 0x9E3779B1u) & 0x80000000u;
}
__device__ __forceinline__ _Float16 split_hi_f16(float zs, uint32_t flip) {
    const float u = __uint_as_float(__float_as_uint(zs) ^ flip);
    const _Float16 h = (_Float16)u;  // to nearest; NaN stays NaN
    uint16_t bits = __builtin_bit_cast(uint16_t, h);
    // towards -inf: a positive value that overshot goes one pattern down, a negative one (or -0) one pattern up
    const uint16_t stepped = (int16_t)bits < 0 ? (uint16_t)(bits + 1) : (uint16_t)(bits - 1);
    bits = (float)h > u ? stepped : bits;
    return __builtin_bit_cast(_Float16, (uint16_t)(bits ^ (uint16_t)(flip >> 16)));
}
template <typename T>
__device__ __forceinline__ T split_hi(float zs, int64_t col) {
    return (T)zs;  // hardware convert: RNE, NaN stays NaN
}
template <>
__device__ __forceinline__ _Float16 split_hi<_Float16>(float zs, int64_t col) {
    return split_hi_f16(zs, split_flip(col));
}
// ... and with the direction of the cell known to the caller (the register kernel keeps one bit per cell of its lane)
template <typename T>
__device__ __forceinline__ T split_hi_flip(float zs, uint32_t flip) {
    return (T)zs;
}
template <>
__device__ __forceinline__ _Float16 split_hi_flip<_Float16>(float zs, uint32_t flip) {
    return split_hi_f16(zs, flip);
}

// x / d for float32 x and d, rounded exactly as the IEEE division rounds it, from r = RN64(1 / d): the product
// RN64(x * r) is within 2^-52 (relative) of x / d, and a quotient of two float32 numbers that is not itself a float32
// number is at least 2^-49 (relative) away from every point where the float32 rounding changes (x - m d is a nonzero
// multiple of the last of the 49 bits of m d for a 25-bit midpoint m), so the float64 product rounds to float32 the
// way the exact quotient does.  Zero, infinite and NaN divisors behave as in x / d (r = inf, 0, NaN).  Three
// instructions instead of the ten of v_div_scale / v_rcp / 4 fma / v_div_fmas / v_div_fixup.
__device__ __forceinline__ float div_by_recip(float x, double r) { return (float)((double)x * r); }

template <typename T>
using vec8 = T __attribute__((ext_vector_type(8)));

// Wave-wide sums and maxima on the DPP network (round 5; until then a __shfl_xor butterfly, i.e. a ds_bpermute — an LDS
// operation — per step, six dependent steps a sum): four DPP steps leave every lane of a 16-lane row with its row's total,
// the four row totals are read with v_readlane and added.  Every lane holds the result.  (The summation order differs from
// the butterfly's: the row statistics of the fills, and with them the last bits of r, are round 5's.)
template <bool IS_MAX, int CTRL>
__device__ __forceinline__ float dpp_step_c(float v) {
    const float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
    return IS_MAX ? fmaxf(v, o) : v + o;
}
template <bool IS_MAX>
__device__ __forceinline__ float row16_all(float v) {  // every lane: the reduction over its row of 16 lanes
    v = dpp_step_c<IS_MAX, 0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_step_c<IS_MAX, 0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_step_c<IS_MAX, 0x141>(v);  // row_half_mirror
    v = dpp_step_c<IS_MAX, 0x140>(v);  // row_mirror
    return v;
}
template <bool IS_MAX, int N>
__device__ __forceinline__ float rowN_all(float v) {  // every lane: the reduction over its 16 (or 8) neighbouring lanes
    v = dpp_step_c<IS_MAX, 0xB1>(v);
    v = dpp_step_c<IS_MAX, 0x4E>(v);
    v = dpp_step_c<IS_MAX, 0x141>(v);
    if (N == 16) v = dpp_step_c<IS_MAX, 0x140>(v);
    return v;
}
template <bool IS_MAX>
__device__ __forceinline__ float wave64_all(float v) {  // every lane: the reduction over the wave
    v = row16_all<IS_MAX>(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return IS_MAX ? fmaxf(fmaxf(r0, r1), fmaxf(r2, r3)) : (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_sum(float v) { return wave64_all<false>(v); }
__device__ __forceinline__ float wave_max(float v) { return wave64_all<true>(v); }

// Dynamic range a single float32 accumulator per cell can take.  The MFMA adds each product into the
// accumulator aligned to the accumulator's exponent: once a huge product (two rows sharing one
// dominant column, e.g. the raw counts of two homopolymers) sits in it, every product below ~2^-24
// of it is dropped, not summed — for a one-hot row that is ALL the other columns, 1/K of r (2.4e-4
// at k = 6), and the adds that are not dropped are truncated at the accumulator's unit.  A row is
// flagged when one column carries at least 1/16 of its energy (max z^2 >= K/16); flagged operands are
// refilled in the float32 layout and take the fp32 kernel, whose accumulation is blocked and rounded
// (skr_operand_fill).  Without a dominant column the accumulator grows gradually and stays inside
// the bar (the contraction also restarts it every 4 096 columns, pearson_bf16.hip).
__device__ __forceinline__ bool row_needs_fp32(float zmax2, float K) {
    // With the huge product in the accumulator from the start, every later MFMA add is truncated at
    // ITS unit — measured on two identical rows with max z^2 = 0.77 K at k = 7: r = 1 - 2.4e-5, i.e.
    // ~0.25 ulp lost per add (the MFMA adder truncates), where the gradual growth of an ordinary
    // r ~ 1 pair costs half of that and everything else far less.
    // Below 1 024 columns a row meets the energy rule all the time (|z| >= 2 at k = 3) and does not need it: the loss is
    // at most one unit of the largest partial sum per MFMA add, 3 K / 32 adds in all — <= 2.9e-6 of an r ~ 1 at
    // K = 256, and half of that at most for an r ~ 0 whose partial sums climbed to K / 2 on the way.
    return K >= 1024.f && zmax2 * 16.f >= K;
}

struct FillArgs {
    const float* x;
    int64_t rows, cols, kt;
    const void* center;  // ck: 0 none, 1 f32, 2 f64
    const void* scale;
    const double* scale_recip;  // 1 / scale in float64 (register kernels with a float32 scale vector: div_by_recip)
    int ck, sk, post, row_standardize;
    float shift;
    float* y;  // optional normalised-count output (may alias x)
    void* out;
    float* diag;  // diag[r] = <z_r, z_r> / K from a float32 tree sum (see skr_pearson_gemm_op)
    float out_scale;  // power of two applied before the split (fp16 halves only, see f16_scale)
    uint32_t* flags;
    const struct NpPlan* np_plan;  // generic kernel: numpy's pairwise summation unrolled for this row width (NULL: tree sums)
};


// ---- numpy's pairwise summation, addition for addition (numpy/_core/src/umath/loops_utils.h.src: @TYPE@_pairwise_sum),
// which is what np.mean / np.std(axis=1) of pearson.py:35-38 run along a row: fewer than 8 values are added one after the
// other; up to 128 go into eight accumulators r[j] += a[i + j] (i in steps of 8) that are folded as
// ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)), the n % 8 leftovers added one by one; more than 128 are split at n/2 rounded
// down to a multiple of 8, recursively.  On rows whose standardisation is ill-conditioned (4 or 16 near-equal values:
// k = 1, 2) the float32 result depends on every one of these roundings, so the rows the generic fill kernel serves
// (every width but the three register-resident ones) are summed in exactly this order: z is then numpy's z bit for bit
// and what is left between the device and the reference is the inner product's own rounding.  (Checked against
// np.add.reduce for every n up to 300 and a dozen larger ones in the CPU test suite.)
constexpr int kNpMaxLeaves = 128;  // n <= 8 192 (a leaf holds 65 .. 128 values once n > 128)
constexpr int kNpExactMaxCols = 8192;
// the recursion for a row width, unrolled on the HOST (it is the same for every row of a launch): the leaves from left to
// right and the additions that join them in post-order
struct NpPlan {
    int n_leaves, prog_len;
    unsigned short leaf_start[kNpMaxLeaves], leaf_n[kNpMaxLeaves];
    unsigned char prog[2 * kNpMaxLeaves];  // 0 = push the next leaf's sum, 1 = add the two on top
};
struct NpScratch {  // per wave, in LDS
    float leaf_sum[kNpMaxLeaves];
    float stack[16];
};

void np_plan_build(NpPlan* p, int n) {
    int sp = 0, st_s[32], st_n[32], st_k[32];  // explicit recursion stack: (start, n, 0 = expand | 1 = emit an add)
    p->n_leaves = 0;
    p->prog_len = 0;
    st_s[sp] = 0, st_n[sp] = n, st_k[sp] = 0, sp++;
    while (sp > 0) {
        sp--;
        const int s0 = st_s[sp], n0 = st_n[sp], k0 = st_k[sp];
        if (k0 == 1) {
            p->prog[p->prog_len++] = 1;
        } else if (n0 <= 128) {
            p->leaf_start[p->n_leaves] = (unsigned short)s0;
            p->leaf_n[p->n_leaves] = (unsigned short)n0;
            p->n_leaves++;
            p->prog[p->prog_len++] = 0;
        } else {
            int n2 = n0 / 2;
            n2 -= n2 % 8;
            st_s[sp] = 0, st_n[sp] = 0, st_k[sp] = 1, sp++;                 // after both halves: add
            st_s[sp] = s0 + n2, st_n[sp] = n0 - n2, st_k[sp] = 0, sp++;     // right half (popped second)
            st_s[sp] = s0, st_n[sp] = n2, st_k[sp] = 0, sp++;               // left half (popped first)
        }
    }
}

// sum of f(0) .. f(n-1) in numpy's order; all 64 lanes call it, all return the sum.  Eight lanes share a leaf (one
// accumulator each), eight leaves per round.
template <class F>
__device__ __forceinline__ float np_pairwise_sum(const NpPlan* __restrict__ p, NpScratch* sc, F f, int lane) {
    const int slot = lane >> 3, j = lane & 7;
    for (int base = 0; base < p->n_leaves; base += 8) {
        const int leaf = base + slot;
        const bool live = leaf < p->n_leaves;
        const int s0 = live ? p->leaf_start[leaf] : 0, n0 = live ? p->leaf_n[leaf] : 0;
        float r = 0.f;
        if (n0 >= 8) {
            // a leaf holds at most 128 values = 16 per accumulator: all 16 reads first (independent), then the chain of adds
            const int n8 = n0 - (n0 % 8);
            float v[16];
#pragma unroll
            for (int m = 0; m < 16; m++) v[m] = 8 * m + 8 <= n8 ? f(s0 + 8 * m + j) : 0.f;
            r = v[0];
#pragma unroll
            for (int m = 1; m < 16; m++) r = 8 * m + 8 <= n8 ? __fadd_rn(r, v[m]) : r;
        } else if (j == 0) {
            for (int i = 0; i < n0; i++) r = __fadd_rn(r, f(s0 + i));  // res = 0.; res += a[i]
        }
        // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)): the lanes of a leaf are neighbours
        float x = __fadd_rn(r, __shfl_down(r, 1, 64));
        float y = __fadd_rn(x, __shfl_down(x, 2, 64));
        float z = __fadd_rn(y, __shfl_down(y, 4, 64));
        if (live && j == 0) {
            if (n0 >= 8) {
                for (int i = n0 - (n0 % 8); i < n0; i++) z = __fadd_rn(z, f(s0 + i));
                sc->leaf_sum[leaf] = z;
            } else {
                sc->leaf_sum[leaf] = r;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float res = 0.f;
    if (lane == 0) {  // the recursion's additions, in its order
        int sp = 0, next = 0;
        for (int i = 0; i < p->prog_len; i++) {
            if (p->prog[i] == 0) {
                sc->stack[sp++] = sc->leaf_sum[next++];
            } else {
                sp--;
                sc->stack[sp - 1] = __fadd_rn(sc->stack[sp - 1], sc->stack[sp]);
            }
        }
        res = sc->stack[0];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return __shfl(res, 0, 64);
}

// One WAVE per row (grid-stride over rows): the row is loaded once with 16-byte coalesced loads,
// parked in a wave-private LDS slice between the passes, and all reductions are wave shuffles, so
// there is no workgroup barrier anywhere.  T = float (zero-padded float32 operand), __bf16 or
// _Float16 (split-interleaved halves).
__device__ __forceinline__ float fill_tail(const FillArgs& a, float v, int64_t c, bool& any_nan) {
    if (a.ck == 1) v = __fsub_rn(v, reinterpret_cast<const float*>(a.center)[c]);
    else if (a.ck == 2) v = (float)((double)v - reinterpret_cast<const double*>(a.center)[c]);
    if (a.sk == 1) v = __fdiv_rn(v, reinterpret_cast<const float*>(a.scale)[c]);
    else if (a.sk == 2) v = (float)((double)v / reinterpret_cast<const double*>(a.scale)[c]);
    if (v != v) any_nan = true;
    if (a.post) {
        v = skr_log2_of_sum1(__fadd_rn(v, a.shift));
    }
    return v;
}

// Rows on TWO tight levels (binary profiles, +-1 patterns, any two values with a jitter up to a percent): after
// standardisation every cell of a level has the same z, every product of two near-copies of such a row nearly the same
// value, and the k-tile sums that reach the MFMA's truncating accumulate are as regular as a clock — the truncations all
// go one way (50 / 50 levels: r = 1.0000122, 1.01 - 1.19 bars at 4 096 and 16 384 columns; other shares 0.25 - 0.63;
// three or more levels 0.13; found by tests/fuzz_pearson.py's structured-row class in round 4).  A standardised row is on two
// levels exactly when the Pearson inequality kurtosis >= skewness^2 + 1 holds with equality (jitter of 1 %: 5e-4 above it;
// three levels: >= 0.25): such rows are 'coherent' like the rows that are mostly one value, the contraction restarts its
// accumulators every 32 k-tiles for them (<= 0.23 bars).  m3, m4: the row's sum of z^3, z^4 over K; var: sum of z^2 over K.
__device__ __forceinline__ bool row_on_two_levels(float var, float m3, float m4) {
    const float sk = m3 / (var * sqrtf(var)), ku = m4 / (var * var);
    return ku - sk * sk - 1.0f < 5e-3f;  // false for NaN rows
}

template <typename T>
__global__ __launch_bounds__(256) void operand_fill_kernel(FillArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int64_t K = a.cols, Kp = a.kt * 32;
    float* row = lds + (size_t)wave * ((K + 3) & ~(int64_t)3);
    const bool vec = (K & 3) == 0;
    // numpy-ordered row sums (above): the plan lives behind the row slices, one per wave
    const NpPlan* plan = a.np_plan;
    const bool np_exact = a.row_standardize && plan != nullptr;
    NpScratch* np_sc = reinterpret_cast<NpScratch*>(lds + (size_t)waves * ((K + 3) & ~(int64_t)3)) + wave;
    bool any_nan = false, overflow = false, outlier = false, coherent = false;
    for (int64_t r = (int64_t)blockIdx.x * waves + wave; r < a.rows; r += (int64_t)gridDim.x * waves) {
        const float* xr = a.x + (size_t)r * K;
        // ---- pass 1: load, elementwise tail of the normalisation, optional write-back, row sum
        float s = 0.f;
        if (vec) {
            for (int64_t c = lane * 4; c < K; c += 256) {
                float4 v = *reinterpret_cast<const float4*>(xr + c);
                v.x = fill_tail(a, v.x, c, any_nan);
                v.y = fill_tail(a, v.y, c + 1, any_nan);
                v.z = fill_tail(a, v.z, c + 2, any_nan);
                v.w = fill_tail(a, v.w, c + 3, any_nan);
                if (a.y) *reinterpret_cast<float4*>(a.y + (size_t)r * K + c) = v;
                *reinterpret_cast<float4*>(row + c) = v;
                s += (v.x + v.y) + (v.z + v.w);
            }
        } else {
            for (int64_t c = lane; c < K; c += 64) {
                const float v = fill_tail(a, xr[c], c, any_nan);
                if (a.y) a.y[(size_t)r * K + c] = v;
                row[c] = v;
                s += v;
            }
        }
        // ---- pass 2: row statistics in the order pearson.py:35-38 computes them
        float mean = 0.f, sd = 1.f;
        if (np_exact) {
            // np.mean(x, 1); x - mean; np.std of that: mean of the centred row, squared deviations from it, sqrt — every sum
            // in numpy's pairwise order, every step a separately rounded float32 operation (numpy/_core/_methods.py: _mean, _var)
            const float kf = (float)K;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the row parked above is read across lanes
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            mean = __fdiv_rn(np_pairwise_sum(plan, np_sc, [&](int c) { return row[c]; }, lane), kf);
            const float m2 = __fdiv_rn(np_pairwise_sum(plan, np_sc, [&](int c) { return __fsub_rn(row[c], mean); }, lane), kf);
            const float var = __fdiv_rn(np_pairwise_sum(plan, np_sc, [&](int c) {
                const float d = __fsub_rn(__fsub_rn(row[c], mean), m2);
                return __fmul_rn(d, d);
            }, lane), kf);
            sd = (float)sqrt((double)var);  // correctly rounded, as np.sqrt
        } else if (a.row_standardize) {
            const float kf = (float)K;
            mean = wave_sum(s) / kf;
            s = 0.f;
            for (int64_t c = lane; c < K; c += 64) s += row[c] - mean;
            const float m2 = wave_sum(s) / kf;
            s = 0.f;
            for (int64_t c = lane; c < K; c += 64) {
                const float d = (row[c] - mean) - m2;
                s += d * d;
            }
            sd = sqrtf(wave_sum(s) / kf);
        }
        float zmax2 = 0.f;  // largest z^2 of the row (pass 3 needs it up front)
        if (sizeof(T) != 4) {
            for (int64_t c = lane; c < K; c += 64) {
                const float zc = (row[c] - mean) / sd;
                zmax2 = fmaxf(zmax2, zc * zc);
            }
            zmax2 = wave_max(zmax2);
            // share of the row held by its minimum (see operand_fill_reg_kernel); equal raw values give equal z
            float vmin = row[0];
            for (int64_t c = lane; c < K; c += 64) vmin = fminf(vmin, row[c]);
            vmin = -wave_max(-vmin);
            // ... or by ANY one value (round 4: a centred row repeats 0, not its minimum — the differential fuzzer's soak found
            // r = -0.46 off by 1.07 bars on rows that were 89 % zeros): if one value fills 85 % of the cells, at least 70 % of
            // the neighbouring pairs are equal (each other cell spoils two pairs at most), whatever the value is
            float same = 0.f, adj = 0.f;
            for (int64_t c = lane; c < K; c += 64) {
                same += (float)(row[c] == vmin);
                if (c + 1 < K) adj += (float)(row[c] == row[c + 1]);
            }
            if (wave_sum(same) >= 0.85f * (float)K || wave_sum(adj) >= 0.70f * (float)(K - 1)) coherent = true;
        }
        // ---- pass 3: emit the operand row, 8 k per lane and step
        float sq = 0.f, m3 = 0.f, m4 = 0.f;
        for (int64_t g = lane; g < a.kt * 4; g += 64) {
            const int64_t tile = g >> 2, sub = g & 3, k0 = tile * 32 + sub * 8;
            float z[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int64_t k = k0 + j;
                float v = k < K ? row[k] : 0.f;
                if (a.row_standardize && k < K) v = __fdiv_rn(__fsub_rn(v, mean), sd);
                z[j] = v;
                sq = __fmaf_rn(v, v, sq);
                if (sizeof(T) != 4) {
                    const float v2 = v * v;
                    m3 = fmaf(v2, v, m3);
                    m4 = fmaf(v2, v2, m4);
                }
            }
            if (sizeof(T) == 4) {
                float* dst = reinterpret_cast<float*>(a.out) + (size_t)r * Kp + k0;
                *reinterpret_cast<float4*>(dst) = make_float4(z[0], z[1], z[2], z[3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(z[4], z[5], z[6], z[7]);
            } else {
                vec8<T> hi, lo;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float zs = z[j] * a.out_scale;
                    if (fabsf(zs) > 65504.f) overflow = true;
                    const T hh = split_hi<T>(zs, k0 + j);
                    hi[j] = hh;
                    lo[j] = (T)(zs - (float)hh);        // exact difference, then RNE
                }
                T* dst = reinterpret_cast<T*>(a.out) + ((size_t)r * a.kt + tile) * 64 + sub * 8;
                *reinterpret_cast<vec8<T>*>(dst) = hi;
                *reinterpret_cast<vec8<T>*>(dst + 32) = lo;
            }
        }
        sq = wave_sum(sq);
        if (lane == 0) a.diag[r] = sq / (float)K;
        if (sizeof(T) != 4 && a.row_standardize && row_on_two_levels(sq / (float)K, wave_sum(m3) / (float)K, wave_sum(m4) / (float)K))
            coherent = true;
        if (sizeof(T) != 4 && row_needs_fp32(zmax2, (float)K)) outlier = true;
    }
    if (any_nan) atomicOr(&a.flags[1], 1u);
    if (overflow) atomicOr(&a.flags[3], 1u);
    if (outlier) atomicOr(&a.flags[4], 1u);
    if (coherent) atomicOr(&a.flags[5], 1u);
}

// One WORKGROUP per row, for rows of 32 KiB and more (k >= 7), where a wave-private LDS slice would
// leave one or two waves per CU.  IN_LDS: the row is parked in LDS between the passes (k = 7: 64 KiB,
// two workgroups per CU); otherwise (k >= 8: 256 KiB and up) the later passes re-read it from global
// memory, where it stays in the L2.  A thread owns the same groups of 8 consecutive columns in every
// pass, so what it parked (in LDS, or in `y`) in pass 1 is its own later on; without `y` and without
// LDS the normalisation tail is simply recomputed from x.
// THREADS: 256, or 1 024 (round 5) for rows too wide for the LDS (k >= 8: 10.3 -> 9.0 ms per 20 000 x 65 536).  A row is a chain of passes, each ending in a workgroup-wide sum: with four waves a pass
// is many rounds of a few loads per thread and the chain's latency is the kernel (49 us per 62 KB row at 2 workgroups per
// CU); sixteen waves put the whole row in flight at once.
template <typename T, bool IN_LDS, int THREADS = 256>
__global__ __launch_bounds__(THREADS) void operand_fill_block_kernel(FillArgs a) {
    extern __shared__ __attribute__((aligned(16))) float rowbuf[];
    constexpr int WAVES = THREADS / 64;
    __shared__ float red[WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t K = a.cols, Kp = a.kt * 32, groups = a.kt * 4;
    const bool vec = (K & 7) == 0;
    bool any_nan = false, overflow = false, outlier = false, coherent = false;
    auto block_sum = [&](float v) -> float {
        v = wave_sum(v);
        if (lane == 0) red[wave] = v;
        __syncthreads();
        float t = (red[0] + red[1]) + (red[2] + red[3]);
        if (WAVES == 16) t = (t + ((red[4] + red[5]) + (red[6] + red[7]))) + (((red[8] + red[9]) + (red[10] + red[11])) + ((red[12] + red[13]) + (red[14] + red[15])));
        __syncthreads();
        return t;
    };
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const int64_t n4 = K >> 2;
    for (int64_t r = blockIdx.x; r < a.rows; r += gridDim.x) {
        const float* xr = a.x + (size_t)r * K;
        float* yr = a.y ? a.y + (size_t)r * K : nullptr;
        bool scratch_nan = false;
        auto val = [&](int64_t c) -> float {
            if (IN_LDS) return rowbuf[c];
            return yr ? yr[c] : fill_tail(a, xr[c], c, scratch_nan);
        };
        // the 8 cells of group g as they were parked in pass 1 (LDS, or y), 16 bytes at a time where the width allows it —
        // round 4: the later passes used to fetch them one float at a time (30 ms per 20 000 x 65 536, 0.5 TB/s); cells past
        // the end of the row read as 0 and are never used
        auto load8 = [&](int64_t g, float (&v)[8]) {
            const int64_t c0 = g * 8;
            if (IN_LDS && c0 + 8 <= K) {  // the LDS copy of the row starts at a 16-byte boundary whatever the width
                const float4 u = *reinterpret_cast<const float4*>(rowbuf + c0), w = *reinterpret_cast<const float4*>(rowbuf + c0 + 4);
                v[0] = u.x, v[1] = u.y, v[2] = u.z, v[3] = u.w, v[4] = w.x, v[5] = w.y, v[6] = w.z, v[7] = w.w;
            } else if (vec && yr) {
                const float* src = yr + c0;
                const float4 u = *reinterpret_cast<const float4*>(src), w = *reinterpret_cast<const float4*>(src + 4);
                v[0] = u.x, v[1] = u.y, v[2] = u.z, v[3] = u.w, v[4] = w.x, v[5] = w.y, v[6] = w.z, v[7] = w.w;
            } else {
#pragma unroll
                for (int jj = 0; jj < 8; jj++) v[jj] = c0 + jj < K ? val(c0 + jj) : 0.f;
            }
        };
        float s = 0.f, vmin = INFINITY, vmax = -INFINITY;
        if (!vec && IN_LDS) {
            // A width that is not a multiple of 8 (5^6, 7^4 ... columns: rows start at 4-byte boundaries only).  Round 5: the
            // row comes in — and the normalised counts go out — in 16-byte pieces, lane after lane (a dword-aligned
            // global_load_dwordx4 is as good as an aligned one), four pieces in flight per thread, and is parked in the
            // LDS; the sums are then taken from the LDS copy in the order they always were (a thread's groups of 8
            // cells, cell by cell): same bits, 8.4 -> 2.x ms per 50 000 x 15 625 (0.14 -> 0.4x of the HBM peak).
            for (int64_t i0 = tid; i0 < n4; i0 += 4 * THREADS) {
                f4u q[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int64_t i = i0 + THREADS * u < n4 ? i0 + THREADS * u : n4 - 1;  // past the end: re-read the last piece
                    q[u] = *reinterpret_cast<const f4u*>(xr + 4 * i);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int64_t i = i0 + THREADS * u;
                    if (i < n4) {
                        const int64_t c = 4 * i;
                        f4u w;
                        w[0] = fill_tail(a, q[u][0], c, any_nan);
                        w[1] = fill_tail(a, q[u][1], c + 1, any_nan);
                        w[2] = fill_tail(a, q[u][2], c + 2, any_nan);
                        w[3] = fill_tail(a, q[u][3], c + 3, any_nan);
                        if (yr) *reinterpret_cast<f4u*>(yr + c) = w;
                        *reinterpret_cast<float4*>(rowbuf + c) = make_float4(w[0], w[1], w[2], w[3]);
                    }
                }
            }
            if (tid < (K & 3)) {  // the last one to three cells
                const int64_t c = 4 * n4 + tid;
                const float v = fill_tail(a, xr[c], c, any_nan);
                if (yr) yr[c] = v;
                rowbuf[c] = v;
            }
            __syncthreads();
            for (int64_t g = tid; g < groups; g += THREADS) {
                float v8[8];
                load8(g, v8);
#pragma unroll
                for (int jj = 0; jj < 8; jj++)
                    if (g * 8 + jj < K) {
                        s += v8[jj];
                        vmin = fminf(vmin, v8[jj]);
                        vmax = fmaxf(vmax, v8[jj]);
                    }
            }
        } else
        for (int64_t g = tid; g < groups; g += THREADS) {
            const int64_t c0 = g * 8;
            // `groups` covers the operand's row PADDED to whole 32-column tiles: a width that is a multiple of 8 but not of
            // 32 (10^4, 22^3, 14^4 ... columns) has up to three groups past the end of the row.  They hold nothing: until
            // round 5 the vector path below read them anyway — the next row's first cells —, added them to the row sum and
            // wrote them "back" through y (found by widening the odd-width test: r 5 bars off at 8 200 columns).
            if (vec && c0 >= K) continue;
            if (vec) {
                float4 u = *reinterpret_cast<const float4*>(xr + c0), w = *reinterpret_cast<const float4*>(xr + c0 + 4);
                u.x = fill_tail(a, u.x, c0, any_nan);
                u.y = fill_tail(a, u.y, c0 + 1, any_nan);
                u.z = fill_tail(a, u.z, c0 + 2, any_nan);
                u.w = fill_tail(a, u.w, c0 + 3, any_nan);
                w.x = fill_tail(a, w.x, c0 + 4, any_nan);
                w.y = fill_tail(a, w.y, c0 + 5, any_nan);
                w.z = fill_tail(a, w.z, c0 + 6, any_nan);
                w.w = fill_tail(a, w.w, c0 + 7, any_nan);
                if (yr) {
                    *reinterpret_cast<float4*>(yr + c0) = u;
                    *reinterpret_cast<float4*>(yr + c0 + 4) = w;
                }
                if (IN_LDS) {
                    *reinterpret_cast<float4*>(rowbuf + c0) = u;
                    *reinterpret_cast<float4*>(rowbuf + c0 + 4) = w;
                }
                s += ((u.x + u.y) + (u.z + u.w)) + ((w.x + w.y) + (w.z + w.w));
                vmin = fminf(fminf(fminf(vmin, fminf(u.x, u.y)), fminf(u.z, u.w)), fminf(fminf(w.x, w.y), fminf(w.z, w.w)));
                vmax = fmaxf(fmaxf(fmaxf(vmax, fmaxf(u.x, u.y)), fmaxf(u.z, u.w)), fmaxf(fmaxf(w.x, w.y), fmaxf(w.z, w.w)));
            } else {
                // widths that are not a multiple of 8 (5^6, 7^4 ... columns): the group's eight loads first, all in flight
                // together (one dependent load per cell made this pass the whole kernel: 8.1 ms per 50 000 x 15 625), then
                // the same arithmetic in the same order
                float raw8[8];
#pragma unroll
                for (int jj = 0; jj < 8; jj++) raw8[jj] = c0 + jj < K ? xr[c0 + jj] : 0.f;
#pragma unroll
                for (int jj = 0; jj < 8; jj++) {
                    const int64_t c = c0 + jj;
                    if (c < K) {
                        const float v = fill_tail(a, raw8[jj], c, any_nan);
                        if (yr) yr[c] = v;
                        if (IN_LDS) rowbuf[c] = v;
                        s += v;
                        vmin = fminf(vmin, v);
                        vmax = fmaxf(vmax, v);
                    }
                }
            }
        }
        float mean = 0.f, sd = 1.f;
        // pass 2 also counts how much of the row one value holds (its minimum, or — neighbouring cells equal — any value:
        // operand_fill_kernel), which needs the row minimum of pass 1
        if (sizeof(T) != 4) {
            vmin = -wave_max(-vmin);
            vmax = wave_max(vmax);
            if (lane == 0) red[wave] = vmin;
            __syncthreads();
            vmin = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
#pragma unroll
            for (int w = 4; w < WAVES; w++) vmin = fminf(vmin, red[w]);
            __syncthreads();
            if (lane == 0) red[wave] = vmax;
            __syncthreads();
            vmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
#pragma unroll
            for (int w = 4; w < WAVES; w++) vmax = fmaxf(vmax, red[w]);
            __syncthreads();
        }
        float same = 0.f, adj = 0.f;
        auto count_repeats = [&](int64_t g, const float (&v)[8]) {
            const int64_t c0 = g * 8;
#pragma unroll
            for (int jj = 0; jj < 8; jj++)
                if (c0 + jj < K) {
                    same += (float)(v[jj] == vmin);
                    if (jj < 7) {
                        if (c0 + jj + 1 < K) adj += (float)(v[jj] == v[jj + 1]);
                    } else if (c0 + 8 < K) {
                        adj += (float)(v[7] == val(c0 + 8));
                    }
                }
        };
        if (a.row_standardize) {
            const float kf = (float)K;
            mean = block_sum(s) / kf;
            s = 0.f;
            for (int64_t g = tid; g < groups; g += THREADS) {
                float v[8];
                load8(g, v);
#pragma unroll
                for (int jj = 0; jj < 8; jj++)
                    if (g * 8 + jj < K) s += v[jj] - mean;
                if (sizeof(T) != 4) count_repeats(g, v);
            }
            const float m2 = block_sum(s) / kf;
            s = 0.f;
            for (int64_t g = tid; g < groups; g += THREADS) {
                float v[8];
                load8(g, v);
#pragma unroll
                for (int jj = 0; jj < 8; jj++)
                    if (g * 8 + jj < K) {
                        const float d = (v[jj] - mean) - m2;
                        s += d * d;
                    }
            }
            sd = sqrtf(block_sum(s) / kf);
        } else if (sizeof(T) != 4) {
            for (int64_t g = tid; g < groups; g += THREADS) {
                float v[8];
                load8(g, v);
                count_repeats(g, v);
            }
        }
        float zmax2 = 0.f;
        if (sizeof(T) != 4) {
            // the largest z^2 of the row sits at its largest or its smallest value: rounding is monotone, so these two
            // evaluations are what the maximum over all cells was (NaN cells never counted; a row without a finite value: 0)
            if (vmin <= vmax) {
                const float z_lo = (vmin - mean) / sd, z_hi = (vmax - mean) / sd;
                zmax2 = fmaxf(z_lo * z_lo, z_hi * z_hi);
                if (!(zmax2 == zmax2)) zmax2 = 0.f;
            }
            const float n_same = block_sum(same), n_adj = block_sum(adj);
            if (n_same >= 0.85f * (float)K || n_adj >= 0.70f * (float)(K - 1)) coherent = true;
        }
        float sq = 0.f, m3 = 0.f, m4 = 0.f;
        for (int64_t g = tid; g < groups; g += THREADS) {
            const int64_t tile = g >> 2, sub = g & 3, k0 = g * 8;
            float z[8];
            load8(g, z);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int64_t k = k0 + j;
                float v = k < K ? z[j] : 0.f;
                if (a.row_standardize && k < K) v = __fdiv_rn(__fsub_rn(v, mean), sd);
                z[j] = v;
                sq = __fmaf_rn(v, v, sq);
                if (sizeof(T) != 4) {
                    const float v2 = v * v;
                    m3 = fmaf(v2, v, m3);
                    m4 = fmaf(v2, v2, m4);
                }
            }
            if (sizeof(T) == 4) {
                float* dst = reinterpret_cast<float*>(a.out) + (size_t)r * Kp + k0;
                *reinterpret_cast<float4*>(dst) = make_float4(z[0], z[1], z[2], z[3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(z[4], z[5], z[6], z[7]);
            } else {
                vec8<T> hi, lo;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float zs = z[j] * a.out_scale;
                    if (fabsf(zs) > 65504.f) overflow = true;
                    const T hh = split_hi<T>(zs, k0 + j);
                    hi[j] = hh;
                    lo[j] = (T)(zs - (float)hh);
                }
                T* dst = reinterpret_cast<T*>(a.out) + ((size_t)r * a.kt + tile) * 64 + sub * 8;
                *reinterpret_cast<vec8<T>*>(dst) = hi;
                *reinterpret_cast<vec8<T>*>(dst + 32) = lo;
            }
        }
        sq = block_sum(sq);  // its barriers also fence the LDS row against the next row's pass 1
        if (tid == 0) a.diag[r] = sq / (float)K;
        if (sizeof(T) != 4 && a.row_standardize) {
            const float s3 = block_sum(m3), s4 = block_sum(m4);
            if (row_on_two_levels(sq / (float)K, s3 / (float)K, s4 / (float)K)) coherent = true;
        }
        if (sizeof(T) != 4 && row_needs_fp32(zmax2, (float)K)) outlier = true;
    }
    if (any_nan) atomicOr(&a.flags[1], 1u);
    if (overflow) atomicOr(&a.flags[3], 1u);
    if (outlier) atomicOr(&a.flags[4], 1u);
    if (coherent) atomicOr(&a.flags[5], 1u);
}

template <typename T>
using vec4h = T __attribute__((ext_vector_type(4)));

// ---- round 5: ANY width up to VPT * 4 096 columns, the row in the registers of a sixteen-wave workgroup ------------------
// The block kernel above parks a row in the LDS (or re-reads it from the L2) and walks it once per statistic, four waves a
// row: a chain of passes during which nothing of the next row is on its way — 26 us per 62 KB row, 1.1 TB/s on 5^6 columns,
// 1.5 TB/s on 4^8.  Here thread t of 1 024 owns the 16-byte pieces t, t + 1 024, ... of the row (VPT of them: 4 for rows
// of up to 16 384 cells, 16 up to 65 536), all loads of a row in flight at once and — VPT = 4 — the NEXT row's pieces
// requested as soon as this row's have been taken over, so that they arrive under this row's statistics and stores; rows
// start on 4-byte boundaries only when the width is odd, which a dword-aligned global_load_dwordx4 does not mind.  Every
// statistic is one wave sum + ONE barrier (two alternating slots of partials).  MODE as in the register kernel below
// (0: rows as they are, 1: float32 centre + scale, 2: + the Log2.post tail); anything else stays with the block kernel.
// The arithmetic per cell is the block kernel's; the row SUMS are taken in another order (16 waves, pieces lane after
// lane), so mean and std of a row — and with them the last bits of r — are those of this kernel wherever it serves.
template <typename T, int VPT, int MODE, bool HASY, int THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(THREADS == 512 ? 2 : 4))) void operand_fill_rowreg_kernel(FillArgs a) {
    constexpr int WAVES = THREADS / 64;
    // 1 024 threads: one workgroup per CU (128 registers a thread), the next row prefetched; 512 threads: two workgroups per
    // CU, each on its own row and in its own phase — no prefetch needed, the other workgroup fills the waits
    constexpr bool PREFETCH = THREADS == 1024 && VPT <= 4;  // (a 65 536-cell row, 64 cells a thread: no room for a second one)
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    __shared__ float red[2][3][WAVES];
    __shared__ float edge[VPT][WAVES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (32-bit column arithmetic: the widest row this kernel takes has 65 536 cells)
    const int K = (int)a.cols, Kp = (int)a.kt * 32;
    const int n4p = Kp >> 2;  // pieces of the padded operand row
    const float kf = (float)K;
    bool any_nan = false, overflow = false, outlier = false, coherent = false;
    int phase = 0;
    // up to three sums (or maxima) over the workgroup with one barrier: the partials of consecutive calls alternate slots
    auto reduce3 = [&](float& p0, float& p1, float& p2, auto m0, auto m1, auto m2) {  // m*: std::true_type = maximum, false_type = sum
        constexpr bool M0 = decltype(m0)::value, M1 = decltype(m1)::value, M2 = decltype(m2)::value;
        p0 = wave64_all<M0>(p0), p1 = wave64_all<M1>(p1), p2 = wave64_all<M2>(p2);
        float(*slot)[WAVES] = red[phase & 1];
        phase++;
        if (lane == 0) slot[0][wave] = p0, slot[1][wave] = p1, slot[2][wave] = p2;
        // A barrier for the LDS only.  __syncthreads() is a workgroup-scope release + acquire around s_barrier, which on
        // gfx9 means s_waitcnt vmcnt(0): every wave would sit at the FIRST barrier of a row until the next row's prefetch
        // — issued a few instructions earlier — had landed, and at every later one until its stores had drained (~1 us a
        // barrier: the fixed 5 us per row this kernel had).  The partials only travel through the LDS.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // every row of 16 lanes folds the 16 partials by itself — one LDS read and four DPP steps per quantity (reading all
        // 48 partials into registers, on top of this row and the next, is what made the kernel spill)
        static_assert(WAVES == 16 || WAVES == 8, "a row of 16 lanes (or half of one) folds the partials");
        p0 = rowN_all<M0, WAVES>(slot[0][lane & (WAVES - 1)]);
        p1 = rowN_all<M1, WAVES>(slot[1][lane & (WAVES - 1)]);
        p2 = rowN_all<M2, WAVES>(slot[2][lane & (WAVES - 1)]);
    };
    const std::true_type kMax;
    const std::false_type kSum;
    // rounding direction of the hi half of each of this thread's VPT * 4 cells, one bit per cell (a hash of the COLUMN: the
    // same for every row)
    constexpr int DIRW = (VPT * 4 + 31) / 32;
    uint32_t dir[DIRW] = {};
    if (sizeof(T) == 2) {
#pragma unroll
        for (int u = 0; u < VPT; u++)
#pragma unroll
            for (int j = 0; j < 4; j++) dir[(u * 4 + j) >> 5] |= (split_flip(4 * (tid + THREADS * u) + j) >> 31) << ((u * 4 + j) & 31);
    }
    // Cells c .. c + 3 of a K-cell float vector WITHOUT a branch: hipcc answers a load inside a branch with s_waitcnt
    // vmcnt(0) right behind it — the four pieces of a row came in one after the other, each at full latency, and the
    // "prefetch" of the next row was waited for on the spot (the fixed ~5 us per row this kernel had).  The address is
    // clamped into the vector (the ragged last piece reads the vector's LAST four cells, pieces past the end likewise) and
    // the lanes are rotated into place; cells at or past K come back as anything — every user masks them with c + j < K.
    // The lanes are NOT rotated into place here but where the cells are used (rot4): a select right behind the load is a
    // wait for it — vmcnt(0), hipcc even made branches of the rotation — and the mixed wave sat out its own prefetch of the
    // next row on the spot while fifteen waves waited for it at the next barrier (2.75 -> 2.2 ms per 50 000 rows of 5^6).
    auto load4 = [&](const float* base, int c) -> f4u {
        const int cc = c + 3 < K ? c : K - 4;
        return *reinterpret_cast<const f4u*>(base + cc);
    };
    // cell j of the piece at c is element min(j + sh, 3) of what load4 brought, sh = c - min(c, K - 4): two select stages
    auto shift_of = [&](int c) { return c + 3 < K ? 0 : c - (K - 4); };
    auto rot4 = [&](auto (&q)[4], int sh) {  // in place
        const bool b0 = sh & 1, b1 = sh & 2;
        const auto a0 = b0 ? q[1] : q[0], a1 = b0 ? q[2] : q[1], a2 = b0 ? q[3] : q[2];
        q[0] = b1 ? a2 : a0, q[1] = b1 ? q[3] : a1, q[2] = b1 ? q[3] : a2;
    };
    // The body of a row, compiled twice (round 5, after the ISA showed ~130 instructions per CELL, most of them the exec-mask
    // bookkeeping of `c + j < K` — sixteen 64-bit lane masks a thread, spilled to VGPR lanes and re-read through
    // v_readlane — and the lane rotation of the clamped loads).  Which cells of a wave's pieces exist is a property of the
    // WAVE: with K = 15 625 the pieces of waves 0 .. 12 are whole in every lane, waves 14 and 15 have three whole pieces
    // and nothing behind them, and only wave 13 holds the ragged piece and the operand's zero padding.  So a wave whose
    // first NP pieces are whole in all 64 lanes and whose others lie past the padded row runs the UNMASKED body over NP
    // pieces (MASKED = false: no comparison with K anywhere but the one neighbour test at the end of a piece); the one or
    // two waves around the end of the row keep the masked body (NP = VPT).  All variants meet at the same barriers.
    auto run = [&](auto np_, auto masked_) {
    constexpr int NP = decltype(np_)::value;
    constexpr bool MASKED = decltype(masked_)::value;
    // only the LAST of the NP pieces can be ragged / padding / void in some lane: the others are whole in the entire wave
    auto M = [&](int u) { return MASKED && u == NP - 1; };
    f4u nx[NP];
    // (unmasked: every address is a wave-uniform base — row, piece u — plus ONE 32-bit byte offset per thread, the
    // "saddr + voffset" form of global_load / global_store: no 64-bit address arithmetic and no address registers per piece)
    const uint32_t tb = (uint32_t)tid * 16u;
    // (pointer arithmetic, not an integer round trip: through uintptr_t the pointer loses its address space and the loads
    // become FLAT ones — counted in lgkmcnt too, i.e. waited for at every LDS-only barrier below)
    auto at = [&](auto* ubase, uint32_t bytes) {
        typedef std::remove_pointer_t<decltype(ubase)> E;
        return reinterpret_cast<E*>(reinterpret_cast<char*>(const_cast<std::remove_const_t<E>*>(ubase)) + bytes);
    };
    auto piece = [&](const float* base, int u) -> f4u {
        if (M(u)) return load4(base, 4 * (tid + THREADS * u));
        return *reinterpret_cast<const f4u*>(at(base + 4 * THREADS * u, tb));
    };
    uint64_t nan_lanes = 0;
    auto unordered = [&](float x, float y) -> uint64_t {  // lanes in which x or y is NaN
        uint64_t lanes;
        asm volatile("v_cmp_u_f32_e64 %0, %1, %2" : "=s"(lanes) : "v"(x), "v"(y));
        return lanes;
    };
    auto fetch = [&](int64_t row) {
        const float* src = a.x + (size_t)row * K;
#pragma unroll
        for (int u = 0; u < NP; u++) nx[u] = piece(src, u);
    };
    // centre and reciprocal scale of piece u: the same for every row.  Piece 0's stay in registers for the whole launch, piece
    // u + 1's are requested while piece u is worked on — loaded at the head of each piece (as until round 5) the L2 latency
    // of three dependent-free loads stood in front of every piece: four exposed round trips a row.
    typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
    struct Cs { f4u m; d2u e0, e1; };
    // (the two vectors through pointers that are made opaque once per row: the loads are loop-invariant, and hoisted out of
    // the row loop by the compiler — all of them, twelve registers a piece — they are what made the 32-piece body spill)
    const float* center_p = reinterpret_cast<const float*>(a.center);
    const double* recip_p = a.scale_recip;
    auto cs_load = [&](int u) -> Cs {
        Cs o;
        const int c = 4 * (tid + THREADS * u);
        o.m = piece(center_p, u);
        const int cc = !M(u) || c + 3 < K ? c : K - 4;  // the same clamping as load4
        const double* rc = M(u) ? recip_p + cc : at(recip_p + 4 * THREADS * u, 2u * tb);
        o.e0 = *reinterpret_cast<const d2u*>(rc), o.e1 = *reinterpret_cast<const d2u*>(rc + 2);
        return o;
    };
    constexpr bool RESIDENT0 = VPT <= 4;  // (64 cells a thread leave no twelve registers for it: piece 0's are loaded per row)
    Cs cs0 = {};
    if (MODE >= 1 && RESIDENT0) cs0 = cs_load(0);
    if (PREFETCH && (int64_t)blockIdx.x < a.rows) fetch(blockIdx.x);
    for (int64_t r = blockIdx.x; r < a.rows; r += gridDim.x) {
        float v[NP][4];
        if (!PREFETCH) fetch(r);
        Cs cs = cs0;
        asm volatile("" : "+s"(center_p), "+s"(recip_p));
        if (MODE >= 1 && !RESIDENT0) cs = cs_load(0);
        // (opaque per row: hoisted out of the row loop, the sixteen sign masks `dir` expands to cost sixteen registers)
        uint32_t dir_r[DIRW];
#pragma unroll
        for (int w = 0; w < DIRW; w++) {
            dir_r[w] = dir[w];
            asm volatile("" : "+v"(dir_r[w]));
        }
#pragma unroll
        for (int u = 0; u < NP; u++) {
            const int c = 4 * (tid + THREADS * u);
            float q[4] = {nx[u][0], nx[u][1], nx[u][2], nx[u][3]};
            if (M(u)) rot4(q, shift_of(c));
#pragma unroll
            for (int j = 0; j < 4; j++) v[u][j] = !M(u) || c + j < K ? q[j] : 0.f;
        }
        // ---- the elementwise tail of the normalisation (kmer_counts.py:169,175,208-209) and the optional write-back
        float s = 0.f, vmin = INFINITY, vmax = -INFINITY;
#pragma unroll
        for (int u = 0; u < NP; u++) {
            const int c = 4 * (tid + THREADS * u);
            Cs cs_next = {};
            if (MODE >= 1 && u + 1 < NP) {
                cs_next = cs_load(u + 1);
                __builtin_amdgcn_sched_barrier(0);  // requested BEFORE this piece's arithmetic, not wherever the scheduler sinks them
            }
            if (MODE >= 1) {
                // x / scale as float(double(x) * (1 / double(scale))): three instructions instead of an IEEE division, the same
                // bits (the register kernel's div_by_recip; the reciprocals are made once per launch: recip64_kernel)
                // centre and reciprocals of the mixed piece: clamped like the cells, rotated by the same shift, here
                float m[4] = {cs.m[0], cs.m[1], cs.m[2], cs.m[3]};
                double d[4] = {cs.e0[0], cs.e0[1], cs.e1[0], cs.e1[1]};
                if (M(u)) rot4(m, shift_of(c)), rot4(d, shift_of(c));
                float t[4];
#pragma unroll
                for (int j = 0; j < 4; j++) t[j] = div_by_recip(__fsub_rn(v[u][j], m[j]), d[j]);
                // NaN among the quotients, tested HERE (volatile): left to the compiler, every `t != t` of a row was deferred to
                // the end of the row and the sixty-four quotients of a 4^8 row kept alive (spilled) next to their logarithms
                if (!M(u)) nan_lanes |= unordered(t[0], t[1]) | unordered(t[2], t[3]);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    // (no branch around a cell: the mixed piece computes every lane and SELECTS — a cell past the row is 0
                    // before and after; as `if (c + j < K) { ... }` each cell was an exec-mask region of its own)
                    const bool ok = !M(u) || c + j < K;
                    if (M(u)) any_nan |= ok && t[j] != t[j];
                    if (MODE == 2) t[j] = skr_log2_of_sum1(__fadd_rn(t[j], a.shift));
                    v[u][j] = ok ? t[j] : 0.f;
                }
            }
            if (HASY && (!M(u) || c < K)) {
                float* yr = at(a.y + (size_t)r * K + 4 * THREADS * u, tb);
                if (!M(u) || c + 3 < K) {
                    *reinterpret_cast<f4u*>(yr) = f4u{v[u][0], v[u][1], v[u][2], v[u][3]};
                } else {
#pragma unroll
                    for (int j = 0; j < 3; j++)
                        if (c + j < K) yr[j] = v[u][j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool ok = !M(u) || c + j < K;
                s += v[u][j];  // (a cell past the row holds 0)
                vmin = fminf(vmin, ok ? v[u][j] : INFINITY);
                vmax = fmaxf(vmax, ok ? v[u][j] : -INFINITY);
            }
            if (sizeof(T) != 4 && lane == 0) edge[u][wave] = v[u][0];  // the first cell of this wave's piece: the wave before needs it
            __builtin_amdgcn_sched_barrier(0);
            cs = cs_next;
        }
        // The next row's pieces, requested HERE: behind this row's centre / scale loads (vmcnt counts in order — a wait for
        // those would otherwise sit out the prefetch too) and in front of everything that only computes and stores.  The last
        // iteration re-reads its own row: an unconditional load (load4).
        if (PREFETCH) fetch(r + gridDim.x < a.rows ? r + gridDim.x : r);
        // ---- row statistics in the order pearson.py:35-38 computes them; minimum and maximum ride on the first barrier
        // the row sum, minimum and maximum ride on ONE barrier
        float nmin = -vmin;
        if (sizeof(T) != 4 || a.row_standardize) reduce3(s, nmin, vmax, kSum, kMax, kMax);
        vmin = -nmin;
        float mean = 0.f, sd = 1.f;
        double rsd = 1.0;
        float same = 0.f, adj = 0.f;
        if (sizeof(T) != 4) {
            // how much of the row one value holds: its minimum, or — neighbouring cells equal — any value (block kernel)
#pragma unroll
            for (int u = 0; u < NP; u++) {
                const int c = 4 * (tid + THREADS * u);
                // the cell after this piece: the next lane's first one; lane 63: the next wave's (the last wave: piece u + 1 of wave 0)
                float nxt = __shfl_down(v[u][0], 1, 64);
                if (lane == 63) nxt = wave + 1 < WAVES ? edge[u][wave + 1] : (u + 1 < VPT ? edge[u + 1 < VPT ? u + 1 : u][0] : 0.f);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool ok = !M(u) || c + j < K;
                    same += (float)(ok && v[u][j] == vmin);
                    const float after = j < 3 ? v[u][j + 1] : nxt;
                    // (the cell after a whole piece is the one comparison with K the unmasked body keeps: the row may end there)
                    const bool ok_after = (!M(u) && j < 3) || c + j + 1 < K;
                    adj += (float)(ok_after && v[u][j] == after);
                }
            }
        }
        if (a.row_standardize) {
            mean = s / kf;
            s = 0.f;
#pragma unroll
            for (int u = 0; u < NP; u++) {
                const int c = 4 * (tid + THREADS * u);
#pragma unroll
                for (int j = 0; j < 4; j++) s += !M(u) || c + j < K ? v[u][j] - mean : 0.f;
            }
            reduce3(s, same, adj, kSum, kSum, kSum);
            const float m2 = s / kf;
            s = 0.f;
#pragma unroll
            for (int u = 0; u < NP; u++) {
                const int c = 4 * (tid + THREADS * u);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float dd = !M(u) || c + j < K ? (v[u][j] - mean) - m2 : 0.f;
                    s += dd * dd;
                }
            }
            float z2 = 0.f, z3 = 0.f;
            reduce3(s, z2, z3, kSum, kSum, kSum);
            sd = sqrtf(s / kf);
            rsd = 1.0 / (double)sd;  // once per row: the quotients below cost three instructions each (div_by_recip)
        } else if (sizeof(T) != 4) {
            float z0 = 0.f;
            reduce3(same, adj, z0, kSum, kSum, kSum);
        }
        float zmax2 = 0.f;
        if (sizeof(T) != 4) {
            if (vmin <= vmax) {
                const float z_lo = (vmin - mean) / sd, z_hi = (vmax - mean) / sd;
                zmax2 = fmaxf(z_lo * z_lo, z_hi * z_hi);
                if (!(zmax2 == zmax2)) zmax2 = 0.f;
            }
            if (same >= 0.85f * kf || adj >= 0.70f * (float)(K - 1)) coherent = true;
        }
        // ---- standardise, split, store (operand rows are zero-padded to whole 32-column tiles)
        float sq = 0.f, m3 = 0.f, m4 = 0.f;
#pragma unroll
        for (int u = 0; u < NP; u++) {
            const int i = tid + THREADS * u, k0 = 4 * i;
            if (M(u) && i >= n4p) continue;
            float z[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float t = v[u][j];  // (0 past the row)
                if (a.row_standardize) t = !M(u) || k0 + j < K ? div_by_recip(__fsub_rn(t, mean), rsd) : 0.f;
                z[j] = t;
                sq = __fmaf_rn(t, t, sq);
                if (sizeof(T) != 4) {
                    const float t2 = t * t;
                    m3 = fmaf(t2, t, m3);
                    m4 = fmaf(t2, t2, m4);
                }
            }
            if (sizeof(T) == 4) {
                float* dst = at(reinterpret_cast<float*>(a.out) + (size_t)r * Kp + 4 * THREADS * u, tb);
                *reinterpret_cast<float4*>(dst) = make_float4(z[0], z[1], z[2], z[3]);
            } else {
                vec4h<T> hi, lo;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float zs = z[j] * a.out_scale;
                    if (fabsf(zs) > 65504.f) overflow = true;
                    const T hh = split_hi_flip<T>(zs, (dir_r[(u * 4 + j) >> 5] << (31 - ((u * 4 + j) & 31))) & 0x80000000u);
                    hi[j] = hh;
                    lo[j] = (T)(zs - (float)hh);
                }
                // piece i = tid + THREADS u sits in tile i / 8, columns 4 (i % 8): (tid / 8) tiles + THREADS / 8 tiles a piece
                T* dst = at(reinterpret_cast<T*>(a.out) + ((size_t)r * a.kt + (THREADS / 8) * u) * 64,
                            (uint32_t)(((tid >> 3) * 64 + 4 * (tid & 7)) * sizeof(T)));
                *reinterpret_cast<vec4h<T>*>(dst) = hi;
                *reinterpret_cast<vec4h<T>*>(dst + 32) = lo;
            }
            __builtin_amdgcn_sched_barrier(0);  // one piece at a time: interleaved, the splits of all pieces are alive at once
        }
        reduce3(sq, m3, m4, kSum, kSum, kSum);
        if (tid == 0) a.diag[r] = sq / kf;
        if (sizeof(T) != 4 && a.row_standardize && row_on_two_levels(sq / kf, m3 / kf, m4 / kf)) coherent = true;
        if (sizeof(T) != 4 && row_needs_fp32(zmax2, kf)) outlier = true;
    }
    if (nan_lanes) any_nan = true;
    };
    // which body this wave runs: pieces [0, whole) are whole in every lane of the wave; piece `whole` is either entirely past
    // the padded row (then so are the ones behind it) or MIXED — the ragged piece, the operand's zero padding, lanes past it
    int whole = 0;
#pragma unroll
    for (int u = 0; u < VPT; u++) {
        const int last = wave * 64 + 63 + THREADS * u;
        if (4 * last + 3 < K && whole == u) whole = u + 1;
    }
    whole = __builtin_amdgcn_readfirstlane(whole);
    const bool mixed = whole < VPT && wave * 64 + THREADS * whole < n4p;
    // (the launch condition — more than 2 x 4 096 columns — makes the first two pieces of every wave whole)
    static_assert(VPT == 4 || VPT == 16, "bodies for 2, 3 and 4 pieces, or for exactly 16 whole ones (4^8 columns)");
    if (VPT == 16) {
        if (whole == VPT && !mixed) run(std::integral_constant<int, VPT>(), std::false_type());
        else __builtin_trap();
    } else if (whole == 4) run(std::integral_constant<int, 4>(), std::false_type());
    else if (whole == 3 && !mixed) run(std::integral_constant<int, 3>(), std::false_type());
    else if (whole == 3) run(std::integral_constant<int, VPT == 4 ? 4 : 1>(), std::true_type());
    else if (whole == 2 && !mixed) run(std::integral_constant<int, 2>(), std::false_type());
    else if (whole == 2) run(std::integral_constant<int, 3>(), std::true_type());
    else __builtin_trap();
    if (any_nan) atomicOr(&a.flags[1], 1u);
    if (overflow) atomicOr(&a.flags[3], 1u);
    if (outlier) atomicOr(&a.flags[4], 1u);
    if (coherent) atomicOr(&a.flags[5], 1u);
}

// Register-resident variant for K = VPL * 256 columns (k = 5: VPL 4, k = 6: VPL 16): a wave keeps
// its whole row in VPL float4 registers per lane — all loads of a row are in flight together, no
// LDS, no barrier, ~100 VGPRs so 20 waves per CU stay resident.  Lane l owns columns
// 256 i + 4 l .. +3, so its halves go out as 8-byte pieces (8 lanes fill the 64-byte hi part of a
// 32-k tile; the lo store fills the other half of the same 128-byte line).

// MODE 0: rows as they are; 1: float32 center + scale; 2: center + scale + Log2.post tail.
// (Compile-time modes: with the runtime ck/sk/post switches of fill_tail inside the unrolled
// row hipcc unswitches the loop into many copies and spills.)
// RW = waves that share a row (1: a wave per row; 4: the workgroup's four waves take 4^7 columns
// together, wave w owning the 256-column pieces w, w + 4, ... and the row sums crossing the waves
// through 16 bytes of LDS and one barrier each).
// X8 (T = _Float16 only; round 4, the opt-in SKR_PREC_F16F8): the "H / X lines" layout of the two-product-unit
// contraction — per 64 columns one 128-byte line of the 64 fp16 hi halves (rounded to nearest: |lo| <= ulp / 2), then one of
// 64 fp8 (e4m3) copies of hi / 128 followed by 64 fp8 copies of lo x 16 (pearson_bf16.hip, NPROD = 2; both stay below 448:
// |hi| <= 2^15, |lo| <= 16) — and a flag for rows in which neighbouring cells
// repeat each other (few-valued rows: the fp8 roundings of a repeated value are the same everywhere and add up).
template <typename T, int VPL, int MODE, int RW, bool HASY, bool X8 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VPL == 16 ? 2 : 4))) void operand_fill_reg_kernel(FillArgs a) {
    static_assert(!X8 || std::is_same<T, _Float16>::value, "the fp8 cross layout pairs with fp16 hi halves");
    __shared__ float red[15][4];
    const int lane = threadIdx.x & 63;
    // the wave index through readfirstlane: the compiler then KNOWS the row index is wave-uniform and keeps every row
    // base in scalar registers (derived from threadIdx it is 'divergent', and all addresses become 64-bit VGPR pairs)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), waves = blockDim.x >> 6;
    constexpr int64_t K = (int64_t)VPL * 256 * RW;
    bool any_nan = false, overflow = false, outlier = false, coherent = false, repeats = false;
    // sum / max over the row; `slot` separates the reductions of one row so that one barrier each is enough
    auto row_sum = [&](float v, int slot) -> float {
        v = wave_sum(v);
        if (RW == 1) return v;
        if (lane == 0) red[slot][wave] = v;
        __syncthreads();
        return (red[slot][0] + red[slot][1]) + (red[slot][2] + red[slot][3]);
    };
    auto row_max = [&](float v, int slot) -> float {
        v = wave_max(v);
        if (RW == 1) return v;
        if (lane == 0) red[slot][wave] = v;
        __syncthreads();
        return fmaxf(fmaxf(red[slot][0], red[slot][1]), fmaxf(red[slot][2], red[slot][3]));
    };
    const int64_t r_first = RW == 1 ? (int64_t)blockIdx.x * waves + wave : (int64_t)blockIdx.x;
    const int piece0 = RW == 1 ? 0 : wave;  // first 256-column piece of this wave
    // rounding direction of the hi half of each of this lane's VPL * 4 cells, one bit per cell (the same for every row)
    uint32_t dir[(VPL * 4 + 31) / 32] = {};
    if (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < VPL; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                dir[(i * 4 + j) >> 5] |= (split_flip((int64_t)(i * RW + piece0) * 256 + lane * 4 + j) >> 31) << ((i * 4 + j) & 31);
    }
    // launched with exactly one row per wave (RW == 1) or per workgroup (RW == 4), see launch_fill: no row loop
    const int64_t r = r_first;
    if (r < a.rows) {
        const float* xr = a.x + (size_t)r * K;
        float4 v[VPL];
#pragma unroll
        for (int i = 0; i < VPL; i++) v[i] = *reinterpret_cast<const float4*>(xr + (i * RW + piece0) * 256 + lane * 4);
        float s = 0.f;
        if (MODE >= 1) {
            // Column vectors (mean, and 1 / std in float64) four 256-column pieces at a time, the next four in flight while
            // these are used: left to the scheduler all 16 pieces are hoisted to the top (64 + 128 registers on top of
            // the row) and the kernel drops to one wave per SIMD.
            constexpr int G = VPL < 4 ? VPL : 4, NG = VPL / G;
            int nan_seen = 0;
            float4 mv[2][G];
            double2 rv[2][G][2];
            auto load_group = [&](int g, int slot) {
#pragma unroll
                for (int q = 0; q < G; q++) {
                    const int64_t c = ((g * G + q) * RW + piece0) * 256 + lane * 4;
                    mv[slot][q] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.center) + c);
                    rv[slot][q][0] = *reinterpret_cast<const double2*>(a.scale_recip + c);
                    rv[slot][q][1] = *reinterpret_cast<const double2*>(a.scale_recip + c + 2);
                }
            };
            load_group(0, 0);
#pragma unroll
            for (int g = 0; g < NG; g++) {
                if (g + 1 < NG) load_group(g + 1, (g + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < G; q++) {
                    const int i = g * G + q;
                    const int64_t c = (i * RW + piece0) * 256 + lane * 4;
                    const float4 m = mv[g & 1][q];
                    v[i].x = div_by_recip(__fsub_rn(v[i].x, m.x), rv[g & 1][q][0].x);
                    v[i].y = div_by_recip(__fsub_rn(v[i].y, m.y), rv[g & 1][q][0].y);
                    v[i].z = div_by_recip(__fsub_rn(v[i].z, m.z), rv[g & 1][q][1].x);
                    v[i].w = div_by_recip(__fsub_rn(v[i].w, m.w), rv[g & 1][q][1].y);
                    nan_seen |= (int)((v[i].x != v[i].x) | (v[i].y != v[i].y) | (v[i].z != v[i].z) | (v[i].w != v[i].w));
                    if (MODE == 2) {
                        v[i].x = skr_log2_of_sum1(__fadd_rn(v[i].x, a.shift));
                        v[i].y = skr_log2_of_sum1(__fadd_rn(v[i].y, a.shift));
                        v[i].z = skr_log2_of_sum1(__fadd_rn(v[i].z, a.shift));
                        v[i].w = skr_log2_of_sum1(__fadd_rn(v[i].w, a.shift));
                    }
                    // (HASY is a template flag: a run-time test of a.y here splits the loop body into 16 basic blocks)
                    if (HASY) *reinterpret_cast<float4*>(a.y + (size_t)r * K + c) = v[i];
                }
                // the NaN tests of this group are finished HERE: as one long OR the optimiser pairs cells of different
                // groups into single unordered compares and keeps all 4 * VPL values from before the log2 alive for them
                asm volatile("" : "+v"(nan_seen));
                __builtin_amdgcn_sched_barrier(0);
            }
            any_nan |= nan_seen != 0;
        }
#pragma unroll
        for (int i = 0; i < VPL; i++) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        if (a.row_standardize) {  // statistics in the order pearson.py:35-38 computes them
            const float kf = (float)K;
            const float mean = row_sum(s, 0) / kf;
            s = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; i++) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
            const float m2 = row_sum(s, 1) / kf;
            s = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; i++) {
                const float dx = v[i].x - m2, dy = v[i].y - m2, dz = v[i].z - m2, dw = v[i].w - m2;
                s += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
            const float sd = sqrtf(row_sum(s, 2) / kf);
            const double rsd = 1.0 / (double)sd;  // once per row; the 4 * VPL quotients below cost 3 instructions each
#pragma unroll
            for (int i = 0; i < VPL; i++) {
                v[i].x = div_by_recip(v[i].x, rsd); v[i].y = div_by_recip(v[i].y, rsd);
                v[i].z = div_by_recip(v[i].z, rsd); v[i].w = div_by_recip(v[i].w, rsd);
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // float64 temporaries of four pieces at a time
            }
        }
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; i++)  // explicit FMAs: every instantiation of the kernel rounds this sum identically
            sq = __fmaf_rn(v[i].w, v[i].w, __fmaf_rn(v[i].z, v[i].z, __fmaf_rn(v[i].y, v[i].y, __fmaf_rn(v[i].x, v[i].x, sq))));
        sq = row_sum(sq, 3);
        if (lane == 0 && (RW == 1 || wave == 0)) a.diag[r] = sq / (float)K;
        if (sizeof(T) != 4 && a.row_standardize) {
            float m3 = 0.f, m4 = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; i++) {
                const float z4[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float z2 = z4[j] * z4[j];
                    m3 = fmaf(z2, z4[j], m3);
                    m4 = fmaf(z2, z2, m4);
                }
            }
            m3 = row_sum(m3, 13);
            m4 = row_sum(m4, 14);
            if (row_on_two_levels(sq / (float)K, m3 / (float)K, m4 / (float)K)) coherent = true;
        }
        if (sizeof(T) != 4) {
            // largest |z| of the row (NaN cells do not count): its square is the largest z^2, and times the operand's
            // power-of-two scale it is the largest value the fp16 halves have to hold
            float zmax = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; i++)
                zmax = fmaxf(fmaxf(zmax, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
            zmax = row_max(zmax, 4);
            if (row_needs_fp32(zmax * zmax, (float)K)) outlier = true;
            if (zmax * a.out_scale > 65504.f) overflow = true;
            // Share of the row held by one repeated value — for count data its minimum, the empty bins.  Near-copies
            // of such a row have only positive products, small ones added to a sum that is already large, and the
            // truncating accumulate then loses up to an ulp of the sum per add (tools/margin_probe.py: the bar is
            // reached at 97 %); the contraction restarts its accumulators twice as often for flagged operands.
            float zmin = v[0].x;
#pragma unroll
            for (int i = 0; i < VPL; i++) zmin = fminf(fminf(zmin, fminf(v[i].x, v[i].y)), fminf(v[i].z, v[i].w));
            zmin = -row_max(-zmin, 5);
            float same = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; i++)
                same += (float)((v[i].x == zmin) + (v[i].y == zmin) + (v[i].z == zmin) + (v[i].w == zmin));
            if (row_sum(same, 6) >= 0.85f * (float)K) coherent = true;
            // ... or by any one value, which need not be the minimum (operand_fill_kernel): of the three neighbouring pairs
            // inside each group of four cells a lane holds, 85 % of one value leaves at least 0.45 K equal
            float eq = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; i++) eq += (float)((v[i].x == v[i].y) + (v[i].y == v[i].z) + (v[i].z == v[i].w));
            eq = row_sum(eq, 7);
            if (eq >= 0.45f * (float)K) coherent = true;
            if (X8 && eq >= (float)K * (1.0f / 256.0f)) repeats = true;
            if constexpr (X8) {
                // ... and rows that repeat values WITHOUT neighbours being equal (a periodic row against a shifted copy
                // of itself: the same pair of values meets in every period and the fp8 roundings of that pair add up).
                // What the averaging needs is many DISTINCT pairs of values per pair of rows, and two rows meet in at
                // least as many distinct pairs as either has distinct values; tools/f8_cross_study.py, rows of period D
                // at 4 096 and 16 384 columns alike: D = 3 / 64 / 513 / 1 025 / 2 049 -> 2.4 / 1.1 / 0.8 / 0.6 / 0.4 bars
                // (all distinct: 0.30 / 0.13).  The row's values are hashed into a bitmap of 2 K bits in the LDS; fewer
                // than 2 048 occupied bits (about 2 100 - 2 350 distinct values) send the row back to the three-product split.
                constexpr int LOGB = K == 1024 ? 11 : K == 4096 ? 13 : 15;
                static_assert((int64_t)1 << LOGB == 2 * K, "the bitmap has two bits per column");
                constexpr int WORDS = (int)(2 * K / 32), STEP = RW == 1 ? 64 : 256;
                __shared__ uint32_t seen_bits[(RW == 1 ? 4 : 1) * WORDS];
                uint32_t* bm = seen_bits + (RW == 1 ? wave * WORDS : 0);
                const int first = RW == 1 ? lane : (int)threadIdx.x;
                for (int w = first; w < WORDS; w += STEP) bm[w] = 0u;
                if (RW == 1) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); else __syncthreads();
#pragma unroll
                for (int i = 0; i < VPL; i++) {
                    const float z4[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t h = (__float_as_uint(z4[j]) * 0x9E3779B1u) >> (32 - LOGB);
                        __hip_atomic_fetch_or(&bm[h >> 5], 1u << (h & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                if (RW == 1) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); else __syncthreads();
                float occ = 0.f;
                for (int w = first; w < WORDS; w += STEP) occ += (float)__popc(bm[w]);
                if (row_sum(occ, 8) < 2048.f) repeats = true;
            }
        }
        // X8: sums over the row of what the fp8 copies lose (decoded back: dh = hi - 128 h8, dl = lo - l8 / 16) and of lo, for the
        // routing rule on the MEANS below (the copies themselves: mean(128 h8) = mean(hi) - mean(dh), and mean(hi) = -mean(lo)
        // in a standardised row; mean(l8 / 16) = mean(lo) - mean(dl))
        float x8_dh = 0.f, x8_lo = 0.f, x8_dl = 0.f, x8_self = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; i++) {
            const int64_t c = (i * RW + piece0) * 256 + lane * 4;
            if (sizeof(T) == 4) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + (size_t)r * K + c) = v[i];
            } else {
                const float z[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
                vec4h<T> hi, lo;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float zs = z[j] * a.out_scale;
                    // X8: hi rounded to NEAREST — the hashed direction exists for rows of one repeated value, which never keep
                    // this layout, and it doubles |lo| (rms 0.58 instead of 0.29 ulp), i.e. the cross term the fp8 copies carry
                    const T hh = X8 ? (T)zs : split_hi_flip<T>(zs, (dir[(i * 4 + j) >> 5] << (31 - ((i * 4 + j) & 31))) & 0x80000000u);
                    hi[j] = hh;
                    lo[j] = (T)(zs - (float)hh);
                }
                if constexpr (X8) {
                    // line 2d: 64 fp16 hi halves of columns 64d .. 64d+63; line 2d+1: their fp8 copies hi / 128, then lo x 16
                    char* line = reinterpret_cast<char*>(a.out) + ((size_t)r * a.kt + 2 * (size_t)(c >> 6)) * 128;
                    const int j = (int)(c & 63);
                    *reinterpret_cast<vec4h<T>*>(line + 2 * j) = hi;
                    int h8 = 0, l8 = 0;
                    h8 = __builtin_amdgcn_cvt_pk_fp8_f32((float)hi[0] * 0x1.0p-7f, (float)hi[1] * 0x1.0p-7f, h8, false);
                    h8 = __builtin_amdgcn_cvt_pk_fp8_f32((float)hi[2] * 0x1.0p-7f, (float)hi[3] * 0x1.0p-7f, h8, true);
                    l8 = __builtin_amdgcn_cvt_pk_fp8_f32((float)lo[0] * 16.f, (float)lo[1] * 16.f, l8, false);
                    l8 = __builtin_amdgcn_cvt_pk_fp8_f32((float)lo[2] * 16.f, (float)lo[3] * 16.f, l8, true);
                    *reinterpret_cast<int*>(line + 128 + j) = h8;
                    *reinterpret_cast<int*>(line + 192 + j) = l8;
                    const float fh[4] = {__builtin_amdgcn_cvt_f32_fp8(h8, 0), __builtin_amdgcn_cvt_f32_fp8(h8, 1),
                                         __builtin_amdgcn_cvt_f32_fp8(h8, 2), __builtin_amdgcn_cvt_f32_fp8(h8, 3)};
                    const float fl[4] = {__builtin_amdgcn_cvt_f32_fp8(l8, 0), __builtin_amdgcn_cvt_f32_fp8(l8, 1),
                                         __builtin_amdgcn_cvt_f32_fp8(l8, 2), __builtin_amdgcn_cvt_f32_fp8(l8, 3)};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float h = (float)hi[q], l = (float)lo[q];
                        const float dh = __fmaf_rn(fh[q], -128.f, h), dl = __fmaf_rn(fl[q], -0.0625f, l);
                        x8_dh += dh;
                        x8_lo += l;
                        x8_dl += dl;
                        // what the fp8 cross term of this row WITH ITSELF is off by: hi lo - (hi - dh)(lo - dl)
                        x8_self += __fmaf_rn(h, dl, dh * (l - dl));
                    }
                } else {
                    T* dst = reinterpret_cast<T*>(a.out) + ((size_t)r * a.kt + (c >> 5)) * 64 + (c & 31);
                    *reinterpret_cast<vec4h<T>*>(dst) = hi;
                    *reinterpret_cast<vec4h<T>*>(dst + 32) = lo;
                }
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (X8) {
            // Tightly clustered values (a few levels with a jitter below the fp16 spacing: every cell of a level gets the
            // same hi, the same fp8 copy of it, and a lo that is the level's offset instead of a pseudo-random remainder):
            // then neither hi - 128 h8 nor lo averages out over a row, and the cross term of two such rows is off by
            // K x mean(hi - 128 h8) x mean(lo) (tools/f8_cross_study.py: 4 levels with a jitter of 1e-4: 3 bars, every value
            // distinct).  The row means go into four maxima over the operand; skr_operand_fill turns their products into
            // a bound on that error and routes the operand back above 0.6 of the bar.  NaN rows do not count.
            const float inv = 1.0f / (float)K;
            // ... and rows whose levels are ALIGNED (the same level in a column for every row: near-copies of each other)
            // meet in the same pairs everywhere — their cross terms are off by what the row's own is, m[3]
            // (tests/fuzz_pearson.py, gen_case_layout kind 3, found it: r = 1.0000176 at 16 384 columns).
            const float m[4] = {row_sum(x8_dh, 9) * inv, row_sum(x8_lo, 10) * inv, row_sum(x8_dl, 11) * inv, row_sum(x8_self, 12) * inv};
            if (lane == 0 && (RW == 1 || wave == 0)) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    // a load first: 4 atomics per row on four words of one cache line serialise the whole launch (2.3 instead
                    // of 0.46 ms at 50 000 rows); a running maximum is raised O(log rows) times
                    const uint32_t bits = __float_as_uint(fabsf(m[q]));
                    if (m[q] == m[q] && bits > __hip_atomic_load(&a.flags[24 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        atomicMax(&a.flags[24 + q], bits);
                }
            }
        }
    }
    if (any_nan) atomicOr(&a.flags[1], 1u);
    if (overflow) atomicOr(&a.flags[3], 1u);
    if (outlier) atomicOr(&a.flags[4], 1u);
    if (coherent) atomicOr(&a.flags[5], 1u);
    if (repeats) atomicOr(&a.flags[6], 1u);
}

// r[i, i] of a self-comparison = <z_i, z_i> / K.  The contraction adds 4 096 squares into one float32
// accumulator per cell in k order (as BLAS does in the reference, whose own diagonal is 4-6e-6 off
// on rows with few distinct values); the tree sum taken while the operand was filled is accurate
// to ~1e-7, so the diagonal is written from it after the contraction.
__global__ __launch_bounds__(256) void patch_diag_kernel(float* __restrict__ C, int64_t ldc, const float* __restrict__ diag,
                                                         int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) C[(size_t)i * ldc + i] = diag[i];
}

int vec_kind(const skr_mat* v, int64_t cols, const char* what, int* kind) {
    *kind = 0;
    if (!v) return SKR_OK;
    SKR_REQUIRE(v->rows * v->cols == cols, "%s vector has %lld entries, matrix has %lld columns", what,
                (long long)(v->rows * v->cols), (long long)cols);
    SKR_REQUIRE(v->dtype == SKR_F32 || v->dtype == SKR_F64, "%s vector must be float32 or float64", what);
    *kind = v->dtype == SKR_F32 ? 1 : 2;
    return SKR_OK;
}

int check_pair(const skr_ctx* ctx, const skr_mat* a, const skr_mat* b) {
    SKR_REQUIRE(ctx && a && b, "NULL argument");
    SKR_REQUIRE(a->ctx == ctx && b->ctx == ctx, "matrix belongs to a different ctx");
    SKR_REQUIRE(a->dtype == b->dtype && (a->dtype == SKR_F32 || a->dtype == SKR_F64),
                "operands must both be float32 or both float64");
    if (a->cols != b->cols)
        return skr_set_error(SKR_ERR_INVALID, "shapes (%lld,%lld) and (%lld,%lld) not aligned: %lld (dim 1) != %lld (dim 1)",
                             (long long)a->rows, (long long)a->cols, (long long)b->rows, (long long)b->cols,
                             (long long)a->cols, (long long)b->cols);
    return SKR_OK;
}

__global__ void recip64_kernel(const float* __restrict__ v, double* __restrict__ r, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) r[i] = 1.0 / (double)v[i];
}

bool is_f32_precision(int p) {
    return p == SKR_PREC_FP32 || p == SKR_PREC_BF16X3 || p == SKR_PREC_F16X3 || p == SKR_PREC_F16F8;
}

// launches the fill kernel that suits the row width and the operand's storage kind
int launch_fill(skr_ctx* ctx, const skr_operand* op, const FillArgs& a_in) {
    const FillArgs& a = a_in;
    const size_t row_floats = (size_t)((a.cols + 3) & ~(int64_t)3);
    const bool wide = row_floats * 4 > 150 * 1024;  // k >= 8: the row does not fit the LDS
    const int waves = (int)std::max<size_t>(1, std::min<size_t>(4, (64 * 1024) / (row_floats * 4)));
    const size_t lds = row_floats * 4 * waves + sizeof(NpScratch) * waves;  // row slices + the numpy-order summations' scratch
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (150 * 1024) / lds));
    const int64_t want = (a.rows + waves - 1) / waves;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cu * per_cu));
    // 4^6 / 4^5 columns with the usual vectors (float32 mean/std computed on the device, or none):
    // the row fits the register file
    int reg_mode = -1;
    if (a.ck == 0 && a.sk == 0 && !a.post && !a.y) reg_mode = 0;
    else if (a.ck == 1 && a.sk == 1) reg_mode = a.post ? 2 : 1;
    if ((a.cols == 16384 || a.cols == 4096 || a.cols == 1024) && reg_mode >= 0) {
        FillArgs a = a_in;
        if (reg_mode >= 1) {  // float64 reciprocals of the scale vector: a quotient then costs 3 instructions (div_by_recip)
            if (ctx->d_recip_len < (size_t)a.cols) {
                if (ctx->d_recip) SKR_HIP(hipFree(ctx->d_recip));
                ctx->d_recip = nullptr;
                ctx->d_recip_len = 0;
                SKR_HIP(hipMalloc((void**)&ctx->d_recip, (size_t)a.cols * sizeof(double)));
                ctx->d_recip_len = (size_t)a.cols;
            }
            hipLaunchKernelGGL(recip64_kernel, dim3((unsigned)((a.cols + 255) / 256)), dim3(256), 0, ctx->stream,
                               reinterpret_cast<const float*>(a.scale), ctx->d_recip, a.cols);
            a.scale_recip = ctx->d_recip;
        }
        const int64_t rows_per_wg = a.cols == 16384 ? 1 : 4;
        // One row per wave (k <= 6: four rows per workgroup) or per workgroup (k = 7), the workgroups dispatched in order:
        // the rows being read and written form a compact front (0.69 -> 0.63 ms at 50 000 x 4 096 against a persistent
        // grid of 5 workgroups per CU; the same effect as in the counting kernel), and with exactly one row per wave the
        // kernel has no row loop for the compiler to pipeline across — a persistent k = 7 variant prefetched the next
        // row into registers this row needs and spilled (2.27 instead of 1.17 ms at 30 000 x 16 384).
        const int64_t all_wgs = (a.rows + rows_per_wg - 1) / rows_per_wg;
        SKR_REQUIRE(all_wgs <= 0x7fffffff, "too many rows for one fill launch (%lld)", (long long)a.rows);
        const unsigned rgrid = (unsigned)std::max<int64_t>(1, all_wgs);
        SkrProfScope prof(ctx, "operand_fill");
#define LAUNCH_REG2(T, V, RW, X)                                                                                           \
    do {                                                                                                                   \
        if (reg_mode == 0) hipLaunchKernelGGL((operand_fill_reg_kernel<T, V, 0, RW, false, X>), dim3(rgrid), dim3(256), 0, ctx->stream, a);      \
        else if (reg_mode == 1 && a.y) hipLaunchKernelGGL((operand_fill_reg_kernel<T, V, 1, RW, true, X>), dim3(rgrid), dim3(256), 0, ctx->stream, a); \
        else if (reg_mode == 1) hipLaunchKernelGGL((operand_fill_reg_kernel<T, V, 1, RW, false, X>), dim3(rgrid), dim3(256), 0, ctx->stream, a); \
        else if (a.y) hipLaunchKernelGGL((operand_fill_reg_kernel<T, V, 2, RW, true, X>), dim3(rgrid), dim3(256), 0, ctx->stream, a);             \
        else hipLaunchKernelGGL((operand_fill_reg_kernel<T, V, 2, RW, false, X>), dim3(rgrid), dim3(256), 0, ctx->stream, a);                    \
    } while (0)
#define LAUNCH_REG(T, X)                                  \
    do {                                                  \
        if (a.cols == 16384) LAUNCH_REG2(T, 16, 4, X);    \
        else if (a.cols == 4096) LAUNCH_REG2(T, 16, 1, X); \
        else LAUNCH_REG2(T, 4, 1, X);                      \
    } while (0)
        if (op->kind == 0) LAUNCH_REG(float, false);
        else if (op->kind == 1) LAUNCH_REG(__bf16, false);
        else if (op->kind == 3) LAUNCH_REG(_Float16, true);
        else LAUNCH_REG(_Float16, false);
#undef LAUNCH_REG
#undef LAUNCH_REG2
        SKR_HIP(hipGetLastError());
    } else if (reg_mode >= 0 && op->kind != 3 && a.cols > 8192 && a.cols <= 16384) {
        // round 5: ANY width of 8 193 .. 16 384 columns but 4^7 (5^6, 10^4, 22^3 ...) — the row in the registers of a
        // sixteen-wave workgroup (operand_fill_rowreg_kernel; multiples of 8 too: the block kernel below takes 3.1 ms where
        // this one takes 1.8 on 50 000 x 15 632); widths below stay with the wave-per-row kernel and its numpy-ordered
        // sums, everything else with the block kernel (a 65 536-cell row in registers, 64 per thread, spills: measured)
        FillArgs a = a_in;
        if (reg_mode >= 1) {  // float64 reciprocals of the scale vector (as for the register kernels above)
            if (ctx->d_recip_len < (size_t)a.cols) {
                if (ctx->d_recip) SKR_HIP(hipFree(ctx->d_recip));
                ctx->d_recip = nullptr;
                ctx->d_recip_len = 0;
                SKR_HIP(hipMalloc((void**)&ctx->d_recip, (size_t)a.cols * sizeof(double)));
                ctx->d_recip_len = (size_t)a.cols;
            }
            hipLaunchKernelGGL(recip64_kernel, dim3((unsigned)((a.cols + 255) / 256)), dim3(256), 0, ctx->stream,
                               reinterpret_cast<const float*>(a.scale), ctx->d_recip, a.cols);
            a.scale_recip = ctx->d_recip;
        }
        SkrProfScope prof(ctx, "operand_fill");
        // one sixteen-wave workgroup per CU (128 registers a thread) that prefetches its next row; two eight-wave workgroups
        // per CU without prefetch were slower (4.18 against 3.47 ms: profiles/r5_generic_width_arms.log)
        const unsigned rgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(a.rows, (int64_t)ctx->num_cu));
#define LAUNCH_ROWREG(T)                                                                                                    \
    do {                                                                                                                    \
        if (reg_mode == 0) hipLaunchKernelGGL((operand_fill_rowreg_kernel<T, 4, 0, false, 1024>), dim3(rgrid), dim3(1024), 0, ctx->stream, a);      \
        else if (reg_mode == 1 && a.y) hipLaunchKernelGGL((operand_fill_rowreg_kernel<T, 4, 1, true, 1024>), dim3(rgrid), dim3(1024), 0, ctx->stream, a); \
        else if (reg_mode == 1) hipLaunchKernelGGL((operand_fill_rowreg_kernel<T, 4, 1, false, 1024>), dim3(rgrid), dim3(1024), 0, ctx->stream, a); \
        else if (a.y) hipLaunchKernelGGL((operand_fill_rowreg_kernel<T, 4, 2, true, 1024>), dim3(rgrid), dim3(1024), 0, ctx->stream, a);             \
        else hipLaunchKernelGGL((operand_fill_rowreg_kernel<T, 4, 2, false, 1024>), dim3(rgrid), dim3(1024), 0, ctx->stream, a);                    \
    } while (0)
        if (op->kind == 0) LAUNCH_ROWREG(float);
        else if (op->kind == 1) LAUNCH_ROWREG(__bf16);
        else LAUNCH_ROWREG(_Float16);
#undef LAUNCH_ROWREG
        SKR_HIP(hipGetLastError());
    } else if (reg_mode >= 0 && op->kind != 3 && a.cols == 65536) {
        // round 5: 4^8 columns — the same kernel with sixteen pieces (64 cells) a thread and no second row in flight: ONE read
        // of the 256 KB row instead of the block kernel's four (its passes re-read the row through an L2 that 256 rows in
        // flight overflow), 8.84 -> 3.79 ms per 20 000 rows in the pipeline form, 9.3 -> 2.3 bare.  Every piece of every
        // wave is whole at this width; other widths above 16 384 would need the mixed-piece bodies for up to sixteen
        // pieces and stay with the block kernel.
        FillArgs a = a_in;
        if (reg_mode >= 1) {
            if (ctx->d_recip_len < (size_t)a.cols) {
                if (ctx->d_recip) SKR_HIP(hipFree(ctx->d_recip));
                ctx->d_recip = nullptr;
                ctx->d_recip_len = 0;
                SKR_HIP(hipMalloc((void**)&ctx->d_recip, (size_t)a.cols * sizeof(double)));
                ctx->d_recip_len = (size_t)a.cols;
            }
            hipLaunchKernelGGL(recip64_kernel, dim3((unsigned)((a.cols + 255) / 256)), dim3(256), 0, ctx->stream,
                               reinterpret_cast<const float*>(a.scale), ctx->d_recip, a.cols);
            a.scale_recip = ctx->d_recip;
        }
        SkrProfScope prof(ctx, "operand_fill");
        const unsigned rgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(a.rows, (int64_t)ctx->num_cu));
#define LAUNCH_ROWREG16(T)                                                                                                  \
    do {                                                                                                                    \
        if (reg_mode == 0) hipLaunchKernelGGL((operand_fill_rowreg_kernel<T, 16, 0, false, 1024>), dim3(rgrid), dim3(1024), 0, ctx->stream, a);      \
        else if (reg_mode == 1 && a.y) hipLaunchKernelGGL((operand_fill_rowreg_kernel<T, 16, 1, true, 1024>), dim3(rgrid), dim3(1024), 0, ctx->stream, a); \
        else if (reg_mode == 1) hipLaunchKernelGGL((operand_fill_rowreg_kernel<T, 16, 1, false, 1024>), dim3(rgrid), dim3(1024), 0, ctx->stream, a); \
        else if (a.y) hipLaunchKernelGGL((operand_fill_rowreg_kernel<T, 16, 2, true, 1024>), dim3(rgrid), dim3(1024), 0, ctx->stream, a);             \
        else hipLaunchKernelGGL((operand_fill_rowreg_kernel<T, 16, 2, false, 1024>), dim3(rgrid), dim3(1024), 0, ctx->stream, a);                    \
    } while (0)
        if (op->kind == 0) LAUNCH_ROWREG16(float);
        else if (op->kind == 1) LAUNCH_ROWREG16(__bf16);
        else LAUNCH_ROWREG16(_Float16);
#undef LAUNCH_ROWREG16
        SKR_HIP(hipGetLastError());
    } else if (row_floats * 4 >= 32 * 1024) {  // k >= 7: one workgroup per row
        SkrProfScope prof(ctx, "operand_fill");
        const size_t blds = wide ? 0 : row_floats * 4;
        const int64_t bper_cu = wide ? 8 : std::max<int64_t>(1, (int64_t)((150 * 1024) / blds));
        const unsigned wgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(a.rows, (int64_t)ctx->num_cu * bper_cu));
        // sixteen-wave workgroups: two per CU at most (2 048 threads), one when the row takes more than half the LDS
        const unsigned wgrid16 = (unsigned)std::max<int64_t>(1, std::min<int64_t>(a.rows, (int64_t)ctx->num_cu * std::min<int64_t>(2, bper_cu)));
#define LAUNCH_BLOCK(T)                                                                                           \
    do {                                                                                                          \
        if (wide) {                                                                                               \
            hipLaunchKernelGGL((operand_fill_block_kernel<T, false, 1024>), dim3(wgrid16), dim3(1024), 0, ctx->stream, a); \
        } else {                                                                                                  \
            SKR_TRY(skr_kernel_lds(ctx, reinterpret_cast<const void*>(operand_fill_block_kernel<T, true>), blds)); \
            hipLaunchKernelGGL((operand_fill_block_kernel<T, true>), dim3(wgrid), dim3(256), blds, ctx->stream, a); \
        }                                                                                                         \
    } while (0)
        if (op->kind == 0) LAUNCH_BLOCK(float);
        else if (op->kind == 1) LAUNCH_BLOCK(__bf16);
        else LAUNCH_BLOCK(_Float16);
#undef LAUNCH_BLOCK
        SKR_HIP(hipGetLastError());
    } else {
        FillArgs a = a_in;
        a.np_plan = nullptr;
        if (a.row_standardize && a.cols <= kNpExactMaxCols) {
            // rows summed in numpy's pairwise order: the recursion for this width, unrolled here once and kept on the device
            if (!ctx->d_np_plan) SKR_HIP(hipMalloc(&ctx->d_np_plan, sizeof(NpPlan)));
            if (ctx->np_plan_cols != a.cols) {
                void* pin = nullptr;
                SKR_TRY(skr_ctx_pinned(ctx, sizeof(NpPlan), &pin));
                np_plan_build(reinterpret_cast<NpPlan*>(pin), (int)a.cols);
                SKR_HIP(hipMemcpyAsync(ctx->d_np_plan, pin, sizeof(NpPlan), hipMemcpyHostToDevice, ctx->stream));
                SKR_TRY(skr_ctx_pinned_used(ctx));
                ctx->np_plan_cols = a.cols;
            }
            a.np_plan = reinterpret_cast<const NpPlan*>(ctx->d_np_plan);
        }
        SkrProfScope prof(ctx, "operand_fill");
#define LAUNCH(T)                                                                                         \
    do {                                                                                                      \
        SKR_TRY(skr_kernel_lds(ctx, reinterpret_cast<const void*>(operand_fill_kernel<T>), lds));             \
        hipLaunchKernelGGL(operand_fill_kernel<T>, dim3(grid), dim3(64 * waves), lds, ctx->stream, a);               \
    } while (0)
        if (op->kind == 0) LAUNCH(float);
        else if (op->kind == 1) LAUNCH(__bf16);
        else LAUNCH(_Float16);
#undef LAUNCH
        SKR_HIP(hipGetLastError());
    }
    return SKR_OK;
}

}  // namespace

extern "C" int skr_operand_create(skr_ctx* ctx, int64_t rows, int64_t cols, int precision, skr_operand** out) {
    SKR_REQUIRE(ctx && out, "NULL argument");
    *out = nullptr;
    SKR_REQUIRE(rows >= 0 && cols > 0, "bad operand shape [%lld, %lld]", (long long)rows, (long long)cols);
    SKR_REQUIRE(is_f32_precision(precision), "operands exist for float32 precisions only (got %d)", precision);
    SKR_TRY(skr_activate(ctx));
    skr_operand* op = new skr_operand();
    op->ctx = ctx;
    op->rows = rows;
    op->cols = cols;
    op->kt = (cols + 31) / 32;
    op->precision = precision;
    op->created_precision = precision;
    // bf16 halves: the split error averages out like 1/sqrt(K): at K >= 1024 (k >= 5) it is inside the parity bar
    // (|dr| <= 2e-6 + 1e-5 |r|); below that the exact-product fp32 MFMA is used.  fp16 halves keep 22 bits of z, so
    // the dropped lo*lo term is bounded by 2^-21 sum|z_i z_j| / K <= 5e-7 whatever K is: the split kernel is used
    // from 64 columns (k = 3) up, where it is faster than the fp32 one (k = 4, 50 000 rows: 6.9 -> 3.2 ms).
    // One float32 accumulator per cell over the whole of K drifts past the bar at K = 65 536 (1e-5 measured in round 1):
    // the split contraction restarts its accumulators every 4 096 columns (pearson_bf16.hip: launch16).
    // SKR_PREC_F16F8 (opt-in): the H / X line layout exists for the three register-resident widths; any other shape is
    // served by the split-fp16 operand it degrades to
    // (4 096 and 16 384 columns: the fp8 roundings average out like 1 / sqrt(K) — measured 0.5 and 0.25 of the bar; at 1 024
    // columns they would not fit it)
    if (precision == SKR_PREC_F16F8 && !(cols == 4096 || cols == 16384)) op->precision = precision = SKR_PREC_F16X3;
    const int64_t split_min = precision == SKR_PREC_F16X3 ? 64 : 1024;
    // Round 4: split-fp16 up to 262 144 columns (k = 8 and 9; SEEKR_SPLIT_MAX_COLS=16384 restores the fp32 kernel for the
    // A/B) — the accumulators restart every 4 096 columns and the chunks' partial sums are added in float32, so the drift of
    // ONE accumulator over the whole of K that sent these shapes to the fp32 kernel in round 1 no longer applies (20 000 x
    // 65 536: the contraction 209 -> 60 ms, worst cell 0.04 of the bar either way; K = 262 144: strict 0.11-0.26, the fp32
    // kernel's own values); bf16 halves and wider rows (k >= 10) keep the fp32 kernel.
    const int64_t split_max = precision == SKR_PREC_F16X3 ? std::max<int64_t>(16384, ctx->knobs.split_max_cols) : 16384;
    if (precision == SKR_PREC_FP32 || cols < split_min || cols > split_max) op->kind = 0;
    else op->kind = precision == SKR_PREC_F16F8 ? 3 : (precision == SKR_PREC_F16X3 ? 2 : 1);
    // fp16 halves: rows are stored times a power of two chosen so that sqrt(K) — the largest value a
    // row-standardised row can hold — lands just below 2^15.  The lo half of a small z then stays a
    // normal fp16 number (without the scale it is subnormal for |z| < 0.125 and z keeps only 3e-8
    // absolute precision); the contraction divides by K s^2.  A function of K alone, so shards
    // exchanged between GPUs agree on it.
    if (op->kind == 2 || op->kind == 3) op->scale = std::exp2f(std::floor(std::log2(32768.0f / std::sqrt((float)cols))));
    const size_t body = ((size_t)rows * op->row_bytes() + 255) & ~(size_t)255;
    const size_t bytes = body + std::max<size_t>((size_t)rows * sizeof(float), 16);  // rows, then diag[rows]
    hipError_t e = hipMalloc(&op->data, bytes);
    if (e != hipSuccess) {
        delete op;
        return skr_set_error(SKR_ERR_NOMEM, "hipMalloc(%zu bytes) for a %lld x %lld operand failed: %s", bytes,
                             (long long)rows, (long long)cols, hipGetErrorString(e));
    }
    op->diag = reinterpret_cast<float*>((char*)op->data + body);
    op->x8_root = op->data;
    *out = op;
    return SKR_OK;
}

extern "C" int skr_operand_free(skr_operand* op) {
    if (!op) return SKR_OK;
    if (op->owner) {
        (void)hipSetDevice(op->ctx->device);
        (void)hipStreamSynchronize(op->ctx->stream);
        (void)hipStreamSynchronize(op->ctx->comm_stream);
        if (op->data) (void)hipFree(op->data);
    }
    delete op;
    return SKR_OK;
}

extern "C" int skr_operand_view(const skr_operand* parent, int64_t row0, int64_t nrows, skr_operand** out) {
    SKR_REQUIRE(parent && out, "NULL argument");
    *out = nullptr;
    SKR_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= parent->rows, "view rows [%lld, %lld) outside 0..%lld",
                (long long)row0, (long long)(row0 + nrows), (long long)parent->rows);
    skr_operand* v = new skr_operand(*parent);
    v->rows = nrows;
    v->owner = false;
    v->data = (char*)parent->data + (size_t)row0 * parent->row_bytes();
    v->diag = parent->diag ? parent->diag + row0 : nullptr;
    *out = v;
    return SKR_OK;
}

extern "C" int skr_operand_as_mat(skr_operand* op, skr_mat** view) {
    SKR_REQUIRE(op && view, "NULL argument");
    skr_mat* m = new skr_mat();
    m->ctx = op->ctx;
    m->rows = op->rows;
    m->cols = op->kt * 32;  // float-sized words per row in either layout
    m->dtype = SKR_F32;
    m->owner = false;
    m->data = op->data;
    *view = m;
    return SKR_OK;
}

extern "C" int skr_operand_fill(skr_ctx* ctx, const skr_mat* x, const skr_mat* center, const skr_mat* scale, int post,
                                float shift, skr_mat* y, int row_standardize, skr_operand* op, int* has_nan) {
    SKR_REQUIRE(ctx && x && op, "NULL argument");
    SKR_REQUIRE(x->ctx == ctx && op->ctx == ctx, "handle belongs to a different ctx");
    SKR_REQUIRE(x->dtype == SKR_F32, "operands are filled from float32 matrices");
    SKR_REQUIRE(x->rows == op->rows && x->cols == op->cols, "operand is [%lld, %lld], matrix is [%lld, %lld]",
                (long long)op->rows, (long long)op->cols, (long long)x->rows, (long long)x->cols);
    if (y) SKR_REQUIRE(y->ctx == ctx && y->dtype == SKR_F32 && y->rows == x->rows && y->cols == x->cols, "bad y");
    FillArgs a;
    a.np_plan = nullptr;
    a.x = (const float*)x->data;
    a.rows = x->rows;
    a.cols = x->cols;
    a.kt = op->kt;
    SKR_TRY(vec_kind(center, x->cols, "center", &a.ck));
    SKR_TRY(vec_kind(scale, x->cols, "scale", &a.sk));
    a.center = center ? center->data : nullptr;
    a.scale = scale ? scale->data : nullptr;
    a.scale_recip = nullptr;
    a.post = post != 0;
    a.shift = shift;
    a.row_standardize = row_standardize != 0;
    a.y = y ? (float*)y->data : nullptr;
    a.out = op->data;
    a.diag = op->diag;
    a.out_scale = op->scale;
    a.flags = ctx->d_flags;
    SKR_TRY(skr_activate(ctx));
    if (has_nan) *has_nan = 0;
    if (x->rows == 0) return SKR_OK;
    // flags: [1] NaN seen, [3] fp16 range exceeded, [4] a row needs more dynamic range than one float32
    // accumulator per cell has, [5] a row is mostly one repeated value ([2] belongs to the counting kernel)
    SKR_HIP(hipMemsetAsync(ctx->d_flags + 1, 0, 4, ctx->stream));
    SKR_HIP(hipMemsetAsync(ctx->d_flags + 3, 0, 16, ctx->stream));  // [6]: neighbouring cells repeat each other (f16f8 only)
    SKR_HIP(hipMemsetAsync(ctx->d_flags + 24, 0, 16, ctx->stream));  // [24..27]: f16f8's maxima of the row means (float bits)
    op->coherent = false;
    // an operand created for the opt-in f16f8 layout whose EARLIER contents were routed back to the three-product split
    // tries its own layout again on new data (the downgrade used to stick for the life of the operand)
    if (op->owner && op->x8_routed_back && op->created_precision == SKR_PREC_F16F8 && op->kind == 2) {
        op->kind = 3;
        op->precision = SKR_PREC_F16F8;
    }
    op->x8_routed_back = false;
    if (op->kind == 3) {
        // the H / X line layout is written by the register kernels alone: float32 vectors computed on the device (or none),
        // rows standardised here (their values are then bounded by sqrt(K): the fp8 copies cannot overflow)
        const bool reg_path = (a.ck == 0 && a.sk == 0 && !a.post && !a.y) || (a.ck == 1 && a.sk == 1);
        if (!reg_path || !a.row_standardize) {
            op->kind = 2;
            op->precision = SKR_PREC_F16X3;
            op->x8_routed_back = true;
        }
    }
    SKR_TRY(launch_fill(ctx, op, a));
    op->diag_valid = true;
    // values are only bounded by sqrt(K) when the rows were standardised here: check the fp16 range otherwise
    const bool check_range = op->kind == 2 && !row_standardize;
    const bool was_x8 = op->kind == 3;
    const bool split = op->kind != 0;
    if (has_nan || check_range || split) {
        SKR_HIP(hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 28 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        SKR_HIP(hipStreamSynchronize(ctx->stream));
        if (has_nan) *has_nan = ctx->h_flags[1] != 0;
        op->coherent = split && ctx->h_flags[5] != 0;
        bool x8_means = false;
        if (was_x8) {
            // error of a cell of r from the row means alone (operand_fill_reg_kernel, X8): 2 (mean(hi - 128 h8) mean(l8 / 16) +
            // mean(128 h8) mean(lo - l8 / 16)) / scale^2, the largest means of the operand standing in for every pair of rows
            float mx[4];  // largest |row mean| of dh, lo, dl and of the row's own cross-term error
            for (int q = 0; q < 4; q++) {
                const uint32_t bits = ctx->h_flags[24 + q];
                mx[q] = *reinterpret_cast<const float*>(&bits);
            }
            const double s2 = (double)op->scale * op->scale;
            // |mean(l8 / 16)| <= |mean lo| + |mean dl|, |mean(128 h8)| <= |mean dh| + |mean lo|; kept with the operand: rows
            // of ANOTHER operand meet these under the same rule (skr_x8_pair_bound: skr_pearson_gemm_op, match_layouts)
            for (int q = 0; q < 3; q++) op->x8_stat[q] = mx[q];
            const double bound = skr_x8_pair_bound(op, op);
            // ... and the largest error of a row's cross term with itself (aligned levels): 3e-6 is a quarter of the bar at
            // r = 1, where such rows meet, and two to three times what unstructured rows reach (K = 4 096: <= 1.8e-6 in 6 000 rows)
            const double self_err = 2.0 * (double)mx[3] / s2;
            x8_means = bound > kX8MeansLimit || self_err > 3e-6;
        }
        if (split && ctx->h_flags[4] != 0) {
            // a row is dominated by so few columns that the split contraction would drop the others
            // (see row_needs_fp32): same storage, float32 layout, and the fp32 kernel from here on.  The
            // normalised counts, if they were asked for, are already in y: refill from them as they are.
            op->kind = 0;
            op->scale = 1.f;
            FillArgs b = a;
            b.out_scale = 1.f;
            if (a.y) {
                b.x = a.y;
                b.y = nullptr;
                b.ck = b.sk = 0;
                b.center = b.scale = nullptr;
                b.post = 0;
            }
            SKR_TRY(launch_fill(ctx, op, b));
            return SKR_OK;
        }
        if (was_x8 && (ctx->h_flags[5] != 0 || ctx->h_flags[6] != 0 || x8_means)) {
            // rows that are mostly one repeated value, whose neighbouring cells repeat each other (raw counts, 0/1 rows), that
            // hold too few distinct values, or whose fp8 roundings do not average out over a row (tight clusters):
            // the fp8 copies of a repeated value carry the same rounding everywhere — the three-product split serves them
            if (!op->owner)
                return skr_set_error(SKR_ERR_UNSUPPORTED, "these rows need the three-product split, and a VIEW cannot change the layout "
                                                          "of the operand it belongs to: fill whole operands with SKR_PREC_F16F8");
            op->kind = 2;
            op->precision = SKR_PREC_F16X3;
            op->x8_routed_back = true;
            FillArgs b = a;
            if (a.y) {  // the normalised counts are already in y: refill from them as they are
                b.x = a.y;
                b.y = nullptr;
                b.ck = b.sk = 0;
                b.center = b.scale = nullptr;
                b.post = 0;
            }
            SKR_HIP(hipMemsetAsync(ctx->d_flags + 5, 0, 4, ctx->stream));
            SKR_TRY(launch_fill(ctx, op, b));
            SKR_HIP(hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            SKR_HIP(hipStreamSynchronize(ctx->stream));
            op->coherent = ctx->h_flags[5] != 0;
            return SKR_OK;
        }
        if (check_range && ctx->h_flags[3] != 0)
            return skr_set_error(SKR_ERR_INVALID,
                                 "a value exceeds the split-fp16 operand range (|v| <= %g at %lld columns); "
                                 "use SKR_PREC_FP32 for rows that are not row-standardised",
                                 65504.0 / op->scale, (long long)op->cols);
    }
    return SKR_OK;
}

/* storage kind an operand ended up with: 0 = float32 (fp32 kernel), 1 = bf16 halves, 2 = fp16 halves */
extern "C" int skr_operand_kind(const skr_operand* op, int* kind) {
    SKR_REQUIRE(op && kind, "NULL argument");
    *kind = op->kind;
    return SKR_OK;
}

/* tag a buffer of the same shape with the storage kind (and scale) of `like`: receive buffers must be
 * read the way the sender wrote them */
// Two f16f8 operands that were filled separately (not views of one allocation, not a receive buffer that adopted its
// shard's layout): the means rule of the fill holds within each, not between them — a's rows may lose much in the fp8
// copy of hi where b's have a lo that does not average out (tests/fuzz_pearson.py found r = 0.00342 off by 1.4 bars at
// 16 384 columns).  Refused here; skr_pearson / skr_pearson_gemm refill both as f16x3 instead (match_layouts).
int skr_x8_pair_check(const skr_operand* a, const skr_operand* b) {
    if (a->kind != 3 || b->kind != 3 || a->x8_root == b->x8_root) return SKR_OK;
    if (skr_x8_pair_bound(a, b) <= kX8MeansLimit) return SKR_OK;
    return skr_set_error(SKR_ERR_UNSUPPORTED, "these two f16f8 operands do not go together (the fp8 roundings of one do not "
                                              "average out against the other: bound %.2e of r): fill both with SKR_PREC_F16X3",
                         skr_x8_pair_bound(a, b));
}

/* The library's own verdict on two f16f8 operands whose three row-mean maxima are set (skr_operand_x8_stats): the bound
 * on a cell of r that follows from the means alone, and whether skr_operand_fill's rule allows it — so that a multi-GPU
 * caller who made the maxima global never re-derives the formula.  b == a: the rows of one matrix against each other.
 * Operands in any other layout: bound 0, ok. */
extern "C" int skr_operand_x8_pair_bound(const skr_operand* a, const skr_operand* b, double* bound, int* ok) {
    SKR_REQUIRE(a && b, "NULL argument");
    const double v = a->kind == 3 && b->kind == 3 ? skr_x8_pair_bound(a, b) : 0.0;
    if (bound) *bound = v;
    if (ok) *ok = v <= kX8MeansLimit ? 1 : 0;
    return SKR_OK;
}

extern "C" int skr_operand_adopt_layout(skr_operand* op, const skr_operand* like) {
    SKR_REQUIRE(op && like && op->cols == like->cols, "operands of different widths");
    op->kind = like->kind;
    op->scale = like->scale;
    op->precision = like->precision;
    op->coherent = like->coherent;
    op->x8_routed_back = false;
    op->diag_valid = false;
    for (int q = 0; q < 3; q++) op->x8_stat[q] = like->x8_stat[q];  // a receive buffer: the caller has made these global
    op->x8_root = like->x8_root;
    return SKR_OK;
}

/* The three row-mean maxima of an operand in the opt-in f16f8 layout (kind 3; zeros otherwise): get them, or set them —
 * shards of one matrix on several GPUs are multiplied against each other, so a multi-GPU caller all-reduces (max) the three
 * values, checks skr_x8 bound on the result and sets it on its shard; receive buffers take it over in
 * skr_operand_adopt_layout.  v: float[3]. */
extern "C" int skr_operand_x8_stats(skr_operand* op, int set, float* v) {
    SKR_REQUIRE(op && v, "NULL argument");
    for (int q = 0; q < 3; q++) {
        if (set) op->x8_stat[q] = v[q];
        v[q] = op->x8_stat[q];
    }
    return SKR_OK;
}

/* The "rows are mostly one repeated value" flag of a prepared operand (skr_operand_fill sets it from this rank's
 * rows): get it, or set it — shards multiplied against each other must agree on it, so a multi-GPU caller
 * all-reduces the flag and sets the result on its shard and receive buffers. */
extern "C" int skr_operand_coherent(skr_operand* op, int set, int* value) {
    SKR_REQUIRE(op && value, "NULL argument");
    if (set) op->coherent = *value != 0;
    *value = op->coherent ? 1 : 0;
    return SKR_OK;
}

extern "C" int skr_pearson_gemm_op(skr_ctx* ctx, const skr_operand* a, const skr_operand* b, int symmetric, skr_mat* r,
                                   int64_t row0, int64_t col0) {
    SKR_REQUIRE(ctx && a && b && r, "NULL argument");
    SKR_REQUIRE(a->ctx == ctx && b->ctx == ctx && r->ctx == ctx, "handle belongs to a different ctx");
    SKR_REQUIRE(a->cols == b->cols && a->kind == b->kind && (a->kind == 0 || a->precision == b->precision),
                "operands were prepared for different shapes or precisions");
    SKR_TRY(skr_x8_pair_check(a, b));
    SKR_REQUIRE(r->dtype == SKR_F32, "result matrix must be float32");
    SKR_REQUIRE(row0 >= 0 && col0 >= 0 && row0 + a->rows <= r->rows && col0 + b->rows <= r->cols,
                "result block [%lld+%lld, %lld+%lld] outside [%lld, %lld]", (long long)row0, (long long)a->rows,
                (long long)col0, (long long)b->rows, (long long)r->rows, (long long)r->cols);
    SKR_TRY(skr_activate(ctx));
    const int64_t M = a->rows, N = b->rows, K = a->cols;
    if (M == 0 || N == 0) return SKR_OK;
    float* C = (float*)r->data + (size_t)row0 * r->cols + col0;
    const bool self = a->data == b->data && M == N;
    const bool lower = symmetric == 2;  // a plain block that carries the bits of the mirror of (b, a): pearson_bf16.hip, LOWER
    if (a->kind == 0) {
        // float32 operands: a_i . b_j and b_j . a_i are the same products in the same k order — nothing to swap
        const int64_t Kp = a->kt * 32;
        SKR_TRY(skr_launch_gemm_f32(ctx, (const float*)a->data, (const float*)b->data, C, M, N, Kp, Kp, Kp, r->cols, K,
                                    symmetric == 1 && self && row0 == col0));
    } else {
        SKR_TRY(skr_launch_gemm_split(ctx, a->precision, a->data, b->data, C, M, N, a->kt, r->cols,
                                      (float)K * a->scale * b->scale, lower ? 4 : (symmetric && self ? 1 : 0), nullptr, 0,
                                      a->coherent || b->coherent));
    }
    if (self && !lower && a->diag_valid && a->diag) {
        hipLaunchKernelGGL(patch_diag_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream, C, r->cols,
                           a->diag, M);
        SKR_HIP(hipGetLastError());
    }
    return SKR_OK;
}

/* Rows of a self-comparison, one stripe at a time (a result larger than the HBM; the rows of one GPU of several):
 * r[row0 + i, j] for the rows i of `a` and ALL rows j of `full`, bit for bit what skr_pearson_gemm_op(full, full, 1, ...)
 * writes into rows a_row0 .. a_row0 + a->rows: the block left of the stripe's diagonal block with the products in the
 * mirror's order (symmetric = 2), the diagonal block as a self-comparison of `a` (its diagonal patched from a's own
 * fill), the block right of it plain.  `a`: a view of those rows of `full`, or — several GPUs — the rank's own shard,
 * of which `full` holds the all-gathered copy. */
extern "C" int skr_pearson_gemm_op_rows(skr_ctx* ctx, const skr_operand* a, const skr_operand* full, int64_t a_row0,
                                        skr_mat* r, int64_t row0) {
    SKR_REQUIRE(ctx && a && full && r, "NULL argument");
    SKR_REQUIRE(a_row0 >= 0 && a_row0 + a->rows <= full->rows, "rows [%lld, %lld) outside the %lld rows of the full operand",
                (long long)a_row0, (long long)(a_row0 + a->rows), (long long)full->rows);
    const int64_t M = a->rows, N = full->rows;
    if (a_row0 > 0) {
        skr_operand left = *full;
        left.rows = a_row0;
        left.owner = false;
        SKR_TRY(skr_pearson_gemm_op(ctx, a, &left, 2, r, row0, 0));
    }
    if (M > 0) {
        // the diagonal block is mirrored inside r: SELF needs its square at the same offset in rows and columns of C
        skr_mat sq = *r;
        sq.owner = false;
        SKR_REQUIRE(row0 >= 0 && row0 + M <= r->rows && a_row0 + M <= r->cols, "stripe [%lld+%lld] outside the result",
                    (long long)row0, (long long)M);
        sq.data = (char*)r->data + ((size_t)row0 * r->cols + (size_t)a_row0) * sizeof(float);
        sq.rows = M;  // [M, ld = r->cols]: the block starts at its own (0, 0)
        SKR_TRY(skr_pearson_gemm_op(ctx, a, a, 1, &sq, 0, 0));
    }
    if (a_row0 + M < N) {
        skr_operand right = *full;
        right.rows = N - a_row0 - M;
        right.owner = false;
        right.data = (char*)full->data + (size_t)(a_row0 + M) * full->row_bytes();
        right.diag = nullptr;
        SKR_TRY(skr_pearson_gemm_op(ctx, a, &right, 0, r, row0, a_row0 + M));
    }
    return SKR_OK;
}

extern "C" int skr_pearson_gemm_op_mirror(skr_ctx* ctx, const skr_operand* a, const skr_operand* b, skr_mat* r,
                                          int64_t row0, int64_t col0, skr_mat* rt, int64_t trow0, int64_t tcol0) {
    SKR_REQUIRE(ctx && a && b && r && rt, "NULL argument");
    SKR_REQUIRE(a->ctx == ctx && b->ctx == ctx && r->ctx == ctx && rt->ctx == ctx, "handle belongs to a different ctx");
    SKR_REQUIRE(a->cols == b->cols && a->kind == b->kind && (a->kind == 0 || a->precision == b->precision),
                "operands were prepared for different shapes or precisions");
    SKR_TRY(skr_x8_pair_check(a, b));
    SKR_REQUIRE(r->dtype == SKR_F32 && rt->dtype == SKR_F32, "result matrices must be float32");
    SKR_REQUIRE(row0 >= 0 && col0 >= 0 && row0 + a->rows <= r->rows && col0 + b->rows <= r->cols,
                "result block [%lld+%lld, %lld+%lld] outside [%lld, %lld]", (long long)row0, (long long)a->rows,
                (long long)col0, (long long)b->rows, (long long)r->rows, (long long)r->cols);
    SKR_REQUIRE(trow0 >= 0 && tcol0 >= 0 && trow0 + b->rows <= rt->rows && tcol0 + a->rows <= rt->cols,
                "mirror block [%lld+%lld, %lld+%lld] outside [%lld, %lld]", (long long)trow0, (long long)b->rows,
                (long long)tcol0, (long long)a->rows, (long long)rt->rows, (long long)rt->cols);
    SKR_TRY(skr_activate(ctx));
    const int64_t M = a->rows, N = b->rows, K = a->cols;
    if (M == 0 || N == 0) return SKR_OK;
    float* C = (float*)r->data + (size_t)row0 * r->cols + col0;
    float* Ct = (float*)rt->data + (size_t)trow0 * rt->cols + tcol0;
    {   // the two blocks must not overlap (same matrix is fine as long as the blocks are disjoint)
        const bool same = r->data == rt->data;
        const bool rows_apart = row0 + M <= trow0 || trow0 + N <= row0;
        const bool cols_apart = col0 + N <= tcol0 || tcol0 + M <= col0;
        SKR_REQUIRE(!same || rows_apart || cols_apart, "block and mirror block overlap");
    }
    if (a->kind == 0) {
        // float32 operands: a_i . b_j and b_j . a_i accumulate the same products in the same k
        // order, so a second contraction gives the bit-identical transposed block
        const int64_t Kp = a->kt * 32;
        SKR_TRY(skr_launch_gemm_f32(ctx, (const float*)a->data, (const float*)b->data, C, M, N, Kp, Kp, Kp, r->cols, K, 0));
        return skr_launch_gemm_f32(ctx, (const float*)b->data, (const float*)a->data, Ct, N, M, Kp, Kp, Kp, rt->cols, K, 0);
    }
    return skr_launch_gemm_split(ctx, a->precision, a->data, b->data, C, M, N, a->kt, r->cols,
                                 (float)K * a->scale * b->scale, 2, Ct, rt->cols, a->coherent || b->coherent);
}

// One of two freshly filled operands fell back to the float32 layout (a row needs the dynamic range of
// the blocked fp32 accumulation): refill the other one the same way so that one kernel serves both.
static int match_layouts(skr_ctx* ctx, const skr_mat* xa, skr_operand* oa, const skr_mat* xb, skr_operand* ob,
                         int row_standardize) {
    // (the bound itself, not skr_x8_pair_check: that one leaves its refusal in the thread's error string, and this
    // function goes on to succeed)
    if (oa->kind == 3 && ob->kind == 3 && oa->x8_root != ob->x8_root && skr_x8_pair_bound(oa, ob) > kX8MeansLimit) {
        for (int q = 0; q < 2; q++) {  // both back to the three-product split
            skr_operand* o = q == 0 ? oa : ob;
            o->kind = 2;
            o->precision = SKR_PREC_F16X3;
            SKR_TRY(skr_operand_fill(ctx, q == 0 ? xa : xb, nullptr, nullptr, 0, 0.f, nullptr, row_standardize, o, nullptr));
        }
        return SKR_OK;
    }
    if (oa->kind == ob->kind) return SKR_OK;
    if (oa->kind != 0 && ob->kind != 0) {  // one of two f16f8 operands degraded to the three-product split: so does the other
        skr_operand* x8 = oa->kind == 3 ? oa : ob;
        x8->kind = 2;
        x8->precision = SKR_PREC_F16X3;
        return skr_operand_fill(ctx, x8 == oa ? xa : xb, nullptr, nullptr, 0, 0.f, nullptr, row_standardize, x8, nullptr);
    }
    skr_operand* split = oa->kind != 0 ? oa : ob;
    const skr_mat* src = oa->kind != 0 ? xa : xb;
    split->kind = 0;
    split->scale = 1.f;
    return skr_operand_fill(ctx, src, nullptr, nullptr, 0, 0.f, nullptr, row_standardize, split, nullptr);
}

// ---- matrix-level entry points built on operands ------------------------------------------------
extern "C" int skr_pearson_gemm(skr_ctx* ctx, const skr_mat* a, const skr_mat* b, int precision, int symmetric,
                                skr_mat* r, int64_t row0, int64_t col0) {
    SKR_TRY(check_pair(ctx, a, b));
    SKR_REQUIRE(r && r->ctx == ctx && r->dtype == a->dtype, "result matrix missing or of the wrong dtype");
    SKR_REQUIRE(row0 >= 0 && col0 >= 0 && row0 + a->rows <= r->rows && col0 + b->rows <= r->cols,
                "result block [%lld+%lld, %lld+%lld] outside [%lld, %lld]", (long long)row0, (long long)a->rows,
                (long long)col0, (long long)b->rows, (long long)r->rows, (long long)r->cols);
    SKR_TRY(skr_activate(ctx));
    if (a->rows == 0 || b->rows == 0) return SKR_OK;
    if (a->cols == 0) return skr_set_error(SKR_ERR_INVALID, "matrices have no columns");
    if (a->dtype == SKR_F64) {
        SKR_REQUIRE(precision == SKR_PREC_F64, "float64 operands need SKR_PREC_F64");
        return skr_launch_gemm_f64(ctx, (const double*)a->data, (const double*)b->data,
                                   (double*)r->data + (size_t)row0 * r->cols + col0, a->rows, b->rows, a->cols, a->cols,
                                   b->cols, r->cols, (double)a->cols, symmetric && row0 == col0);
    }
    SKR_REQUIRE(is_f32_precision(precision),
                "float32 operands need SKR_PREC_FP32, SKR_PREC_BF16X3, SKR_PREC_F16X3 or SKR_PREC_F16F8");
    const bool same = a->data == b->data && a->rows == b->rows;
    skr_operand *oa = nullptr, *ob = nullptr;
    int rc = skr_operand_create(ctx, a->rows, a->cols, precision, &oa);
    if (rc == SKR_OK) rc = skr_operand_fill(ctx, a, nullptr, nullptr, 0, 0.f, nullptr, 0, oa, nullptr);
    if (rc == SKR_OK && !same) {
        rc = skr_operand_create(ctx, b->rows, b->cols, precision, &ob);
        if (rc == SKR_OK) rc = skr_operand_fill(ctx, b, nullptr, nullptr, 0, 0.f, nullptr, 0, ob, nullptr);
    }
    if (rc == SKR_OK && !same) rc = match_layouts(ctx, a, oa, b, ob, 0);
    if (rc == SKR_OK) rc = skr_pearson_gemm_op(ctx, oa, same ? oa : ob, symmetric, r, row0, col0);
    skr_operand_free(oa);
    skr_operand_free(ob);
    return rc;
}

/* r[row0 + i, col0 + j] = <a_i, b_j> / K for float64 rows that are standardised already and may be zero-padded (the
 * stored width a->cols >= K; skr_pearson pads to whole 16-column stages): the contraction of skr_pearson's float64 path
 * as a step of its own, for callers that produce r one row stripe at a time or hold the rows of b on several GPUs.
 * A product of two doubles has one rounding, so a_i . b_j and b_j . a_i are the same bits: stripes need no swapped form. */
extern "C" int skr_pearson_gemm_f64(skr_ctx* ctx, const skr_mat* a, const skr_mat* b, int64_t K, int symmetric, skr_mat* r,
                                    int64_t row0, int64_t col0) {
    SKR_TRY(check_pair(ctx, a, b));
    SKR_REQUIRE(a->dtype == SKR_F64, "float64 rows only");
    SKR_REQUIRE(r && r->ctx == ctx && r->dtype == SKR_F64, "result matrix missing or not float64");
    SKR_REQUIRE(K > 0 && K <= a->cols, "K = %lld outside 1..%lld (the stored width)", (long long)K, (long long)a->cols);
    SKR_REQUIRE(row0 >= 0 && col0 >= 0 && row0 + a->rows <= r->rows && col0 + b->rows <= r->cols,
                "result block [%lld+%lld, %lld+%lld] outside [%lld, %lld]", (long long)row0, (long long)a->rows,
                (long long)col0, (long long)b->rows, (long long)r->rows, (long long)r->cols);
    SKR_TRY(skr_activate(ctx));
    if (a->rows == 0 || b->rows == 0) return SKR_OK;
    return skr_launch_gemm_f64(ctx, (const double*)a->data, (const double*)b->data,
                               (double*)r->data + (size_t)row0 * r->cols + col0, a->rows, b->rows, a->cols, a->cols, b->cols,
                               r->cols, (double)K, symmetric && a->data == b->data && a->rows == b->rows);
}

extern "C" int skr_pearson(skr_ctx* ctx, const skr_mat* counts1, const skr_mat* counts2, int row_standardize,
                           int precision, skr_mat* r) {
    SKR_TRY(check_pair(ctx, counts1, counts2));
    SKR_REQUIRE(r, "result matrix is NULL");
    SKR_REQUIRE(r->rows == counts1->rows && r->cols == counts2->rows, "result must be [%lld, %lld]",
                (long long)counts1->rows, (long long)counts2->rows);
    if (counts1->cols == 0) return skr_set_error(SKR_ERR_INVALID, "matrices have no columns");
    const bool same = counts1 == counts2;
    if (counts1->dtype == SKR_F64) {
        if (!row_standardize) return skr_pearson_gemm(ctx, counts1, counts2, precision, 0, r, 0, 0);
        SKR_REQUIRE(precision == SKR_PREC_F64, "float64 operands need SKR_PREC_F64");
        SKR_REQUIRE(r->dtype == SKR_F64 && r->ctx == ctx, "result matrix must be a float64 matrix of this ctx");
        if (counts1->rows == 0 || counts2->rows == 0) return SKR_OK;
        // standardised rows are zero-padded to whole 16-k stages of the tiled contraction (zeros add nothing)
        const int64_t K = counts1->cols, Kp = (K + 15) / 16 * 16;
        skr_mat *z1 = nullptr, *z2 = nullptr;
        int rc = skr_mat_create(ctx, counts1->rows, Kp, SKR_F64, &z1);
        if (rc == SKR_OK) rc = skr_row_standardize(ctx, counts1, z1);
        if (rc == SKR_OK && !same) {
            rc = skr_mat_create(ctx, counts2->rows, Kp, SKR_F64, &z2);
            if (rc == SKR_OK) rc = skr_row_standardize(ctx, counts2, z2);
        }
        if (rc == SKR_OK) {
            const skr_mat* zb = same ? z1 : z2;
            rc = skr_launch_gemm_f64(ctx, (const double*)z1->data, (const double*)zb->data, (double*)r->data, z1->rows,
                                     zb->rows, Kp, Kp, Kp, r->cols, (double)K, same);
        }
        skr_mat_free(z1);
        skr_mat_free(z2);
        return rc;
    }
    SKR_REQUIRE(is_f32_precision(precision), "float32 counts need a float32 precision");
    SKR_REQUIRE(r->dtype == SKR_F32 && r->ctx == ctx, "result matrix must be a float32 matrix of this ctx");
    if (counts1->rows == 0 || counts2->rows == 0) return SKR_OK;
    skr_operand *o1 = nullptr, *o2 = nullptr;
    int rc = skr_operand_create(ctx, counts1->rows, counts1->cols, precision, &o1);
    if (rc == SKR_OK) rc = skr_operand_fill(ctx, counts1, nullptr, nullptr, 0, 0.f, nullptr, row_standardize, o1, nullptr);
    if (rc == SKR_OK && !same) {
        rc = skr_operand_create(ctx, counts2->rows, counts2->cols, precision, &o2);
        if (rc == SKR_OK) rc = skr_operand_fill(ctx, counts2, nullptr, nullptr, 0, 0.f, nullptr, row_standardize, o2, nullptr);
    }
    if (rc == SKR_OK && !same) rc = match_layouts(ctx, counts1, o1, counts2, o2, row_standardize);
    if (rc == SKR_OK) rc = skr_pearson_gemm_op(ctx, o1, same ? o1 : o2, same, r, 0, 0);
    skr_operand_free(o1);
    skr_operand_free(o2);
    return rc;
}

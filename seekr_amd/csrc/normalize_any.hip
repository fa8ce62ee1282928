// BasicCounter.center() / standardize() / log2_norm() on a hand-assigned count matrix whose dtype is NOT float32
// (kmer_counts.py:165-192 work on whatever `self.counts` holds: a float64 matrix read from a CSV, an integer matrix, a
// float16 one; SURVEY §8(b): "those methods must work on an arbitrary host array").  get_counts() itself always produces
// float32 and never comes here — these kernels are the drop-in surface, not the hot path: one thread per column for the
// statistics, a grid-stride loop for the elementwise steps, written for exactness against numpy, not for bandwidth.
//
// What numpy does, per dtype (numpy 2.2, `_methods._mean` / `_var`, observed against the imported reference):
//   float64   every step in float64: column sums add the rows one after the other (axis 0 of a C-contiguous matrix),
//             mean = sum / N;  std: m' = sum / N, d = x - m', d = d * d, v = sum(d) / N, sqrt(v).
//   integers  the same chain on the values converted to float64 (np.mean / np.std use float64 for integer input); the
//             in-place `counts -= mean` / `counts /= std` then RAISE (float64 does not cast to an integer matrix) — the
//             host code replays that exception; only `counts -= <integer vector>` and log2_norm are legal.
//   float16   np.mean: float32 sums of the widened values, sum / N rounded to float32 then to float16;  np.std: every step
//             a float16 operation (evaluated in float32 and rounded to half — double rounding is innocuous, 24 >= 2*11+2):
//             m' = half(sum16 / N) with the quotient taken in float64, d = x - m', d * d, sum16(d) / N, sqrt.
//   log2_norm `counts += 1` in the matrix's own type (integers wrap), then np.log2 whose result type follows numpy's
//             loops: float64 -> float64, float16 -> float16, 8-bit integers -> float16, 16-bit -> float32, wider -> float64.
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cmath>
#include <cstring>

#include "common.hpp"

#pragma clang fp contract(off)

namespace {

enum NpType {  // = SKR_NP_* of include/seekr_hip.h
    NP_F16 = 0, NP_F32 = 1, NP_F64 = 2, NP_I8 = 3, NP_I16 = 4, NP_I32 = 5, NP_I64 = 6, NP_U8 = 7, NP_U16 = 8, NP_U32 = 9,
    NP_U64 = 10, NP_BOOL = 11, NP_COUNT = 12
};

__host__ __device__ inline int np_size(int t) {
    switch (t) {
        case NP_I8: case NP_U8: case NP_BOOL: return 1;
        case NP_F16: case NP_I16: case NP_U16: return 2;
        case NP_F32: case NP_I32: case NP_U32: return 4;
        default: return 8;
    }
}
inline bool np_is_int(int t) { return t >= NP_I8 && t <= NP_BOOL; }

// ---- float16 as numpy evaluates it: widen to float32 (exact), operate, round to nearest even
__device__ __forceinline__ float h2f(uint16_t h) {
    return __half2float(__ushort_as_half(h));
}
__device__ __forceinline__ uint16_t f2h(float f) { return __half_as_ushort(__float2half_rn(f)); }
__device__ __forceinline__ float rh(float f) { return h2f(f2h(f)); }  // f rounded to half precision, kept in a float
// float64 -> float16 in ONE rounding (npy_double_to_half): the double is first taken to float32 with round-to-odd — the
// sticky bit keeps what a second rounding needs (24 >= 11 + 2) — then to half with round-to-nearest-even
__device__ __forceinline__ uint16_t d2h(double d) {
    float f = __double2float_rz(d);
    if ((double)f != d) f = __uint_as_float(__float_as_uint(f) | 1u);
    return f2h(f);
}

// ---- one cell of the matrix, whatever its type, as float64 (integers: the C conversion numpy's cast loops use)
__device__ __forceinline__ double load_f64(const void* x, int64_t i, int t) {
    switch (t) {
        case NP_F64: return reinterpret_cast<const double*>(x)[i];
        case NP_F32: return (double)reinterpret_cast<const float*>(x)[i];
        case NP_F16: return (double)h2f(reinterpret_cast<const uint16_t*>(x)[i]);
        case NP_I8: return (double)reinterpret_cast<const int8_t*>(x)[i];
        case NP_I16: return (double)reinterpret_cast<const int16_t*>(x)[i];
        case NP_I32: return (double)reinterpret_cast<const int32_t*>(x)[i];
        case NP_I64: return (double)reinterpret_cast<const long long*>(x)[i];
        case NP_U8: case NP_BOOL: return (double)reinterpret_cast<const uint8_t*>(x)[i];
        case NP_U16: return (double)reinterpret_cast<const uint16_t*>(x)[i];
        case NP_U32: return (double)reinterpret_cast<const uint32_t*>(x)[i];
        default: return (double)reinterpret_cast<const unsigned long long*>(x)[i];
    }
}
// an integer cell as 64 bits (sign- or zero-extended) and back (the low bytes: two's-complement wrap, numpy's cast)
__device__ __forceinline__ unsigned long long load_int(const void* x, int64_t i, int t) {
    switch (t) {
        case NP_I8: return (unsigned long long)(long long)reinterpret_cast<const int8_t*>(x)[i];
        case NP_I16: return (unsigned long long)(long long)reinterpret_cast<const int16_t*>(x)[i];
        case NP_I32: return (unsigned long long)(long long)reinterpret_cast<const int32_t*>(x)[i];
        case NP_U8: case NP_BOOL: return reinterpret_cast<const uint8_t*>(x)[i];
        case NP_U16: return reinterpret_cast<const uint16_t*>(x)[i];
        case NP_U32: return reinterpret_cast<const uint32_t*>(x)[i];
        default: return reinterpret_cast<const unsigned long long*>(x)[i];
    }
}
__device__ __forceinline__ void store_int(void* x, int64_t i, int t, unsigned long long v) {
    switch (np_size(t)) {
        case 1: reinterpret_cast<uint8_t*>(x)[i] = (uint8_t)v; break;
        case 2: reinterpret_cast<uint16_t*>(x)[i] = (uint16_t)v; break;
        case 4: reinterpret_cast<uint32_t*>(x)[i] = (uint32_t)v; break;
        default: reinterpret_cast<unsigned long long*>(x)[i] = v; break;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Column statistics: thread c walks column c from the first row to the last (np.add.reduce along axis 0 of a C-contiguous
// matrix adds row after row into one accumulator per column); neighbouring threads read neighbouring cells.
// what: 0 = np.mean(axis=0), 1 = np.std(axis=0).  `out`: float16 for a float16 matrix, float64 otherwise.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colstat_any_kernel(const void* __restrict__ x, int64_t rows, int64_t cols, int t, int what,
                                                          void* __restrict__ out) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    if (t == NP_F16) {
        const uint16_t* xs = reinterpret_cast<const uint16_t*>(x);
        uint16_t* o = reinterpret_cast<uint16_t*>(out);
        if (what == 0) {  // _mean: umr_sum(dtype=float32), true_divide by N (evaluated in float64, stored float32), .astype(float16)
            float acc = 0.0f;
            for (int64_t i = 0; i < rows; i++) acc = __fadd_rn(acc, h2f(xs[i * cols + c]));
            o[c] = f2h((float)((double)acc / (double)rows));
            return;
        }
        // _var: every intermediate is a float16 array
        float acc = 0.0f;  // holds half-representable values
        for (int64_t i = 0; i < rows; i++) acc = rh(__fadd_rn(acc, h2f(xs[i * cols + c])));
        const float m = h2f(d2h((double)acc / (double)rows));
        acc = 0.0f;
        for (int64_t i = 0; i < rows; i++) {
            float d = rh(__fsub_rn(h2f(xs[i * cols + c]), m));
            d = rh(__fmul_rn(d, d));
            acc = rh(__fadd_rn(acc, d));
        }
        const float v = h2f(d2h((double)acc / (double)rows));
        o[c] = f2h((float)sqrt((double)v));  // half sqrt: the float32 square root, correctly rounded, rounded to half
        return;
    }
    double* o = reinterpret_cast<double*>(out);
    double acc = 0.0;
    for (int64_t i = 0; i < rows; i++) acc += load_f64(x, i * cols + c, t);
    const double m = acc / (double)rows;
    if (what == 0) {
        o[c] = m;
        return;
    }
    acc = 0.0;
    for (int64_t i = 0; i < rows; i++) {
        double d = load_f64(x, i * cols + c, t) - m;
        d = d * d;
        acc += d;
    }
    o[c] = sqrt(acc / (double)rows);
}

// ---------------------------------------------------------------------------------------------------------------
// Elementwise, in place.  op: 0 = x -= vec[col], 1 = x /= vec[col] (float matrices; vec float32 or float64 — the type
// numpy's promotion gives the operation), 2 = integer x -= vec[col] (64-bit wrap), 3 = x += 1 then y = log2(x).
// flags[1] |= 1 when a NaN is stored (standardize's warning).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void elementwise_any_kernel(void* __restrict__ x, int64_t rows, int64_t cols, int t, int op,
                                                              const void* __restrict__ vec, int vec_f64, void* __restrict__ y,
                                                              int y_t, uint32_t* __restrict__ flags) {
    const int64_t total = rows * cols;
    bool any_nan = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i % cols;
        if (op == 0 || op == 1) {
            if (t == NP_F64) {
                double* xd = reinterpret_cast<double*>(x);
                const double v = reinterpret_cast<const double*>(vec)[c];
                const double r = op == 0 ? xd[i] - v : xd[i] / v;
                any_nan |= (r != r);
                xd[i] = r;
            } else {  // float16: evaluated in the promoted type, rounded once to half
                uint16_t* xh = reinterpret_cast<uint16_t*>(x);
                uint16_t r;
                if (vec_f64) {
                    const double v = reinterpret_cast<const double*>(vec)[c];
                    const double a = (double)h2f(xh[i]);
                    r = d2h(op == 0 ? a - v : a / v);
                } else {
                    const float v = reinterpret_cast<const float*>(vec)[c];
                    const float a = h2f(xh[i]);
                    r = f2h(op == 0 ? __fsub_rn(a, v) : __fdiv_rn(a, v));
                }
                any_nan |= ((r & 0x7fffu) > 0x7c00u);
                xh[i] = r;
            }
        } else if (op == 2) {
            store_int(x, i, t, load_int(x, i, t) - reinterpret_cast<const unsigned long long*>(vec)[c]);
        } else {  // log2_norm (kmer_counts.py:189-192)
            if (t == NP_F64) {
                double* xd = reinterpret_cast<double*>(x);
                const double s = xd[i] + 1.0;
                xd[i] = s;
                reinterpret_cast<double*>(y)[i] = log2(s);
            } else if (t == NP_F16) {
                uint16_t* xh = reinterpret_cast<uint16_t*>(x);
                const float s = rh(__fadd_rn(h2f(xh[i]), 1.0f));
                xh[i] = f2h(s);
                reinterpret_cast<uint16_t*>(y)[i] = f2h(skr_log2_cr(s));
            } else {
                store_int(x, i, t, load_int(x, i, t) + 1ull);
                const double s = load_f64(x, i, t);  // the wrapped value, as numpy's cast to the loop's float type sees it
                if (y_t == NP_F16) reinterpret_cast<uint16_t*>(y)[i] = f2h(skr_log2_cr((float)s));
                else if (y_t == NP_F32) reinterpret_cast<float*>(y)[i] = skr_log2_cr((float)s);
                else reinterpret_cast<double*>(y)[i] = log2(s);
            }
        }
    }
    if (__any(any_nan) && (threadIdx.x & 63) == 0) atomicOr(&flags[1], 1u);
}

// ---------------------------------------------------------------------------------------------------------------
// The same statistics when numpy reduces COLUMN BY COLUMN: a column-major (Fortran-ordered) matrix — what
// `DataFrame.values` of a CSV hands out — or a single column.  There axis 0 is the fast axis, the reduction becomes the
// inner loop, and numpy's float loops add a column in their PAIRWISE order (umath loops: fewer than 8 values one after the
// other from 0; up to 128 in eight strided accumulators folded ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus leftovers; longer
// runs split at n/2 rounded down to a multiple of 8) — the order operand.hip reproduces for the ROWS of pearson.py:35-38.
// For 50 000 rows the two orders differ by up to 1e-5 relative in float32: not a rounding detail.  x is column-major here
// (cell (i, j) at j * rows + i); float32 and float64 only (no cast, so numpy makes one call per column: no buffer chunks).
// F maps a cell to the value that is summed (the cell itself; or its squared deviation from the column's mean).
// ---------------------------------------------------------------------------------------------------------------
template <typename T, typename A, typename F>
__device__ T np_pairwise(const A* a, int64_t n, F f) {  // f: a cell (type A) -> the value that is summed (type T)
    if (n < 8) {
        T res = (T)0;
        for (int64_t i = 0; i < n; i++) res = res + f(a[i]);
        return res;
    }
    if (n <= 128) {
        T r[8];
        for (int j = 0; j < 8; j++) r[j] = f(a[j]);
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] = r[j] + f(a[i + j]);
        T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res = res + f(a[i]);
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    const T left = np_pairwise<T, A, F>(a, n2, f);
    return left + np_pairwise<T, A, F>(a + n2, n - n2, f);
}

template <typename T>
__global__ __launch_bounds__(64) void colstat_pairwise_kernel(const T* __restrict__ x, int64_t rows, int64_t cols, int what,
                                                             T* __restrict__ out) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    const T* col = x + c * rows;
    // (the quotient by N: numpy divides the float32 sum by an intp scalar in float64 and stores float32 — one rounding of the
    // exact quotient either way, 53 >= 2 * 24 + 2)
    // numpy's reduction hands its inner loop the column in pieces of the iterator's buffer size (np.getbufsize(): 8 192
    // elements) also when nothing is cast: res = ((0 + pairwise(piece 0)) + pairwise(piece 1)) + ... — observed: at 50 000
    // float32 rows the one-piece sum is a different number
    constexpr int64_t kNpBuf = 8192;
    auto chunked = [&](auto f) {
        T res = (T)0;
        for (int64_t i0 = 0; i0 < rows; i0 += kNpBuf) res = res + np_pairwise<T, T>(col + i0, rows - i0 < kNpBuf ? rows - i0 : kNpBuf, f);
        return res;
    };
    const T m = (T)((double)chunked([](T v) { return v; }) / (double)rows);
    if (what == 0) {
        out[c] = m;
        return;
    }
    const T s = chunked([m](T v) {
        const T d = v - m;
        return d * d;
    });
    const T var = (T)((double)s / (double)rows);
    out[c] = (T)sqrt((double)var);
}

// A column-major float16 matrix.  np.mean: the column is cast to float32 in pieces of the buffer size and added pairwise in
// float32 (umr_sum(dtype=float32)).  np.std: numpy's half loop adds a piece pairwise IN FLOAT32 accumulators
// (HALF_pairwise_sum returns a float) and rounds to half once per piece — out = half(float(out) + piece) — every other
// step as in colstat_any_kernel (half arrays, quotients by N in float64 rounded once to half).
__global__ __launch_bounds__(64) void colstat_pairwise_f16_kernel(const uint16_t* __restrict__ x, int64_t rows, int64_t cols, int what,
                                                                 uint16_t* __restrict__ out) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    const uint16_t* col = x + c * rows;
    constexpr int64_t kNpBuf = 8192;
    auto piece = [&](int64_t i0, auto f) { return np_pairwise<float, uint16_t>(col + i0, rows - i0 < kNpBuf ? rows - i0 : kNpBuf, f); };
    if (what == 0) {
        float res = 0.0f;
        for (int64_t i0 = 0; i0 < rows; i0 += kNpBuf) res = __fadd_rn(res, piece(i0, [](uint16_t v) { return h2f(v); }));
        out[c] = f2h((float)((double)res / (double)rows));
        return;
    }
    float res = 0.0f;  // half-representable between pieces
    for (int64_t i0 = 0; i0 < rows; i0 += kNpBuf) res = rh(__fadd_rn(res, piece(i0, [](uint16_t v) { return h2f(v); })));
    const float m = h2f(d2h((double)res / (double)rows));
    res = 0.0f;
    for (int64_t i0 = 0; i0 < rows; i0 += kNpBuf)
        res = rh(__fadd_rn(res, piece(i0, [m](uint16_t v) {
                     const float d = rh(__fsub_rn(h2f(v), m));
                     return rh(__fmul_rn(d, d));
                 })));
    const float var = h2f(d2h((double)res / (double)rows));
    out[c] = f2h((float)sqrt((double)var));
}

struct DevBuf {  // a device allocation freed on every way out
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) {
        SKR_HIP(hipMalloc(&p, bytes ? bytes : 1));
        return SKR_OK;
    }
};

int check_any(const skr_ctx* ctx, const void* x, int64_t rows, int64_t cols, int t) {
    SKR_REQUIRE(ctx, "ctx is NULL");
    SKR_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    SKR_REQUIRE(x || rows * cols == 0, "matrix is NULL");
    SKR_REQUIRE(t >= 0 && t < NP_COUNT, "unknown element type %d", t);
    return SKR_OK;
}

}  // namespace

extern "C" int skr_host_colstat(skr_ctx* ctx, const void* x, int64_t rows, int64_t cols, int np_type, int what, void* out) {
    SKR_TRY(check_any(ctx, x, rows, cols, np_type));
    SKR_REQUIRE(out || cols == 0, "out is NULL");
    SKR_REQUIRE(what == 0 || what == 1, "what must be 0 (mean) or 1 (std)");
    SKR_REQUIRE(np_type != NP_F32, "float32 matrices take skr_colsum_seq (the tuned path)");
    if (cols == 0) return SKR_OK;
    SKR_TRY(skr_activate(ctx));
    const size_t bytes = (size_t)rows * (size_t)cols * (size_t)np_size(np_type);
    const size_t out_elem = np_type == NP_F16 ? 2 : 8;
    DevBuf dx, dout;
    SKR_TRY(dx.alloc(bytes));
    SKR_TRY(dout.alloc((size_t)cols * out_elem));
    if (bytes) SKR_HIP(hipMemcpyAsync(dx.p, x, bytes, hipMemcpyHostToDevice, ctx->stream));
    {
        SkrProfScope prof(ctx, "colstat_any");
        hipLaunchKernelGGL(colstat_any_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, ctx->stream, dx.p, rows, cols,
                           np_type, what, dout.p);
    }
    SKR_HIP(hipGetLastError());
    SKR_HIP(hipMemcpyAsync(out, dout.p, (size_t)cols * out_elem, hipMemcpyDeviceToHost, ctx->stream));
    SKR_HIP(hipStreamSynchronize(ctx->stream));
    return SKR_OK;
}

extern "C" int skr_host_colstat_colmajor(skr_ctx* ctx, const void* x, int64_t rows, int64_t cols, int np_type, int what, void* out) {
    SKR_TRY(check_any(ctx, x, rows, cols, np_type));
    SKR_REQUIRE(out || cols == 0, "out is NULL");
    SKR_REQUIRE(what == 0 || what == 1, "what must be 0 (mean) or 1 (std)");
    SKR_REQUIRE(np_type == NP_F16 || np_type == NP_F32 || np_type == NP_F64,
                "column-major statistics exist for float16, float32 and float64 (integers: as float64)");
    if (cols == 0) return SKR_OK;
    SKR_TRY(skr_activate(ctx));
    const size_t elem = (size_t)np_size(np_type);
    const size_t bytes = (size_t)rows * (size_t)cols * elem;
    DevBuf dx, dout;
    SKR_TRY(dx.alloc(bytes));
    SKR_TRY(dout.alloc((size_t)cols * elem));
    if (bytes) SKR_HIP(hipMemcpyAsync(dx.p, x, bytes, hipMemcpyHostToDevice, ctx->stream));
    {
        SkrProfScope prof(ctx, "colstat_pairwise");
        const dim3 grid((unsigned)((cols + 63) / 64));
        if (np_type == NP_F16)
            hipLaunchKernelGGL(colstat_pairwise_f16_kernel, grid, dim3(64), 0, ctx->stream, (const uint16_t*)dx.p, rows, cols, what, (uint16_t*)dout.p);
        else if (np_type == NP_F32)
            hipLaunchKernelGGL(colstat_pairwise_kernel<float>, grid, dim3(64), 0, ctx->stream, (const float*)dx.p, rows, cols, what, (float*)dout.p);
        else
            hipLaunchKernelGGL(colstat_pairwise_kernel<double>, grid, dim3(64), 0, ctx->stream, (const double*)dx.p, rows, cols, what, (double*)dout.p);
    }
    SKR_HIP(hipGetLastError());
    SKR_HIP(hipMemcpyAsync(out, dout.p, (size_t)cols * elem, hipMemcpyDeviceToHost, ctx->stream));
    SKR_HIP(hipStreamSynchronize(ctx->stream));
    return SKR_OK;
}

extern "C" int skr_host_apply(skr_ctx* ctx, void* x, int64_t rows, int64_t cols, int np_type, int op, const void* vec, int vec_is_f64,
                              void* y, int y_np_type, int* has_nan) {
    SKR_TRY(check_any(ctx, x, rows, cols, np_type));
    SKR_REQUIRE(op >= 0 && op <= 3, "unknown op %d", op);
    SKR_REQUIRE(np_type != NP_F32, "float32 matrices take skr_apply (the tuned path)");
    if (has_nan) *has_nan = 0;
    const bool is_int = np_is_int(np_type);
    if (op == 0 || op == 1) SKR_REQUIRE(!is_int, "op %d needs a float matrix", op);
    if (op == 2) SKR_REQUIRE(is_int && np_type != NP_BOOL, "op 2 needs an integer matrix");
    if (op == 3) {
        const int want = np_type == NP_F64 ? NP_F64 : np_type == NP_F16 ? NP_F16 : np_size(np_type) == 1 ? NP_F16 : np_size(np_type) == 2 ? NP_F32 : NP_F64;
        SKR_REQUIRE(np_type != NP_BOOL, "numpy refuses `+= 1` on a bool matrix");
        SKR_REQUIRE(y_np_type == want, "np.log2 of element type %d gives type %d, not %d", np_type, want, y_np_type);
        SKR_REQUIRE(y || rows * cols == 0, "y is NULL");
    } else {
        SKR_REQUIRE(vec || cols == 0, "vec is NULL");
        if (np_type == NP_F64) SKR_REQUIRE(vec_is_f64, "a float64 matrix is operated on in float64");
    }
    const int64_t total = rows * cols;
    if (total == 0) return SKR_OK;
    SKR_TRY(skr_activate(ctx));
    const size_t bytes = (size_t)total * (size_t)np_size(np_type);
    const size_t vec_bytes = op == 3 ? 0 : (size_t)cols * (op == 2 || vec_is_f64 ? 8 : 4);
    const size_t y_bytes = op == 3 ? (size_t)total * (size_t)np_size(y_np_type) : 0;
    DevBuf dx, dv, dy;
    SKR_TRY(dx.alloc(bytes));
    SKR_TRY(dv.alloc(vec_bytes));
    SKR_TRY(dy.alloc(y_bytes));
    SKR_HIP(hipMemcpyAsync(dx.p, x, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (vec_bytes) SKR_HIP(hipMemcpyAsync(dv.p, vec, vec_bytes, hipMemcpyHostToDevice, ctx->stream));
    SKR_HIP(hipMemsetAsync(ctx->d_flags + 1, 0, 4, ctx->stream));
    {
        SkrProfScope prof(ctx, "elementwise_any");
        const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, (int64_t)ctx->num_cu * 8));
        hipLaunchKernelGGL(elementwise_any_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, dx.p, rows, cols, np_type, op,
                           (const void*)dv.p, vec_is_f64, dy.p, y_np_type, ctx->d_flags);
    }
    SKR_HIP(hipGetLastError());
    SKR_HIP(hipMemcpyAsync(x, dx.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (y_bytes) SKR_HIP(hipMemcpyAsync(y, dy.p, y_bytes, hipMemcpyDeviceToHost, ctx->stream));
    SKR_HIP(hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    SKR_HIP(hipStreamSynchronize(ctx->stream));
    if (has_nan) *has_nan = (int)(ctx->h_flags[1] & 1u);
    return SKR_OK;
}

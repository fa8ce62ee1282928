// K4 + K5 — column statistics in numpy's reduction order and the elementwise transforms of
// BasicCounter (kmer_counts.py:165-209).
//
// Parity here is defined by rounding, not mathematics:
//   * np.mean/np.std(axis=0) on the C-contiguous float32 matrix add the rows one after the
//     other into one float32 accumulator per column.  colsum_seq_kernel reproduces exactly
//     that chain; a tree reduction would be *more* accurate and fail parity at N >= 50k.
//   * every elementwise step is a separately rounded float32 operation (no FMA contraction).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <vector>

#include "common.hpp"

#pragma clang fp contract(off)

namespace {

constexpr int kColsPerWG = 16;    // 64-byte column strip per workgroup: 4^6 columns -> 256 workgroups, one per CU
constexpr int kTileRows = 512;    // rows staged in LDS per step (32 KiB)
constexpr int kDepth = 6;         // tiles in flight per workgroup (register ring): 192 KiB of HBM loads (4 -> 6: 0.24 -> 0.21-0.23 ms per pass)
constexpr int kThreads = 256;      // staging threads (waves 1..4); wave 0 of the workgroup only walks
constexpr int kWgThreads = kThreads + 64;
constexpr int kLanesPerRow = kColsPerWG / 4;             // 16-byte loads
constexpr int kRowsPerPass = kThreads / kLanesPerRow;    // rows covered by one load instruction
constexpr int kLoads = kTileRows / kRowsPerPass;         // loads per thread per tile

enum CenterKind { C_NONE = 0, C_F32 = 1, C_F64 = 2 };

template <int KIND>
__device__ __forceinline__ float sub_center(float x, const void* vec, int64_t col) {
    if (KIND == C_F32) return __fsub_rn(x, reinterpret_cast<const float*>(vec)[col]);
    if (KIND == C_F64) return (float)((double)x - reinterpret_cast<const double*>(vec)[col]);
    return x;
}

template <int KIND>
__device__ __forceinline__ float div_scale(float x, const void* vec, int64_t col) {
    if (KIND == C_F32) return __fdiv_rn(x, reinterpret_cast<const float*>(vec)[col]);
    if (KIND == C_F64) return (float)((double)x / reinterpret_cast<const double*>(vec)[col]);
    return x;
}

// ---------------------------------------------------------------------------------------
// Sequential column sums.  The chain acc = fl32(acc + x[i, j]) over i is inherently serial per
// column, so the kernel is organised around keeping HBM busy while one lane per column walks:
//   * a workgroup owns a strip of 16 columns (64 B per row); 4^6 columns give 256 workgroups;
//   * waves 1..4 (256 threads) stream tiles of 512 rows x 16 columns (16-byte loads, t() applied
//     on the way) through a ring of kDepth register tiles into a double-buffered LDS tile, so
//     192 KiB of loads per workgroup are in flight while the walk proceeds;
//   * wave 0 does nothing but walk: lanes 0..15 go down the current LDS tile row by row, each
//     extending one column's float32 chain.  The chain of dependent v_add_f32 paces the kernel
//     (not HBM latency), so the walker carries no staging work: one barrier per tile is all that
//     stands between two tiles of its chain.
// ---------------------------------------------------------------------------------------
// MINOUT (first pass only: no centre, no square): the staging waves also keep the minimum of the raw values of their
// columns (a thread always loads the same four columns) and each of the four waves writes its minima as one row of
// `colmin` [4, cols] (NaN where it met a NaN).  Rounding is monotone, so the minimum of the NORMALISED matrix over a
// column is the normalised minimum of the raw column (for scale >= 0): the Log2.post shift |min z| (kmer_counts.py:208)
// then needs a scan of these four rows instead of a pass over the matrix.
// ---- the chain across GPUs without a transfer in between (skr_chain): rank g's walker of strip s waits — the staging
// ring already full — until rank g-1 has put the strip's 16 running sums into rank g's mailbox (peer stores over xGMI
// into uncached memory + an epoch word per strip), continues the chain over its own rows and passes the sums on the
// same way; the last rank writes the finished sums into every rank's result box.  Against ncclSend / ncclRecv of the
// vector between two kernels (tools/chain_bench.py: 41 us per hop + 24 us of kernel start, x 7 hops x 3 passes) a hop is
// one peer store and one poll.
constexpr int kMaxChainRanks = 16;
struct ChainLink {
    const float* carry_in = nullptr;       // this rank's mailbox (written by rank - 1); null on rank 0 / unchained
    const uint32_t* carry_flag = nullptr;  // [strips]
    float* next_data = nullptr;            // rank + 1's mailbox; null on the last rank
    uint32_t* next_flag = nullptr;
    float* result_data[kMaxChainRanks] = {};  // last rank only: every other rank's result box
    uint32_t* result_flag[kMaxChainRanks] = {};
    int n_result = 0;
    uint32_t epoch = 0;
    uint32_t* err = nullptr;  // set when a wait gave up (a peer died): the host raises instead of the kernel hanging
};

// The wait is bounded in TIME, not in polls: s_memrealtime counts at 100 MHz whatever the shader clock does, so a wait
// gives up after kChainWaitTicks = 10 s — long enough for a peer that is still packing its FASTA, short enough that a
// dead peer is an error and not a hang.  A link that gave up raises `err`; the host folds that flag into the step's
// verdict all-reduce, so EVERY rank raises together (the sums a timed-out link passes on are garbage).
constexpr unsigned long long kChainWaitTicks = 1000000000ull;
__device__ __forceinline__ void chain_wait(const uint32_t* flag, uint32_t epoch, uint32_t* err) {
    // relaxed polls (an acquire per poll would invalidate the caches under the staging waves), one acquire at the end
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const uint32_t v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)(v - epoch) >= 0) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            return;
        }
        if (__builtin_amdgcn_s_memrealtime() - t0 > kChainWaitTicks) break;
        __builtin_amdgcn_s_sleep(4);
    }
    if (err) atomicOr(err, 1u);
}
__device__ __forceinline__ float chain_load(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void chain_store(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the wave's lanes < 16 hold one column each of strip `strip`: write them to every box the link names (the next rank's
// carry box, or on the last rank all result boxes), ONE system-scope release for all of them, then raise the epoch words
__device__ __forceinline__ void chain_post_all(const ChainLink& link, int64_t strip, int64_t col, bool mine, float v, int lane) {
    if (!link.next_data && link.n_result == 0) return;
    if (mine) {
        if (link.next_data) chain_store(link.next_data + col, v);
        for (int r = 0; r < link.n_result; r++) chain_store(link.result_data[r] + col, v);
    }
    __threadfence_system();
    if (lane == 0) {
        if (link.next_flag) __hip_atomic_store(link.next_flag + strip, link.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int r = 0; r < link.n_result; r++)
            __hip_atomic_store(link.result_flag[r] + strip, link.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// 16 bytes of a row whose start is only 4-byte aligned (a column count that is not a multiple of four: 5^6, 7^4 ...):
// the hardware takes a dword-aligned global_load_dwordx4 as it takes an aligned one, the type tells the compiler so
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float (*TilePtr)[kColsPerWG][kTileRows + 4];

// VEC: the strip is full (16 columns inside the matrix): unguarded 16-byte loads.  Decided per WORKGROUP since round 5
// (colsum_seq_kernel below): a width that is not a multiple of 16 used to send every strip down the guarded element-wise
// path (50 000 x 15 625: 1.13 ms per pass, 2.7 TB/s); now only its last, ragged strip goes there.
template <int CK, bool SQUARE, bool VEC, bool MINOUT>
__device__ __forceinline__ void colsum_seq_body(TilePtr tile, int64_t strip, const float* __restrict__ x, int64_t rows,
                                                int64_t cols, const void* __restrict__ center,
                                                const float* __restrict__ center2, float* __restrict__ acc,
                                                float* __restrict__ colmin, const ChainLink& link) {
    const int64_t col0 = strip * kColsPerWG;
    const int64_t n_tiles = (rows + kTileRows - 1) / kTileRows;
    // both roles run the tile loop to a multiple of kDepth: the stagers' loop then has no conditional load in
    // it, which is what lets hipcc keep counted vmcnt waits (three tiles in flight behind the one being stored)
    const int64_t n_pad = (n_tiles + kDepth - 1) / kDepth * kDepth;
    if (threadIdx.x < 64) {  // ---- the walker wave
        const int wl = threadIdx.x;
        const bool mine = wl < kColsPerWG && col0 + wl < cols;
        float running = mine ? acc[col0 + wl] : 0.f;
        for (int64_t t = -kDepth; t < n_pad; t++) {
            __syncthreads();  // tile t is in buffer t & 1 (the first kDepth rounds only fill the stagers' ring)
            if (t == 0 && link.carry_flag) {
                // the previous rank's sums of this strip: waited for HERE, with kDepth tiles of this rank's rows already
                // on their way (the stagers stand at the next barrier), not in front of the launch
                if (wl == 0) chain_wait(link.carry_flag + strip, link.epoch, link.err);
                running = mine ? chain_load(link.carry_in + col0 + wl) : 0.f;
            }
            if (wl < kColsPerWG && t >= 0 && t < n_tiles) {
                const int64_t left = rows - t * kTileRows;
                const int nr = (int)(left < kTileRows ? left : kTileRows);
                const float* colp = &tile[t & 1][wl][0];
                if (nr == kTileRows) {
                    // software-pipelined: the ds_read_b128 of the next 64 rows are in flight while the
                    // 64 dependent adds of the current ones retire (an LDS read takes ~130 cycles; left
                    // to the compiler it was exposed once per 64 rows: 13 cycles per row instead of ~8)
                    // (8 reads per batch: lgkmcnt counts at most 15 outstanding LDS operations)
                    constexpr int kBatch = 8, kBatches = kTileRows / (4 * kBatch);
                    float4 q[2][kBatch];
#pragma unroll
                    for (int i = 0; i < kBatch; i++) q[0][i] = *reinterpret_cast<const float4*>(colp + 4 * i);
#pragma unroll
                    for (int b = 0; b < kBatches; b++) {
                        if (b + 1 < kBatches) {
#pragma unroll
                            for (int i = 0; i < kBatch; i++)
                                q[(b + 1) & 1][i] = *reinterpret_cast<const float4*>(colp + (b + 1) * 4 * kBatch + 4 * i);
                        }
                        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch above the adds it overlaps with
#pragma unroll
                        for (int i = 0; i < kBatch; i++) {
                            running = __fadd_rn(running, q[b & 1][i].x);
                            running = __fadd_rn(running, q[b & 1][i].y);
                            running = __fadd_rn(running, q[b & 1][i].z);
                            running = __fadd_rn(running, q[b & 1][i].w);
                        }
                    }
                } else {
                    const int nr4 = nr & ~3;
                    for (int r = 0; r < nr4; r += 4) {
                        const float4 q = *reinterpret_cast<const float4*>(colp + r);
                        running = __fadd_rn(running, q.x);
                        running = __fadd_rn(running, q.y);
                        running = __fadd_rn(running, q.z);
                        running = __fadd_rn(running, q.w);
                    }
                    for (int r = nr4; r < nr; r++) running = __fadd_rn(running, colp[r]);
                }
            }
            // the stagers refill buffer t & 1 with tile t + 2 only after the barrier of tile t + 1,
            // which this wave reaches when it is done here
        }
        if (mine) acc[col0 + wl] = running;
        chain_post_all(link, strip, col0 + wl, mine, running, wl);
        return;
    }
    // ---- the staging waves
    const int tid = threadIdx.x - 64;
    const int lane_c4 = (tid % kLanesPerRow) * 4;  // 4 consecutive columns handled by this thread
    const int lane_r = tid / kLanesPerRow;         // row inside a kRowsPerPass-row slab

    float c2[4] = {0, 0, 0, 0};
    if (SQUARE && center2) {
        for (int j = 0; j < 4; j++)
            if (col0 + lane_c4 + j < cols) c2[j] = center2[col0 + lane_c4 + j];
    }

    float4 mn = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    bool saw_nan[4] = {false, false, false, false};
    auto load_tile = [&](int64_t row_base, float4 (&regs)[kLoads]) {
        if (VEC) {
#pragma unroll
            for (int s = 0; s < kLoads; s++) {
                int64_t r = row_base + s * kRowsPerPass + lane_r;
                r = r < rows ? r : rows - 1;
                const f4u q = *reinterpret_cast<const f4u*>(x + (size_t)r * cols + col0 + lane_c4);
                regs[s] = make_float4(q[0], q[1], q[2], q[3]);
            }
            if (MINOUT) {  // rows past the end re-read the last row: real values, harmless for a minimum
#pragma unroll
                for (int s = 0; s < kLoads; s++) {
                    mn.x = fminf(mn.x, regs[s].x);
                    mn.y = fminf(mn.y, regs[s].y);
                    mn.z = fminf(mn.z, regs[s].z);
                    mn.w = fminf(mn.w, regs[s].w);
                    saw_nan[0] |= regs[s].x != regs[s].x;
                    saw_nan[1] |= regs[s].y != regs[s].y;
                    saw_nan[2] |= regs[s].z != regs[s].z;
                    saw_nan[3] |= regs[s].w != regs[s].w;
                }
            }
        } else {  // ragged strip / odd column count: element-wise, guarded (small matrices only)
#pragma unroll
            for (int s = 0; s < kLoads; s++) {
                int64_t r = row_base + s * kRowsPerPass + lane_r;
                r = r < rows ? r : rows - 1;
                const float* p = x + (size_t)r * cols;
                const int64_t c = col0 + lane_c4;
                float4 v;
                v.x = p[c + 0 < cols ? c + 0 : cols - 1];
                v.y = p[c + 1 < cols ? c + 1 : cols - 1];
                v.z = p[c + 2 < cols ? c + 2 : cols - 1];
                v.w = p[c + 3 < cols ? c + 3 : cols - 1];
                regs[s] = v;
            }
        }
        if (MINOUT && !VEC) {  // columns past the end re-read the last column, rows past the end the last row: real values
#pragma unroll
            for (int s = 0; s < kLoads; s++) {
                mn.x = fminf(mn.x, regs[s].x);
                mn.y = fminf(mn.y, regs[s].y);
                mn.z = fminf(mn.z, regs[s].z);
                mn.w = fminf(mn.w, regs[s].w);
                saw_nan[0] |= regs[s].x != regs[s].x;
                saw_nan[1] |= regs[s].y != regs[s].y;
                saw_nan[2] |= regs[s].z != regs[s].z;
                saw_nan[3] |= regs[s].w != regs[s].w;
            }
        }
    };
    auto transform = [&](float v, int j) -> float {
        int64_t c = col0 + lane_c4 + j;
        c = c < cols ? c : cols - 1;  // columns past the end are computed but never written back
        float t = sub_center<CK>(v, center, c);
        if (SQUARE) {
            const float d = center2 ? __fsub_rn(t, c2[j]) : t;
            t = __fmul_rn(d, d);
        }
        return t;
    };
    auto store_tile = [&](int buf, const float4 (&regs)[kLoads]) {
#pragma unroll
        for (int s = 0; s < kLoads; s++) {
            float4 v = regs[s];
            if (CK != C_NONE || SQUARE) {
                v.x = transform(v.x, 0);
                v.y = transform(v.y, 1);
                v.z = transform(v.z, 2);
                v.w = transform(v.w, 3);
            }
            const int r = s * kRowsPerPass + lane_r;
            tile[buf][lane_c4 + 0][r] = v.x;
            tile[buf][lane_c4 + 1][r] = v.y;
            tile[buf][lane_c4 + 2][r] = v.z;
            tile[buf][lane_c4 + 3][r] = v.w;
        }
    };

    // Every load below is unconditional (row indices clamped; tiles past the end re-read the last row and are
    // never walked): a branch around a load makes hipcc drain vmcnt(0) at the next use and serialises the ring.
    // ... and so does a prologue that fills the ring in front of the loop (hipcc then waits vmcnt(0) at the loop
    // head).  The loop therefore starts one round early with an empty ring: that round stores zeros nobody
    // walks and issues the loads of tiles 0 .. kDepth-1.
    float4 ring[kDepth][kLoads] = {};
    for (int64_t t0 = -kDepth; t0 < n_pad; t0 += kDepth) {
#pragma unroll
        for (int d = 0; d < kDepth; d++) {  // static ring index: the tiles stay in registers
            store_tile(d & 1, ring[d]);     // kDepth is even, so d & 1 == t & 1; waits only for this tile's (oldest) loads
            __syncthreads();
            load_tile((t0 + d + kDepth) * kTileRows, ring[d]);
        }
    }
    if (MINOUT) {
        // lanes with the same lane % kLanesPerRow hold the same four columns: fold them inside the wave
        float m[4] = {mn.x, mn.y, mn.z, mn.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float v = saw_nan[j] ? NAN : m[j];
            for (int off = kLanesPerRow; off < 64; off <<= 1) {
                const float o = __shfl_xor(v, off, 64);
                v = (v != v || o != o) ? NAN : fminf(v, o);
            }
            m[j] = v;
        }
        const int swave = tid >> 6;  // 0..3: the row of colmin this staging wave owns
        if ((tid & 63) < kLanesPerRow) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (col0 + lane_c4 + j < cols) colmin[(size_t)swave * cols + col0 + lane_c4 + j] = m[j];
        }
    }
}

template <int CK, bool SQUARE, bool MINOUT = false>
__global__ __launch_bounds__(kWgThreads) void colsum_seq_kernel(const float* __restrict__ x, int64_t rows,
                                                              int64_t cols, const void* __restrict__ center,
                                                              const float* __restrict__ center2,
                                                              float* __restrict__ acc, float* __restrict__ colmin,
                                                              const ChainLink link) {
    // column-major tile so that the walker fetches 4 consecutive rows of its column with one
    // ds_read_b128; +4 floats of padding per column keep the 16 walker lanes on distinct banks
    __shared__ __attribute__((aligned(16))) float tile[2][kColsPerWG][kTileRows + 4];
    // Workgroups are dealt to the 8 XCDs round-robin by blockIdx, and each XCD has its own L2.  A strip is
    // 64 bytes wide — half a cache line — so neighbouring strips placed on different XCDs make HBM deliver
    // every line twice.  G neighbouring strips (G * 64 contiguous bytes of each row) go to the same XCD.
    int64_t strip = blockIdx.x;
    {
        constexpr int G = 4;
        const int64_t full = ((int64_t)gridDim.x / (8 * G)) * (8 * G);
        if (strip < full) {
            const int64_t xcd = strip % 8, slot = strip / 8;
            strip = ((slot / G) * 8 + xcd) * G + (slot % G);
        }
    }
    if ((strip + 1) * kColsPerWG <= cols)  // uniform over the workgroup: no branch around any load inside the bodies
        colsum_seq_body<CK, SQUARE, true, MINOUT>(tile, strip, x, rows, cols, center, center2, acc, colmin, link);
    else
        colsum_seq_body<CK, SQUARE, false, MINOUT>(tile, strip, x, rows, cols, center, center2, acc, colmin, link);
}

// A rank without rows still passes the sums on: one thread per column, one wave per four strips.
__global__ __launch_bounds__(64) void chain_forward_kernel(int64_t cols, float* __restrict__ acc, const ChainLink link) {
    const int lane = threadIdx.x, wl = lane & 15;
    const int64_t strip = (int64_t)blockIdx.x * 4 + (lane >> 4);
    const int64_t col = strip * kColsPerWG + wl;
    const bool mine = col < cols;
    float v = mine ? acc[col] : 0.f;
    if (link.carry_flag && strip * kColsPerWG < cols) {
        if (wl == 0) chain_wait(link.carry_flag + strip, link.epoch, link.err);
    }
    __builtin_amdgcn_wave_barrier();
    if (link.carry_flag && mine) v = chain_load(link.carry_in + col);
    if (mine) acc[col] = v;
    if (strip * kColsPerWG >= cols) return;
    chain_post_all(link, strip, col, mine, v, wl);
}

// Every rank but the last: the finished sums arrive in the result box, strip by strip; copy them into acc.
__global__ __launch_bounds__(64) void chain_result_kernel(int64_t cols, float* __restrict__ acc, const float* __restrict__ box,
                                                          const uint32_t* __restrict__ flag, uint32_t epoch, uint32_t* err) {
    const int lane = threadIdx.x, wl = lane & 15;
    const int64_t strip = (int64_t)blockIdx.x * 4 + (lane >> 4);
    const int64_t col = strip * kColsPerWG + wl;
    if (strip * kColsPerWG >= cols) return;
    if (wl == 0) chain_wait(flag + strip, epoch, err);
    __builtin_amdgcn_wave_barrier();
    if (col < cols) acc[col] = chain_load(box + col);
}

__global__ void vec_finish_kernel(float* v, int64_t cols, float n, int take_sqrt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cols) return;
    float q = __fdiv_rn(v[i], n);
    // correctly rounded f32 sqrt: f64 sqrt then one rounding (53 >= 2*24+2 bits: double rounding is innocuous);
    // __fsqrt_rn would lower to the ~1 ulp native v_sqrt_f32 here
    if (take_sqrt) q = (float)sqrt((double)q);
    v[i] = q;
}

// ---------------------------------------------------------------------------------------
// Elementwise: y = post(scale(center(pre(x)))) and the NaN-propagating global minimum of z.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t order_bits(float f) {  // monotone float -> uint map
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
inline float unorder_bits(uint32_t u) {  // host side inverse
    const uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    union {
        uint32_t u32;
        float f32;
    } cv;
    cv.u32 = b;
    return cv.f32;
}

template <int CK, int SK, bool PRE, bool POST, bool WRITE, bool MIN>
__global__ __launch_bounds__(256) void elementwise_kernel(const float* __restrict__ x, int64_t rows, int64_t cols,
                                                          const void* __restrict__ center,
                                                          const void* __restrict__ scale, float shift,
                                                          float* __restrict__ y, uint32_t* __restrict__ flags) {
    const int64_t total = rows * cols;
    const bool vec_ok = (cols % 4) == 0;
    float vmin = INFINITY;
    bool any_nan = false;
    auto one = [&](float v, int64_t c) -> float {
        if (PRE) v = skr_log2_cr(__fadd_rn(v, 1.0f));
        v = sub_center<CK>(v, center, c);
        v = div_scale<SK>(v, scale, c);
        if (v != v) any_nan = true;
        if (MIN) vmin = fminf(vmin, v);
        if (POST) {
            v = skr_log2_of_sum1(__fadd_rn(v, shift));
        }
        return v;
    };
    if (vec_ok) {
        const int64_t n4 = total / 4;
        const int64_t c4 = cols / 4;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t c = (i % c4) * 4;
            float4 v = reinterpret_cast<const float4*>(x)[i];
            v.x = one(v.x, c);
            v.y = one(v.y, c + 1);
            v.z = one(v.z, c + 2);
            v.w = one(v.w, c + 3);
            if (WRITE) reinterpret_cast<float4*>(y)[i] = v;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            const float v = one(x[i], i % cols);
            if (WRITE) y[i] = v;
        }
    }
    // block reduction of (min, nan)
    uint32_t enc = order_bits(vmin);
    uint32_t nanf = any_nan ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_down(enc, off, 64);
        enc = o < enc ? o : enc;
        nanf |= __shfl_down(nanf, off, 64);
    }
    __shared__ uint32_t s_enc[4], s_nan[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        s_enc[wave] = enc;
        s_nan[wave] = nanf;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t e = s_enc[0], f = s_nan[0];
        for (int w = 1; w < 4; w++) {
            e = s_enc[w] < e ? s_enc[w] : e;
            f |= s_nan[w];
        }
        if (MIN) atomicMin(&flags[0], e);
        if (f) atomicOr(&flags[1], 1u);
    }
}

int vec_kind(const skr_mat* v, int64_t cols, const char* what, int* kind) {
    if (!v) {
        *kind = C_NONE;
        return SKR_OK;
    }
    SKR_REQUIRE(v->rows * v->cols == cols, "%s vector has %lld entries, matrix has %lld columns", what,
                (long long)(v->rows * v->cols), (long long)cols);
    SKR_REQUIRE(v->dtype == SKR_F32 || v->dtype == SKR_F64, "%s vector must be float32 or float64", what);
    *kind = v->dtype == SKR_F32 ? C_F32 : C_F64;
    return SKR_OK;
}

template <int CK, int SK, bool PRE, bool POST, bool WRITE, bool MIN>
void launch_elem(skr_ctx* ctx, const skr_mat* x, const skr_mat* center, const skr_mat* scale, float shift, float* y) {
    const int64_t total = x->rows * x->cols;
    int64_t blocks = (total / 4 + 255) / 256;
    blocks = std::max<int64_t>(1, std::min<int64_t>(blocks, (int64_t)ctx->num_cu * 8));
    hipLaunchKernelGGL((elementwise_kernel<CK, SK, PRE, POST, WRITE, MIN>), dim3((unsigned)blocks), dim3(256), 0,
                       ctx->stream, (const float*)x->data, x->rows, x->cols, center ? center->data : nullptr,
                       scale ? scale->data : nullptr, shift, y, ctx->d_flags);
}

template <int CK, int SK>
void dispatch_elem(skr_ctx* ctx, const skr_mat* x, const skr_mat* c, const skr_mat* s, bool pre, bool post, bool write,
                   bool mn, float shift, float* y) {
    if (!write) {  // min / nan scan only
        launch_elem<CK, SK, false, false, false, true>(ctx, x, c, s, shift, y);
        return;
    }
    if (pre && post) launch_elem<CK, SK, true, true, true, false>(ctx, x, c, s, shift, y);
    else if (pre) launch_elem<CK, SK, true, false, true, false>(ctx, x, c, s, shift, y);
    else if (post) launch_elem<CK, SK, false, true, true, false>(ctx, x, c, s, shift, y);
    else launch_elem<CK, SK, false, false, true, false>(ctx, x, c, s, shift, y);
    (void)mn;
}

int run_elem(skr_ctx* ctx, const skr_mat* x, const skr_mat* c, const skr_mat* s, bool pre, bool post, bool write,
             float shift, float* y) {
    int ck, sk;
    SKR_TRY(vec_kind(c, x->cols, "center", &ck));
    SKR_TRY(vec_kind(s, x->cols, "scale", &sk));
#define CASE(CK_, SK_) \
    if (ck == CK_ && sk == SK_) dispatch_elem<CK_, SK_>(ctx, x, c, s, pre, post, write, !write, shift, y)
    CASE(C_NONE, C_NONE);
    CASE(C_NONE, C_F32);
    CASE(C_NONE, C_F64);
    CASE(C_F32, C_NONE);
    CASE(C_F32, C_F32);
    CASE(C_F32, C_F64);
    CASE(C_F64, C_NONE);
    CASE(C_F64, C_F32);
    CASE(C_F64, C_F64);
#undef CASE
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

int reset_flags(skr_ctx* ctx) {
    // flags[0] = +inf in ordered encoding is 0xFF800000 -> use all ones (above every float), flags[1] = 0
    SKR_HIP(hipMemsetAsync(ctx->d_flags, 0xFF, 4, ctx->stream));
    SKR_HIP(hipMemsetAsync(ctx->d_flags + 1, 0, 4, ctx->stream));
    return SKR_OK;
}

int read_flags(skr_ctx* ctx) {
    SKR_HIP(hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    SKR_HIP(hipStreamSynchronize(ctx->stream));
    return SKR_OK;
}

int check_f32(const skr_ctx* ctx, const skr_mat* m, const char* what) {
    SKR_REQUIRE(m, "%s is NULL", what);
    SKR_REQUIRE(m->ctx == ctx, "%s belongs to a different ctx", what);
    SKR_REQUIRE(m->dtype == SKR_F32, "%s must be float32", what);
    return SKR_OK;
}

}  // namespace

// One launch of the column-sum kernel: plain (link empty), with the column minima (colmin != NULL: first pass only), or
// as a link of the chain across GPUs.
static int launch_colsum(skr_ctx* ctx, const skr_mat* x, const skr_mat* center, const skr_mat* center2, int square, skr_mat* acc,
                         skr_mat* colmin, const ChainLink& link) {
    SKR_REQUIRE(ctx, "ctx is NULL");
    SKR_TRY(check_f32(ctx, x, "x"));
    SKR_TRY(check_f32(ctx, acc, "acc"));
    SKR_REQUIRE(acc->rows * acc->cols == x->cols, "acc must hold one float per column");
    int ck;
    SKR_TRY(vec_kind(center, x->cols, "center", &ck));
    if (center2) {
        SKR_REQUIRE(square, "center2 is only used with square != 0");
        SKR_TRY(check_f32(ctx, center2, "center2"));
        SKR_REQUIRE(center2->rows * center2->cols == x->cols, "center2 must hold one float per column");
    }
    if (colmin) {
        SKR_TRY(check_f32(ctx, colmin, "colmin"));
        SKR_REQUIRE(!center && !square, "the column minima ride on the first pass (no centre, no square)");
        SKR_REQUIRE(colmin->rows == 4 && colmin->cols == x->cols, "colmin must be [4, %lld]", (long long)x->cols);
        if (x->rows == 0) return skr_set_error(SKR_ERR_UNSUPPORTED, "no rows: no column minima");
    }
    SKR_TRY(skr_activate(ctx));
    if (x->cols == 0) return SKR_OK;
    const unsigned grid = (unsigned)((x->cols + kColsPerWG - 1) / kColsPerWG);
    if (x->rows == 0) {  // no rows: acc stays as it is — but a link of the chain still passes the sums on
        if (link.carry_flag || link.next_data || link.n_result) {
            hipLaunchKernelGGL(chain_forward_kernel, dim3((grid + 3) / 4), dim3(64), 0, ctx->stream, x->cols, (float*)acc->data, link);
            SKR_HIP(hipGetLastError());
        }
        return SKR_OK;
    }
    const float* c2 = center2 ? (const float*)center2->data : nullptr;
    const void* c1 = center ? center->data : nullptr;
    float* cm = colmin ? (float*)colmin->data : nullptr;
    SkrProfScope prof(ctx, square ? "colsum_seq_sq" : "colsum_seq");
#define LAUNCH(CK_, SQ_, MIN_)                                                                                     \
    hipLaunchKernelGGL((colsum_seq_kernel<CK_, SQ_, MIN_>), dim3(grid), dim3(kWgThreads), 0, ctx->stream,         \
                       (const float*)x->data, x->rows, x->cols, c1, c2, (float*)acc->data, cm, link)
    if (colmin) {
        LAUNCH(C_NONE, false, true);
    } else if (square) {
        if (ck == C_NONE) LAUNCH(C_NONE, true, false);
        else if (ck == C_F32) LAUNCH(C_F32, true, false);
        else LAUNCH(C_F64, true, false);
    } else {
        if (ck == C_NONE) LAUNCH(C_NONE, false, false);
        else if (ck == C_F32) LAUNCH(C_F32, false, false);
        else LAUNCH(C_F64, false, false);
    }
#undef LAUNCH
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

extern "C" int skr_colsum_seq(skr_ctx* ctx, const skr_mat* x, const skr_mat* center, const skr_mat* center2,
                              int square, skr_mat* acc) {
    return launch_colsum(ctx, x, center, center2, square, acc, nullptr, ChainLink{});
}

// First pass of the column statistics with the column minima of the raw matrix as a by-product (see MINOUT above).
// colmin: float32 [4, cols]; min over its four rows = the column's minimum (NaN if the column holds a NaN).
extern "C" int skr_colsum_seq_colmin(skr_ctx* ctx, const skr_mat* x, skr_mat* acc, skr_mat* colmin) {
    SKR_REQUIRE(colmin, "colmin is NULL");
    return launch_colsum(ctx, x, nullptr, nullptr, 0, acc, colmin, ChainLink{});
}

// ---------------------------------------------------------------------------------------
// skr_chain: the mailboxes of the column-sum chain across GPUs (ChainLink above).  Every rank owns one block of
// UNCACHED device memory — carry box + result box, one float per column each, and an epoch word per strip for either —
// exports it as a HIP IPC handle and opens the other ranks' blocks; a kernel then stores straight into a peer's block.
// ---------------------------------------------------------------------------------------
struct skr_chain {
    skr_ctx* ctx = nullptr;
    int64_t cols_cap = 0, strips_cap = 0;
    char* block = nullptr;  // own mailbox
    size_t bytes = 0;
    int nranks = 1, rank = 0;
    char* peer[kMaxChainRanks] = {};     // every rank's mailbox as seen from here (peer[rank] == block)
    bool opened[kMaxChainRanks] = {};    // came from hipIpcOpenMemHandle (to be closed)
    uint32_t epoch = 0;
    uint32_t* d_err = nullptr;
    float* carry(char* b) const { return (float*)b; }
    float* result(char* b) const { return (float*)b + cols_cap; }
    uint32_t* carry_flag(char* b) const { return (uint32_t*)((float*)b + 2 * cols_cap); }
    uint32_t* result_flag(char* b) const { return carry_flag(b) + strips_cap; }
};

extern "C" int skr_chain_create(skr_ctx* ctx, int64_t cols_cap, skr_chain** out) {
    SKR_REQUIRE(ctx && out && cols_cap > 0, "bad argument");
    *out = nullptr;
    SKR_TRY(skr_activate(ctx));
    skr_chain* c = new skr_chain();
    c->ctx = ctx;
    c->cols_cap = (cols_cap + kColsPerWG - 1) / kColsPerWG * kColsPerWG;
    c->strips_cap = c->cols_cap / kColsPerWG;
    c->bytes = ((size_t)(2 * c->cols_cap + 2 * c->strips_cap) * 4 + 4095) & ~(size_t)4095;
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, c->bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        delete c;
        return skr_set_error(SKR_ERR_HIP, "hipExtMallocWithFlags(%zu bytes, uncached) for the chain mailbox: %s", c->bytes, hipGetErrorString(e));
    }
    c->block = (char*)p;
    e = hipMemsetAsync(c->block, 0, c->bytes, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(c->block);
        delete c;
        return skr_set_error(SKR_ERR_HIP, "clearing the chain mailbox: %s", hipGetErrorString(e));
    }
    c->d_err = ctx->d_flags + 20;  // one free word of the ctx flag block (0-7 kernels' flags, 8-15 tile queues, 32-63 all-reduce)
    c->peer[0] = c->block;
    *out = c;
    return SKR_OK;
}

extern "C" int skr_chain_export(skr_chain* c, char handle[64]) {
    SKR_REQUIRE(c && handle, "NULL argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "a HIP IPC memory handle is expected to be 64 bytes");
    SKR_TRY(skr_activate(c->ctx));
    hipIpcMemHandle_t h;
    SKR_HIP(hipIpcGetMemHandle(&h, c->block));
    memcpy(handle, &h, 64);
    return SKR_OK;
}

static void chain_close_peers(skr_chain* c) {
    for (int g = 0; g < kMaxChainRanks; g++) {
        if (c->opened[g] && c->peer[g]) (void)hipIpcCloseMemHandle(c->peer[g]);
        c->opened[g] = false;
        c->peer[g] = nullptr;
    }
}

// handles: [nranks][64] as exported by every rank (this rank's own entry is not opened)
extern "C" int skr_chain_connect(skr_chain* c, int nranks, int rank, const char* handles) {
    SKR_REQUIRE(c && handles, "NULL argument");
    SKR_REQUIRE(nranks >= 1 && nranks <= kMaxChainRanks && rank >= 0 && rank < nranks, "bad rank %d of %d (at most %d)", rank, nranks,
                kMaxChainRanks);
    SKR_TRY(skr_activate(c->ctx));
    chain_close_peers(c);
    for (int g = 0; g < nranks; g++) {
        if (g == rank) {
            c->peer[g] = c->block;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)g * 64, 64);
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            chain_close_peers(c);
            return skr_set_error(SKR_ERR_COMM, "hipIpcOpenMemHandle of rank %d's chain mailbox: %s", g, hipGetErrorString(e));
        }
        c->peer[g] = (char*)p;
        c->opened[g] = true;
    }
    c->nranks = nranks;
    c->rank = rank;
    return SKR_OK;
}

// the ranks of a single-process emulation (tools/chain_bench.py, tests): all[g] lives in this process
extern "C" int skr_chain_connect_local(skr_chain* c, int nranks, int rank, skr_chain* const* all) {
    SKR_REQUIRE(c && all && nranks >= 1 && nranks <= kMaxChainRanks && rank >= 0 && rank < nranks, "bad argument");
    chain_close_peers(c);
    for (int g = 0; g < nranks; g++) {
        SKR_REQUIRE(all[g] && all[g]->cols_cap == c->cols_cap, "rank %d's mailbox has another size", g);
        c->peer[g] = all[g]->block;
    }
    c->nranks = nranks;
    c->rank = rank;
    return SKR_OK;
}

extern "C" int skr_chain_free(skr_chain* c) {
    if (!c) return SKR_OK;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    chain_close_peers(c);
    if (c->block) (void)hipFree(c->block);
    delete c;
    return SKR_OK;
}

// One pass of the chain on this rank (kmer_counts.py:168,174 over all ranks' rows in global row order): the column-sum
// kernel as a link — it waits for rank - 1's sums inside the kernel and stores its own into rank + 1's mailbox — and, on
// every rank but the last, the copy of the finished sums from the result box into `acc`.  On return (stream order) acc
// holds the sums over ALL ranks' rows, the same bits on every rank.  Every rank must make the same sequence of calls.
// Test-only (SEEKR_CHAIN_HOST_WAIT under SEEKR_TEST_HOOKS): the host polls the epoch words of the first `strips` strips
// until all of them have reached `epoch`, so that the kernel launched next finds its mailbox full and never spins.
static int chain_host_wait(skr_chain* c, const uint32_t* flags, unsigned strips, uint32_t epoch) {
    skr_ctx* ctx = c->ctx;
    std::vector<uint32_t> h(strips);
    for (int tries = 0; tries < 600000; tries++) {  // ~ 30 s
        SKR_HIP(hipMemcpyAsync(h.data(), flags, (size_t)strips * 4, hipMemcpyDeviceToHost, ctx->comm_stream));
        SKR_HIP(hipStreamSynchronize(ctx->comm_stream));
        bool all = true;
        for (unsigned i = 0; i < strips && all; i++) all = (int32_t)(h[i] - epoch) >= 0;
        if (all) return SKR_OK;
        usleep(50);
    }
    return skr_set_error(SKR_ERR_COMM, "the chain mailbox was not filled within 30 s (epoch %u)", epoch);
}

static int chain_result(skr_chain* c, skr_mat* acc, int64_t cols) {
    skr_ctx* ctx = c->ctx;
    if (c->rank >= c->nranks - 1 || cols <= 0) return SKR_OK;
    const unsigned strips = (unsigned)((cols + kColsPerWG - 1) / kColsPerWG);
    if (ctx->knobs.chain_host_wait) SKR_TRY(chain_host_wait(c, c->result_flag(c->block), strips, c->epoch));
    SkrProfScope prof(ctx, "chain_result_wait");
    hipLaunchKernelGGL(chain_result_kernel, dim3((strips + 3) / 4), dim3(64), 0, ctx->stream, cols, (float*)acc->data,
                       (const float*)c->result(c->block), (const uint32_t*)c->result_flag(c->block), c->epoch, c->d_err);
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

// defer_result != 0 (single-process emulation of several ranks on ONE stream): the wait for the result box is left to a
// later skr_chain_result, issued once the last rank's link is in the stream too
extern "C" int skr_chain_result(skr_chain* c, skr_mat* acc) {
    SKR_REQUIRE(c && acc, "NULL argument");
    SKR_TRY(check_f32(c->ctx, acc, "acc"));
    SKR_TRY(skr_activate(c->ctx));
    return chain_result(c, acc, acc->rows * acc->cols);
}

extern "C" int skr_colsum_seq_chain(skr_chain* c, const skr_mat* x, const skr_mat* center, const skr_mat* center2, int square,
                                    skr_mat* acc, skr_mat* colmin, int defer_result) {
    SKR_REQUIRE(c && x && acc, "NULL argument");
    skr_ctx* ctx = c->ctx;
    SKR_REQUIRE(x->cols <= c->cols_cap, "the chain was created for at most %lld columns", (long long)c->cols_cap);
    SKR_TRY(skr_activate(ctx));
    const int P = c->nranks, g = c->rank, last = P - 1;
    c->epoch++;
    ChainLink link;
    link.epoch = c->epoch;
    link.err = c->d_err;
    if (g > 0) {
        link.carry_in = c->carry(c->block);
        link.carry_flag = c->carry_flag(c->block);
    }
    if (g < last) {
        link.next_data = c->carry(c->peer[g + 1]);
        link.next_flag = c->carry_flag(c->peer[g + 1]);
    } else {
        for (int r = 0; r < last; r++) {
            link.result_data[link.n_result] = c->result(c->peer[r]);
            link.result_flag[link.n_result] = c->result_flag(c->peer[r]);
            link.n_result++;
        }
    }
    if (ctx->knobs.chain_host_wait && g > 0 && x->cols > 0)
        SKR_TRY(chain_host_wait(c, link.carry_flag, (unsigned)((x->cols + kColsPerWG - 1) / kColsPerWG), c->epoch));
    SKR_TRY(launch_colsum(ctx, x, center, center2, square, acc, colmin, link));
    if (!defer_result) SKR_TRY(chain_result(c, acc, x->cols));
    return SKR_OK;
}

// 1 when a wait of the chain gave up since the last call (a peer never delivered); clears the mark.  Synchronises.
extern "C" int skr_chain_check(skr_chain* c, int* timed_out) {
    SKR_REQUIRE(c && timed_out, "NULL argument");
    SKR_TRY(skr_activate(c->ctx));
    uint32_t v = 0;
    SKR_HIP(hipMemcpyAsync(&v, c->d_err, 4, hipMemcpyDeviceToHost, c->ctx->stream));  // (never the NULL stream: pack.hip, upload_seqs)
    SKR_HIP(hipStreamSynchronize(c->ctx->stream));
    if (v) {
        SKR_HIP(hipMemsetAsync(c->d_err, 0, 4, c->ctx->stream));
        SKR_HIP(hipStreamSynchronize(c->ctx->stream));
    }
    *timed_out = v != 0;
    return SKR_OK;
}

extern "C" int skr_vec_finish(skr_ctx* ctx, skr_mat* v, int64_t n, int take_sqrt) {
    SKR_REQUIRE(ctx, "ctx is NULL");
    SKR_TRY(check_f32(ctx, v, "v"));
    SKR_TRY(skr_activate(ctx));
    const int64_t cols = v->rows * v->cols;
    if (cols == 0) return SKR_OK;
    hipLaunchKernelGGL(vec_finish_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, ctx->stream,
                       (float*)v->data, cols, (float)n, take_sqrt);
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

extern "C" int skr_min_nan(skr_ctx* ctx, const skr_mat* x, const skr_mat* center, const skr_mat* scale,
                           float* min_out, int* has_nan) {
    SKR_REQUIRE(ctx, "ctx is NULL");
    SKR_TRY(check_f32(ctx, x, "x"));
    SKR_TRY(skr_activate(ctx));
    SKR_TRY(reset_flags(ctx));
    if (x->rows * x->cols > 0) {
        SkrProfScope prof(ctx, "elementwise_min");
        SKR_TRY(run_elem(ctx, x, center, scale, false, false, /*write=*/false, 0.f, nullptr));
    }
    SKR_TRY(read_flags(ctx));
    const bool nan = ctx->h_flags[1] != 0;
    if (has_nan) *has_nan = nan ? 1 : 0;
    if (min_out) *min_out = nan ? NAN : unorder_bits(ctx->h_flags[0]);
    return SKR_OK;
}

extern "C" int skr_apply(skr_ctx* ctx, const skr_mat* x, int pre, const skr_mat* center, const skr_mat* scale,
                         int post, float shift, skr_mat* y, int* has_nan) {
    SKR_REQUIRE(ctx, "ctx is NULL");
    SKR_TRY(check_f32(ctx, x, "x"));
    SKR_TRY(check_f32(ctx, y, "y"));
    SKR_REQUIRE(x->rows == y->rows && x->cols == y->cols, "x and y shapes differ");
    SKR_TRY(skr_activate(ctx));
    SKR_TRY(reset_flags(ctx));
    if (x->rows * x->cols > 0) {
        SkrProfScope prof(ctx, "elementwise_apply");
        SKR_TRY(run_elem(ctx, x, center, scale, pre != 0, post != 0, /*write=*/true, shift, (float*)y->data));
    }
    if (has_nan) {
        SKR_TRY(read_flags(ctx));
        *has_nan = ctx->h_flags[1] != 0;
    }
    return SKR_OK;
}

extern "C" int skr_normalize(skr_ctx* ctx, skr_mat* x, int log2_mode, int mean_mode, const skr_mat* mean_vec,
                             int std_mode, const skr_mat* std_vec, skr_mat* mean_out, skr_mat* std_out,
                             int* has_nan) {
    SKR_REQUIRE(ctx, "ctx is NULL");
    SKR_TRY(check_f32(ctx, x, "x"));
    SKR_REQUIRE(log2_mode >= SKR_LOG2_NONE && log2_mode <= SKR_LOG2_POST, "bad log2 mode %d", log2_mode);
    SKR_REQUIRE(mean_mode >= 0 && mean_mode <= 2 && std_mode >= 0 && std_mode <= 2, "bad mean/std mode");
    SKR_REQUIRE(mean_mode != 2 || mean_vec, "mean_mode 2 needs mean_vec");
    SKR_REQUIRE(std_mode != 2 || std_vec, "std_mode 2 needs std_vec");
    SKR_REQUIRE(mean_mode != 1 || mean_out, "mean_mode 1 needs mean_out");
    SKR_REQUIRE(std_mode != 1 || std_out, "std_mode 1 needs std_out");
    SKR_TRY(skr_activate(ctx));
    if (has_nan) *has_nan = 0;
    const int64_t n = x->rows;
    // Log2.pre is applied eagerly (the statistics are statistics of log2(count + 1), :201-202)
    if (log2_mode == SKR_LOG2_PRE) SKR_TRY(skr_apply(ctx, x, 1, nullptr, nullptr, 0, 0.f, x, nullptr));
    // centre (:165-169): the matrix itself stays raw; the transform is replayed by later passes
    const skr_mat* center = nullptr;
    // Log2.post with a mean computed here and a scale that is computed (>= 0) or absent: the first pass brings the raw
    // column minima along and the minimum of z is looked for among their normalised values (skr_colsum_seq_colmin)
    skr_mat* colmin = nullptr;
    const bool want_colmin = log2_mode == SKR_LOG2_POST && mean_mode == 1 && std_mode != 2 && n > 0;
    if (mean_mode == 1) {
        SKR_TRY(check_f32(ctx, mean_out, "mean_out"));
        SKR_TRY(skr_mat_fill_zero(mean_out));
        if (want_colmin) {
            SKR_TRY(skr_mat_create(ctx, 4, x->cols, SKR_F32, &colmin));
            const int rc = skr_colsum_seq_colmin(ctx, x, mean_out, colmin);
            if (rc != SKR_OK) {
                skr_mat_free(colmin);
                return rc;
            }
        } else {
            SKR_TRY(skr_colsum_seq(ctx, x, nullptr, nullptr, 0, mean_out));
        }
        SKR_TRY(skr_vec_finish(ctx, mean_out, n, 0));
        center = mean_out;
    } else if (mean_mode == 2) {
        center = mean_vec;
    }
    // standardise (:171-175): np.std of the centred matrix = its own mean m', then squares
    const skr_mat* scale = nullptr;
    if (std_mode == 1) {
        SKR_TRY(check_f32(ctx, std_out, "std_out"));
        skr_mat* mprime = nullptr;
        SKR_TRY(skr_mat_create(ctx, 1, x->cols, SKR_F32, &mprime));
        int rc = skr_mat_fill_zero(mprime);
        if (rc == SKR_OK) rc = skr_colsum_seq(ctx, x, center, nullptr, 0, mprime);
        if (rc == SKR_OK) rc = skr_vec_finish(ctx, mprime, n, 0);
        if (rc == SKR_OK) rc = skr_mat_fill_zero(std_out);
        if (rc == SKR_OK) rc = skr_colsum_seq(ctx, x, center, mprime, 1, std_out);
        if (rc == SKR_OK) rc = skr_vec_finish(ctx, std_out, n, 1);
        skr_mat_free(mprime);
        if (rc != SKR_OK && colmin) skr_mat_free(colmin);
        SKR_TRY(rc);
        scale = std_out;
    } else if (std_mode == 2) {
        scale = std_vec;
    }
    float shift = 0.f;
    if (log2_mode == SKR_LOG2_POST) {
        float mn = 0.f;
        const int rc = skr_min_nan(ctx, colmin ? colmin : x, center, scale, &mn, nullptr);
        if (colmin) skr_mat_free(colmin);
        colmin = nullptr;
        SKR_TRY(rc);
        shift = fabsf(mn);  // NaN stays NaN (np.abs(np.min(...)), :208)
    }
    if (colmin) skr_mat_free(colmin);
    int nan_after_scale = 0;
    if (center || scale || log2_mode == SKR_LOG2_POST)
        SKR_TRY(skr_apply(ctx, x, 0, center, scale, log2_mode == SKR_LOG2_POST, shift, x,
                          (scale && has_nan) ? &nan_after_scale : nullptr));
    if (has_nan) *has_nan = nan_after_scale;
    return SKR_OK;
}

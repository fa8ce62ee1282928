// Internal declarations shared by the translation units of libseekr_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/seekr_hip.h"

// A/B switches of the measurements in DESIGN §4, read from the environment ONCE, when the ctx is created
// (INTEGRATION.md lists them); a launch never calls getenv.
struct SkrKnobs {
    bool gemm_persist = true;    // SEEKR_GEMM_PERSIST=0: one workgroup per tile instead of the persistent grid
    int gemm_chunk_tiles = 0;    // SEEKR_GEMM_CHUNK_TILES: k tiles per accumulator restart (0 = the operand's own choice)
    int gemm_reserve_cus = -1;   // SEEKR_GEMM_RESERVE_CUS: CUs left to RCCL's kernels (-1 = 8 with a communicator, else 0)
    int gemm_subtile = 4;        // SEEKR_GEMM_SUBTILE: order in which an XCD's CUs take the tiles of a super-tile (pearson_bf16.hip: tile_of_block)
    int gemm_epilogue = 3;       // SEEKR_GEMM_EPILOGUE: 0 = every tile through the general epilogue loop; 1 / 2 / 3 = the lean epilogue with 64- / 128- / 256-byte row runs (pearson_bf16.hip, round 6; 3 measured best)
    int gemm_wave_tile = 0;      // SEEKR_GEMM_WAVE_TILE=1: the 4-wave 128 x 128 wave-tile arm (libseekr_hip_diag.so only; tools/gemm_bench.py --diag-lib)
    int host_wait = 0;           // SEEKR_HOST_WAIT=spin|yield|block (1|2|3): hipSetDeviceFlags when the ctx is created; 0 = the runtime's own policy
    int count_percu = 0;         // SEEKR_COUNT_PERCU: cap on resident workgroups per CU (0 = none)
    int count_persist = 0;       // SEEKR_COUNT_PERSIST=1: persistent grid at k <= 6; 2: one workgroup per sequence at k = 7 too
    bool count_legacy = false;   // SEEKR_COUNT_LEGACY=1: the round-1 counting kernel
    int count_wps = 0;           // SEEKR_COUNT_WPS: waves per sequence (0 = by k)
    int64_t split_max_cols = 262144;    // SEEKR_SPLIT_MAX_COLS: widest row the split-fp16 contraction takes (k = 9; 16384 = rounds 1-3: k >= 8 on the fp32 kernel)
    bool count_generic_global = false;  // SEEKR_COUNT_GENERIC_GLOBAL=1: any-alphabet counting on the round-1 path (histogram in HBM)
    int count_generic_wgs = 0;          // SEEKR_COUNT_GENERIC_WGS=1|2: workgroups per CU of the any-alphabet LDS counter (0: chosen from the shape)
    bool count_k8_global = false;  // SEEKR_COUNT_K8_GLOBAL=1: k = 8 on the round-1 path (histogram in the output row, L2 atomics)
    int count_occ = 0;           // SEEKR_COUNT_OCC: cap on one-wave workgroups per CU of the non-persistent launch (0 = what the LDS allows)
    bool chain_host_wait = false;  // SEEKR_CHAIN_HOST_WAIT=1 (only under SEEKR_TEST_HOOKS=1): the column-sum chain waits for its
                                   // mailbox on the HOST before it launches — several ranks sharing one GPU (tests) cannot wait
                                   // inside a kernel: the waiting kernel fills the CUs and the one it waits for never starts
};

struct skr_ctx {
    int device = 0;
    SkrKnobs knobs;
    int diag_mode = 0;  // libseekr_hip_diag.so only (skr_gemm_diag_mode); the production library never reads it
    // kernels whose dynamic-LDS limit has been raised (hipFuncSetAttribute is called once per kernel and size, not per launch)
    std::map<const void*, int> lds_attr;
    std::map<std::pair<const void*, size_t>, int> occupancy;  // hipOccupancyMaxActiveBlocksPerMultiprocessor results
    hipStream_t stream = nullptr;       // compute stream
    hipStream_t comm_stream = nullptr;  // RCCL traffic, overlapped with compute
    // downloads that run beside the compute stream (skr_mat_download_at / skr_mat_write_rows_at: stripe s of r goes to
    // the host while stripe s + 1 is contracted); created on first use.  A "mark" is an event recorded on the compute
    // stream (skr_ctx_mark): the copy waits for that point only, not for what was enqueued after it.
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> marks;   // slot -> event (created once, recycled)
    std::vector<int> free_marks;
    std::vector<uint8_t> mark_live;
    void* h_copy[2] = {nullptr, nullptr};  // pinned staging of skr_mat_write_rows_at
    size_t h_copy_bytes = 0;
    int num_cu = 256;
    // small device scratch: [0] encoded min, [1] nan flag, [2] error flag, ...
    uint32_t* d_flags = nullptr;
    uint32_t* h_flags = nullptr;  // pinned mirror
    double* d_recip = nullptr;    // float64 reciprocals of a scale vector (skr_operand_fill), grown on demand
    size_t d_recip_len = 0;
    void* d_np_plan = nullptr;    // numpy's pairwise summation unrolled for np_plan_cols columns (operand.hip: NpPlan)
    int64_t np_plan_cols = 0;
    // workspaces owned by the ctx and grown on demand
    void* ws = nullptr;
    size_t ws_bytes = 0;
    // pinned host staging for small index tables uploaded asynchronously (skr_ctx_pinned): `pin_done` marks the end
    // of the last copy out of it, so the next user waits for that copy only, never for the stream
    void* h_pin = nullptr;
    size_t h_pin_bytes = 0;
    hipEvent_t pin_done = nullptr;
    // profiling
    bool prof = false;
    struct ProfRec {
        std::string name;
        hipEvent_t start, stop;
    };
    std::vector<ProfRec> prof_recs;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> event_pool;
    // RCCL (loaded lazily with dlopen; see comm.cpp)
    void* comm = nullptr;
    int nranks = 1, rank = 0;
    // tickets: completion events of exchanges on comm_stream.  Slots are recycled (the events too), so a long-running
    // loop neither leaks events nor grows the table; a ticket id carries the slot's generation.
    struct Ticket {
        hipEvent_t ev = nullptr;
        uint32_t gen = 0;
        bool live = false;
        bool vec = false;  // a statistic vector of the column-sum chain (booked apart from the operand shifts)
    };
    std::vector<Ticket> tickets;
    std::vector<int> free_tickets;
    // small device blocks (statistic vectors, a few KiB) handed back by skr_mat_free and reused by skr_mat_create:
    // no hipFree / hipMalloc, so no host synchronisation per step.  `comm_done` orders the next user (compute stream)
    // behind anything the communication stream may still read from the block.
    struct SmallBlock {
        void* p = nullptr;
        size_t bytes = 0;
        hipEvent_t comm_done = nullptr;
    };
    std::vector<SmallBlock> small_blocks;
    // a matrix may be freed by another host thread than the one driving this ctx (Python's collector runs where it
    // likes; seekr_amd/multi.py has one thread per GPU): the list above is the only ctx state a free touches
    std::mutex small_lock;
};

struct skr_mat {
    skr_ctx* ctx = nullptr;
    int64_t rows = 0, cols = 0;
    int dtype = SKR_F32;
    void* data = nullptr;
    bool owner = true;  // views created by skr_mat_view do not own `data`
    size_t bytes() const { return (size_t)rows * (size_t)cols * elem(); }
    size_t elem() const { return dtype == SKR_F64 ? 8 : 4; }
};

struct skr_seqs {
    skr_ctx* ctx = nullptr;
    int64_t n = 0;
    int64_t total_bases = 0;
    int64_t max_len = 0;
    int64_t n_words = 0;       // packed 2-bit words (16 bases each), per-sequence word aligned
    int64_t n_mask_words = 0;  // validity bit words (32 bases each), only for sequences with invalid bases
    uint32_t* d_packed = nullptr;
    int64_t* d_word_off = nullptr;  // [n+1]
    int64_t* d_len = nullptr;       // [n]
    uint32_t* d_mask = nullptr;
    int64_t* d_mask_off = nullptr;  // [n], -1 when the sequence is all-alphabet
    std::vector<int64_t> h_len;
    std::string headers;  // '\n' joined (FASTA input only)
};

int skr_set_error(int code, const char* fmt, ...);

#define SKR_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return skr_set_error(e_ == hipErrorOutOfMemory ? SKR_ERR_NOMEM : SKR_ERR_HIP,      \
                                 "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                                 __LINE__);                                                    \
    } while (0)

#define SKR_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) return skr_set_error(SKR_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define SKR_TRY(call)             \
    do {                          \
        int rc_ = (call);         \
        if (rc_ != SKR_OK) return rc_; \
    } while (0)

// Scoped kernel timer: records a start/stop event pair on the ctx stream (or the given one: the communication
// stream's transfers) when profiling is on.
struct SkrProfScope {
    skr_ctx* ctx;
    int idx = -1;
    hipStream_t stream;
    SkrProfScope(skr_ctx* c, const char* name, hipStream_t on = nullptr);
    ~SkrProfScope();
};

int skr_ctx_workspace(skr_ctx* ctx, size_t bytes, void** out);
// raise a kernel's dynamic shared memory limit to at least `bytes` (first launch of that kernel on this ctx only)
int skr_kernel_lds(skr_ctx* ctx, const void* kern, size_t bytes);
// pinned host buffer of at least `bytes`, free to be overwritten (the previous asynchronous copy out of it has
// finished); after enqueueing a copy from it on ctx->stream call skr_ctx_pinned_used
int skr_ctx_pinned(skr_ctx* ctx, size_t bytes, void** out);
int skr_ctx_pinned_used(skr_ctx* ctx);
// GEMM launchers (pearson.hip, pearson_bf16.hip), used by operand.hip
int skr_launch_gemm_f32(skr_ctx* ctx, const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t Kp,
                        int64_t lda, int64_t ldb, int64_t ldc, int64_t K, int symmetric);
int skr_launch_gemm_f64(skr_ctx* ctx, const double* A, const double* B, double* C, int64_t M, int64_t N, int64_t K,
                        int64_t lda, int64_t ldb, int64_t ldc, double kdiv, int symmetric);
int skr_launch_gemm_split(skr_ctx* ctx, int precision, const void* As, const void* Bs, float* C, int64_t M, int64_t N,
                          int64_t kt, int64_t ldc, float kdiv, int mode, float* Ct, int64_t ldct, bool coherent = false);

// k tiles per accumulator restart of the split contraction (pearson_bf16.hip: launch16): 128 (4 096 columns), 32 for
// operands whose rows are mostly one repeated value; SEEKR_GEMM_CHUNK_TILES overrides both.  One rule for the
// contraction and for every caller that must know whether a launch will have more than one k chunk (fused_edges.hip).
inline int64_t skr_gemm_chunk_tiles(const skr_ctx* ctx, bool coherent) {
    return ctx->knobs.gemm_chunk_tiles ? ctx->knobs.gemm_chunk_tiles : (coherent ? 32 : 128);
}

// Where the EDGES mode of the split contraction appends the cells that survive a threshold (pearson_bf16.hip).
struct SkrEdgeSink {
    unsigned long long* keys;   // row << 32 | column (global indices)
    float* vals;
    unsigned long long* count;  // cells found (may exceed cap: then the list is incomplete and the caller retries)
    unsigned long long cap;
    int64_t row_global0, col_global0;
    float cutoff;
    int upper;
};
int skr_launch_gemm_edges(skr_ctx* ctx, int precision, const void* As, const void* Bs, float* C, int64_t M, int64_t N,
                          int64_t kt, int64_t ldc, float kdiv, bool coherent, const SkrEdgeSink& sink);

// A Pearson operand prepared for the matrix cores.  Storage is kt*32 float-sized words per row
// (kt = ceil(cols/32)) in both layouts: zero-padded float32, or split-interleaved 16-bit halves
// (per 32-wide k tile: 32 hi then 32 lo = one 128-byte line).
struct skr_operand {
    skr_ctx* ctx = nullptr;
    int64_t rows = 0, cols = 0, kt = 0;
    int precision = SKR_PREC_FP32;
    int created_precision = SKR_PREC_FP32;  // what skr_operand_create was asked for (a fill may route an f16f8 operand back to f16x3)
    bool x8_routed_back = false;  // ... and did so for the rows it saw LAST: the next fill tries the f16f8 layout again (not after skr_operand_adopt_layout: an explicit choice)
    int kind = 0;  // 0 = float32 padded, 1 = bf16 halves, 2 = fp16 halves
    void* data = nullptr;
    float scale = 1.f;        // stored values = z * scale (fp16 halves only; a function of cols)
    float* diag = nullptr;    // [rows] <z_i, z_i>/K, written by skr_operand_fill (not exchanged between GPUs)
    bool diag_valid = false;  // false for buffers that only ever received rows from a peer
    bool coherent = false;    // rows are mostly one repeated value: the contraction restarts its accumulators twice as often
    bool owner = true;
    // kind 3 (the opt-in SKR_PREC_F16F8 layout) only: the largest |row mean| of what the fp8 copies lose and of lo over the
    // rows filled into this operand (dh = hi - 128 h8, lo, dl = lo - l8 / 16; operand.hip, X8), and the allocation the rows
    // live in — two operands of different allocations are multiplied only if the products of their means bound the error
    float x8_stat[3] = {0.f, 0.f, 0.f};
    const void* x8_root = nullptr;
    size_t row_bytes() const { return (size_t)kt * 128; }
};
// error of a cell of r from the row means alone when rows of a meet rows of b in the f16f8 layout (see operand.hip)
inline double skr_x8_pair_bound(const skr_operand* a, const skr_operand* b) {
    const double Da = a->x8_stat[0], La = a->x8_stat[1], la = a->x8_stat[2];
    const double Db = b->x8_stat[0], Lb = b->x8_stat[1], lb = b->x8_stat[2];
    return (Da * (Lb + lb) + (Da + La) * lb + Db * (La + la) + (Db + Lb) * la) / ((double)a->scale * b->scale);
}
constexpr double kX8MeansLimit = 0.6 * 2e-6;  // of the bar at r = 0
int skr_x8_pair_check(const skr_operand* a, const skr_operand* b);  // operand.hip
int skr_activate(const skr_ctx* ctx);
// the copy stream, made to wait for `mark` (skr_ctx_mark; consumed here) or, mark < 0, for everything enqueued on the
// compute stream so far
int skr_copy_stream_after(skr_ctx* ctx, int64_t mark, hipStream_t* out);

// float32 log2 rounded from a float64 evaluation: correctly rounded except for near-ties of the
// f64 result, which is the closest a device can get to numpy's log2 (SVML / libm are correctly
// rounded for >90 % of inputs; ocml's f32 log2f is systematically 1 ulp off on many of the
// discrete per-kb count values, which shifts the float32 column statistics of Log2.pre by >1e-5).
__device__ __forceinline__ float skr_log2_cr(float x) { return (float)log2((double)x); }
// Log2.post is the last step (no statistics are taken of its output), so the <= 1 ulp float32
// log2 is enough there and an order of magnitude cheaper than the float64 evaluation.
__device__ __forceinline__ float skr_log2_fast(float x) { return log2f(x); }
// log2(u + 1) of Log2.post.  The float32 sum u + 1 is zero or at least 2^-24 in magnitude — never subnormal — so the
// scaling log2f wraps around v_log_f32 for subnormal arguments can never act: the bare instruction returns the same
// bits (five instructions fewer per cell in kernels that are bound by their instruction count).
__device__ __forceinline__ float skr_log2_of_sum1(float u) { return __builtin_amdgcn_logf(__fadd_rn(u, 1.0f)); }

// K7 (fast path) — Pearson contraction on the bf16 matrix cores with split operands.
//
// Every float32 z is split as z = hi + lo + O(2^-17 |z|) with hi = bf16(z), lo = bf16(z - hi).
// r = sum_k z1 z2 is then accumulated in float32 from NPROD bf16 products per k:
//     NPROD = 3 : hi*hi + hi*lo + lo*hi            (|error| <= ~3e-6 on r ~ 1, ~2e-7 rms elsewhere)
//     NPROD = 4 : ... + lo*lo                      (error ~4e-7 = what float32 BLAS gives)
//     NPROD = 2 : "f16f8" (round 4, opt-in): hi*hi on the fp16 MFMA, and the two cross products hi*lo + lo*hi as ONE
//                 block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3) into the same accumulator: two
//                 product-units per k instead of three.  Operand layout "H / X lines" (operand.hip, kind 3): per 64
//                 columns one 128-byte line of 64 fp16 hi values and one of 64 + 64 fp8 copies of hi / 128 and lo x 16;
//                 the k loop alternates H stages (two 16x16x32 fp16 MFMAs per accumulator tile) and X stages (one
//                 16x16x128 fp8 MFMA: A = [hi8 | lo8], B = [lo8 | hi8], block scales 2^3 x 2^0 put it in the
//                 accumulator's units).  Same staging, same 128 bytes per row and line, same LDS reads.
// on v_mfma_f32_16x16x32_bf16 / _f16, which run 16x the f32-input MFMA rate; a single bf16
// product (2.7e-4) is nowhere near the 1e-5 parity bar.
//
// Operand layout ("split-interleaved", produced by operand.hip): for every row and every
// 32-wide k tile, 32 hi values followed by 32 lo values — one 128-byte line — so that staging a
// k tile of a row is one full cache line and a 1-KiB LDS-DMA piece covers 8 rows.
//
// Geometry: block tile 256 x 256, BK = 32, LDS 2 stages x (256+256) rows x 128 B = 128 KiB, one
// workgroup per CU; 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 (8 x 4 MFMA tiles of 16 x 16,
// 128 accumulator registers), two waves per SIMD cover each other's LDS waits.  Per 32-k tile a
// wave issues 24 ds_read_b128 and 32*NPROD MFMAs of 16 cycles.  (The 32x32x16 shape has the same
// LDS bytes per flop but the chip holds a lower clock on it: -13 % measured; see
// MI355X_MICROARCH.md, DVFS give-back.)
// The LDS image is lane-linear (LDS-DMA), so the bank swizzle chunk ^= (row >> 1) & 7 is applied
// on the source address and again on the read (same involution), as in the fp32 kernel.
//
// Output modes: PLAIN writes C; SELF (a == b) computes the tiles on and above the diagonal and
// mirrors them into the same matrix; CROSS writes C and the transposed block into a second
// matrix Ct — r(h,g) = r(g,h)^T costs stores, not a second contraction (distributed half-ring).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

namespace {

constexpr int TM = 256, TN = 256;
constexpr int kRowBytes = 128;                       // one k tile of one row: 32 hi + 32 lo bf16
constexpr int kStageBytes = (TM + TN) * kRowBytes;   // 64 KiB
enum { PLAIN = 0, SELF = 1, CROSS = 2, EDGES = 3, LOWER = 4 };
// LOWER: a PLAIN block whose cells carry the bits the MIRROR of the swapped product would: C[i, j] = what PLAIN(b, a) puts
// at [j, i].  Per accumulator the products enter in the order lo x hi, hi x lo, hi x hi (A's half named first), so a_i . b_j
// and b_j . a_i differ in their last bits; SELF keeps the value computed with the row of the SMALLER index as A for both
// cells of a pair.  A row stripe of a self-comparison (r larger than the HBM, or the rows of one GPU of several:
// skr_pearson_gemm_op_rows) reproduces those bits with LOWER left of its diagonal block — the two cross products swap
// places, nothing else changes — SELF on it and PLAIN right of it.

// EDGES mode: the epilogue does not store the tile; cells that kmer_leiden.py:94-96 would leave non-zero
// (`ld_sim[ld_sim < cutoff] = 0; np.fill_diagonal(ld_sim, 0)`: kept iff !(v < cutoff), v != 0, off the diagonal; with
// `upper` only column > row) are appended to a list — one ballot per accumulator register, one atomic per wave and
// register that holds a hit — in no particular order; the host sorts the list by (row, column).
using EdgeSink = SkrEdgeSink;  // common.hpp
template <typename T>
using vec8 = T __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

// Tile order.  Blocks are dealt round-robin to the 8 XCDs (b % 8 labels the XCD group); the i-th
// block of an XCD is placed so that 32 consecutive ones form an 8 x 4 sub-tile (sharing 8 A and
// 4 B panels in that XCD's L2) and the 8 sub-tiles of the 8 XCDs form one 16 x 16 super-tile
// whose 32 panels (~134 MB at K=4096) fit the 256 MB Infinity Cache.  Speed only, never
// correctness: blocks that fall outside the matrix (or below the diagonal when SYM) exit.
// `shape` (SEEKR_GEMM_SUBTILE, an A/B knob): which 32 tiles of the super-tile an XCD works on at a time and
// in which order its CUs take them — 4 (default since round 3): 8 x 4, tile COLUMN fastest; 0: 8 x 4, tile row fastest
// (rounds 1-2); 1: the same region as two 4 x 4 halves; 2: 16 x 2; 3: 4 x 8; 5 / 6: Morton order.  The 32 CUs of an XCD
// drift apart by more k tiles than its 4 MB L2 holds, so what counts is which tiles START together: measured at 50 000
// rows (PMC, profiles/r3_subtile_shapes.txt) L2 hit rate 48 % (0) / 63 % (1) / 66 % (4), HBM-side traffic 89.6 / 61.5 /
// 55.3 GB per launch (7.7 / 5.3 / 4.75 x algorithmic), launch 22.3 / 21.9 / 21.8 ms.
__device__ __forceinline__ bool tile_of_block(int64_t bid, int64_t super_n, int64_t tiles_m, int64_t tiles_n,
                                              int64_t* tm, int64_t* tn, int shape = 0) {
    const int64_t xcd = bid & 7, i = bid >> 3;
    const int64_t s = i >> 5, j = i & 31;
    const int64_t sr = s / super_n, sc = s % super_n;
    if (shape == 1) {
        const int64_t sub = 2 * xcd + (j >> 4);  // 16 sub-tiles of 4 x 4
        *tm = sr * 16 + (sub & 3) * 4 + (j & 3);
        *tn = sc * 16 + (sub >> 2) * 4 + ((j >> 2) & 3);
    } else if (shape == 2) {
        *tm = sr * 16 + (j & 15);
        *tn = sc * 16 + xcd * 2 + (j >> 4);
    } else if (shape == 3) {  // 4 x 8, rows fastest
        *tm = sr * 16 + (xcd >> 1) * 4 + (j & 3);
        *tn = sc * 16 + (xcd & 1) * 8 + (j >> 2);
    } else if (shape == 4) {  // 8 x 4 like the default, but the tile COLUMN fastest
        *tm = sr * 16 + (xcd & 1) * 8 + (j >> 2);
        *tn = sc * 16 + (xcd >> 1) * 4 + (j & 3);
    } else if (shape == 5) {  // the default 8 x 4 region in Morton (Z) order: neighbours in slot order share panels both ways
        *tm = sr * 16 + (xcd & 1) * 8 + ((j & 1) | ((j >> 1) & 2) | ((j >> 2) & 4));
        *tn = sc * 16 + (xcd >> 1) * 4 + (((j >> 1) & 1) | ((j >> 2) & 2));
    } else if (shape == 6) {  // two 4 x 4 halves like 1, each in Morton order
        const int64_t sub = 2 * xcd + (j >> 4);
        *tm = sr * 16 + (sub & 3) * 4 + ((j & 1) | ((j >> 1) & 2));
        *tn = sc * 16 + (sub >> 2) * 4 + (((j >> 1) & 1) | ((j >> 2) & 2));
    } else {
        *tm = sr * 16 + (xcd & 1) * 8 + (j & 7);
        *tn = sc * 16 + (xcd >> 1) * 4 + (j >> 3);
    }
    return *tm < tiles_m && *tn < tiles_n;
}

// T = __bf16 (8-bit significand halves) or _Float16 (11-bit halves: hi + lo carry 22 bits, so the
// three-product sum is float32-grade; |z| <= sqrt(K) << 65504 and fp16 subnormals are kept).
// Lane l holds A[row l&15][k = 8(l>>4)+j] / B likewise; D: col = l&15, row = 4(l>>4)+e.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma16x16(vec8<__bf16> a, vec8<__bf16> b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4v mfma16x16(vec8<_Float16> a, vec8<_Float16> b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// The same instruction with the accumulator PINNED in the accumulation registers ("+a"): the 4-wave geometry keeps 256
// accumulators per lane, which is the whole AGPR half of a one-wave-per-SIMD register file; left to the register
// allocator (builtin form) hipcc keeps part of them in VGPRs and shuffles them through v_accvgpr_write / scratch inside
// the k loop (264 scratch accesses and 660 accvgpr moves per two k tiles).  hipcc does not model what is inside the
// string (cdna_hip_programming.md §5.7): consecutive statements here never touch the same accumulator (an accumulate
// chain would need no wait state either), operands come from ds_reads the compiler does wait for, and the reads of the
// accumulators after the k loop are fenced by hand (mfma_drain).
__device__ __forceinline__ void mfma16x16_pinned(f32x4v& c, vec8<__bf16> a, vec8<__bf16> b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16x16_pinned(f32x4v& c, vec8<_Float16> a, vec8<_Float16> b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// lane <- lane ^ 1 (CTRL 0xB1 = quad_perm [1,0,3,2]) / lane ^ 2 (0x4E = [2,3,0,1]) inside each quad of lanes: one DPP move
template <int CTRL>
__device__ __forceinline__ float quad_perm(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
// an MFMA's result may be read by anything but the next MFMA's C operand only 12+ wait states after issue (8-pass XDL)
__device__ __forceinline__ void mfma_drain() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// PERSIST: one workgroup per CU for the whole launch.  Tile slots are handed out by eight counters,
// one per XCD group (slot i of group x is block id 8 i + x of the order above, so an XCD keeps
// walking 8 x 4 sub-tiles in its own L2); a workgroup drains the queue of the XCD it runs on
// (HW_REG_XCC_ID) and then helps with the others, so every slot is processed exactly once whatever
// the placement.  What this buys over one workgroup per tile: the stores of tile t drain while the
// main loop of tile t+1 runs (with one 128 KiB workgroup per CU nothing else overlaps them), and
// the slots a self-comparison skips cost one atomic instead of a workgroup launch.
// DIAG (instantiated only with -DSEEKR_DIAG, i.e. in libseekr_hip_diag.so for tools/gemm_diag.py): lane 0 of each workgroup stamps
// s_memtime (shader cycles) and s_memrealtime (100 MHz) around the k loop and the epilogue of every tile into a buffer of
// its own (MI355X_MICROARCH.md, DVFS give-back item 6); no output value depends on a stamp.
// WAVES = 8 (default): 2 x 4 waves, wave tile 128 x 64, two waves per SIMD.  WAVES = 4 (the A/B arm of VERDICT r2 #3,
// SEEKR_GEMM_WAVE_TILE=1): 2 x 2 waves, wave tile 128 x 128, ONE wave per SIMD with the whole 512-register file (256
// accumulators); a wave then reads 32 instead of 24 fragments per k tile but there are half as many waves: 128 instead of
// 192 KiB of ds_read per CU and k tile.  With nobody to cover its LDS latency the k loop is software-pipelined by hand
// (below); per accumulator the products are added in the same order, so r is the same bits.
template <typename T, int NPROD, int MODE, bool PERSIST, bool DIAG = false, int WAVES = 8>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void pearson_gemm_split16_kernel(
    const T* __restrict__ A, const T* __restrict__ B, float* __restrict__ C, float* __restrict__ Ct,
    int64_t M, int64_t N, int64_t kt, int64_t ldc, int64_t ldct, float kdiv, int64_t tiles_m, int64_t tiles_n,
    int64_t super_n, uint32_t* __restrict__ queues, int64_t slots_per_queue, int64_t pitch_tiles, int flags,
    unsigned long long* __restrict__ diag, const EdgeSink es) {
    const int accumulate = flags & 1;  // later k chunks add to what the earlier ones left in C; bit 1: last chunk; bits 2-3: lean epilogue (0 = off, 1-3 = 64 / 128 / 256-byte row runs); bits 8..: tile order shape
    static_assert(WAVES == 8 || (WAVES == 4 && NPROD == 3), "4-wave geometry: three products only");
    static_assert(NPROD != 2 || (std::is_same<T, _Float16>::value && WAVES == 8), "f16f8: fp16 hi lines, 8-wave geometry");
    constexpr int WN = WAVES == 8 ? 4 : 2, MT = 8, NT = WAVES == 8 ? 4 : 8, PP = 32 / WAVES;  // waves as 2 x WN, wave tile 128 x 16 NT
    constexpr int WTN = NT * 16;                                                                // wave tile width
    constexpr bool SYM = MODE == SELF;
    constexpr bool SWAPPED = MODE == LOWER;  // the roles of A and B in the order of the products (comment at the enum)
    static_assert(!SWAPPED || WAVES == 8, "LOWER exists for the 8-wave geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int64_t s_bid;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int home = PERSIST ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7) : 0;  // HW_REG_XCC_ID[3:0]
    int helping = 0;  // queues tried so far: home, home+1, ...
  for (;;) {
    int64_t bid = blockIdx.x;
    if (PERSIST) {
        if (tid == 0) {
            int64_t got = -1;
            while (helping < 8) {
                const int q = (home + helping) & 7;
                const uint32_t i = atomicAdd(&queues[q], 1u);
                if ((int64_t)i < slots_per_queue) {
                    got = ((int64_t)i << 3) | q;
                    break;
                }
                helping++;
            }
            s_bid = got;
        }
        __syncthreads();
        bid = s_bid;
        __syncthreads();  // s_bid may be rewritten as soon as this slot turns out to be empty
        if (bid < 0) return;
    }
    int64_t tm, tn;
    if (!tile_of_block(bid, super_n, tiles_m, tiles_n, &tm, &tn, flags >> 8) || (SYM && tn < tm)) {  // outside, or the mirror writes it
        if (PERSIST) continue;
        return;
    }
    unsigned long long st[6];
    if (DIAG) {
        st[0] = __builtin_amdgcn_s_memtime();
        st[1] = __builtin_amdgcn_s_memrealtime();
    }
    const int64_t row_base = tm * TM, col_base = tn * TN;
    // K x scale^2 is a power of two for every 4-letter alphabet: the division is then an exact multiplication
    // (ten instructions fewer per cell of the epilogue); any other divisor is divided by, as np.inner(...)/K is
    const float rk = (__float_as_uint(kdiv) & 0x007FFFFFu) == 0 && kdiv > 0x1.0p-100f && kdiv < 0x1.0p100f ? 1.0f / kdiv : 0.f;
    const int64_t pitch = pitch_tiles * 64;  // elements per operand row; kt may be a chunk of it

    // staging: wave w moves pieces PP*w .. PP*w+PP-1 (8 rows x one 128-byte line each) of the A tile
    // and of the B tile.  Uniform tile base (SGPRs) + 32-bit per-lane byte offset (one VGPR per piece).
    const char* a_tile = reinterpret_cast<const char*>(A + (size_t)row_base * pitch);
    const char* b_tile = reinterpret_cast<const char*>(B + (size_t)col_base * pitch);
    uint32_t a_voff[PP], b_voff[PP];
#pragma unroll
    for (int p = 0; p < PP; p++) {
        const int row = (wave * PP + p) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int64_t ra = std::min<int64_t>(row, M - 1 - row_base);  // rows past the end re-read the last one
        const int64_t rb = std::min<int64_t>(row, N - 1 - col_base);
        a_voff[p] = (uint32_t)(ra * pitch * 2 + chunk * 16);
        b_voff[p] = (uint32_t)(rb * pitch * 2 + chunk * 16);
    }
    auto stage = [&](int buf, int64_t tile, bool half = false) {  // half: diagnostic only (a timing experiment, below)
        char* abase = smem + buf * kStageBytes;
        char* bbase = abase + TM * kRowBytes;
        const uint32_t toff = (uint32_t)tile * kRowBytes;  // folded into the 32-bit lane offset: saddr + voffset form
#pragma unroll
        for (int p = 0; p < PP; p++) {
            if (DIAG && half && p >= PP / 2) break;
            lds_dma16(a_tile + (a_voff[p] + toff), abase + (wave * PP + p) * 1024);
            lds_dma16(b_tile + (b_voff[p] + toff), bbase + (wave * PP + p) * 1024);
        }
    };

    const int q = lane >> 4;  // k quarter: this lane's fragment is k = 8q .. 8q+7 of the 32-k tile
    int a_off[MT], a_swz[MT], b_off[NT], b_swz[NT];
#pragma unroll
    for (int t = 0; t < MT; t++) {
        const int ra = wm * 128 + t * 16 + (lane & 15);
        a_off[t] = ra * kRowBytes;
        a_swz[t] = (ra >> 1) & 7;
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int rb = wn * WTN + t * 16 + (lane & 15);
        b_off[t] = TM * kRowBytes + rb * kRowBytes;
        b_swz[t] = (rb >> 1) & 7;
    }

    f32x4v acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[i][j][e] = 0.f;

    // Two LDS stages: tile t+1 streams in by LDS-DMA while tile t is consumed; one barrier per k
    // tile.  The loop runs at the clock the chip grants an MFMA-dense body (1.9-1.95 GHz effective,
    // MI355X_MICROARCH.md "DVFS give-back"): a four-phase software pipeline that reads the next
    // phase's fragments under the current phase's MFMAs and moves the barrier before the last
    // phase (no MFMA ever waits on a fresh LDS read) ran in the same 25.1 vs 25.2 ms, so the
    // simple form stays.
    int cur = 0;
    stage(0, 0);
    if (WAVES == 4 && kt > 1) stage(1, 1);  // the 4-wave loop keeps two stages in flight (below)
    __syncthreads();
    if (DIAG) st[2] = __builtin_amdgcn_s_memtime();
    const bool no_dma = DIAG && (diag[1] & 1);  // diagnostic only: k loop without its staging traffic (results meaningless)
    const bool same_tile = DIAG && (diag[1] & 2);  // diagnostic only: every stage re-loads k tile 0 (served by the nearest cache)
    const bool two_units = DIAG && (diag[1] & 8);  // diagnostic only: TIMING CEILING of a two-product-unit split (results meaningless)
    if (DIAG && WAVES == 8 && two_units) {
        // VERDICT r3 #5, measured before anything is built around it: the two cross products (lo x hi, hi x lo: two
        // 16-bit MFMAs per k tile and accumulator tile) replaced by ONE v_mfma_i32_16x16x64_i8 — the cycles of one 16-bit
        // MFMA (MI355X_MICROARCH.md: I8 16x16x64 = the cycles of the BF16 form at twice the K) — fed with the bytes of
        // the lo fragments, into the same accumulator registers.  Same staging, same LDS reads, same register
        // pressure as an fp8-cross variant would have, 64 instead of 96 MFMA slots per wave and k tile: what the k
        // loop would cost with 2 instead of 3 product-units per k.
        typedef int i32x4v __attribute__((ext_vector_type(4)));
        for (int64_t t = 0; t < kt; t++) {
            if (t + 1 < kt) stage(cur ^ 1, t + 1);
            const char* base = smem + cur * kStageBytes;
            vec8<T> ahi[MT], bhi[NT];
            i32x4v a8[MT], b8[NT];
#pragma unroll
            for (int i = 0; i < NT; i++) {
                bhi[i] = *reinterpret_cast<const vec8<T>*>(base + b_off[i] + ((q ^ b_swz[i]) << 4));
                b8[i] = *reinterpret_cast<const i32x4v*>(base + b_off[i] + (((4 + q) ^ b_swz[i]) << 4));
            }
#pragma unroll
            for (int i = 0; i < MT; i++) {
                ahi[i] = *reinterpret_cast<const vec8<T>*>(base + a_off[i] + ((q ^ a_swz[i]) << 4));
                a8[i] = *reinterpret_cast<const i32x4v*>(base + a_off[i] + (((4 + q) ^ a_swz[i]) << 4));
            }
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    i32x4v ci = __builtin_bit_cast(i32x4v, acc[mt][nt]);
                    ci = __builtin_amdgcn_mfma_i32_16x16x64_i8(a8[mt], b8[nt], ci, 0, 0, 0);
                    acc[mt][nt] = mfma16x16(ahi[mt], bhi[nt], __builtin_bit_cast(f32x4v, ci));
                }
            __syncthreads();
            cur ^= 1;
        }
    } else if constexpr (WAVES == 8 && NPROD == 2) {
        // f16f8: lines alternate H (64 fp16 hi values of 64 columns) and X ([64 x hi8 | 64 x lo8] of the same columns); kt
        // (lines) is even for every width this layout exists for.  Per accumulator: hi*hi of columns 0-31, of 32-63, then
        // the 128 cross products of the 64 columns in one instruction.
        typedef int i32x8v __attribute__((ext_vector_type(8)));
        typedef int i32x4v __attribute__((ext_vector_type(4)));
        constexpr int kScaleA = 0x82828282, kScaleB = 0x7f7f7f7f;  // E8M0 block scales 2^3 (= 128 / 16) and 2^0, the same in every lane
        // DIAG flag 16 (libseekr_hip_diag.so, timing only): the X line staged HALF — what the loop would cost if that line
        // carried lo8 alone (hi8 derived from the H line's fp16 values in registers): 192 instead of 256 bytes per 64 columns
        const bool half_x = DIAG && (diag[1] & 16);
        for (int64_t t = 0; t < kt; t += 2) {
            {   // H stage
                if (t + 1 < kt) stage(cur ^ 1, t + 1, half_x);
                const char* base = smem + cur * kStageBytes;
                vec8<T> a0[MT], a1[MT], b0[NT], b1[NT];
#pragma unroll
                for (int i = 0; i < NT; i++) {
                    b0[i] = *reinterpret_cast<const vec8<T>*>(base + b_off[i] + ((q ^ b_swz[i]) << 4));
                    b1[i] = *reinterpret_cast<const vec8<T>*>(base + b_off[i] + (((4 + q) ^ b_swz[i]) << 4));
                }
#pragma unroll
                for (int i = 0; i < MT; i++) {
                    a0[i] = *reinterpret_cast<const vec8<T>*>(base + a_off[i] + ((q ^ a_swz[i]) << 4));
                    a1[i] = *reinterpret_cast<const vec8<T>*>(base + a_off[i] + (((4 + q) ^ a_swz[i]) << 4));
                }
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) {
                        acc[mt][nt] = mfma16x16(a0[mt], b0[nt], acc[mt][nt]);
                        acc[mt][nt] = mfma16x16(a1[mt], b1[nt], acc[mt][nt]);
                    }
                __syncthreads();
                cur ^= 1;
            }
            if (t + 1 < kt) {   // X stage
                if (t + 2 < kt) stage(cur ^ 1, t + 2);
                const char* base = smem + cur * kStageBytes;
                i32x8v a8[MT], b8[NT];
                // A lane group q reads bytes 32q.., B the other half of the line; LOWER: the other way round, so that k position
                // for k position the products are those of the swapped call (the block scales are powers of two: exact)
                const int qa = SWAPPED ? 2 * (q ^ 2) : 2 * q, qb = SWAPPED ? 2 * q : 2 * (q ^ 2);
#pragma unroll
                for (int i = 0; i < NT; i++) {
                    const i32x4v lo4 = *reinterpret_cast<const i32x4v*>(base + b_off[i] + ((qb ^ b_swz[i]) << 4));
                    const i32x4v hi4 = *reinterpret_cast<const i32x4v*>(base + b_off[i] + (((qb + 1) ^ b_swz[i]) << 4));
                    b8[i] = i32x8v{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                }
#pragma unroll
                for (int i = 0; i < MT; i++) {
                    const i32x4v lo4 = *reinterpret_cast<const i32x4v*>(base + a_off[i] + ((qa ^ a_swz[i]) << 4));
                    const i32x4v hi4 = *reinterpret_cast<const i32x4v*>(base + a_off[i] + (((qa + 1) ^ a_swz[i]) << 4));
                    a8[i] = i32x8v{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                }
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nt = 0; nt < NT; nt++)
                        acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[mt], b8[nt], acc[mt][nt], 0, 0, 0, kScaleA, 0,
                                                                                       kScaleB);
                __syncthreads();
                cur ^= 1;
            }
        }
    } else if constexpr (WAVES == 8) {
        for (int64_t t = 0; t < kt; t++) {
            if (t + 1 < kt && !no_dma) stage(cur ^ 1, same_tile ? 0 : t + 1);
            const char* base = smem + cur * kStageBytes;
            vec8<T> ahi[MT], alo[MT], bhi[NT], blo[NT];
    #pragma unroll
            for (int i = 0; i < NT; i++) {
                bhi[i] = *reinterpret_cast<const vec8<T>*>(base + b_off[i] + ((q ^ b_swz[i]) << 4));
                blo[i] = *reinterpret_cast<const vec8<T>*>(base + b_off[i] + (((4 + q) ^ b_swz[i]) << 4));
            }
    #pragma unroll
            for (int i = 0; i < MT; i++) {
                ahi[i] = *reinterpret_cast<const vec8<T>*>(base + a_off[i] + ((q ^ a_swz[i]) << 4));
                alo[i] = *reinterpret_cast<const vec8<T>*>(base + a_off[i] + (((4 + q) ^ a_swz[i]) << 4));
            }
    #pragma unroll
            for (int mt = 0; mt < MT; mt++)
    #pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    if (NPROD >= 4) acc[mt][nt] = mfma16x16(alo[mt], blo[nt], acc[mt][nt]);
                    if (SWAPPED) {
                        acc[mt][nt] = mfma16x16(ahi[mt], blo[nt], acc[mt][nt]);
                        acc[mt][nt] = mfma16x16(alo[mt], bhi[nt], acc[mt][nt]);
                    } else {
                        acc[mt][nt] = mfma16x16(alo[mt], bhi[nt], acc[mt][nt]);
                        acc[mt][nt] = mfma16x16(ahi[mt], blo[nt], acc[mt][nt]);
                    }
                    acc[mt][nt] = mfma16x16(ahi[mt], bhi[nt], acc[mt][nt]);
                }
            __syncthreads();
            cur ^= 1;
        }
    } else {
        // One wave per SIMD: every fragment a phase multiplies was requested a whole phase (1 024 MFMA cycles) earlier.
        // Tile t: phase 1 = Alo x Bhi, phase 2 = Ahi x Blo, [barrier: stage t+1 has landed], phase 3 = Ahi x Bhi — per
        // accumulator the order of the 8-wave kernel.  Alo(t+1) is read into Alo's registers during phase 3 (free after
        // phase 1), Bhi(t+1) into Blo's (free after phase 2); Ahi(t), Blo(t) are read during phase 1.  The two B sets swap
        // roles from one tile to the next, so the loop is unrolled by two.
        // With one wave per SIMD every cycle the wave spends issuing anything but an MFMA is a cycle the matrix core
        // idles (clumped in front of the phases, the 32 ds_reads, 16 LDS-DMA pieces and their address arithmetic cost
        // 26 %: 28.1 vs 22.3 ms).  So (a) the non-MFMA instructions are dealt out BETWEEN the MFMAs, one per two or four of
        // them, pinned by sched_barriers; (b) they need no vector arithmetic: the stage buffer is a compile-time constant
        // (even k tiles use buffer 0: the body is instantiated per buffer), so every ds_read is base register +
        // immediate, and the k tile's offset goes into the scalar base of the LDS-DMA.
        mfma_drain();  // the zeroed accumulators (v_accvgpr_write) are settled before the first statement reads them
        vec8<T> alo[MT], ahi[MT], b0[NT], b1[NT];
        // per-lane LDS addresses within a stage: rows differ by multiples of 16, so the swizzle term is the same for all
        // eight fragments of an operand half
        const int a_lane_hi = a_off[0] + ((q ^ a_swz[0]) << 4), a_lane_lo = a_off[0] + (((4 + q) ^ a_swz[0]) << 4);
        const int b_lane_hi = b_off[0] + ((q ^ b_swz[0]) << 4), b_lane_lo = b_off[0] + (((4 + q) ^ b_swz[0]) << 4);
        auto rd = [&](int buf, int lane_off, int i) {
            return *reinterpret_cast<const vec8<T>*>(smem + buf * kStageBytes + lane_off + i * 16 * kRowBytes);
        };
        auto dma = [&](int buf, int64_t tile, int g) {  // piece g of 16: A pieces even, B pieces odd
            const int p = g >> 1;
            const char* src = ((g & 1) ? b_tile : a_tile) + (size_t)tile * kRowBytes;  // uniform: scalar base
            char* dst = smem + buf * kStageBytes + ((g & 1) ? TM * kRowBytes : 0) + (wave * PP + p) * 1024;
            lds_dma16(src + ((g & 1) ? b_voff[p] : a_voff[p]), dst);
        };
#pragma unroll
        for (int i = 0; i < MT; i++) alo[i] = rd(0, a_lane_lo, i);
#pragma unroll
        for (int i = 0; i < NT; i++) b0[i] = rd(0, b_lane_hi, i);
        // STEADY: k tiles t+1 and t+2 exist (no conditions inside the interleaved schedule)
        auto body = [&](vec8<T>(&bhi)[NT], vec8<T>(&blo)[NT], int64_t t, auto cur_c, auto steady_c) {
            constexpr int CUR = decltype(cur_c)::value;
            constexpr bool STEADY = decltype(steady_c)::value;
            // Alo(t), Bhi(t) were requested a phase ago and have long arrived; said explicitly (lgkmcnt(0), free here):
            // with 16 younger reads in flight the compiler could only express "all done" in the 4-bit counter
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_sched_barrier(0);
            // phase 1: Alo x Bhi, and under it the reads of Blo(t), Ahi(t): one after every four MFMAs
#pragma unroll
            for (int g = 0; g < 16; g++) {
#pragma unroll
                for (int u = 0; u < 4; u++) mfma16x16_pinned(acc[(4 * g + u) / NT][(4 * g + u) % NT], alo[(4 * g + u) / NT], bhi[(4 * g + u) % NT]);
                if (g < NT) blo[g] = rd(CUR, b_lane_lo, g);
                else ahi[g - NT] = rd(CUR, a_lane_hi, g - NT);
                __builtin_amdgcn_sched_barrier(0);
            }
            // phase 2: Ahi x Blo
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) mfma16x16_pinned(acc[mt][nt], ahi[mt], blo[nt]);
            __builtin_amdgcn_sched_barrier(0);
            // this wave's pieces of stage t+1 have landed (vmcnt(0), said explicitly: in one of the two instances of this
            // body hipcc's own LDS-DMA tracking put no wait in front of the barrier), then everybody's
            __builtin_amdgcn_s_waitcnt(0x0070);
            __syncthreads();  // nobody reads stage t any more
            __builtin_amdgcn_sched_barrier(0);
            // phase 3: Ahi x Bhi; under its first half the reads of Bhi(t+1) (into Blo's registers) and Alo(t+1), under its
            // second half the LDS-DMA of stage t+2 into the buffer tile t just left — a whole k tile ahead of its barrier
            if constexpr (STEADY) {
#pragma unroll
                for (int g = 0; g < 32; g++) {
#pragma unroll
                    for (int u = 0; u < 2; u++) mfma16x16_pinned(acc[(2 * g + u) / NT][(2 * g + u) % NT], ahi[(2 * g + u) / NT], bhi[(2 * g + u) % NT]);
                    if (g < NT) blo[g] = rd(CUR ^ 1, b_lane_hi, g);
                    else if (g < 16) alo[g - NT] = rd(CUR ^ 1, a_lane_lo, g - NT);
                    else dma(CUR, t + 2, g - 16);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                if (t + 2 < kt) {
#pragma unroll
                    for (int g = 0; g < 16; g++) dma(CUR, t + 2, g);
                }
                if (t + 1 < kt) {
#pragma unroll
                    for (int i = 0; i < NT; i++) blo[i] = rd(CUR ^ 1, b_lane_hi, i);
#pragma unroll
                    for (int i = 0; i < MT; i++) alo[i] = rd(CUR ^ 1, a_lane_lo, i);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) mfma16x16_pinned(acc[mt][nt], ahi[mt], bhi[nt]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        using std::integral_constant;
        int64_t t = 0;
        for (; t + 3 < kt; t += 2) {  // even tiles live in buffer 0
            body(b0, b1, t, integral_constant<int, 0>{}, integral_constant<bool, true>{});
            body(b1, b0, t + 1, integral_constant<int, 1>{}, integral_constant<bool, true>{});
        }
        for (; t + 1 < kt; t += 2) {
            body(b0, b1, t, integral_constant<int, 0>{}, integral_constant<bool, false>{});
            body(b1, b0, t + 1, integral_constant<int, 1>{}, integral_constant<bool, false>{});
        }
        if (t < kt) body(b0, b1, t, integral_constant<int, 0>{}, integral_constant<bool, false>{});
        mfma_drain();
    }
    if (DIAG) st[3] = __builtin_amdgcn_s_memtime();
    if (MODE == EDGES) {
        // pass 1 marks this lane's surviving cells (128 bits), one atomic per wave reserves room for all of them,
        // pass 2 writes them (a list in no particular order: the host sorts it).  One atomic per wave and TILE: with one
        // per accumulator register a dense block (a few % of the cells kept) spent 250 us per tile in dependent atomics.
        // Cheap first: a cell can only survive if !(acc < cutoff x K) — exact when K x scale^2 is a power of two —
        // so the exact test (global indices, diagonal, block edge) runs for the few candidates only.
        const float thr = rk != 0.f && !accumulate ? es.cutoff * kdiv : -INFINITY;
        auto cell = [&](int mt, int nt, int e, float& w, unsigned long long& key) -> bool {
            const int64_t n = col_base + wn * WTN + nt * 16 + (lane & 15);
            const int64_t m = row_base + wm * 128 + mt * 16 + 4 * q + e;
            const bool inside = m < M && n < N;
            w = rk != 0.f ? acc[mt][nt][e] * rk : acc[mt][nt][e] / kdiv;
            if (accumulate && inside) w = C[(size_t)m * ldc + n] + w;  // earlier k chunks left their sum in C
            const int64_t grow = es.row_global0 + m, gcol = es.col_global0 + n;
            key = ((unsigned long long)grow << 32) | (unsigned long long)gcol;
            return inside && !(w < es.cutoff) && w != 0.f && (es.upper ? gcol > grow : gcol != grow);
        };
        uint32_t bits[MT * NT / 8] = {};
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (!(acc[mt][nt][e] < thr)) {  // NaN passes, as it does in numpy's `ld_sim < cutoff`
                        float w;
                        unsigned long long key;
                        const int idx = (mt * NT + nt) * 4 + e;
                        if (cell(mt, nt, e, w, key)) bits[idx >> 5] |= 1u << (idx & 31);
                    }
                }
        uint32_t mine = 0;
#pragma unroll
        for (int w = 0; w < MT * NT / 8; w++) mine += __popc(bits[w]);
        uint32_t incl = mine;  // inclusive scan over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        const uint32_t total = __shfl(incl, 63, 64);
        if (total) {  // wave-uniform
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(es.count, (unsigned long long)total);
            base = __shfl(base, 0, 64);
            unsigned long long pos = base + (incl - mine);
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int idx = (mt * NT + nt) * 4 + e;
                        if (bits[idx >> 5] & (1u << (idx & 31))) {
                            float w;
                            unsigned long long key;
                            (void)cell(mt, nt, e, w, key);
                            if (pos < es.cap) {
                                es.keys[pos] = key;
                                es.vals[pos] = w;
                            }
                            pos++;
                        }
                    }
        }
        if (!PERSIST) return;
        continue;
    }
    // The mirror (SELF: the tile below the diagonal; CROSS: the transposed block in Ct) is written by the LAST k chunk only,
    // as a copy of the finished value: the earlier chunks of a K > 4 096 call then neither store nor re-load it (round 4;
    // counters had the accumulating chunks of a k = 7 self-comparison 16 % slower than the first one, the waves parked on
    // the mirror's loads — 16 rows x 64 bytes per instruction — while a PLAIN block paid 1.7 %: DESIGN §4).
    const bool last_chunk = (flags & 2) != 0;
    const bool mirror = (MODE == CROSS || (SYM && tm != tn)) && last_chunk && !(DIAG && (diag[1] & 4));  // DIAG flag 4: timing without the mirror stores
    auto quad_transpose = [&](float (&v)[4], int j) {  // 4 x 4 inside each quad of lanes; its own inverse
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            const float send = (j & 1) ? v[e] : v[e + 1];
            // lane ^ 1 / lane ^ 2 inside a quad (ds_bpermute; DPP quad_perm moves were measured 1 % slower in round 5 and reverted)
            const float recv = __shfl_xor(send, 1, 64);
            if (j & 1) v[e] = recv; else v[e + 1] = recv;
        }
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float send = (j & 2) ? v[e] : v[e + 2];
            const float recv = __shfl_xor(send, 2, 64);
            if (j & 2) v[e] = recv; else v[e + 2] = recv;
        }
    };
    auto store_mirror = [&](const float (&v)[4], int64_t n, int64_t m0) {  // r[n, m0..m0+3]: the lane's 4 rows are contiguous in the mirror
        if (n >= N) return;
        float* dst = Ct + (size_t)n * ldct + m0;
        if (m0 + 3 < M && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
            __builtin_nontemporal_store(f32x4v{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4v*>(dst));
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (m0 + e < M) dst[e] = v[e];
        }
    };
    // The LEAN epilogue (round 6): what nearly every tile of a launch is — whole, off the diagonal, one k chunk, the divisor
    // a power of two, rows of r 16-byte aligned — stored without a test, a 64-bit multiply or a branch per cell.  The
    // general loop below spends ~130 instructions per accumulator tile on bounds, alignment, the diagonal and the
    // multiply-or-divide choice (4 200 per wave and tile, two waves per SIMD: the 16 us of a tile's epilogue were issue
    // slots, not stores); here an accumulator tile is 4 multiplies, the quad transpose and two 16-byte stores whose
    // addresses are a scalar base per (mt | nt) + one per-lane 32-bit offset + an immediate.  Same values, same cells,
    // same cells: r is the same bits.  flags bits 2-3 (SEEKR_GEMM_EPILOGUE: 0 = off, 1 / 2 / 3 = the run length
    // below) select it; every other tile takes the general loop.
    const bool lean = (flags & 12) && (!accumulate || ((flags >> 2) & 3) == 3) && rk != 0.f && row_base + TM <= M && col_base + TN <= N && !(SYM && tm == tn) &&
                      ldc < (int64_t(1) << 26) && ((reinterpret_cast<uintptr_t>(C) | (uintptr_t)(ldc * 4)) & 15) == 0 &&
                      (!mirror || (ldct < (int64_t(1) << 26) && ((reinterpret_cast<uintptr_t>(Ct) | (uintptr_t)(ldct * 4)) & 15) == 0));
    if (DIAG && (diag[1] & 32)) {
        // diagnostic only (round 6): NO store at all — the timing ceiling of any scheme that hides the epilogue under the
        // next tile's k loop; the accumulators are still read, r stays unwritten
        float sink = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) sink += acc[mt][nt][0] + acc[mt][nt][1] + acc[mt][nt][2] + acc[mt][nt][3];
        if (sink == 123.456f) C[0] = sink;
    } else if (lean) {
        // Lane geometry of a 16 x 16 accumulator tile: lane = 16 rho + c holds column c, rows 4 rho .. 4 rho + 3 (c = 4 g + j).
        // RUN = bytes of one row of r that one store instruction covers: 64 is what the MFMA layout gives after the 4 x 4
        // transpose inside each quad of lanes (16 rows x 64 B per instruction); with G = RUN / 64 neighbouring accumulator tiles
        // exchanged between lane groups first an instruction covers 8 rows x 128 B or 4 rows x 256 B.  tools/micro/tile_store
        // (stores alone, this tile walk, whole chip): 3.7 TB/s with 64-byte runs, 4.6 with 128, 5.3 with 256 — HBM takes
        // half-line writes badly, and the write stream competes with the k loops' staging loads of every other CU.
        //   direct tile (rows of r): v_permlane32_swap pairs tile nt with nt + 1 (lane halves), v_permlane16_swap pairs of pairs;
        //   mirror (columns of the tile become rows): DPP row_ror:8 / row_shr:4 / row_shl:4 pair tile mt with mt + 1, pairs of pairs.
        // All of it moves finished values between lanes: r is the same bits whatever RUN (tools/epilogue_check.py).
        // (the lane's geometry is re-derived HERE for every tile: taken from `lane` directly, the dozen values below are loop
        // invariants of the persistent loop, hoisted above it and kept alive through the k loop — where the kernel has no
        // register to spare: 256 VGPRs and 600 bytes of scratch instead of 215 and none)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int c = lane_e & 15, g = c >> 2, j = lane_e & 3, rho = lane_e >> 4;
        const bool odd = (j & 1) != 0, upper = (j & 2) != 0, c_hi = (c & 8) != 0, c_mid = (c & 4) != 0;
        char* cw = reinterpret_cast<char*>(C + (size_t)(row_base + wm * 128) * ldc + col_base + wn * WTN);  // wave-uniform
        const size_t d_step = (size_t)ldc * 64;  // 16 rows down
        char* tw[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
            tw[nt] = reinterpret_cast<char*>(Ct + (size_t)(col_base + wn * WTN + nt * 16) * ldct + row_base + wm * 128);
        auto swap32 = [](float& a, float& b) {  // a's lanes 32-63 <-> b's lanes 0-31
            // (the two results are copied into scalars first: __builtin_bit_cast(float, r[1]) straight from the vector element
            // reads element 0 with this clang — both outputs of the swap then compile to the first)
            const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
            const unsigned r0 = r[0], r1 = r[1];
            a = __builtin_bit_cast(float, r0);
            b = __builtin_bit_cast(float, r1);
        };
        auto swap16 = [](float& a, float& b) {  // a's odd 16-lane rows <-> b's even ones (row 1 <-> row 0, row 3 <-> row 2)
            const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
            const unsigned r0 = r[0], r1 = r[1];
            a = __builtin_bit_cast(float, r0);
            b = __builtin_bit_cast(float, r1);
        };
        auto put = [](const float (&v)[4], char* base, uint32_t off) {
            __builtin_nontemporal_store(f32x4v{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4v*>(base + (size_t)off));
        };
        // every accumulator is scaled where it is stored — twice for a tile with a mirror, by the SAME factor under two
        // names: one name, and the compiler merges the two multiplications into 128 scaled copies that live from the first
        // pass to the second (256 VGPRs + scratch instead of 215)
        float rk_m = rk;
        asm volatile("" : "+v"(rk_m));
        auto body = [&](auto with_mirror, auto run_c, auto acc_c) {
            constexpr int G = decltype(run_c)::value / 64;  // accumulator tiles per row run
            // ACC: a later k chunk of a row wider than one accumulator run (k >= 7: 16 384 columns = four chunks): what the
            // earlier chunks left in C is added before the store (256-byte runs only).  The old cells of a group are
            // requested PF groups ahead — a wave has 12 loads of 1 KiB in flight — and, when this is the last chunk of a
            // tile with a mirror, the finished values are brought back into the accumulator's layout (the exchanges below
            // are their own inverses) and stored to the mirror from there.
            constexpr bool ACC = decltype(acc_c)::value;
            static_assert(!ACC || G == 4, "the accumulating lean epilogue exists for 256-byte runs");
            constexpr int PF = 3;
            auto quad_transpose4 = [&](float (&v)[4]) {  // 4 x 4 inside each quad of lanes (DPP); its own inverse
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const float t0 = quad_perm<0xB1>(v[e]), t1 = quad_perm<0xB1>(v[e + 1]);
                    v[e + 1] = odd ? v[e + 1] : t0;
                    v[e] = odd ? t1 : v[e];
                }
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float t0 = quad_perm<0x4E>(v[e]), t1 = quad_perm<0x4E>(v[e + 2]);
                    v[e + 2] = upper ? v[e + 2] : t0;
                    v[e] = upper ? t1 : v[e];
                }
            };
            // ---- the direct tile
            const int tile_d = G == 1 ? 0 : (G == 2 ? (rho >> 1) : (((rho & 1) << 1) | (rho >> 1)));  // which tile of its group a lane stores
            const int row_d = G == 1 ? 4 * rho + j : (G == 2 ? 4 * (rho & 1) + j : j);                 // ... and which of the first rows
            const uint32_t d_off = (uint32_t)((row_d * ldc + 16 * tile_d + 4 * g) * 4);
            const size_t d_rows = (size_t)ldc * (16 / G) * 4;  // from one store of a group to the next: 16 / G rows down
            const uint32_t m_off1 = (uint32_t)((c * ldct + 4 * rho) * 4);  // ACC + mirror: the lane's four rows in row c of the mirror tile
            f32x4v old_c[ACC ? PF : 1][4];
            auto request = [&](int mt) {  // the cells of C that group mt will be added to, in the order its four stores go out
                if constexpr (ACC) {
                    const char* base = cw + (size_t)mt * d_step;
#pragma unroll
                    for (int x = 0; x < 4; x++)
                        old_c[mt % PF][x] = *reinterpret_cast<const f32x4v*>(base + x * d_rows + (size_t)d_off);
                }
            };
            if constexpr (ACC) {
#pragma unroll
                for (int mt = 0; mt < PF - 1; mt++) request(mt);
            }
            char* cwm = cw;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                if constexpr (ACC) {
                    if (mt + PF - 1 < MT) request(mt + PF - 1);
                }
#pragma unroll
                for (int nb = 0; nb < NT / G; nb++) {
                    float t[G][4];
#pragma unroll
                    for (int i = 0; i < G; i++) {
#pragma unroll
                        for (int e = 0; e < 4; e++) t[i][e] = acc[mt][nb * G + i][e] * rk;
                        quad_transpose4(t[i]);  // lane j of a quad ends up with row 4 rho + j, four columns
                    }
                    char* base = cwm + nb * (64 * G);
                    if constexpr (G == 1) {
                        put(t[0], base, d_off);
                    } else if constexpr (G == 2) {
#pragma unroll
                        for (int e = 0; e < 4; e++) swap32(t[0][e], t[1][e]);  // t[0]: rows 0-7 of both tiles, t[1]: rows 8-15
                        put(t[0], base, d_off);
                        put(t[1], base + d_rows, d_off);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            swap32(t[0][e], t[1][e]);
                            swap32(t[2][e], t[3][e]);
                            swap16(t[0][e], t[2][e]);  // t[0]: rows 0-3 of the four tiles, t[2]: rows 4-7
                            swap16(t[1][e], t[3][e]);  // t[1]: rows 8-11, t[3]: rows 12-15
                        }
                        if constexpr (ACC) {  // stores go out in the order t[0], t[2], t[1], t[3]: so were the loads
                            constexpr int at[4] = {0, 2, 1, 3};
#pragma unroll
                            for (int x = 0; x < 4; x++)
#pragma unroll
                                for (int e = 0; e < 4; e++) t[at[x]][e] = old_c[mt % PF][x][e] + t[at[x]][e];
                        }
                        put(t[0], base, d_off);
                        put(t[2], base + d_rows, d_off);
                        put(t[1], base + 2 * d_rows, d_off);
                        put(t[3], base + 3 * d_rows, d_off);
                        if constexpr (ACC && decltype(with_mirror)::value) {  // the finished values, back where the MFMA left the partial ones
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                swap16(t[1][e], t[3][e]);
                                swap16(t[0][e], t[2][e]);
                                swap32(t[2][e], t[3][e]);
                                swap32(t[0][e], t[1][e]);
                            }
                            // ... and straight out to the mirror, 16 rows x 64 bytes per instruction (writing them back into the
                            // accumulator registers for a pass with 256-byte runs made the allocator spill 100 registers)
#pragma unroll
                            for (int i = 0; i < 4; i++) {
                                quad_transpose4(t[i]);
                                put(t[i], tw[i] + mt * 64, m_off1);
                            }
                        }
                    }
                    // one group at a time: left to itself the scheduler interleaves the 32 independent groups of this 3 000-
                    // instruction block until all 256 registers are in use, and the allocator then spills accumulators
                    __builtin_amdgcn_sched_barrier(0);
                }
                cwm += d_step;
            }
            // ---- the mirror: a lane's four rows are four consecutive cells of row (column index) of Ct
            if constexpr (decltype(with_mirror)::value && !ACC) {
                const int row_m = G == 1 ? c : (G == 2 ? (c & 7) : (c & 3));
                const int tile_m = G == 1 ? 0 : (G == 2 ? (c >> 3) : ((((c >> 2) & 1) << 1) | (c >> 3)));
                const uint32_t m_off = (uint32_t)((row_m * ldct + 16 * tile_m + 4 * rho) * 4);
                const size_t m_rows = (size_t)ldct * (16 / G) * 4;
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
#pragma unroll
                    for (int mb = 0; mb < MT / G; mb++) {
                        float u[G][4];
#pragma unroll
                        for (int i = 0; i < G; i++)
#pragma unroll
                            for (int e = 0; e < 4; e++) u[i][e] = acc[mb * G + i][nt][e] * rk_m;
                        char* base = tw[nt] + mb * (64 * G);
                        if constexpr (G == 1) {
                            put(u[0], base, m_off);
                        } else {
                            // columns 0-7 of tile i meet columns 0-7 of tile i + 1 (lane c <-> c ^ 8): a = mirror rows 0-7, b = rows 8-15
                            auto pair8 = [&](float (&x)[4], float (&y)[4]) {
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    const float xr = quad_perm<0x128>(x[e]), yr = quad_perm<0x128>(y[e]);  // row_ror:8
                                    x[e] = c_hi ? yr : x[e];
                                    y[e] = c_hi ? y[e] : xr;
                                }
                            };
                            pair8(u[0], u[1]);
                            if constexpr (G == 2) {
                                put(u[0], base, m_off);
                                put(u[1], base + m_rows, m_off);
                            } else {
                                pair8(u[2], u[3]);
                                // ... and the two pairs (lane c <-> c ^ 4): x = mirror rows c & 3 of four tiles, y = rows 4 + (c & 3)
                                auto pair4 = [&](float (&x)[4], float (&y)[4]) {
#pragma unroll
                                    for (int e = 0; e < 4; e++) {
                                        const float yd = quad_perm<0x114>(y[e]), xu = quad_perm<0x104>(x[e]);  // row_shr:4 (from lane c - 4), row_shl:4 (from c + 4)
                                        x[e] = c_mid ? yd : x[e];
                                        y[e] = c_mid ? y[e] : xu;
                                    }
                                };
                                pair4(u[0], u[2]);  // u[0]: mirror rows 0-3, u[2]: rows 4-7
                                pair4(u[1], u[3]);  // u[1]: rows 8-11, u[3]: rows 12-15
                                put(u[0], base, m_off);
                                put(u[2], base + m_rows, m_off);
                                put(u[1], base + 2 * m_rows, m_off);
                                put(u[3], base + 3 * m_rows, m_off);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        };
        const int run = (flags >> 2) & 3;  // 1: 64-byte runs, 2: 128, 3: 256
        using std::false_type;
        using std::integral_constant;
        using std::true_type;
        if (accumulate) {  // (lean only with run == 3: the condition above)
            if (mirror) body(true_type{}, integral_constant<int, 256>{}, true_type{});
            else body(false_type{}, integral_constant<int, 256>{}, true_type{});
        } else if (mirror) {
            if (run == 3) body(true_type{}, integral_constant<int, 256>{}, false_type{});
            else if (run == 2) body(true_type{}, integral_constant<int, 128>{}, false_type{});
            else body(true_type{}, integral_constant<int, 64>{}, false_type{});
        } else {
            if (run == 3) body(false_type{}, integral_constant<int, 256>{}, false_type{});
            else if (run == 2) body(false_type{}, integral_constant<int, 128>{}, false_type{});
            else body(false_type{}, integral_constant<int, 64>{}, false_type{});
        }
    } else
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int64_t n = col_base + wn * WTN + nt * 16 + (lane & 15);
            const int64_t m0 = row_base + wm * 128 + mt * 16 + 4 * q;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = rk != 0.f ? acc[mt][nt][e] * rk : acc[mt][nt][e] / kdiv;
            if (SYM && tm == tn) {
                // diagonal tile: hi*lo and lo*hi enter the accumulator in a different order for
                // (i,j) and (j,i); keep the upper element and mirror it so r is exactly symmetric
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int64_t m = m0 + e;
                    if (n < N && m < M && n >= m) {
                        const float w = accumulate ? C[(size_t)m * ldc + n] + v[e] : v[e];
                        C[(size_t)m * ldc + n] = w;
                        if (n > m) Ct[(size_t)n * ldct + m] = w;
                    }
                }
            } else {
                if (mirror && !accumulate) store_mirror(v, n, m0);  // a single-chunk call: the value is final as it stands
                // direct tile: transpose 4x4 inside each quad of lanes (lane j of a quad ends up with
                // row m0+j, columns c0..c0+3) so it is written with 16-byte stores too — the epilogue
                // is store-issue bound and this quarters its instruction count
                const int j = lane & 3;
                quad_transpose(v, j);
                const int64_t mrow = m0 + j, ncol = n - j;
                float* dst = C + (size_t)mrow * ldc + ncol;
                if (mrow < M) {
                    if (ncol + 3 < N && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                        f32x4v w{v[0], v[1], v[2], v[3]};
                        if (accumulate) w += *reinterpret_cast<const f32x4v*>(dst);
                        __builtin_nontemporal_store(w, reinterpret_cast<f32x4v*>(dst));
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = w[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (ncol + e < N) {
                                if (accumulate) v[e] = dst[e] + v[e];
                                dst[e] = v[e];
                            }
                    }
                }
                if (mirror && accumulate) {  // the last of several chunks: the finished values, back in the accumulator's layout
                    quad_transpose(v, j);
                    store_mirror(v, n, m0);
                }
            }
        }
    if (DIAG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tile's stores have left the CU
        st[4] = __builtin_amdgcn_s_memtime();
        st[5] = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            const unsigned long long rec = atomicAdd(&diag[0], 1ull);
            if (rec < (1ull << 16)) {
                unsigned long long* d = diag + 8 + rec * 8;
                for (int i = 0; i < 6; i++) d[i] = st[i];
                d[6] = (unsigned long long)tm << 32 | (unsigned long long)tn;
                d[7] = (unsigned long long)home << 32 | blockIdx.x;
            }
        }
    }
    if (!PERSIST) return;
  }
}

struct SplitOut {
    float* C;
    int64_t ldc;
    float* Ct;   // SELF: == C; CROSS: the matrix receiving the transposed block; PLAIN: unused
    int64_t ldct;
};

// one launch over `ktc` k tiles starting at the operands' current tile (pitch: `kt_pitch` tiles per row)
template <typename T, int NPROD, int MODE>
int launch_chunk(skr_ctx* ctx, const T* Ac, const T* Bc, const SplitOut& o, int64_t M, int64_t N, int64_t ktc, int64_t kt_pitch,
                 float K, int accumulate, const EdgeSink& es) {
    const int64_t tiles_m = (M + TM - 1) / TM, tiles_n = (N + TN - 1) / TN;
    const int64_t super_m = (tiles_m + 15) / 16, super_n = (tiles_n + 15) / 16;
    const int64_t slots = super_m * super_n * 256;
    if (ctx->knobs.gemm_persist && slots > ctx->num_cu) {
        uint32_t* queues = ctx->d_flags + 8;  // eight counters, zeroed per launch
        SKR_HIP(hipMemsetAsync(queues, 0, 8 * sizeof(uint32_t), ctx->stream));
        auto kern = pearson_gemm_split16_kernel<T, NPROD, MODE, true>;
        unsigned long long* diag = nullptr;
#ifdef SEEKR_DIAG
        // libseekr_hip_diag.so only (python -m seekr_amd.build --diag; tools/gemm_diag.py): the production library holds
        // neither the stamping instance nor the switches below, which make r meaningless.  Never with an edge sink
        // active: the stamps live in the ctx workspace, which is where the sink's list is.
        if (std::is_same<T, _Float16>::value && (NPROD == 3 || NPROD == 2) && (MODE == PLAIN || MODE == SELF) && ctx->diag_mode && !es.count) {
            void* ws = nullptr;
            SKR_TRY(skr_ctx_workspace(ctx, (size_t)(8 + 8 * 65536) * 8, &ws));
            diag = (unsigned long long*)ws;
            SKR_HIP(hipMemsetAsync(diag, 0, 64, ctx->stream));
            // experiments: 2 = k loop without staging, 3 = every stage re-loads k tile 0, 4 = self mode without the mirror
            // stores, 5 = two product-units per k (one int8 MFMA in place of the two cross products: timing ceiling)
            const int dmode = ctx->diag_mode;
            // 6 (f16f8 operands only) = the X line staged half (timing ceiling of a layout without the hi8 copies)
            // 7 = no epilogue stores at all (timing ceiling of hiding the epilogue: round 6)
            if (dmode >= 2 && dmode <= 7)
                SKR_HIP(hipMemsetAsync(diag + 1, dmode == 2 ? 1 : (dmode == 3 ? 2 : (dmode == 4 ? 4 : (dmode == 5 ? 8 : (dmode == 6 ? 16 : 32)))), 1, ctx->stream));
            kern = pearson_gemm_split16_kernel<T, NPROD, (MODE == SELF ? SELF : PLAIN), true, true>;
        }
#endif
        unsigned threads = 512;
#ifdef SEEKR_DIAG
        if constexpr (NPROD == 3 && MODE != LOWER) {
            // A/B arm (libseekr_hip_diag.so only; measured 13 % slower): 4 waves x 128 x 128 (kernel comment)
            if (ctx->knobs.gemm_wave_tile == 1 && !diag) {
                kern = pearson_gemm_split16_kernel<T, NPROD, MODE, true, false, 4>;
                threads = 256;
            }
        }
#endif
        SKR_TRY(skr_kernel_lds(ctx, reinterpret_cast<const void*>(kern), 2 * kStageBytes));
        // A resident workgroup owns its CU outright (8 waves x ~248 VGPRs, 128 KiB LDS): with RCCL traffic
        // in flight on the communication stream a few CUs are left free, or the send/recv kernels of
        // shift s+1 could not start before this launch ends and the overlap of §5 would be lost.
        const int reserve = ctx->knobs.gemm_reserve_cus >= 0 ? ctx->knobs.gemm_reserve_cus : (ctx->nranks > 1 ? 8 : 0);
        const unsigned grid = (unsigned)std::max(8, ctx->num_cu - reserve);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 2 * kStageBytes, ctx->stream, Ac, Bc, o.C, o.Ct, M, N, ktc, o.ldc, o.ldct, K,
                           tiles_m, tiles_n, super_n, queues, slots / 8, kt_pitch, accumulate | ((ctx->knobs.gemm_epilogue & 3) << 2) | (ctx->knobs.gemm_subtile << 8), diag, es);
    } else {
        auto kern = pearson_gemm_split16_kernel<T, NPROD, MODE, false>;
        SKR_TRY(skr_kernel_lds(ctx, reinterpret_cast<const void*>(kern), 2 * kStageBytes));
        hipLaunchKernelGGL(kern, dim3((unsigned)slots), dim3(512), 2 * kStageBytes, ctx->stream, Ac, Bc, o.C, o.Ct, M, N, ktc, o.ldc,
                           o.ldct, K, tiles_m, tiles_n, super_n, (uint32_t*)nullptr, (int64_t)0, kt_pitch,
                           accumulate | ((ctx->knobs.gemm_epilogue & 3) << 2) | (ctx->knobs.gemm_subtile << 8), (unsigned long long*)nullptr, es);
    }
    SKR_HIP(hipGetLastError());
    return SKR_OK;
}

template <typename T, int NPROD, int MODE>
int launch16(skr_ctx* ctx, const T* A, const T* B, const SplitOut& o, int64_t M, int64_t N, int64_t kt, float K,
             const char* name, bool coherent, const EdgeSink* sink = nullptr) {
    // One float32 accumulator per cell is restarted every 4 096 columns: the MFMA adder truncates each
    // add at the accumulator's unit (~0.25 ulp lost per add, measured; `chunk_tiles` = 32, i.e. every 1 024 columns (64 until round 4: a soak case sat at 0.75 of the bar with 64, 0.20 with 32), for
    // operands whose rows are mostly one repeated value: tools/margin_probe.py), a bias that grows with the number
    // of adds and with the accumulator — 5e-6 relative on an r ~ 1 pair at K = 4 096, four times that at
    // 16 384 in one go.  Later chunks add their partial result to C in the epilogue (rounded float32 adds).
    const int64_t kChunkTiles = skr_gemm_chunk_tiles(ctx, coherent);
    if (NPROD == 2 && ((kChunkTiles | kt) & 1))  // H and X lines come in pairs: a chunk must not end between them
        return skr_set_error(SKR_ERR_INVALID, "f16f8 operands need an even number of lines per row and per k chunk (%lld, %lld)",
                             (long long)kt, (long long)kChunkTiles);
    const EdgeSink es = sink ? *sink : EdgeSink{};
    SkrProfScope prof(ctx, name);
    for (int64_t t0 = 0; t0 < kt; t0 += kChunkTiles) {
        const int64_t ktc = std::min(kChunkTiles, kt - t0);
        const T* Ac = A + t0 * 64;
        const T* Bc = B + t0 * 64;
        const int accumulate = (t0 > 0 ? 1 : 0) | (t0 + kChunkTiles >= kt ? 2 : 0);  // bit 0: add to C; bit 1: the last chunk (writes the mirror)
        if (MODE == EDGES && t0 + kChunkTiles < kt) {
            // not the last k chunk: its partial sums go to C like a plain block; only the last chunk thresholds
            if (!o.C) return skr_set_error(SKR_ERR_INVALID, "rows of %lld k tiles in chunks of %lld need a scratch block", (long long)kt, (long long)kChunkTiles);
            SKR_TRY((launch_chunk<T, NPROD, PLAIN>(ctx, Ac, Bc, o, M, N, ktc, kt, K, accumulate, es)));
        } else {
            SKR_TRY((launch_chunk<T, NPROD, MODE>(ctx, Ac, Bc, o, M, N, ktc, kt, K, accumulate, es)));
        }
    }
    return SKR_OK;
}

template <typename T, int NPROD>
int gemm_split(skr_ctx* ctx, const T* As, const T* Bs, const SplitOut& o, int64_t M, int64_t N, int64_t kt, float K,
               int mode, const char* name, bool coherent, const EdgeSink* sink = nullptr) {
    switch (mode) {
        case EDGES: return launch16<T, NPROD, EDGES>(ctx, As, Bs, o, M, N, kt, K, name, coherent, sink);
        case SELF: return launch16<T, NPROD, SELF>(ctx, As, Bs, o, M, N, kt, K, name, coherent);
        case CROSS: return launch16<T, NPROD, CROSS>(ctx, As, Bs, o, M, N, kt, K, name, coherent);
        case LOWER: return launch16<T, NPROD, LOWER>(ctx, As, Bs, o, M, N, kt, K, name, coherent);
        default: return launch16<T, NPROD, PLAIN>(ctx, As, Bs, o, M, N, kt, K, name, coherent);
    }
}

}  // namespace

// A, B: split-interleaved operands ([rows, kt, {hi,lo}, 32] 16-bit halves) produced by operand.hip.
// K: the divisor (columns x the operands' storage scales).  mode 0: C = A B^T / K; 1: A == B, one triangle computed and mirrored inside C; 2: C as mode 0 and
// Ct[j * ldct + i] = C[i * ldc + j] as well; 4: C as mode 0 with the bits of the swapped call's mirror (LOWER, above).
int skr_launch_gemm_split(skr_ctx* ctx, int precision, const void* As, const void* Bs, float* C, int64_t M, int64_t N,
                          int64_t kt, int64_t ldc, float K, int mode, float* Ct, int64_t ldct, bool coherent) {
    SplitOut o{C, ldc, mode == SELF ? C : Ct, mode == SELF ? ldc : ldct};
    switch (precision) {
        case SKR_PREC_BF16X3:
            return gemm_split<__bf16, 3>(ctx, (const __bf16*)As, (const __bf16*)Bs, o, M, N, kt, K, mode,
                                         "pearson_gemm_bf16x3", coherent);
        case SKR_PREC_F16X3:
            return gemm_split<_Float16, 3>(ctx, (const _Float16*)As, (const _Float16*)Bs, o, M, N, kt, K, mode,
                                           "pearson_gemm_f16x3", coherent);
        case SKR_PREC_F16F8:
            return gemm_split<_Float16, 2>(ctx, (const _Float16*)As, (const _Float16*)Bs, o, M, N, kt, K, mode,
                                           "pearson_gemm_f16f8", coherent);
        default: return skr_set_error(SKR_ERR_INVALID, "not a split precision: %d", precision);
    }
}

// As skr_launch_gemm_split in PLAIN geometry, but the epilogue thresholds instead of storing (EDGES mode).  C is only
// touched when the rows have more than one k chunk: the earlier chunks leave their partial sums there.
int skr_launch_gemm_edges(skr_ctx* ctx, int precision, const void* As, const void* Bs, float* C, int64_t M, int64_t N,
                          int64_t kt, int64_t ldc, float K, bool coherent, const SkrEdgeSink& sink) {
    SplitOut o{C, ldc, nullptr, 0};
    switch (precision) {
        case SKR_PREC_BF16X3:
            return gemm_split<__bf16, 3>(ctx, (const __bf16*)As, (const __bf16*)Bs, o, M, N, kt, K, EDGES, "pearson_gemm_bf16x3",
                                         coherent, &sink);
        case SKR_PREC_F16X3:
            return gemm_split<_Float16, 3>(ctx, (const _Float16*)As, (const _Float16*)Bs, o, M, N, kt, K, EDGES,
                                           "pearson_gemm_f16x3", coherent, &sink);
        case SKR_PREC_F16F8:
            return gemm_split<_Float16, 2>(ctx, (const _Float16*)As, (const _Float16*)Bs, o, M, N, kt, K, EDGES,
                                           "pearson_gemm_f16f8", coherent, &sink);
        default: return skr_set_error(SKR_ERR_INVALID, "not a split precision: %d", precision);
    }
}

#ifdef SEEKR_DIAG
// libseekr_hip_diag.so only: copies the stamp records of the last contraction launch to the host.
// out: [max_records][8] uint64 = {memtime t0, memrealtime t0, memtime k-loop start, memtime k-loop end, memtime end,
// memrealtime end, tm<<32|tn, xcc<<32|workgroup}; *n_records = records written by the kernel.
extern "C" int skr_gemm_diag_read(skr_ctx* ctx, unsigned long long* out, int64_t max_records, int64_t* n_records) {
    SKR_REQUIRE(ctx && out && n_records && max_records >= 0, "NULL argument");
    SKR_TRY(skr_activate(ctx));
    SKR_REQUIRE(ctx->ws && ctx->ws_bytes >= (size_t)(8 + 8 * 65536) * 8, "no diagnostic launch has run");
    SKR_HIP(hipStreamSynchronize(ctx->stream));
    unsigned long long n = 0;
    SKR_HIP(hipMemcpyAsync(&n, ctx->ws, 8, hipMemcpyDeviceToHost, ctx->stream));
    SKR_HIP(hipStreamSynchronize(ctx->stream));
    *n_records = (int64_t)n;
    const int64_t take = std::min<int64_t>(std::min<int64_t>((int64_t)n, 65536), max_records);
    if (take > 0) {
        SKR_HIP(hipMemcpyAsync(out, (char*)ctx->ws + 64, (size_t)take * 64, hipMemcpyDeviceToHost, ctx->stream));
        SKR_HIP(hipStreamSynchronize(ctx->stream));
    }
    return SKR_OK;
}

// 0 = production kernels; 1 = stamps only (r stays valid); 2-7 = timing experiments that make r meaningless (above)
extern "C" int skr_gemm_diag_mode(skr_ctx* ctx, int mode) {
    SKR_REQUIRE(ctx && mode >= 0 && mode <= 7, "mode 0..7");
    ctx->diag_mode = mode;
    return SKR_OK;
}
#endif

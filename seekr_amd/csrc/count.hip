// K2 + K3 — k-mer counting: sliding-window 4^k indexer, per-sequence LDS histogram, and the
// per-kb scaling fused into the histogram flush (kmer_counts.py:140-151, 194-202).
//
// count_rows_kernel (k <= 8, the tuned path).  A work item is one sequence (or one 8 192-window tile
// of a long sequence) and is owned by ONE WAVE at k <= 6 (64-thread workgroups: no barrier anywhere,
// 17 independent waves per CU at k = 6) or by a 4-wave workgroup at k = 7 and 8.  Lane l of a sweep takes the 16
// windows that start in packed word 64*sweep + l; it holds that word and the next (coalesced loads,
// the first sweeps prefetched one item ahead).  The packer stores the first base in the top bits, so
// window j of the pair is r_j = v_alignbit(hi, lo, 32 - 2j) and its column the top 2k bits of r_j:
// no branch, five vector instructions per window.  Bins are 16-bit counters packed two to an LDS
// word (bin b and bin b + 4^k/2 share word b mod 4^k/2: the top bit of the column picks the half):
// 8 KiB per item at k = 6, 32 KiB at k = 7, 128 KiB at k = 8 — an item never has more than 8 192 windows, so a bin cannot
// overflow.  Counting is ds_add_u32 (no return).  When all 64 lanes of a sweep hold the same two
// packed words (homopolymers and every repeat whose period divides 16 bases) the 64 x 16 atomics —
// 64 lanes on one address each — are replaced by 16 adds of 64 from one lane (wave-level aggregation
// by readfirstlane + ballot).  Windows past the end or over a non-alphabet base are redirected to a
// trash word instead of being branched around.
// The flush reads four words per lane step (8 bins), zeroes them, converts the counts through a
// 16-entry per-item table of the reference's float32 per-kb values (built by 16 lanes only when the
// window count differs from the previous item's) and streams the two 16-byte pieces of the dense row
// with nontemporal stores — the row write (4*4^k bytes per sequence) is the algorithmic traffic that
// bounds the kernel.
//
// Long sequences (more than 8 192 windows) are cut into tiles of 8 192 windows (the k-1 bases of halo
// are simply the next packed words) that are counted like sequences into a scratch matrix of uint32
// partial histograms; reduce_tiles_kernel sums a sequence's tiles (chunks of 32 in parallel) and convert_long_kernel
// writes the row.  A chromosome-sized
// sequence thus spreads over the whole chip instead of serialising on one CU.
//
// count_kmers_kernel (the round-1 kernel: one 256-thread workgroup per sequence, uint32 bins) is kept
// for float64 output and, with GLOBAL = true, for k >= 9 (4^k sixteen-bit bins no longer fit the LDS): it counts
// straight into the sequence's output row, used as a uint32 histogram in HBM (zeroed by a memset first,
// L2 atomics), and converts the row in place.
#include <algorithm>
#include <cstdlib>
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

// ---------------------------------------------------------------------------------------
// The tuned path (k <= 8): see the file header.
// ---------------------------------------------------------------------------------------
constexpr int kItemWindows = 8192;  // windows per work item: longer sequences are cut into tiles of this many; also keeps a 16-bit bin from overflowing

struct CountArgs {
    const uint32_t* packed;
    const int64_t* word_off;
    const int64_t* len;
    const uint32_t* mask;
    const int64_t* mask_off;
    const int64_t* item_seq;    // TILES: sequence of each tile
    const int64_t* item_word0;  // TILES: first packed word of the tile inside its sequence
    int64_t n_items;            // sequences, or tiles
    void* out;                  // rows [n_seqs, 4^k] of the OUT type; TILES: uint32 [n_items, 4^k] partial histograms
    int k;
};

__device__ __forceinline__ void lds_add_u32(uint32_t* lds_base, uint32_t byte_addr, uint32_t v) {
    (void)__hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(lds_base) + byte_addr), v, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);  // result unused: ds_add_u32
}

// (Round 4 measured two other orders of the row flush — the row strictly in ascending address order, the hi pieces parked
// in registers, with nontemporal or ordinary stores: 2 % at best at k = 6, 12-23 % slower at k = 7; DESIGN §4 — and
// removed them again: a lane step stores its lo piece and its hi piece back to back.)
template <int OUT, int WPS, bool TILES>
__global__ __launch_bounds__(WPS * 64) void count_rows_kernel(const CountArgs a) {
    constexpr int T = WPS * 64;
    constexpr int P = WPS == 1 ? 2 : 1;  // sweeps of packed words prefetched one item ahead (2 048 / 4 096 bases)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    const int k = a.k;
    const uint32_t nbins = 1u << (2 * k);
    const uint32_t nwords = nbins >> 1;                 // two 16-bit bins per word: bin b and bin b + nwords
    const uint32_t hist_words = nwords < 4 ? 4 : nwords;
    uint32_t* hist = lds;                               // [hist_words] | trash [64] | tab [16]
    const uint32_t trash_addr = (hist_words + (tid & 63)) * 4;  // one word per lane: no two lanes of a wave collide on it
    float* tab = reinterpret_cast<float*>(lds + hist_words + 64);
    const uint32_t sh = 30 - 2 * k;                     // (r >> sh) & amask = byte address of the window's word
    const uint32_t amask = (nwords - 1) << 2;
    const uint32_t win_mask = (1u << k) - 1u;           // k consecutive validity bits

    for (uint32_t w = tid * 4; w < hist_words + 64; w += T * 4) *reinterpret_cast<uint4*>(&hist[w]) = make_uint4(0, 0, 0, 0);

    // The packed words of the NEXT item are fetched while the current one is flushed.  Unconditional loads,
    // indices clamped into the padded arrays: a branch around them would make hipcc drain vmcnt(0) at the join.
    int64_t n_seq = 0, nL = 0, n_woff = 0, n_moff = -1, n_w0 = 0;
    uint32_t n_hi[P], n_lo[P];
    auto windows_of = [&](int64_t L, int64_t w0) -> int64_t {  // windows this item counts
        const int64_t Wtot = L - k + 1;
        int64_t Wi = Wtot - (w0 << 4);
        if (TILES) Wi = Wi < kItemWindows ? Wi : kItemWindows;
        else if (Wtot > kItemWindows) Wi = 0;  // a long sequence: its tiles are counted by the TILES launch
        return Wi > 0 ? Wi : 0;
    };
    auto prefetch = [&](int64_t it) {
        it = it < a.n_items ? it : a.n_items - 1;
        n_seq = TILES ? a.item_seq[it] : it;
        n_w0 = TILES ? a.item_word0[it] : 0;
        nL = a.len[n_seq];
        n_woff = a.word_off[n_seq] + n_w0;
        n_moff = a.mask_off[n_seq];
        const int64_t nww = (windows_of(nL, n_w0) + 15) >> 4;
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int64_t w = p * T + tid;
            const int64_t wc = w < nww ? w : nww;
            n_hi[p] = a.packed[n_woff + wc];
            n_lo[p] = a.packed[n_woff + wc + 1];
        }
    };
    prefetch(blockIdx.x);
    if (WPS > 1) __syncthreads();
    int64_t tab_W = INT64_MIN;  // window count the table was built for
    // current item: metadata and first sweeps in registers.  The loads of item i+1 are issued at the START of
    // item i's counting and consumed (moved into c_*) BEFORE item i's row is stored: a wave never waits for
    // its own stores — vmcnt counts loads and stores together, in order, so a load issued after a row's stores
    // could only be waited for with all of them.
    int64_t seq = n_seq, L = nL, woff = n_woff, moff = n_moff, w0 = n_w0;
    uint32_t c_hi[P], c_lo[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        c_hi[p] = n_hi[p];
        c_lo[p] = n_lo[p];
    }

    for (int64_t it = blockIdx.x; it < a.n_items; it += gridDim.x) {
        if (it + gridDim.x < a.n_items) prefetch(it + gridDim.x);  // in flight during the counting (persistent grids only)
        const int64_t Wtot = L - k + 1;  // windows, counting every character (kmer_counts.py:143-144)
        const int64_t Wi = windows_of(L, w0);
        const bool skip = !TILES && Wtot > kItemWindows;
        const int64_t nww = (Wi + 15) >> 4;
        const double inc = Wtot > 0 ? 1000.0 / (double)Wtot : 0.0;
        if ((OUT == OUT_F32 || OUT == OUT_F32_LOG2) && Wtot != tab_W) {
            // the item's output value for every small count (almost all bins): 16 lanes do the float64 work,
            // the flush just looks it up; sets of equal-length sequences build it once
            if (tid < kTabSize) {
                float t = per_kb_value((uint32_t)tid, inc);
                if (OUT == OUT_F32_LOG2) t = skr_log2_cr(t + 1.0f);  // kmer_counts.py:189-192: counts += 1; log2
                tab[tid] = t;
            }
            tab_W = Wtot;
        }

        auto sweep = [&](int64_t base, uint32_t hi, uint32_t lo) {
            const int64_t w = base + tid;
            if (moff < 0 && Wi - ((base + T - 1) << 4) >= 16) {
                // every lane of the workgroup has 16 whole windows.  Wave-level aggregation: if all 64 lanes
                // hold the same two words, each of the 16 columns would get 64 adds on one address
                const uint32_t h0 = __builtin_amdgcn_readfirstlane(hi), l0 = __builtin_amdgcn_readfirstlane(lo);
                const bool same = __builtin_amdgcn_ballot_w64(((hi ^ h0) | (lo ^ l0)) != 0) == 0;
                if (same) {
                    if ((tid & 63) == 0) {
#pragma unroll
                        for (int j = 0; j < 16; j++) {
                            const uint32_t r = j ? __builtin_amdgcn_alignbit(hi, lo, 32 - 2 * j) : hi;
                            lds_add_u32(hist, (r >> sh) & amask, (int32_t)r < 0 ? 0x400000u : 64u);
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const uint32_t r = j ? __builtin_amdgcn_alignbit(hi, lo, 32 - 2 * j) : hi;
                        lds_add_u32(hist, (r >> sh) & amask, (int32_t)r < 0 ? 0x10000u : 1u);
                    }
                }
            } else {
                // a sweep that holds the end of the item or non-alphabet bases: windows that do not count are
                // sent to a trash word (still counted in W: kmer_counts.py:143-149)
                const int64_t left = Wi - (w << 4);
                const int lim = left < 0 ? 0 : (left > 16 ? 16 : (int)left);
                uint32_t invalid = 0;  // bit j: base 16(w0+w)+j is not in the alphabet
                if (moff >= 0) {
                    const int64_t aw = w0 + (w < nww ? w : nww);
                    const uint32_t* mwords = a.mask + moff + (aw >> 1);
                    invalid = (uint32_t)(((unsigned long long)mwords[0] | ((unsigned long long)mwords[1] << 32)) >> ((aw & 1) * 16));
                }
                if (lim > 0) {  // lanes past the end of the item do nothing
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const uint32_t r = j ? __builtin_amdgcn_alignbit(hi, lo, 32 - 2 * j) : hi;
                        const bool ok = j < lim && ((invalid >> j) & win_mask) == 0;
                        lds_add_u32(hist, ok ? ((r >> sh) & amask) : trash_addr, (int32_t)r < 0 ? 0x10000u : 1u);
                    }
                }
            }
        };
#pragma unroll
        for (int p = 0; p < P; p++)
            if ((int64_t)p * T < nww) sweep((int64_t)p * T, c_hi[p], c_lo[p]);
        for (int64_t base = (int64_t)P * T; base < nww; base += T) {  // longer items
            const int64_t w = base + tid;
            const int64_t wc = w < nww ? w : nww;
            sweep(base, a.packed[woff + wc], a.packed[woff + wc + 1]);
        }
        // the next item's words have arrived (only the previous row's stores are older): take them now
        const int64_t t_seq = n_seq, t_L = nL, t_woff = n_woff, t_moff = n_moff, t_w0 = n_w0;
        uint32_t t_hi[P], t_lo[P];
#pragma unroll
        for (int p = 0; p < P; p++) {
            asm volatile("v_mov_b32 %0, %1" : "=v"(t_hi[p]) : "v"(n_hi[p]));
            asm volatile("v_mov_b32 %0, %1" : "=v"(t_lo[p]) : "v"(n_lo[p]));
        }
        if (WPS > 1) __syncthreads();
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // one wave: the LDS executes its instructions in order

        // ---- flush: bins -> output values, dense row to HBM; the bins are zeroed on the way out
        if (!skip) {
            auto value_of = [&](uint32_t n) -> float {
                if (n < (uint32_t)kTabSize) return tab[n];
                float t = per_kb_value(n, inc);
                if (OUT == OUT_F32_LOG2) t = skr_log2_cr(t + 1.0f);
                return t;
            };
            const size_t row = (size_t)(TILES ? it : seq) * nbins;
            typedef float f4 __attribute__((ext_vector_type(4)));
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            for (uint32_t w4 = tid * 4; w4 < nwords; w4 += T * 4) {
                const uint4 c = *reinterpret_cast<const uint4*>(&hist[w4]);
                *reinterpret_cast<uint4*>(&hist[w4]) = make_uint4(0, 0, 0, 0);
                if (nwords < 4) {  // k = 1: two words, four bins
                    const uint32_t cw[2] = {c.x, c.y};
                    for (int i = 0; i < 2; i++) {
                        if (OUT == OUT_U32) {
                            reinterpret_cast<uint32_t*>(a.out)[row + i] = cw[i] & 0xFFFFu;
                            reinterpret_cast<uint32_t*>(a.out)[row + 2 + i] = cw[i] >> 16;
                        } else {
                            reinterpret_cast<float*>(a.out)[row + i] = value_of(cw[i] & 0xFFFFu);
                            reinterpret_cast<float*>(a.out)[row + 2 + i] = value_of(cw[i] >> 16);
                        }
                    }
                } else if (OUT == OUT_U32) {
                    u4* dst = reinterpret_cast<u4*>(reinterpret_cast<uint32_t*>(a.out) + row + w4);
                    const u4 lo4{c.x & 0xFFFFu, c.y & 0xFFFFu, c.z & 0xFFFFu, c.w & 0xFFFFu};
                    const u4 hi4{c.x >> 16, c.y >> 16, c.z >> 16, c.w >> 16};
                    if (TILES) {  // read again in a moment by reduce_tiles_kernel: leave them in the L2
                        dst[0] = lo4;
                        *reinterpret_cast<u4*>(reinterpret_cast<uint32_t*>(dst) + nwords) = hi4;
                    } else {
                        __builtin_nontemporal_store(lo4, dst);
                        __builtin_nontemporal_store(hi4, reinterpret_cast<u4*>(reinterpret_cast<uint32_t*>(dst) + nwords));
                    }
                } else {
                    f4 lo4, hi4;
                    if (((c.x | c.y | c.z | c.w) & 0xFFF0FFF0u) == 0) {  // all eight counts below 16: table
                        lo4 = f4{tab[c.x & 15u], tab[c.y & 15u], tab[c.z & 15u], tab[c.w & 15u]};
                        hi4 = f4{tab[c.x >> 16], tab[c.y >> 16], tab[c.z >> 16], tab[c.w >> 16]};
                    } else {
                        lo4 = f4{value_of(c.x & 0xFFFFu), value_of(c.y & 0xFFFFu), value_of(c.z & 0xFFFFu), value_of(c.w & 0xFFFFu)};
                        hi4 = f4{value_of(c.x >> 16), value_of(c.y >> 16), value_of(c.z >> 16), value_of(c.w >> 16)};
                    }
                    // the row is written once and not read again by this kernel: keep it out of the L2
                    f4* dst = reinterpret_cast<f4*>(reinterpret_cast<float*>(a.out) + row + w4);
                    __builtin_nontemporal_store(lo4, dst);
                    __builtin_nontemporal_store(hi4, reinterpret_cast<f4*>(reinterpret_cast<float*>(dst) + nwords));
                }
            }
        }
        if (WPS > 1) __syncthreads();  // zeroed bins visible before the next item is counted
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        seq = t_seq, L = t_L, woff = t_woff, moff = t_moff, w0 = t_w0;
#pragma unroll
        for (int p = 0; p < P; p++) {
            c_hi[p] = t_hi[p];
            c_lo[p] = t_lo[p];
        }
    }
}

// Sum of a long sequence's tile histograms, in two steps so that a single chromosome-sized sequence still fills the
// chip: (1) workgroup (bin block, sequence, chunk of 32 tiles) adds its tiles and atomically adds the result to the
// sequence's uint32 sum row; (2) the sum row becomes the output row (same per-kb arithmetic as the flush above).
constexpr int kReduceChunk = 32;
__global__ __launch_bounds__(256) void reduce_tiles_kernel(const uint32_t* __restrict__ partial, const int64_t* __restrict__ tile_begin,
                                                           int k, uint32_t* __restrict__ sums) {
    const uint32_t nbins = 1u << (2 * k);
    const int64_t ls = blockIdx.y;
    const int64_t t0 = tile_begin[ls] + (int64_t)blockIdx.z * kReduceChunk;
    const int64_t t1 = std::min<int64_t>(tile_begin[ls + 1], t0 + kReduceChunk);
    if (t0 >= t1) return;
    for (uint32_t b = blockIdx.x * 256 + threadIdx.x; b < nbins; b += gridDim.x * 256) {
        uint32_t n = 0;
#pragma unroll 8
        for (int64_t t = t0; t < t1; t++) n += partial[(size_t)t * nbins + b];
        if (n) atomicAdd(&sums[(size_t)ls * nbins + b], n);
    }
}

template <int OUT>
__global__ __launch_bounds__(256) void convert_long_kernel(const uint32_t* __restrict__ sums, const int64_t* __restrict__ long_seq,
                                                           const int64_t* __restrict__ len, int k, void* __restrict__ out) {
    const uint32_t nbins = 1u << (2 * k);
    const int64_t ls = blockIdx.y;
    const int64_t seq = long_seq[ls];
    const double inc = 1000.0 / (double)(len[seq] - k + 1);
    for (uint32_t b = blockIdx.x * 256 + threadIdx.x; b < nbins; b += gridDim.x * 256) {
        const uint32_t n = sums[(size_t)ls * nbins + b];
        if (OUT == OUT_U32) {
            reinterpret_cast<uint32_t*>(out)[(size_t)seq * nbins + b] = n;
        } else {
            float v = per_kb_value(n, inc);
            if (OUT == OUT_F32_LOG2) v = skr_log2_cr(v + 1.0f);
            reinterpret_cast<float*>(out)[(size_t)seq * nbins + b] = v;
        }
    }
}

// k <= 8, float32 / uint32 output: count_rows_kernel over the sequences, then the tiles of the long ones.
template <int OUT, int WPS>
int launch_rows(skr_ctx* ctx, const skr_seqs* s, int k, void* out) {
    const uint32_t nbins = 1u << (2 * k);
    const size_t lds = ((size_t)std::max<uint32_t>(4u, nbins >> 1) + 64 + kTabSize) * 4;
    auto grid_for_kernel = [&](const void* kern, int threads, int64_t items, unsigned* grid) -> int {
        SKR_TRY(skr_kernel_lds(ctx, kern, lds));
        int per_cu = 0;
        auto occ = ctx->occupancy.find({kern, lds});
        if (occ == ctx->occupancy.end()) {
            SKR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds));
            ctx->occupancy[{kern, lds}] = per_cu;
        } else {
            per_cu = occ->second;
        }
        per_cu = std::max(1, std::min(per_cu, 2048 / threads));
        // Row streams in flight chip-wide: every wave (k <= 6) writes its own 16 KiB row.  12 per CU (3 072 rows, a 50 MB
        // window) wrote 6-10 % faster than the 19 the LDS allows, most clearly right after a kernel that left the caches
        // dirty (tools/count_bench.py --pre gemm: 0.155 vs 0.170 ms at 50 000 x 2 kb) — the HBM prefers fewer streams.
        per_cu = std::min(per_cu, 768 / threads);  // 12 waves per CU: 12 rows (k <= 6) or 3 rows of 64 KiB (k = 7) in flight
        if (ctx->knobs.count_percu) per_cu = std::min(per_cu, ctx->knobs.count_percu);  // A/B knob
        // persistent, statically strided items: every workgroup must be resident from the start
        *grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(items, (int64_t)ctx->num_cu * per_cu));
        return SKR_OK;
    };
    CountArgs a{s->d_packed, s->d_word_off, s->d_len, s->d_mask, s->d_mask_off, nullptr, nullptr, s->n, out, k};
    unsigned grid = 1;
    auto kern = count_rows_kernel<OUT, WPS, false>;
    SKR_TRY(grid_for_kernel(reinterpret_cast<const void*>(kern), WPS * 64, s->n, &grid));
    // One wave per sequence (k <= 6): NOT persistent — one workgroup per sequence, dispatched by the hardware in order, as
    // many resident as the LDS allows (19 per CU at k = 6).  The rows being written then form a compact front that
    // moves through the matrix, which is what the HBM writes fastest (tools/micro/store_pattern.hip); a persistent
    // grid whose waves stride through the sequences drifts apart.  Measured behind a contraction, 50 000 x 2 kb:
    // 0.146 ms against 0.165 for the persistent grid of 12 waves per CU (which needs the cap: 19 are slower still),
    // k = 5: 0.100 vs 0.108.
    // k = 7 (four waves per sequence): one workgroup per sequence as well since round 4 — 0.323 against 0.340 ms for the
    // persistent grid (30 000 x 5 kb behind a contraction: 0.775 against 0.737 of 8 TB/s); SEEKR_COUNT_PERSIST=1 restores it
    // (k = 8: one 128-KiB workgroup per CU — the persistent grid stays: a fresh workgroup per sequence would zero its 128 KiB of
    // bins every time, 0.63 against 0.70 of 8 TB/s for 20 000 x 2 kb)
    const bool persistent = ctx->knobs.count_persist == 1 || (ctx->knobs.count_persist == 0 && k >= 8);  // A/B knob
    if (!persistent) grid = (unsigned)std::min<int64_t>(s->n, 0x7fffffff);
    size_t lds_launch = lds;
    // Round 4: at k = 6 the LDS would let 19 one-wave workgroups share a CU; SEVENTEEN (enforced by asking for 9.25 KiB of
    // LDS each) write the rows 7-8 % faster behind a contraction — 0.138-0.140 ms against 0.149-0.150 for 50 000 x 2 kb,
    // 0.76 against 0.70 of 8 TB/s, two runs of tools/count_bench.py --pre gemm (profiles/r4_count_occupancy.log: 18 and 19
    // per CU 0.150, 17 and 16 0.139, 15 and 14 0.146, 12 0.160, 8 0.187) — fewer row streams, no SIMD with a fifth wave
    // for long.  Inside the bench step 17 measured 0.141-0.142 ms against 0.144-0.145 for 16 (two runs each), so 17 it is.
    // Smaller k (2 KiB of bins and less) are fastest unrestricted.  SEEKR_COUNT_OCC overrides.
    const int occ = ctx->knobs.count_occ > 0 ? ctx->knobs.count_occ : (WPS == 1 && k == 6 ? 17 : 0);
    if (!persistent && occ > 0) {
        lds_launch = std::max(lds, ((size_t)160 * 1024 / (size_t)occ) & ~(size_t)255);
        SKR_TRY(skr_kernel_lds(ctx, reinterpret_cast<const void*>(kern), lds_launch));
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WPS * 64), lds_launch, ctx->stream, a);
    SKR_HIP(hipGetLastError());
    if (s->max_len - k + 1 <= kItemWindows) return SKR_OK;

    // ---- long sequences: tiles of kItemWindows windows -> uint32 partial histograms -> reduce + convert
    std::vector<int64_t> long_seq, tile_begin{0}, item_seq, item_word0;
    int64_t max_tiles_of_one = 0;
    auto run_batch = [&]() -> int {
        const int64_t n_long = (int64_t)long_seq.size(), n_tiles = (int64_t)item_seq.size();
        if (n_long == 0) return SKR_OK;
        // index tables: one pinned staging buffer, one asynchronous copy
        const size_t n_idx = (size_t)(2 * n_tiles + 2 * n_long + 1);
        const size_t sums_off = (n_idx * sizeof(int64_t) + 255) & ~(size_t)255;
        const size_t part_off = (sums_off + (size_t)n_long * nbins * 4 + 255) & ~(size_t)255;
        void* ws = nullptr;
        SKR_TRY(skr_ctx_workspace(ctx, part_off + (size_t)n_tiles * nbins * 4, &ws));
        void* pin = nullptr;
        SKR_TRY(skr_ctx_pinned(ctx, n_idx * sizeof(int64_t), &pin));
        int64_t* h = reinterpret_cast<int64_t*>(pin);
        memcpy(h, item_seq.data(), (size_t)n_tiles * 8);
        memcpy(h + n_tiles, item_word0.data(), (size_t)n_tiles * 8);
        memcpy(h + 2 * n_tiles, long_seq.data(), (size_t)n_long * 8);
        memcpy(h + 2 * n_tiles + n_long, tile_begin.data(), (size_t)(n_long + 1) * 8);
        int64_t* d_item_seq = reinterpret_cast<int64_t*>(ws);
        int64_t* d_item_word0 = d_item_seq + n_tiles;
        int64_t* d_long_seq = d_item_word0 + n_tiles;
        int64_t* d_tile_begin = d_long_seq + n_long;
        uint32_t* d_sums = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ws) + sums_off);
        uint32_t* d_partial = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ws) + part_off);
        SKR_HIP(hipMemcpyAsync(d_item_seq, h, n_idx * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
        SKR_TRY(skr_ctx_pinned_used(ctx));
        SKR_HIP(hipMemsetAsync(d_sums, 0, (size_t)n_long * nbins * 4, ctx->stream));
        CountArgs t{s->d_packed, s->d_word_off, s->d_len, s->d_mask, s->d_mask_off, d_item_seq, d_item_word0, n_tiles, d_partial, k};
        auto tkern = count_rows_kernel<OUT_U32, 4, true>;
        unsigned tgrid = 1;
        SKR_TRY(grid_for_kernel(reinterpret_cast<const void*>(tkern), 256, n_tiles, &tgrid));
        hipLaunchKernelGGL(tkern, dim3(tgrid), dim3(256), lds, ctx->stream, t);
        SKR_HIP(hipGetLastError());
        const unsigned bx = (unsigned)std::min<uint32_t>((nbins + 255) / 256, 64);
        const unsigned chunks = (unsigned)std::min<int64_t>(65535, (max_tiles_of_one + kReduceChunk - 1) / kReduceChunk);
        hipLaunchKernelGGL(reduce_tiles_kernel, dim3(bx, (unsigned)n_long, std::max(1u, chunks)), dim3(256), 0, ctx->stream, d_partial,
                           d_tile_begin, k, d_sums);
        SKR_HIP(hipGetLastError());
        hipLaunchKernelGGL(convert_long_kernel<OUT>, dim3(bx, (unsigned)n_long), dim3(256), 0, ctx->stream, d_sums, d_long_seq,
                           s->d_len, k, out);
        SKR_HIP(hipGetLastError());
        long_seq.clear();
        item_seq.clear();
        item_word0.clear();
        tile_begin.assign(1, 0);
        max_tiles_of_one = 0;
        return SKR_OK;
    };
    const int64_t batch_tiles = std::max<int64_t>(1, ((int64_t)1 << 31) / ((int64_t)nbins * 4));  // ~2 GB of partial histograms
    for (int64_t i = 0; i < s->n; i++) {
        const int64_t W = s->h_len[i] - k + 1;
        if (W <= kItemWindows) continue;
        const int64_t tiles = (W + kItemWindows - 1) / kItemWindows;
        if (!item_seq.empty() && (int64_t)item_seq.size() + tiles > batch_tiles) SKR_TRY(run_batch());
        if ((int64_t)long_seq.size() >= 65535) SKR_TRY(run_batch());  // gridDim.y limit
        long_seq.push_back(i);
        max_tiles_of_one = std::max(max_tiles_of_one, tiles);
        for (int64_t t = 0; t < tiles; t++) {
            item_seq.push_back(i);
            item_word0.push_back(t * (kItemWindows / 16));
        }
        tile_begin.push_back((int64_t)item_seq.size());
    }
    return run_batch();
}

template <int OUT>
int launch_count(skr_ctx* ctx, const skr_seqs* s, int k, void* out, const char* name) {
    // k = 8 on the LDS path (round 4): 65 536 sixteen-bit bins packed two to a word are 128 KiB — one 4-wave workgroup per
    // CU — and an item never has more than 8 192 windows; SEEKR_COUNT_K8_GLOBAL=1 keeps the round-1 path for the A/B
    const bool lds_k8 = k == 8 && OUT != OUT_F64 && !ctx->knobs.count_k8_global;
    if (k > 7 && !lds_k8) {  // histogram in the output row itself
        if (s->n < 1) return SKR_OK;
        SKR_HIP(hipMemsetAsync(out, 0, (size_t)s->n * ((size_t)4 << (2 * k)), ctx->stream));
        const int64_t grid = std::min<int64_t>(s->n, (int64_t)ctx->num_cu * 8);
        SkrProfScope prof(ctx, name);
        hipLaunchKernelGGL((count_kmers_kernel<OUT, true>), dim3((unsigned)grid), dim3(kThreads), 0, ctx->stream, s->d_packed,
                           s->d_word_off, s->d_len, s->d_mask, s->d_mask_off, s->n, k, out, ctx->d_flags);
        SKR_HIP(hipGetLastError());
        return SKR_OK;
    }
    if (s->n < 1) return SKR_OK;
    const bool legacy = ctx->knobs.count_legacy && s->max_len - k + 1 <= 65535;  // A/B knob (tools/count_bench.py): the round-1 kernel
    if (OUT != OUT_F64 && !legacy) {
        SkrProfScope prof(ctx, name);
        // k <= 6: one wave per sequence (8 KiB of bins at k = 6); k = 7 / 8: 32 / 128 KiB of bins shared by four waves
        const int wps = ctx->knobs.count_wps ? ctx->knobs.count_wps : (k <= 6 ? 1 : 4);  // A/B knob
        constexpr int O = OUT == OUT_F64 ? OUT_F32 : OUT;  // (never instantiated for float64)
        return wps == 1 ? launch_rows<O, 1>(ctx, s, k, out) : launch_rows<O, 4>(ctx, s, k, out);
    }
    const size_t lds = (size_t)4 << (2 * k);
    SKR_TRY(skr_kernel_lds(ctx, reinterpret_cast<const void*>(count_kmers_kernel<OUT, false>), lds));
    // as many resident workgroups as LDS allows, capped by the wave limit (8 x 256 threads / CU)
    int per_cu = (int)std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1));
    if (per_cu < 1) per_cu = 1;
    int64_t grid = std::min<int64_t>(s->n, (int64_t)ctx->num_cu * per_cu);
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


// ---- round 4: A^k <= 16 384 columns (5 letters up to k = 6, 20 amino acids up to k = 3) count in the LDS -------------
// One 256-thread workgroup per sequence, persistent grid.  The sequence goes through the LDS in chunks of kGenChunk
// characters, translated to letter codes ONCE per character (lookup table in the LDS; -1 = not in the alphabet) instead
// of k times per window; a thread takes every 256th window of the chunk, builds its column from k code bytes and adds
// to a uint32 bin (ds_add_u32; uint32, so a sequence of any length is one item).  The flush converts with the reference's
// per-kb arithmetic (16-entry table per sequence, per_kb_value above), zeroes the bins and streams the row out with
// nontemporal stores — coalesced dwords: rows of A^k floats are 4- but not 16-byte aligned.  Against the round-1 form
// (memset of a uint32 histogram in HBM + L2 atomics + a conversion pass: >= 3 x the row bytes, plus the upload inside the
// call) the device traffic is the row write alone: 1 + 4 A^k / L bytes per base (SURVEY 8d's ASCII figure).
constexpr int kGenChunk = 4096;
constexpr int64_t kGenLdsBins = 36864;  // 144 KiB of bins + 8.6 KiB of tables and codes: ONE workgroup per CU above 19 K bins (round 5: was
                                        // 16 384 — 7^5, 3^9, 12^4, 8^5 columns went through the histogram in HBM at 0.04 of the peak)

constexpr int kGenThreads = 1024;  // two workgroups of 16 waves per CU at 62.5 KiB of bins: the flush is a latency-bound loop of
                                   // dword stores, so what counts is how many of them are in flight (256 threads: 1.32 ms for
                                   // 50 000 x 5^6 rows = 0.31 of HBM)
// (Round 5 measured 16-bit bins, two to a word, with four 512-thread workgroups per CU instead of two of 1 024: 1.16 ms against
// 0.76 ms — the sub-dword LDS reads and writes of the flush cost more than the halved footprint buys;
// profiles/r5_generic_width_arms.log.  Not kept.)
template <typename OutT, bool LOG2>
// (at most 72 SGPRs: with the 86 the compiler would take, the CU holds ONE sixteen-wave workgroup of this kernel instead of two —
// hipOccupancyMaxActiveBlocksPerMultiprocessor says two either way; measured with per-workgroup start times: the second 256
// workgroups started when the first 256 had ended.  The surplus goes to VGPR lanes, 18 of them.)
__global__ __launch_bounds__(kGenThreads) __attribute__((amdgpu_num_vgpr(56), amdgpu_num_sgpr(72))) void count_generic_lds_kernel(const unsigned char* __restrict__ bases,
                                                                  const int64_t* __restrict__ offsets, int64_t n_seqs, int k,
                                                                  int alen, uint32_t nbins, GenericLut lut,
                                                                  OutT* __restrict__ out, int fast_tab, uint32_t row_bins,
                                                                  uint32_t bin_lo) {
    // `nbins` bins of the row live in the LDS: bins [bin_lo, bin_lo + nbins) of a row of `row_bins` cells.  A row wider than the
    // LDS (5^7, 6^6 ... columns) is counted in several launches, each over ALL windows and one range of bins.
    extern __shared__ __attribute__((aligned(16))) uint32_t glds[];
    const uint32_t words_pad = (nbins + 3u) & ~3u;
    uint32_t* bins = glds;                                                 // [nbins]
    float* tab = reinterpret_cast<float*>(glds + words_pad);               // [16]
    int8_t* lutb = reinterpret_cast<int8_t*>(glds + words_pad + kTabSize);  // [256]
    int8_t* codes = lutb + 256;                                            // [kGenChunk + 64]
    const int tid = threadIdx.x;
    for (uint32_t b = tid; b < words_pad; b += kGenThreads) glds[b] = 0;
    if (tid < 256) lutb[tid] = lut.code[tid];
    __syncthreads();
    // Round 5: the characters of the NEXT sequence's first chunk are requested at the TOP of this sequence's turn (its
    // offsets were read one turn earlier still), translated into a SECOND code buffer after this sequence's histogram and
    // found there, ready, by the next turn: no load stands in front of a sequence, the translate phase and its barrier are
    // gone, and — the point — nothing at the top of a turn depends on vmcnt any more.  The first form (characters requested
    // before the flush, consumed at the top of the next turn) made every wave wait there for vmcnt(0): the loads are older
    // than the flush's stores, but the number of stores a wave issues in a loop is not known to the compiler, so "the loads
    // have landed" became "everything has landed" and each turn sat out the write latency of its own row.  All barriers of
    // the loop wait for the LDS only (every hazard in it is an LDS one).
    constexpr int kGenPre = (kGenChunk + 64 + kGenThreads - 1) / kGenThreads;
    int8_t* codes_next = codes + kGenChunk + 64;
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    const int64_t stride = gridDim.x;
    // characters in flight: `pre` for the next sequence (requested one turn ago), `pre2` for the one after it (requested at the
    // top of this turn).  The requests are UNCONDITIONAL loads of exactly kGenPre bytes a thread (addresses clamped into the
    // buffer): the wait in front of the translation can then be "all but the kGenPre youngest" — with loads inside
    // `i < n ? ... : 0` branches the compiler cannot count them and waits for everything, i.e. for the request it has just made.
    unsigned char pre[kGenPre], pre2[kGenPre];
    const int64_t last_byte = std::max<int64_t>(offsets[n_seqs] - 1, 0);
    auto read_offsets = [&](int64_t sn, int64_t& off, int64_t& len) {
        off = 0, len = 0;
        if (sn < n_seqs) {
            off = offsets[sn];
            len = offsets[sn + 1] - off;
        }
    };
    // (32-bit lane arithmetic throughout the loop: the kernel is bound by VALU issue — a wave64 instruction takes four cycles —
    // and 64-bit compares / adds per character and per window were a third of what a turn issued)
    auto request = [&](unsigned char (&dst)[kGenPre], int64_t off) {  // the first chunk's characters of the sequence at `off`
        // (a trailing empty sequence has off == total: the address itself is clamped too, or bases[total] — one past the
        // allocation — would be read and thrown away, a fault when `total` ends on an allocation boundary)
        const int64_t at = std::min<int64_t>(off, last_byte);
        const unsigned char* base = bases + at;  // wave-uniform
        const uint32_t lim = (uint32_t)std::min<int64_t>(last_byte - at, kGenChunk + 64);
#pragma unroll
        for (int j = 0; j < kGenPre; j++) dst[j] = base[std::min<uint32_t>((uint32_t)tid + (uint32_t)kGenThreads * j, lim)];
    };
    auto translate = [&](int8_t* dst, const unsigned char (&src)[kGenPre], int64_t len) {
        const uint32_t n_first = (uint32_t)std::min<int64_t>(std::min<int64_t>(len, kGenChunk + k - 1), kGenChunk + 64);
#pragma unroll
        for (int j = 0; j < kGenPre; j++) {
            const uint32_t i = (uint32_t)tid + (uint32_t)kGenThreads * j;
            if (i < n_first) dst[i] = lutb[src[j]];
        }
    };
    int64_t cur_off, cur_len, nxt_off, nxt_len, nn_off, nn_len;
    read_offsets(blockIdx.x, cur_off, cur_len);
    read_offsets(blockIdx.x + stride, nxt_off, nxt_len);
    read_offsets(blockIdx.x + 2 * stride, nn_off, nn_len);
    request(pre, cur_off);
    translate(codes, pre, cur_len);
    request(pre, nxt_off);
    lds_barrier();
    // one sequence; `use` holds the NEXT sequence's characters (requested a turn ago), `req` takes those of the one after it —
    // the two register sets swap roles from turn to turn (a copy would be a wait for the request just made)
    auto turn = [&](int64_t s, unsigned char (&use)[kGenPre], unsigned char (&req)[kGenPre], int8_t* codes, int8_t* codes_next) {
        const unsigned char* seq = bases + cur_off;
        const int64_t len = cur_len;
        request(req, nn_off);  // the sequence after next: a whole turn to land
        int64_t n3_off, n3_len;
        read_offsets(s + 3 * stride, n3_off, n3_len);
        const int64_t W = len - k + 1;  // windows, counting every character (kmer_counts.py:143-144)
        const double inc = W > 0 ? 1000.0 / (double)W : 0.0;
        if (!std::is_same<OutT, uint32_t>::value && sizeof(OutT) == 4 && tid < kTabSize) {
            float t = per_kb_value((uint32_t)tid, inc);
            if (LOG2) t = skr_log2_cr(t + 1.0f);
            tab[tid] = t;
        }
        for (int64_t c0 = 0; c0 < W; c0 += kGenChunk) {
            if (c0 > 0) {  // later chunks of a long sequence: translated here, into the buffer the first chunk came in
                const uint32_t n_char = (uint32_t)std::min<int64_t>(std::min<int64_t>(len - c0, kGenChunk + k - 1), kGenChunk + 64);
                const unsigned char* chunk = seq + c0;  // wave-uniform
                for (uint32_t i = tid; i < n_char; i += kGenThreads) codes[i] = lutb[chunk[i]];
                lds_barrier();
            }
            const uint32_t n_win = (uint32_t)std::min<int64_t>(W - c0, kGenChunk);
            for (uint32_t w = tid; w < n_win; w += kGenThreads) {
                // (idx < row_bins <= 2^24 and alen <= 127: a 24-bit multiply-add, full rate, instead of v_mad_u64_u32; two
                // letters per turn of the loop: two LDS reads in flight instead of one read -> wait -> multiply)
                uint32_t idx = 0;
                int bad = 0, p = 0;
                for (; p + 2 <= k; p += 2) {
                    const int c = codes[w + p], d = codes[w + p + 1];
                    bad |= c | d;  // the sign bit survives: any letter outside the alphabet
                    idx = __umul24(__umul24(idx, (uint32_t)alen) + (uint32_t)(c & 127), (uint32_t)alen) + (uint32_t)(d & 127);
                }
                if (p < k) {
                    const int c = codes[w + p];
                    bad |= c;
                    idx = __umul24(idx, (uint32_t)alen) + (uint32_t)(c & 127);
                }
                idx -= bin_lo;  // (wraps for a bin below the range)
                if (bad >= 0 && idx < nbins) (void)__hip_atomic_fetch_add(&bins[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            lds_barrier();  // the bins are complete (and `codes` may be overwritten by the next chunk); the table is visible
        }
        translate(codes_next, use, nxt_len);  // BEFORE the flush issues its stores: the turn's one wait for vmcnt (all but `req`)
        if (W <= 0) lds_barrier();  // a sequence without a single window: the table still has to reach every wave's flush
                                    // (no chunk, hence no other barrier in between; found by the differential fuzzer)
        OutT* row = out + (size_t)s * row_bins + bin_lo;
        auto value_of = [&](uint32_t n) -> OutT {
            if (std::is_same<OutT, uint32_t>::value) return (OutT)n;
            if (sizeof(OutT) == 8) return (OutT)per_kb_value_f64(n, inc);
            float t;
            if (n < (uint32_t)kTabSize) {
                t = tab[n];
            } else {
                t = per_kb_value(n, inc);
                if (LOG2) t = skr_log2_cr(t + 1.0f);
            }
            return (OutT)t;
        };
        if (sizeof(OutT) == 4) {
            // Round 5: a group is four bins that are ALIGNED IN THE LDS (b = 4 g: one ds_read_b128, one ds_write_b128 of
            // zeros) and goes out as one 16-byte store to wherever the row puts it — rows of A^k four-byte cells start on
            // 4-byte boundaries only, and a dword-aligned global_store_dwordx4 is as good as an aligned one.  Until now the
            // groups were aligned in MEMORY instead: four ds_read_b32 + four ds_write_b32 each, and a head / tail path.
            typedef OutT v4u __attribute__((ext_vector_type(4), aligned(4)));
            const uint32_t groups = nbins >> 2;
            for (uint32_t g = tid; g < groups; g += kGenThreads) {
                const uint32_t b = 4 * g;
                const uint4 n = *reinterpret_cast<const uint4*>(bins + b);
                *reinterpret_cast<uint4*>(bins + b) = make_uint4(0, 0, 0, 0);
                v4u o;
                if (fast_tab && !std::is_same<OutT, uint32_t>::value && sizeof(OutT) == 4 && __all((n.x | n.y | n.z | n.w) < (uint32_t)kTabSize)) {
                    // every count of the wave's 256 bins is in the table (a row of a few thousand windows over thousands of
                    // bins: nearly always) — four table reads in flight and no branch per value.  Only where two workgroups
                    // share a CU (`fast_tab`): measured on one box, 0.287 -> 0.272 ms at 3 125 bins and 0.324 -> 0.30 at 8 000,
                    // but 0.60 -> 0.69 ms at 15 625 bins with one workgroup per CU
                    o = v4u{(OutT)tab[n.x], (OutT)tab[n.y], (OutT)tab[n.z], (OutT)tab[n.w]};
                } else {
                    o = v4u{value_of(n.x), value_of(n.y), value_of(n.z), value_of(n.w)};
                }
                __builtin_nontemporal_store(o, reinterpret_cast<v4u*>(row + b));
            }
            if (tid < (nbins & 3u)) {  // the last one to three cells
                const uint32_t b = 4 * groups + tid;
                const uint32_t n = bins[b];
                bins[b] = 0;
                row[b] = value_of(n);
            }
        } else {
            for (uint32_t b = tid; b < nbins; b += kGenThreads) {
                const uint32_t n = bins[b];
                bins[b] = 0;
                __builtin_nontemporal_store(value_of(n), row + b);
            }
        }
        lds_barrier();  // zeroed bins, the table and the next sequence's codes are settled before the next turn
        cur_off = nxt_off, cur_len = nxt_len, nxt_off = nn_off, nxt_len = nn_len, nn_off = n3_off, nn_len = n3_len;
    };
    for (int64_t s = blockIdx.x; s < n_seqs; s += 2 * stride) {
        turn(s, pre, pre2, codes, codes_next);  // (the code buffers swap roles with the register sets: fixed LDS addresses in
        if (s + stride < n_seqs) turn(s + stride, pre2, pre, codes_next, codes);  // each body, not a pointer that is exchanged)
    }
}

}  // namespace

// ASCII sequences resident on the device (any alphabet): what skr_count_generic_dev counts from, any number of times.
struct skr_aseqs {
    skr_ctx* ctx = nullptr;
    int64_t n = 0;
    size_t total = 0;
    unsigned char* d_bases = nullptr;
    int64_t* d_off = nullptr;  // [n + 1], relative to d_bases
    std::vector<int64_t> h_len;
};

extern "C" int skr_aseqs_create(skr_ctx* ctx, const char* bases, const int64_t* offsets, int64_t n, skr_aseqs** out) {
    SKR_REQUIRE(ctx && out, "NULL argument");
    *out = nullptr;
    SKR_REQUIRE(n >= 0 && (n == 0 || offsets), "bad sequence count / offsets");
    for (int64_t i = 0; i < n; i++) SKR_REQUIRE(offsets[i + 1] >= offsets[i], "offsets must be non-decreasing (at %lld)", (long long)i);
    SKR_REQUIRE(n == 0 || bases || offsets[n] == offsets[0], "bases is NULL");
    SKR_TRY(skr_activate(ctx));
    skr_aseqs* a = new skr_aseqs();
    a->ctx = ctx;
    a->n = n;
    a->total = n ? (size_t)(offsets[n] - offsets[0]) : 0;
    a->h_len.resize((size_t)n);
    std::vector<int64_t> rel((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; i++) {
        a->h_len[(size_t)i] = offsets[i + 1] - offsets[i];
        rel[(size_t)i + 1] = offsets[i + 1] - offsets[0];
    }
    hipError_t e = hipMalloc((void**)&a->d_bases, a->total + 64);  // slack: the counter's prefetch is clamped, this is the belt
    if (e == hipSuccess) e = hipMalloc((void**)&a->d_off, (size_t)(n + 1) * sizeof(int64_t));
    if (e == hipSuccess && a->total) e = hipMemcpyAsync(a->d_bases, bases + offsets[0], a->total, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(a->d_off, rel.data(), (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // `rel` and the caller's buffers may go
    if (e != hipSuccess) {
        if (a->d_bases) (void)hipFree(a->d_bases);
        if (a->d_off) (void)hipFree(a->d_off);
        delete a;
        return skr_set_error(e == hipErrorOutOfMemory ? SKR_ERR_NOMEM : SKR_ERR_HIP, "uploading %lld ASCII sequences: %s", (long long)n,
                             hipGetErrorString(e));
    }
    *out = a;
    return SKR_OK;
}

extern "C" int skr_aseqs_free(skr_aseqs* a) {
    if (!a) return SKR_OK;
    (void)skr_activate(a->ctx);
    (void)hipStreamSynchronize(a->ctx->stream);
    if (a->d_bases) (void)hipFree(a->d_bases);
    if (a->d_off) (void)hipFree(a->d_off);
    delete a;
    return SKR_OK;
}

extern "C" int skr_count_generic_dev(skr_ctx* ctx, const skr_aseqs* a, const char* alphabet, int alen, int k, int log2_pre,
                                     skr_mat* out) {
    SKR_REQUIRE(ctx && a && alphabet && out && out->ctx == ctx && a->ctx == ctx, "NULL or foreign argument");
    SKR_REQUIRE(alen >= 1 && alen <= 127 && k >= 1, "alphabet of 1..127 letters and k >= 1 expected");
    double bins_d = 1.0;
    for (int p = 0; p < k; p++) bins_d *= alen;
    if (bins_d > (double)(1 << 26))
        return skr_set_error(SKR_ERR_UNSUPPORTED, "%d^%d columns: rows are supported up to 2^26 columns", alen, k);
    const int64_t nbins = (int64_t)bins_d, n = a->n;
    SKR_REQUIRE(out->rows == n && out->cols == nbins, "output must be [%lld, %lld], got [%lld, %lld]", (long long)n,
                (long long)nbins, (long long)out->rows, (long long)out->cols);
    SKR_REQUIRE(!(log2_pre && out->dtype != SKR_F32), "log2_pre is implemented for SKR_F32 output only");
    if (n == 0) return SKR_OK;
    if (out->dtype != SKR_U32)
        for (int64_t L : a->h_len)
            if (L == k - 1)  // kmer_counts.py:144
                return skr_set_error(SKR_ERR_ZERODIV, "division by zero");  // the text Python gives `1000 / 0` (kmer_counts.py:144)
    GenericLut lut;
    memset(lut.code, -1, sizeof(lut.code));
    // a repeated letter keeps its LAST position, as the reference's dict {kmer: index} does (:122)
    for (int c = 0; c < alen; c++) lut.code[(unsigned char)alphabet[c]] = (int8_t)c;
    SKR_TRY(skr_activate(ctx));
    if (nbins <= ((int64_t)1 << 24) && k <= 64 && !ctx->knobs.count_generic_global) {
        // the histogram — or, for rows wider than the LDS (round 5: 5^7, 6^6, 20^4 ... columns, until then counted with L2
        // atomics into a histogram in HBM at 0.04 of the peak), one RANGE of its bins per launch — lives in the LDS: the row
        // write is the only traffic that matters, the characters (a few kB a sequence) are simply read once per range
        const int64_t range = std::min<int64_t>(nbins, kGenLdsBins);
        const size_t lds = (size_t)((range + 3) & ~(int64_t)3) * 4 + kTabSize * 4 + 256 + 2 * (kGenChunk + 64);
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2048 / kGenThreads, ((size_t)160 * 1024) / lds));
        // Two workgroups per CU overlap one's flush with the other's histogram — except when the rows are long and the
        // sequences short (the flush is nearly all of a turn): measured at 50 000 x 2 kb, 14 641 / 15 625 bins: 0.735 / 0.770 ms
        // with two workgroups per CU against 0.563 / 0.594 with one; 10 000 bins and fewer, or 20 kb sequences: two win
        // (0.378 against 0.479 ms at 10 000 bins).  (tools/count_generic_bench.py)
        double mean_len = 0.0;
        for (int64_t L : a->h_len) mean_len += (double)L;
        mean_len /= (double)n;
        int want_per_cu = (range >= 12288 && 2.0 * mean_len < (double)range) ? 1 : per_cu;
        if (ctx->knobs.count_generic_wgs >= 1) want_per_cu = std::min(per_cu, ctx->knobs.count_generic_wgs);  // A/B knob
        const unsigned grid = (unsigned)std::min<int64_t>(n, (int64_t)ctx->num_cu * want_per_cu);
        SkrProfScope prof(ctx, "count_generic");
#define SKR_GEN_LAUNCH(T, LG)                                                                                              \
    do {                                                                                                                   \
        auto kern = count_generic_lds_kernel<T, LG>;                                                                       \
        SKR_TRY(skr_kernel_lds(ctx, reinterpret_cast<const void*>(kern), lds));                                            \
        for (int64_t lo = 0; lo < nbins; lo += range)                                                                      \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kGenThreads), lds, ctx->stream, a->d_bases, a->d_off, n, k, alen,       \
                               (uint32_t)std::min<int64_t>(range, nbins - lo), lut, (T*)out->data, want_per_cu > 1 ? 1 : 0,   \
                               (uint32_t)nbins, (uint32_t)lo);                                                               \
    } while (0)
        if (out->dtype == SKR_U32) SKR_GEN_LAUNCH(uint32_t, false);
        else if (out->dtype == SKR_F64) SKR_GEN_LAUNCH(double, false);
        else if (log2_pre) SKR_GEN_LAUNCH(float, true);
        else SKR_GEN_LAUNCH(float, false);
#undef SKR_GEN_LAUNCH
        SKR_HIP(hipGetLastError());
        return SKR_OK;
    }
    // wider rows: uint32 scratch histogram in HBM, a batch of sequences at a time (L2 atomics), then the conversion pass
    uint32_t* d_hist = nullptr;
    const int64_t batch = std::max<int64_t>(1, std::min<int64_t>(n, ((int64_t)256 << 20) / (nbins * 4)));
    hipError_t e = hipMalloc((void**)&d_hist, (size_t)batch * nbins * 4);
    for (int64_t s0 = 0; s0 < n && e == hipSuccess; s0 += batch) {
        const int64_t ns = std::min(batch, n - s0);
        e = hipMemsetAsync(d_hist, 0, (size_t)ns * nbins * 4, ctx->stream);
        if (e != hipSuccess) break;
        SkrProfScope prof(ctx, "count_generic");
        hipLaunchKernelGGL(count_generic_kernel, dim3((unsigned)std::min<int64_t>(ns, (int64_t)ctx->num_cu * 8)), dim3(kThreads),
                           0, ctx->stream, a->d_bases, a->d_off, s0, ns, k, alen, nbins, lut, d_hist);
        const dim3 cgrid((unsigned)std::min<int64_t>((nbins + kThreads - 1) / kThreads, 64), (unsigned)std::min<int64_t>(ns, 4096));
        if (out->dtype == SKR_U32)
            hipLaunchKernelGGL((convert_generic_kernel<uint32_t, false>), cgrid, dim3(kThreads), 0, ctx->stream, d_hist, a->d_off, s0,
                               ns, k, nbins, (uint32_t*)out->data);
        else if (out->dtype == SKR_F64)
            hipLaunchKernelGGL((convert_generic_kernel<double, false>), cgrid, dim3(kThreads), 0, ctx->stream, d_hist, a->d_off, s0, ns,
                               k, nbins, (double*)out->data);
        else if (log2_pre)
            hipLaunchKernelGGL((convert_generic_kernel<float, true>), cgrid, dim3(kThreads), 0, ctx->stream, d_hist, a->d_off, s0, ns,
                               k, nbins, (float*)out->data);
        else
            hipLaunchKernelGGL((convert_generic_kernel<float, false>), cgrid, dim3(kThreads), 0, ctx->stream, d_hist, a->d_off, s0, ns,
                               k, nbins, (float*)out->data);
        e = hipGetLastError();
    }
    (void)hipStreamSynchronize(ctx->stream);
    if (d_hist) (void)hipFree(d_hist);
    if (e != hipSuccess) return skr_set_error(SKR_ERR_HIP, "generic-alphabet counting failed: %s", hipGetErrorString(e));
    return SKR_OK;
}

// Host buffers in: upload, count, release (the drop-in API's path: BasicCounter with an alphabet the 2-bit kernels do not
// cover).  Callers that count the same sequences again keep a skr_aseqs instead.
extern "C" int skr_count_generic(skr_ctx* ctx, const char* bases, const int64_t* offsets, int64_t n, const char* alphabet,
                                 int alen, int k, int log2_pre, skr_mat* out) {
    SKR_REQUIRE(ctx && alphabet && out && out->ctx == ctx, "NULL or foreign argument");
    skr_aseqs* a = nullptr;
    SKR_TRY(skr_aseqs_create(ctx, bases, offsets, n, &a));
    const int rc = skr_count_generic_dev(ctx, a, alphabet, alen, k, log2_pre, out);
    (void)skr_aseqs_free(a);
    return rc;
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
            return skr_set_error(SKR_ERR_ZERODIV, "division by zero");  // the text Python gives `1000 / 0` (kmer_counts.py:144)
    if (out->dtype == SKR_F64) {
        if (k > 7) return skr_set_error(SKR_ERR_UNSUPPORTED, "float64 count output is implemented for k <= 7");
        return launch_count<OUT_F64>(ctx, s, k, out->data, "count_kmers_f64");
    }
    if (log2_pre) return launch_count<OUT_F32_LOG2>(ctx, s, k, out->data, "count_kmers_f32_log2");
    return launch_count<OUT_F32>(ctx, s, k, out->data, "count_kmers_f32");
}

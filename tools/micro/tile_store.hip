// What bounds the epilogue of the split contraction (pearson_bf16.hip)?  A persistent 512-thread workgroup per CU writes the
// 256 x 256 tiles of a 50 000 x 50 000 float32 self-comparison (tiles on and above the diagonal + their mirrors: 10 GB), every
// wave its 128 x 64 part of the tile and its 64 x 128 part of the mirror, with 16-byte nontemporal stores whose 64 lanes
// cover R rows x (1024 / R) bytes: R = 16 is what the kernel's MFMA layout gives (after the quad transpose), 8 / 4 would
// take one or two more lane exchanges.  No arithmetic, no loads: the rate the stores alone can reach, chip-wide and (grid = 1)
// for one CU by itself.
//   hipcc --offload-arch=gfx950 -O3 -o tile_store tile_store.hip && ./tile_store
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int R, bool MIRROR, bool NT>
__global__ __launch_bounds__(512) void tiles(float* out, long n, int tiles_n, unsigned* counter, long n_slots) {
    __shared__ long s_slot;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3;
    constexpr int LPR = 64 / R;  // lanes per row: each 16 bytes
    for (;;) {
        if (threadIdx.x == 0) s_slot = atomicAdd(counter, 1u);
        __syncthreads();
        const long slot = s_slot;
        __syncthreads();
        if (slot >= n_slots) return;
        const long tm = slot / tiles_n, tn = slot % tiles_n;
        if (tn < tm) continue;
        const f4 v{(float)slot, 1.f, 2.f, 3.f};
        // direct: rows tm*256 + wm*128 .. +128, cols tn*256 + wn*64 .. +64 (256 bytes per row)
        {
            float* base = out + (tm * 256 + wm * 128) * n + tn * 256 + wn * 64;
            constexpr int CH = (LPR * 4 > 64) ? 64 : LPR * 4;          // floats per row chunk covered by one instruction
            constexpr int RR = 256 / CH;                               // rows per instruction (64 lanes x 4 floats = 256 floats)
#pragma unroll 4
            for (int i = 0; i < 128 * 64 / 256; i++) {                 // 32 instructions
                const int e = i * 256 + lane * 4;                      // element index in a [128 / RR blocks][RR rows][CH] walk
                const int blk = e / (RR * CH), in = e % (RR * CH);
                const int chunks_per_row = 64 / CH;
                const int row = (blk / chunks_per_row) * RR + in / CH, col = (blk % chunks_per_row) * CH + in % CH;
                if (tm * 256 + wm * 128 + row < n && tn * 256 + wn * 64 + col + 3 < n) {
                    if (NT) __builtin_nontemporal_store(v, (f4*)(base + (long)row * n + col));
                    else *(f4*)(base + (long)row * n + col) = v;
                }
            }
        }
        if (MIRROR && tm != tn) {  // rows tn*256 + wn*64 .. +64, cols tm*256 + wm*128 .. +128 (512 bytes per row)
            float* base = out + (tn * 256 + wn * 64) * n + tm * 256 + wm * 128;
            constexpr int CH = (LPR * 4 > 128) ? 128 : LPR * 4;
            constexpr int RR = 256 / CH;
#pragma unroll 4
            for (int i = 0; i < 32; i++) {
                const int e = i * 256 + lane * 4;
                const int blk = e / (RR * CH), in = e % (RR * CH);
                const int chunks_per_row = 128 / CH;
                const int row = (blk / chunks_per_row) * RR + in / CH, col = (blk % chunks_per_row) * CH + in % CH;
                if (tn * 256 + wn * 64 + row < n && tm * 256 + wm * 128 + col + 3 < n) {
                    if (NT) __builtin_nontemporal_store(v, (f4*)(base + (long)row * n + col));
                    else *(f4*)(base + (long)row * n + col) = v;
                }
            }
        }
    }
}

int main() {
    const long n = 50000;
    const int tiles_n = (int)((n + 255) / 256);
    float* out; CK(hipMalloc(&out, (size_t)n * n * 4));
    unsigned* counter; CK(hipMalloc(&counter, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern, int grid, long n_slots, double gb) {
        std::vector<float> ms;
        for (int it = 0; it < 9; it++) {
            (void)hipMemsetAsync(counter, 0, 4);
            (void)hipEventRecord(e0);
            kern<<<grid, 512>>>(out, n, tiles_n, counter, n_slots);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float t; (void)hipEventElapsedTime(&t, e0, e1);
            if (it >= 2) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-46s grid %3d  median %8.4f ms  %7.0f GB/s  (%.1f B/clk/CU at 2.0 GHz)\n", name, grid, ms[ms.size() / 2], gb / ms[ms.size() / 2] * 1e3,
               gb * 1e9 / (ms[ms.size() / 2] * 1e-3) / grid / 2.0e9);
        return 0;
    };
    const long all = (long)tiles_n * tiles_n;
    const double gb_all = (double)n * n * 4 / 1e9;
    // chip-wide: all tiles of the upper triangle + mirrors
    run("R=16 (64 B runs, today) nt + mirror", tiles<16, true, true>, 256, all, gb_all);
    run("R=8  (128 B runs) nt + mirror", tiles<8, true, true>, 256, all, gb_all);
    run("R=4  (256 B runs) nt + mirror", tiles<4, true, true>, 256, all, gb_all);
    run("R=2  (256/512 B runs) nt + mirror", tiles<2, true, true>, 256, all, gb_all);
    run("R=16 plain stores + mirror", tiles<16, true, false>, 256, all, gb_all);
    run("R=8  plain stores + mirror", tiles<8, true, false>, 256, all, gb_all);
    run("R=4  plain stores + mirror", tiles<4, true, false>, 256, all, gb_all);
    // one CU by itself: the first tile row only (196 tiles, 195 mirrors)
    const double gb_row = (double)(2 * tiles_n - 1) * 256 * 256 * 4 / 1e9;
    run("one CU: R=16 nt + mirror", tiles<16, true, true>, 1, tiles_n, gb_row);
    run("one CU: R=8  nt + mirror", tiles<8, true, true>, 1, tiles_n, gb_row);
    run("one CU: R=4  nt + mirror", tiles<4, true, true>, 1, tiles_n, gb_row);
    run("one CU: R=2  nt + mirror", tiles<2, true, true>, 1, tiles_n, gb_row);
    run("one CU: R=16 plain + mirror", tiles<16, true, false>, 1, tiles_n, gb_row);
    run("one CU: R=4 plain + mirror", tiles<4, true, false>, 1, tiles_n, gb_row);
    run("8 CUs: R=16 nt + mirror", tiles<16, true, true>, 8, 8L * tiles_n, gb_row * 7.9);
    run("8 CUs: R=4 nt + mirror", tiles<4, true, true>, 8, 8L * tiles_n, gb_row * 7.9);
    run("64 CUs: R=16 nt + mirror", tiles<16, true, true>, 64, 64L * tiles_n, gb_row * 56);
    run("64 CUs: R=4 nt + mirror", tiles<4, true, true>, 64, 64L * tiles_n, gb_row * 56);
    return 0;
}

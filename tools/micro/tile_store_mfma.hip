// Can the epilogue's stores drain UNDER matrix work when they are dealt out between the MFMAs (the arm profiles/
// r6_epilogue_fused.log 5c names and did not build: "only stores dealt out BETWEEN the MFMAs of the next tile's first k steps
// could move them under matrix work")?  The tile walk of tile_store.hip (50 000 x 50 000 float32 self-comparison, 256 x 256
// tiles + mirrors, 4 rows x 256 B per store instruction, persistent 512-thread workgroup per CU), every wave additionally
// issuing the MFMAs of a K-column contraction per tile (96 v_mfma_f32_16x16x32_f16 per 32-k tile, operands from registers: no
// LDS, no staging loads) — either BEFORE the tile's stores (what the kernel does), or with one store instruction after every
// (96 * kt / 64) MFMAs, or alone.  kt = 8 is k = 4 (256 columns), kt = 32 is k = 5, kt = 128 is k = 6.
//   hipcc --offload-arch=gfx950 -O3 -o tile_store_mfma tile_store_mfma.hip && ./tile_store_mfma
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: MFMAs then stores; 1: interleaved; 2: stores only; 3: MFMAs only
// WG = 512: one workgroup per CU, all eight waves in the same phase (the kernel).  WG = 256 / 128: two / four independent
// workgroups per CU, each taking a 128-row half (64-row quarter... here: the same per-wave work, a slot = tile x part) — their
// phases drift apart, so some waves issue stores while others multiply.
template <int MODE, int WG = 512>
__global__ __launch_bounds__(WG) void tiles(float* out, long n, int tiles_n, unsigned* counter, long n_slots, int kt, float seed) {
    __shared__ long s_slot;
    constexpr int PARTS = 512 / WG;
    const int lane = threadIdx.x & 63, wave0 = threadIdx.x >> 6;
    f4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = f4{seed, seed, seed, seed};
    h8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            a[i][e] = (_Float16)(seed * (float)(lane + i));
            b[i][e] = (_Float16)(seed * (float)(lane - i));
        }
    const int per_store = 96 * kt / 64;  // MFMAs between two store instructions when interleaved
    for (;;) {
        if (threadIdx.x == 0) s_slot = atomicAdd(counter, 1u);
        __syncthreads();
        const long slot = s_slot / PARTS;
        const int wave = wave0 + (int)(s_slot % PARTS) * (WG / 64), wm = wave >> 2, wn = wave & 3;
        __syncthreads();
        if (slot >= n_slots) break;
        const long tm = slot / tiles_n, tn = slot % tiles_n;
        if (tn < tm) continue;
        const bool whole = tm * 256 + 256 <= n && tn * 256 + 256 <= n;
        float* d_base = out + (tm * 256 + wm * 128 + (lane >> 4)) * n + tn * 256 + wn * 64 + (lane & 15) * 4;
        float* m_base = out + (tn * 256 + wn * 64 + (lane >> 5)) * n + tm * 256 + wm * 128 + (lane & 31) * 4;
        auto store = [&](int i) {  // 64 instructions: 32 direct (4 rows x 256 B), 32 mirror (2 rows x 512 B)
            if (!whole) return;
            const f4 v = acc[i & 15];
            if (i < 32) __builtin_nontemporal_store(v, (f4*)(d_base + (long)(i * 4) * n));
            else if (tm != tn) __builtin_nontemporal_store(v, (f4*)(m_base + (long)((i - 32) * 2) * n));
        };
        auto mfmas = [&](int count) {
            for (int j = 0; j < count; j += 16) {
#pragma unroll
                for (int u = 0; u < 16; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 3], b[u >> 2], acc[u], 0, 0, 0);
            }
        };
        if (MODE == 0) {
            mfmas(96 * kt);
            for (int i = 0; i < 64; i++) store(i);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 64; i++) {
                mfmas(per_store);
                store(i);
            }
        } else if (MODE == 2) {
            for (int i = 0; i < 64; i++) store(i);
        } else {
            mfmas(96 * kt);
        }
    }
    if (seed == 123.f) {
        f4 t = acc[0];
#pragma unroll
        for (int u = 1; u < 16; u++) t += acc[u];
        out[threadIdx.x] = t[0] + t[1] + t[2] + t[3];
    }
}

int main() {
    const long n = 50000;
    const int tiles_n = (int)((n + 255) / 256);
    float* out; CK(hipMalloc(&out, (size_t)n * n * 4));
    unsigned* counter; CK(hipMalloc(&counter, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long all = (long)tiles_n * tiles_n;
    auto run = [&](const char* name, auto kern, int kt, int wg = 512) {
        std::vector<float> ms;
        for (int it = 0; it < 9; it++) {
            (void)hipMemsetAsync(counter, 0, 4);
            (void)hipEventRecord(e0);
            kern<<<256 * (512 / wg), wg>>>(out, n, tiles_n, counter, all, kt, 1e-3f);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float t; (void)hipEventElapsedTime(&t, e0, e1);
            if (it >= 2) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("kt %3d  %-28s median %8.4f ms\n", kt, name, ms[ms.size() / 2]);
        return 0;
    };
    for (int kt : {8, 16, 32, 128}) {
        run("mfma only", tiles<3>, kt);
        run("stores only", tiles<2>, kt);
        run("mfma THEN stores (today)", tiles<0>, kt);
        run("stores between mfmas", tiles<1>, kt);
        run("2 WGs of 4 waves: then", tiles<0, 256>, kt, 256);
        run("2 WGs of 4 waves: between", tiles<1, 256>, kt, 256);
        run("4 WGs of 2 waves: then", tiles<0, 128>, kt, 128);
    }
    return 0;
}

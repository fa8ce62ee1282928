// What the memory system delivers for the column-sum kernel's read pattern alone (no LDS, no walker): every workgroup
// streams a strip of `piece` bytes per row through `depth` loads in flight per thread.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/strip_read.hip -o tools/micro/strip_read
//   ./strip_read [rows] [cols] -> table over piece bytes {64,128,256,512}, XCD grouping G, loads in flight
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(256) void strip_read(const float* __restrict__ x, long rows, long cols, int piece_floats, int G,
                                                  float* __restrict__ out) {
    long strip = blockIdx.x;
    const long full = ((long)gridDim.x / (8 * G)) * (8 * G);
    if (strip < full) {
        const long xcd = strip % 8, slot = strip / 8;
        strip = ((slot / G) * 8 + xcd) * G + (slot % G);
    }
    const int lanes_per_row = piece_floats / 4;
    const int rows_per_pass = 256 / lanes_per_row;
    const long col = strip * piece_floats + (threadIdx.x % lanes_per_row) * 4;
    const long r0 = threadIdx.x / lanes_per_row;
    float4 acc = make_float4(0, 0, 0, 0);
    for (long r = r0; r < rows; r += (long)rows_per_pass * DEPTH) {
        float4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            long rr = r + (long)d * rows_per_pass;
            rr = rr < rows ? rr : rows - 1;
            v[d] = *reinterpret_cast<const float4*>(x + rr * cols + col);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            acc.x += v[d].x; acc.y += v[d].y; acc.z += v[d].z; acc.w += v[d].w;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}

int main(int argc, char** argv) {
    const long rows = argc > 1 ? atol(argv[1]) : 50000, cols = argc > 2 ? atol(argv[2]) : 4096;
    float *x, *out;
    CK(hipMalloc((void**)&x, rows * cols * 4));
    CK(hipMalloc((void**)&out, 1 << 20));
    CK(hipMemset(x, 0, rows * cols * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int pieces[] = {64, 128, 256, 512};
    const int groups[] = {1, 2, 4, 8, 16};
    printf("%ld x %ld floats (%.2f GB)\n", rows, cols, rows * cols * 4 / 1e9);
    for (int pb : pieces) {
        const int pf = pb / 4;
        const unsigned grid = (unsigned)(cols / pf);
        for (int G : groups) {
            for (int depth : {8, 16, 32}) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; rep++) {
                    CK(hipEventRecord(a));
                    if (depth == 8) hipLaunchKernelGGL(strip_read<8>, dim3(grid), dim3(256), 0, 0, x, rows, cols, pf, G, out);
                    else if (depth == 16) hipLaunchKernelGGL(strip_read<16>, dim3(grid), dim3(256), 0, 0, x, rows, cols, pf, G, out);
                    else hipLaunchKernelGGL(strip_read<32>, dim3(grid), dim3(256), 0, 0, x, rows, cols, pf, G, out);
                    CK(hipEventRecord(b));
                    CK(hipEventSynchronize(b));
                    float ms;
                    CK(hipEventElapsedTime(&ms, a, b));
                    if (rep > 0 && ms < best) best = ms;
                }
                printf("piece %3d B  G %2d  depth %2d  grid %4u : %.3f ms = %.2f TB/s\n", pb, G, depth, grid, best, rows * cols * 4 / best / 1e9);
            }
        }
    }
    return 0;
}

// Dependent v_add_f32 chains: cycles per add for 1, 2 and 4 interleaved chains in one wave (s_memtime).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/add_chain tools/micro/add_chain.hip && /tmp/add_chain
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CH>
__global__ void chain(const float* x, float* out, unsigned long long* cyc, int n) {
    float a[CH];
    float v[8];
    for (int j = 0; j < 8; j++) v[j] = x[(threadIdx.x + j) & 63];
    // 64 adds per chain and loop trip: the loop's own scalar instructions weigh < 3 %
    for (int c = 0; c < CH; c++) a[c] = x[c];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 64; j++)
#pragma unroll
            for (int c = 0; c < CH; c++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(v[j & 7]));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int c = 0; c < CH; c++) s += a[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float *x, *out; unsigned long long* cyc;
    hipMalloc(&x, 256); hipMalloc(&out, 256); hipMalloc(&cyc, 8);
    hipMemset(x, 0, 256);
    const int n = 20000;
    unsigned long long h;
#define RUN(CH) for (int rep = 0; rep < 2; rep++) { chain<CH><<<1, 64>>>(x, out, cyc, n); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); } \
    printf("chains %d: %.2f shader cycles per add-round (%d adds), %.2f per add\n", CH, (double)h / (n * 64.0), CH, (double)h / (n * 64.0 * CH));
    RUN(1) RUN(2) RUN(4)
    // wall-clock version: many launches timed with events
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
#define WALL(CH) hipEventRecord(e0); chain<CH><<<1, 64>>>(x, out, cyc, 200000); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); \
    printf("chains %d: %.3f ms for %d add-rounds -> %.2f ns per round\n", CH, ms, 200000 * 64, ms * 1e6 / (200000 * 64.0));
    WALL(1) WALL(2) WALL(4)
    return 0;
}

// The column-sum walker alone (normalize.hip: colsum_seq_kernel, wave 0): 16 lanes go down a 512-row LDS tile, one
// ds_read_b128 per 4 rows and 4 dependent v_add_f32, software-pipelined 8 reads ahead.  Cycles per tile without any
// staging wave or barrier: what is left of a pass's time above it is synchronisation with the stagers.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/walker tools/micro/walker.hip && /tmp/walker
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kTileRows = 512;

template <bool BARRIER>
__global__ __launch_bounds__(320) void walk(float* out, unsigned long long* cyc, int n_tiles) {
    __shared__ __attribute__((aligned(16))) float tile[2][16][kTileRows + 4];
    for (int i = threadIdx.x; i < 2 * 16 * (kTileRows + 4); i += blockDim.x) (&tile[0][0][0])[i] = (float)(i & 7) * 0.25f;
    __syncthreads();
    if (threadIdx.x >= 64) {
        if (BARRIER)
            for (int t = 0; t < n_tiles; t++) __syncthreads();
        return;
    }
    const int wl = threadIdx.x;
    float running = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < n_tiles; t++) {
        if (BARRIER) __syncthreads();
        if (wl < 16) {
            const float* colp = &tile[t & 1][wl][0];
            constexpr int kBatch = 8, kBatches = kTileRows / (4 * kBatch);
            float4 q[2][kBatch];
#pragma unroll
            for (int i = 0; i < kBatch; i++) q[0][i] = *reinterpret_cast<const float4*>(colp + 4 * i);
#pragma unroll
            for (int b = 0; b < kBatches; b++) {
                if (b + 1 < kBatches) {
#pragma unroll
                    for (int i = 0; i < kBatch; i++) q[(b + 1) & 1][i] = *reinterpret_cast<const float4*>(colp + (b + 1) * 4 * kBatch + 4 * i);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < kBatch; i++) {
                    running = __fadd_rn(running, q[b & 1][i].x);
                    running = __fadd_rn(running, q[b & 1][i].y);
                    running = __fadd_rn(running, q[b & 1][i].z);
                    running = __fadd_rn(running, q[b & 1][i].w);
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (wl < 16) out[wl] = running;
    if (wl == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc; unsigned long long h;
    (void)hipMalloc(&out, 256); (void)hipMalloc(&cyc, 8);
    const int n = 2000;
    for (int rep = 0; rep < 2; rep++) { walk<false><<<1, 320>>>(out, cyc, n); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); }
    printf("walker alone:            %.0f cycles per 512-row tile = %.2f per row\n", (double)h / n, (double)h / n / kTileRows);
    for (int rep = 0; rep < 2; rep++) { walk<true><<<1, 320>>>(out, cyc, n); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); }
    printf("walker + idle barriers:  %.0f cycles per 512-row tile = %.2f per row\n", (double)h / n, (double)h / n / kTileRows);
    return 0;
}

// Matrix-core rate under load: dense loops of v_mfma_f32_16x16x32_f16, v_mfma_i32_16x16x64_i8 and the 2:2 mix
// a split contraction with int8 cross terms would issue, operands in registers, random data, 2 waves per SIMD.
// Reports time and instruction rate per variant (the clock the chip grants differs per instruction mix).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void rate_kernel(const uint32_t* __restrict__ seed, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s[8];
    for (int i = 0; i < 8; i++) s[i] = seed[(t * 8 + i) & 0xFFFF];
    h8 ah, bh;
    for (int i = 0; i < 8; i++) {
        ah[i] = (_Float16)((float)(s[i] & 0xFFFF) / 65536.f - 0.5f);
        bh[i] = (_Float16)((float)(s[i] >> 16) / 65536.f - 0.5f);
    }
    i4v ai = {(int)s[0], (int)s[1], (int)s[2], (int)s[3]}, bi = {(int)s[4], (int)s[5], (int)s[6], (int)s[7]};
    f4 accf[8];
    i4 acci[8];
    for (int j = 0; j < 8; j++) {
        accf[j] = f4{0, 0, 0, 0};
        acci[j] = i4{0, 0, 0, 0};
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 0) {  // 2 x f16 per slot
                accf[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, accf[j], 0, 0, 0);
                accf[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, accf[j], 0, 0, 0);
            } else if (MODE == 1) {  // 2 x i8
                acci[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ai, bi, acci[j], 0, 0, 0);
                acci[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bi, ai, acci[j], 0, 0, 0);
            } else {  // 1 f16 + 1 i8
                accf[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, accf[j], 0, 0, 0);
                acci[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ai, bi, acci[j], 0, 0, 0);
            }
        }
    }
    float r = 0;
    for (int j = 0; j < 8; j++) r += accf[j][0] + accf[j][3] + (float)acci[j][1];
    out[t] = r;
}

int main() {
    const int iters = 20000, blocks = 256 * 1, threads = 512;  // 8 waves per CU = 2 per SIMD
    std::vector<uint32_t> h(65536);
    srand(1);
    for (auto& v : h) v = (uint32_t)rand() * 2654435761u;
    uint32_t* d_seed;
    float* d_out;
    hipMalloc(&d_seed, h.size() * 4);
    hipMalloc(&d_out, (size_t)blocks * threads * 4);
    hipMemcpy(d_seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[3] = {"f16 16x16x32 only", "i8 16x16x64 only", "1 f16 + 1 i8 alternating"};
    for (int rep = 0; rep < 2; rep++)
        for (int mode = 0; mode < 3; mode++) {
            for (int w = 0; w < 3; w++) {  // a second of back-to-back launches before the timed ones
                if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(threads), 0, 0, d_seed, d_out, iters);
                if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(threads), 0, 0, d_seed, d_out, iters);
                if (mode == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(blocks), dim3(threads), 0, 0, d_seed, d_out, iters);
            }
            hipEventRecord(e0);
            for (int w = 0; w < 5; w++) {
                if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(threads), 0, 0, d_seed, d_out, iters);
                if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(threads), 0, 0, d_seed, d_out, iters);
                if (mode == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(blocks), dim3(threads), 0, 0, d_seed, d_out, iters);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 5;
            const double instr = (double)blocks * (threads / 64) * iters * 16.0;  // wave-level MFMA instructions
            printf("%-28s %8.3f ms  %.3f G MFMA instr/s  (f16-equivalent %.0f TFLOP/s if all were 16x16x32 f16)\n", names[mode], ms,
                   instr / ms / 1e6, instr * 16 * 16 * 32 * 2 / (ms * 1e-3) / 1e12);
        }
    return 0;
}

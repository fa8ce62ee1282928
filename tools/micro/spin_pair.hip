// Can a kernel that waits for a word in uncached device memory be released by a kernel launched LATER on another stream
// of the same process?  (The peer-mailbox chain needs this between GPUs; tools/chain_bench.py --concurrent needs it on one.)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/spin_pair.hip -o /tmp/spin_pair && /tmp/spin_pair [wgs] [threads] [lds_kib]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void waiter(const unsigned* flag, unsigned* out, long max_polls) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) {
        long i = 0;
        for (; i < max_polls; i++) {
            if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 7u) break;
            __builtin_amdgcn_s_sleep(4);
        }
        if (blockIdx.x == 0) out[0] = i < max_polls ? 1u : 0u, out[1] = (unsigned)i;
        lds[0] = 1;
    }
    __syncthreads();
}
__global__ void setter(unsigned* flag) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) lds[0] = 1;
}
int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256, threads = argc > 2 ? atoi(argv[2]) : 320;
    const size_t lds = (size_t)(argc > 3 ? atoi(argv[3]) : 66) * 1024;
    unsigned *flag, *out;
    CK(hipExtMallocWithFlags((void**)&flag, 4096, hipDeviceMallocUncached));
    CK(hipMalloc((void**)&out, 64));
    CK(hipMemset(flag, 0, 4096));
    CK(hipMemset(out, 0, 64));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void*)waiter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)setter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(waiter, dim3(wgs), dim3(threads), lds, a, flag, out, 1L << 22);
    hipLaunchKernelGGL(setter, dim3(wgs), dim3(threads), lds, b, flag);
    CK(hipDeviceSynchronize());
    unsigned h[2];
    CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
    printf("waiter %s after %u polls (wgs %d, threads %d, lds %zu KiB)\n", h[0] ? "released" : "TIMED OUT", h[1], wgs, threads, lds / 1024);
    return 0;
}

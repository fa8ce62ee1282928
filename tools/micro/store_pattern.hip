// How fast can 50 000 rows of 16 KiB be WRITTEN, by access pattern?  (count.hip's flush is a pure row write.)
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// one workgroup of T threads per row, non-persistent; NT: nontemporal
template <int T, bool NT, bool SPLIT>
__global__ __launch_bounds__(T) void per_row(float* out, int row_floats) {
    float* row = out + (size_t)blockIdx.x * row_floats;
    const int half = row_floats / 2;
    for (int i = threadIdx.x * 4; i < (SPLIT ? half : row_floats); i += T * 4) {
        f4 v{(float)i, 1.f, 2.f, 3.f};
        if (NT) { __builtin_nontemporal_store(v, (f4*)(row + i)); if (SPLIT) __builtin_nontemporal_store(v, (f4*)(row + half + i)); }
        else { *(f4*)(row + i) = v; if (SPLIT) *(f4*)(row + half + i) = v; }
    }
}
// persistent: grid workgroups of T threads, rows strided by the grid
template <int T, bool NT, bool SPLIT>
__global__ __launch_bounds__(T) void persistent(float* out, int row_floats, int n_rows, int delay) {
    const int half = row_floats / 2;
    for (int r = blockIdx.x; r < n_rows; r += gridDim.x) {
        float* row = out + (size_t)r * row_floats;
        for (int i = threadIdx.x * 4; i < (SPLIT ? half : row_floats); i += T * 4) {
            f4 v{(float)i, 1.f, 2.f, 3.f};
            if (NT) { __builtin_nontemporal_store(v, (f4*)(row + i)); if (SPLIT) __builtin_nontemporal_store(v, (f4*)(row + half + i)); }
            else { *(f4*)(row + i) = v; if (SPLIT) *(f4*)(row + half + i) = v; }
            for (int d = 0; d < delay; d++) __builtin_amdgcn_s_sleep(1);
        }
    }
}
// persistent, rows handed out by an atomic counter (dynamic, in order)
template <int T, bool NT, bool SPLIT>
__global__ __launch_bounds__(T) void persistent_dyn(float* out, int row_floats, int n_rows, unsigned* counter, int chunk) {
    const int half = row_floats / 2;
    __shared__ int s_r;
    for (;;) {
        if (threadIdx.x == 0) s_r = (int)atomicAdd(counter, (unsigned)chunk);
        __syncthreads();
        const int r0 = s_r;
        __syncthreads();
        if (r0 >= n_rows) return;
        for (int r = r0; r < r0 + chunk && r < n_rows; r++) {
            float* row = out + (size_t)r * row_floats;
            for (int i = threadIdx.x * 4; i < (SPLIT ? half : row_floats); i += T * 4) {
                f4 v{(float)i, 1.f, 2.f, 3.f};
                if (NT) { __builtin_nontemporal_store(v, (f4*)(row + i)); if (SPLIT) __builtin_nontemporal_store(v, (f4*)(row + half + i)); }
                else { *(f4*)(row + i) = v; if (SPLIT) *(f4*)(row + half + i) = v; }
            }
        }
    }
}
int main() {
    const int n_rows = 50000, row_floats = 4096;
    float* out; CK(hipMalloc(&out, (size_t)n_rows * row_floats * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double gb = (double)n_rows * row_floats * 4 / 1e9;
    auto report = [&](const char* name, std::vector<float>& ms) {
        std::sort(ms.begin(), ms.end());
        printf("%-44s median %.4f ms  min %.4f ms  %.0f GB/s\n", name, ms[ms.size() / 2], ms[0], gb / ms[ms.size() / 2] * 1e3);
    };
#define RUN(name, launch) { std::vector<float> ms; for (int it = 0; it < 12; it++) { CK(hipEventRecord(e0)); launch; CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); if (it >= 2) ms.push_back(t); } report(name, ms); }
    RUN("memset", CK(hipMemsetAsync(out, 0, (size_t)n_rows * row_floats * 4)));
    RUN("per_row<256> plain", (per_row<256, false, false><<<n_rows, 256>>>(out, row_floats)));
    RUN("per_row<256> nt", (per_row<256, true, false><<<n_rows, 256>>>(out, row_floats)));
    RUN("per_row<256> nt split", (per_row<256, true, true><<<n_rows, 256>>>(out, row_floats)));
    RUN("per_row<64> nt", (per_row<64, true, false><<<n_rows, 64>>>(out, row_floats)));
    RUN("per_row<64> nt split", (per_row<64, true, true><<<n_rows, 64>>>(out, row_floats)));
    RUN("per_row<512> nt", (per_row<512, true, false><<<n_rows, 512>>>(out, row_floats)));
    RUN("per_row<1024> nt", (per_row<1024, true, false><<<n_rows, 1024>>>(out, row_floats)));
    for (int per_cu : {1, 2, 4, 8}) {
        char nm[96];
        snprintf(nm, 96, "persistent<256> nt, %d WG/CU", per_cu);
        RUN(nm, (persistent<256, true, false><<<256 * per_cu, 256>>>(out, row_floats, n_rows, 0)));
        snprintf(nm, 96, "persistent<256> nt split, %d WG/CU", per_cu);
        RUN(nm, (persistent<256, true, true><<<256 * per_cu, 256>>>(out, row_floats, n_rows, 0)));
    }
    unsigned* counter; CK(hipMalloc(&counter, 4));
    for (int per_cu : {4, 8, 16, 19, 32}) {
        char nm[96];
        snprintf(nm, 96, "persistent<64> nt split, %d WG/CU", per_cu);
        RUN(nm, (persistent<64, true, true><<<256 * per_cu, 64>>>(out, row_floats, n_rows, 0)));
        for (int chunk : {1, 4}) {
            snprintf(nm, 96, "persistent_dyn<64> nt split, %d WG/CU, chunk %d", per_cu, chunk);
            RUN(nm, ((void)hipMemsetAsync(counter, 0, 4), persistent_dyn<64, true, true><<<256 * per_cu, 64>>>(out, row_floats, n_rows, counter, chunk)));
        }
    }
    for (int per_cu : {1, 2, 4, 8}) {
        char nm[96];
        for (int chunk : {1, 4}) {
            snprintf(nm, 96, "persistent_dyn<256> nt split, %d WG/CU, chunk %d", per_cu, chunk);
            RUN(nm, ((void)hipMemsetAsync(counter, 0, 4), persistent_dyn<256, true, true><<<256 * per_cu, 256>>>(out, row_floats, n_rows, counter, chunk)));
        }
        snprintf(nm, 96, "persistent_dyn<512> nt split, %d WG/CU, chunk 8", per_cu);
        if (per_cu <= 4) RUN(nm, ((void)hipMemsetAsync(counter, 0, 4), persistent_dyn<512, true, true><<<256 * per_cu, 512>>>(out, row_floats, n_rows, counter, 8)));
    }
    for (int per_cu : {1, 2, 4}) {
        char nm[96];
        snprintf(nm, 96, "persistent<512> nt split, %d WG/CU", per_cu);
        RUN(nm, (persistent<512, true, true><<<256 * per_cu, 512>>>(out, row_floats, n_rows, 0)));
        snprintf(nm, 96, "persistent<512> plain, %d WG/CU", per_cu);
        RUN(nm, (persistent<512, false, false><<<256 * per_cu, 512>>>(out, row_floats, n_rows, 0)));
    }
    return 0;
}

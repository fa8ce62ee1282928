// What v_cvt_scalef32_pk_fp8_f16 does with its scale (divide or multiply), and its rounding: prints fp8 bytes for a few f16
// inputs next to v_cvt_pk_fp8_f32 of the same value / 128.   hipcc --offload-arch=gfx950 -O2 -o cvt_fp8 cvt_fp8.hip && ./cvt_fp8
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, int n, unsigned* a, unsigned* b) {
    const int i = threadIdx.x;
    if (i >= n) return;
    h2 v = {(_Float16)in[i], (_Float16)in[i]};
    s2 p = {0, 0};
    p = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p, v, 128.0f, false);
    a[i] = (unsigned short)p[0] & 0xFF;
    int q = 0;
    q = __builtin_amdgcn_cvt_pk_fp8_f32(in[i] * (1.0f / 128.0f), 0.f, q, false);
    b[i] = q & 0xFF;
}
int main() {
    const float h[] = {256.f, 128.f, 1.f, 100.f, 33000.f, -512.f, 3.1f, 0.01f, 1000.f, 20000.f, 7.f, 448.f * 128.f, 57000.f};
    const int n = sizeof(h) / sizeof(h[0]);
    float* d; unsigned *a, *b;
    hipMalloc(&d, sizeof(h)); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, n, a, b);
    unsigned ha[32], hb[32];
    hipMemcpy(ha, a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, b, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%10.3f  scalef32(f16, 128) -> 0x%02x   pk_fp8_f32(x/128) -> 0x%02x %s\n", h[i], ha[i], hb[i], ha[i] == hb[i] ? "" : "DIFFERENT");
    return 0;
}

// What the cross-lane moves of the contraction's lean epilogue do, printed lane by lane (gfx950):
// v_permlane32_swap, v_permlane16_swap, DPP row_ror:8, row_shr:4, row_shl:4, quad_perm.
//   hipcc --offload-arch=gfx950 -O3 -o lane_moves lane_moves.hip && ./lane_moves
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int l = threadIdx.x;
    unsigned a = 100 + l, b = 200 + l;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[0 * 64 + l] = r[0]; out[1 * 64 + l] = r[1];
    auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[2 * 64 + l] = s[0]; out[3 * 64 + l] = s[1];
    out[4 * 64 + l] = __builtin_amdgcn_update_dpp(0, (int)a, 0x128, 0xF, 0xF, true);
    out[5 * 64 + l] = __builtin_amdgcn_update_dpp(0, (int)a, 0x114, 0xF, 0xF, true);
    out[6 * 64 + l] = __builtin_amdgcn_update_dpp(0, (int)a, 0x104, 0xF, 0xF, true);
    out[7 * 64 + l] = __builtin_amdgcn_update_dpp(0, (int)a, 0xB1, 0xF, 0xF, true);
    out[8 * 64 + l] = __builtin_amdgcn_update_dpp(0, (int)a, 0x4E, 0xF, 0xF, true);
}
int main() {
    int* d; hipMalloc(&d, 9 * 64 * 4);
    k<<<1, 64>>>(d);
    int h[9 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[9] = {"permlane32_swap -> first ", "permlane32_swap -> second", "permlane16_swap -> first ", "permlane16_swap -> second",
                            "dpp row_ror:8 (a)", "dpp row_shr:4 (a)", "dpp row_shl:4 (a)", "dpp quad_perm[1,0,3,2]", "dpp quad_perm[2,3,0,1]"};
    for (int i = 0; i < 9; i++) {
        printf("%s:", names[i]);
        for (int l = 0; l < 64; l++) printf(" %d", h[i * 64 + l]);
        printf("\n");
    }
    return 0;
}

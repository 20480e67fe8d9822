// dpp_probe.hip -- one-off probe: do the whole-wave DPP controls (wave_rol:1 = 0x134, wave_shl:1 = 0x130) work on gfx950,
// and what does row_shl:n with row_mask/bank_mask write?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *o) {
    int l = threadIdx.x;
    int v = 100 + l;
    o[l] = __builtin_amdgcn_update_dpp(-1, v, 0x134, 0xf, 0xf, false);          // wave_rol:1
    o[64 + l] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);     // wave_shl:1
    o[128 + l] = __builtin_amdgcn_update_dpp(l, v, 0x108, 0x1, 0x1, false);     // row_shl:8, row 0 / bank 0 only, old = lane id
    o[192 + l] = __builtin_amdgcn_update_dpp(l, v, 0xE4, 0x1, 0x1, false);      // quad_perm identity, row 0 / bank 0 only
    o[256 + l] = __builtin_amdgcn_ds_bpermute(4 * ((l + 3) & 63), v);           // rotate left by 3 through the LDS crossbar
}
int main() {
    int *d, h[320];
    (void)hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[5] = {"wave_rol:1", "wave_shl:1", "row_shl:8 rm1 bm1", "quad id rm1 bm1", "bpermute rol3"};
    for (int t = 0; t < 5; ++t) {
        printf("%-18s:", names[t]);
        for (int l = 0; l < 64; ++l) if (l < 6 || (l >= 14 && l < 18) || l >= 61) printf(" [%d]=%d", l, h[t * 64 + l]);
        printf("\n");
    }
    printf("err %d\n", (int)hipGetLastError());
    return 0;
}

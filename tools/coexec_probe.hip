// coexec_probe.hip -- one-off probe: do independent VALU ops (DPP moves) issue in the shadow of MFMAs on gfx950?
// 12 waves per CU on all CUs; per iteration N_MFMA MFMAs (4x4x1: 2 passes, 16x16x1: 8 passes) interleaved with N_VALU DPP moves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float dpp(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true)); }
template <int KIND, int NV>   // KIND 0: 8x 4x4x1 per iteration, KIND 1: 2x 16x16x1 per iteration (same flops); NV DPP moves per iteration
__global__ __launch_bounds__(768) void k(float *out, int iters) {
    int l = threadIdx.x;
    float a = l * 0.001f, b = l * 0.002f, s = 0;
    float t[8];
    for (int i = 0; i < 8; ++i) t[i] = l + i;
    f32x4 acc4[8];
    f32x16 acc16[2];
    for (int i = 0; i < 8; ++i) acc4[i] = (f32x4){0, 0, 0, 0};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc16[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc4[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc4[i], 4, 0, 0);
                if (i < NV) t[i] = dpp(t[i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc16[i] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc16[i], 2, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) if (i * 4 + j < NV) t[i * 4 + j] = dpp(t[i * 4 + j]);
            }
        }
    }
    for (int i = 0; i < 8; ++i) s += acc4[i][0] + t[i];
    for (int i = 0; i < 2; ++i) s += acc16[i][0] + acc16[i][15];
    out[blockIdx.x * 768 + l] = s;
}
template <int KIND, int NV>
void run(float *o) {
    const int iters = 40000;
    hipLaunchKernelGGL((k<KIND, NV>), 256, 768, 0, 0, o, 10);
    (void)hipDeviceSynchronize();
    auto w0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL((k<KIND, NV>), 256, 768, 0, 0, o, iters);
    (void)hipDeviceSynchronize();
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
    printf("%s + %d DPP per 4096 flop/lane-group: %.2f ns per iteration per SIMD (3 waves)  err %d\n", KIND ? "2x 16x16x1" : "8x 4x4x1  ", NV, us * 1e3 / (3.0 * iters),
           (int)hipGetLastError());
}
int main() {
    float *o;
    (void)hipMalloc(&o, 256 * 768 * 4);
    run<0, 0>(o); run<0, 4>(o); run<0, 8>(o);
    run<1, 0>(o); run<1, 4>(o); run<1, 8>(o);
    return 0;
}

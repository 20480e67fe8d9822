// mfma164_probe.hip -- sustained rate of v_mfma_f32_16x16x4_f32 on gfx950 with the accumulator population of cconv16_kernels.hip
// (8 waves per CU = 2 per SIMD, 50 independent accumulators per wave, random operands, long run): ns per MFMA per SIMD and the
// clock the chip holds while doing it (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512, 2) void ktime(float *out, unsigned long long *clk, int iters, const float *src) {
    const int l = threadIdx.x;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(l * 7 + i * 131) & 4095]; b[i] = src[(l * 13 + i * 257 + 1) & 4095]; }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 512 + l] = s;
    if (l == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}
int main() {
    float *t, *src;
    unsigned long long *clk, h[512];
    (void)hipMalloc(&t, 256 * 512 * 4); (void)hipMalloc(&clk, 4096); (void)hipMalloc(&src, 16384);
    float hs[4096];
    srand(3);
    for (int i = 0; i < 4096; ++i) hs[i] = (float)((rand() % 2000001) - 1000000) * 1e-6f;
    (void)hipMemcpy(src, hs, 16384, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        const int iters = rep == 0 ? 100 : 40000;
        auto w0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(ktime<50>, 256, 512, 0, 0, t, clk, iters, src);
        (void)hipDeviceSynchronize();
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
        (void)hipMemcpy(h, clk, 4096, hipMemcpyDeviceToHost);
        double n = 2.0 * 50 * iters;                        // MFMAs per SIMD (2 waves per SIMD)
        printf("iters %d: %.1f us, %.2f ns per MFMA per SIMD, %.1f TFLOP/s; in-kernel clock %.0f MHz (%.1f cycles per MFMA)\n", iters, us, us * 1e3 / n,
               2048.0 * n * 1024 / us * 1e-6, (double)h[0] / (double)h[1] * 100.0, (double)h[0] / n);
    }
    return 0;
}

// calib_kernels.hip -- two calibration figures for bench.py: what THIS box's matrix pipe and memory deliver right now, measured in the
// bench's own process before the timed region, so that `value / calib` is comparable between the boxes of the pool (they differ by 1-3 %, which is more
// than most kernel changes of a round).  Not part of the codec path; no counterpart in the reference.
#include "common.h"

typedef float calib_f32x4 __attribute__((ext_vector_type(4)));
// 8 waves per CU (two per SIMD, the occupancy of the encode-order kernels), 8 independent accumulator quads per wave, v_mfma_f32_16x16x4_f32 only
__global__ __launch_bounds__(512) void k_calib_mfma(float *out, int iters) {
    const int l = threadIdx.x;
    const float a = l * 0.001f, b = l * 0.002f;
    calib_f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (calib_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[(long)blockIdx.x * 512 + l] = s;
}

// fp32 MFMA rate of the whole chip (16x16x4: 2048 flops per instruction and wave): *tflops.  ~50 ms.
LIC360_API int lic360_calib_mfma_f32(void *stream, double *tflops, int *cus_out) {
    ARG_CHECK(tflops);
    hipStream_t s = (hipStream_t)stream;
    int dev = 0, cus = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus_out) *cus_out = cus;
    float *buf = nullptr;
    HIP_TRY(hipMalloc((void **)&buf, (size_t)cus * 512 * sizeof(float)));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const int iters = 40000;
    hipLaunchKernelGGL(k_calib_mfma, dim3(cus), dim3(512), 0, s, buf, 100);
    LAUNCH_CHECK();
    HIP_TRY(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_calib_mfma, dim3(cus), dim3(512), 0, s, buf, iters);
    LAUNCH_CHECK();
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *tflops = 2048.0 * 8.0 * iters * 8.0 * cus / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(buf);
    return 0;
}

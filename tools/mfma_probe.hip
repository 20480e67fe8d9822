// mfma_probe.hip -- one-off hardware probe (not part of the product): lane layout, A-broadcast (cbsz/abid)
// and fmaf-exactness of v_mfma_f32_4x4x1_16b_f32 on gfx950.  Build: hipcc --offload-arch=gfx950 -o mfma_probe mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *a, const float *b, const float *c, float *d0, float *d1, float *d2) {
    int l = threadIdx.x;
    f32x4 cc = {c[l * 4], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3]};
    f32x4 r0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], cc, 0, 0, 0);
    f32x4 r1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], cc, 4, 0, 0);
    f32x4 r2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], cc, 4, 3, 0);
    for (int i = 0; i < 4; ++i) { d0[l * 4 + i] = r0[i]; d1[l * 4 + i] = r1[i]; d2[l * 4 + i] = r2[i]; }
}
__global__ void ktime(float *out, int iters) {
    int l = threadIdx.x;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = l * 0.001f, b = l * 0.002f;
    long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, 0, 0);
    }
    long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[l] = s;
    if (l == 0) out[64] = (float)(t1 - t0) / (iters * 8.0f);
}
int main() {
    float ha[64], hb[64], hc[256], h0[256], h1[256], h2[256];
    srand(1);
    auto rnd = []() { return (float)((rand() % 2000001) - 1000000) * 1.2345e-6f * (1 + rand() % 7); };
    for (int i = 0; i < 64; ++i) { ha[i] = rnd(); hb[i] = rnd(); }
    for (int i = 0; i < 256; ++i) hc[i] = rnd();
    float *a, *b, *c, *d0, *d1, *d2, *t;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&c, 1024); hipMalloc(&d0, 1024); hipMalloc(&d1, 1024); hipMalloc(&d2, 1024); hipMalloc(&t, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice); hipMemcpy(c, hc, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, a, b, c, d0, d1, d2);
    hipMemcpy(h0, d0, 1024, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, 1024, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, 1024, hipMemcpyDeviceToHost);
    int bad0 = 0, bad1 = 0, bad2 = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            float e0 = fmaf(ha[4 * (l / 4) + r], hb[l], hc[l * 4 + r]);
            float e1 = fmaf(ha[r], hb[l], hc[l * 4 + r]);
            float e2 = fmaf(ha[12 + r], hb[l], hc[l * 4 + r]);
            bad0 += memcmp(&e0, &h0[l * 4 + r], 4) != 0;
            bad1 += memcmp(&e1, &h1[l * 4 + r], 4) != 0;
            bad2 += memcmp(&e2, &h2[l * 4 + r], 4) != 0;
        }
    printf("layout D[l][r]=fma(A[4*(l/4)+r],B[l],C[l][r]): mismatches cbsz0=%d  cbsz4/abid0=%d  cbsz4/abid3=%d (of 256)\n", bad0, bad1, bad2);
    hipLaunchKernelGGL(ktime, 1, 64, 0, 0, t, 100000);
    float ht[65];
    hipMemcpy(ht, t, 260, hipMemcpyDeviceToHost);
    printf("cycles per 4x4x1 mfma (8 independent accumulators, 1 wave): %.2f\n", ht[64]);
    return 0;
}

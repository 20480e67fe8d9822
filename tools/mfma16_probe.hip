// mfma16_probe.hip -- one-off hardware probe: lane layout, A-broadcast (cbsz/abid), fmaf-exactness and throughput of
// v_mfma_f32_16x16x1_4b_f32 vs v_mfma_f32_4x4x1_16b_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <chrono>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CBSZ, int ABID>
__global__ void k(const float *a, const float *b, const float *c, float *d) {
    int l = threadIdx.x;
    f32x16 cc;
    for (int i = 0; i < 16; ++i) cc[i] = c[l * 16 + i];
    f32x16 r = __builtin_amdgcn_mfma_f32_16x16x1f32(a[l], b[l], cc, CBSZ, ABID, 0);
    for (int i = 0; i < 16; ++i) d[l * 16 + i] = r[i];
}
template <int KIND>
__global__ __launch_bounds__(768) void ktime(float *out, int iters) {
    int l = threadIdx.x;
    float a = l * 0.001f, b = l * 0.002f, s = 0;
    if constexpr (KIND == 0) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc[i], 2, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    }
    out[blockIdx.x * 768 + l] = s;
}
template <int CBSZ, int ABID>
int check(const float *ha, const float *hb, const float *hc, float *a, float *b, float *c, float *d) {
    static float h[1024];
    hipLaunchKernelGGL((k<CBSZ, ABID>), 1, 64, 0, 0, a, b, c, d);
    (void)hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int blk = 0; blk < 4; ++blk)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                int lane = j + 16 * (i / 4), reg = 4 * blk + i % 4;
                int ablk = CBSZ ? ABID : blk;
                float e = fmaf(ha[16 * ablk + i], hb[16 * blk + j], hc[lane * 16 + reg]);
                bad += memcmp(&e, &h[lane * 16 + reg], 4) != 0;
            }
    return bad;
}
int main() {
    float ha[64], hb[64], hc[1024];
    srand(1);
    auto rnd = []() { return (float)((rand() % 2000001) - 1000000) * 1.2345e-6f * (1 + rand() % 7); };
    for (int i = 0; i < 64; ++i) { ha[i] = rnd(); hb[i] = rnd(); }
    for (int i = 0; i < 1024; ++i) hc[i] = rnd();
    float *a, *b, *c, *d, *t;
    (void)hipMalloc(&a, 256); (void)hipMalloc(&b, 256); (void)hipMalloc(&c, 4096); (void)hipMalloc(&d, 4096); (void)hipMalloc(&t, 256 * 768 * 4);
    (void)hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb, 256, hipMemcpyHostToDevice); (void)hipMemcpy(c, hc, 4096, hipMemcpyHostToDevice);
    printf("16x16x1 layout D_b[i][j] @ lane j+16*(i/4), reg 4b+i%%4 = fma(A[16b'+i], B[16b+j], C): mismatches of 1024: cbsz0 %d, cbsz2/abid0 %d, abid1 %d, abid2 %d, abid3 %d\n",
           check<0, 0>(ha, hb, hc, a, b, c, d), check<2, 0>(ha, hb, hc, a, b, c, d), check<2, 1>(ha, hb, hc, a, b, c, d), check<2, 2>(ha, hb, hc, a, b, c, d),
           check<2, 3>(ha, hb, hc, a, b, c, d));
    for (int kind = 0; kind < 2; ++kind) {
        const int iters = 20000;
        if (kind == 0) hipLaunchKernelGGL(ktime<0>, 256, 768, 0, 0, t, 10); else hipLaunchKernelGGL(ktime<1>, 256, 768, 0, 0, t, 10);
        (void)hipDeviceSynchronize();
        auto w0 = std::chrono::steady_clock::now();
        if (kind == 0) hipLaunchKernelGGL(ktime<0>, 256, 768, 0, 0, t, iters); else hipLaunchKernelGGL(ktime<1>, 256, 768, 0, 0, t, iters);
        (void)hipDeviceSynchronize();
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
        double n = 3.0 * (kind == 0 ? 8 : 4) * iters;       // MFMAs per SIMD (3 waves per SIMD)
        double flop = (kind == 0 ? 512.0 : 2048.0) * n * 4 * 256;
        printf("%s: %.2f ns per MFMA per SIMD, %.1f TFLOP/s whole chip (12 waves/CU, 256 CUs), err %d\n", kind == 0 ? "4x4x1_16b " : "16x16x1_4b", us * 1e3 / n, flop / us * 1e-6,
               (int)hipGetLastError());
    }
    return 0;
}

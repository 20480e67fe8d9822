// gdn_kernels.hip -- generalised divisive normalisation in one pass (reference lic360_operator/GDN.py:66-100, which runs
// x*x, a 1x1 conv2d, sqrt and a division as four torch kernels):  norm[i,p] = sqrt(beta[i] + sum_j gamma[i,j] * x[j,p]^2),
// y = x / norm (or x * norm for the inverse transform).  C x C x P MACs at 2 flops each against 8 bytes per element moved: 96 flop/B at
// C = 192 -- above the fp32 vector ridge (157 TFLOP/s / 8 TB/s = 20 flop/B), so the pass is bound by the FMAs, not by HBM; with
// v_pk_fma_f32 the vector pipe has the same fp32 peak as the matrix pipe and needs no operand shuffling.
//   * a workgroup owns 64 positions of one image and all C channels: the x tile goes to LDS once ([C][64], 48 KB at C = 192; every
//     load issued before the first LDS write), gamma streams through LDS in slabs of 16 input channels, the next slab's loads in
//     flight during the current slab's MFMAs; wave w owns positions 16 w .. 16 w + 15 and all C / 16 output-channel tiles: one
//     v_mfma_f32_16x16x4_f32 per (tile, four input channels), the B operand squared on the way in; the result overwrites the x tile
//     and leaves row-wise, 16 bytes per lane;
//   * the sum over j runs in ascending j for every output (a fixed order; torch's conv sums in its own, so parity is to 1e-5).
// C in {16, 32, 48, 64, 96, 128, 192} (the transforms use 192 and 96); 60 KB of LDS at C = 192.
#include "common.h"

#define GDN_PT 64                                   // positions per workgroup: 16 per wave
#define GDN_JS 16                                   // input channels per gamma slab
#define GDN_GP 17                                   // pitch of a gamma slab row (16 + 1: the A-operand reads walk rows, the pad spreads the banks)
typedef float gdn_f4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x4_f32: D[16 x 16] += A[16 x 4] B[4 x 16]; lane l holds A[l & 15][l >> 4], B[l >> 4][l & 15], D rows 4 (l >> 4) + v, column
// l & 15.  Rows = output channels, columns = positions, k = four consecutive input channels (summed in ascending order).
template <int CT, bool VEC>                         // CT = C / 16 output-channel tiles per wave; VEC: 16-byte global accesses (P % 4 == 0)
__global__ __launch_bounds__(256) void k_gdn(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
                                             float *__restrict__ out, int C, long P, int inverse) {
    extern __shared__ float lds[];
    float *xs = lds, *gs = lds + (size_t)C * GDN_PT;                        // x tile [C][64] (overwritten by the result) | gamma slab [C][17]
    constexpr int XQ = CT * 16 * (GDN_PT / 4) / 256;                        // quads of the x tile per thread (12 at C = 192)
    constexpr int GQ = CT;                                                   // gamma slab elements per thread (C * 16 / 256)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
    const long n = blockIdx.y, p0 = (long)blockIdx.x * GDN_PT;
    const float *xn = x + n * C * P;
    float *on = out + n * C * P;
    // ---- x tile: every load is issued before the first LDS write (one memory latency for the whole tile); past the end: zeros
    gdn_f4 xr[XQ];
#pragma unroll
    for (int k = 0; k < XQ; ++k) {
        const int e = tid + 256 * k, j = e / (GDN_PT / 4), q = e - j * (GDN_PT / 4);
        const long p = p0 + 4 * q;
        if (VEC && p + 3 < P) xr[k] = *(const gdn_f4 *)(xn + (long)j * P + p);
        else {
#pragma unroll
            for (int t = 0; t < 4; ++t) xr[k][t] = p + t < P ? xn[(long)j * P + p + t] : 0.0f;
        }
    }
    // gamma slab 0 into registers while the x tile lands
    float gr[GQ];
#pragma unroll
    for (int k = 0; k < GQ; ++k) { const int e = tid + 256 * k; gr[k] = gamma[(long)(e / GDN_JS) * C + (e % GDN_JS)]; }
#pragma unroll
    for (int k = 0; k < XQ; ++k) { const int e = tid + 256 * k; *(gdn_f4 *)(xs + (e / (GDN_PT / 4)) * GDN_PT + 4 * (e % (GDN_PT / 4))) = xr[k]; }
    gdn_f4 acc[CT];
#pragma unroll
    for (int a = 0; a < CT; ++a) acc[a] = (gdn_f4){0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < C; j0 += GDN_JS) {
        __syncthreads();                                                     // the previous slab is consumed (first pass: the x tile is written)
#pragma unroll
        for (int k = 0; k < GQ; ++k) { const int e = tid + 256 * k; gs[(e / GDN_JS) * GDN_GP + (e % GDN_JS)] = gr[k]; }
        if (j0 + GDN_JS < C) {                                               // the next slab's loads fly during this slab's MFMAs
#pragma unroll
            for (int k = 0; k < GQ; ++k) { const int e = tid + 256 * k; gr[k] = gamma[(long)(e / GDN_JS) * C + j0 + GDN_JS + (e % GDN_JS)]; }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < GDN_JS / 4; ++ks) {
            const float v = xs[(j0 + 4 * ks + kq) * GDN_PT + 16 * wave + col], b = v * v;
#pragma unroll
            for (int a = 0; a < CT; ++a)
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(gs[(16 * a + col) * GDN_GP + 4 * ks + kq], b, acc[a], 0, 0, 0);
        }
    }
    __syncthreads();                                                         // every wave is done reading x^2 operands
    // ---- y = x / sqrt(norm) written over the x tile (each cell by the lane that owns it), then stored row-wise, 16 bytes per lane
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = 16 * a + 4 * kq + v;
            float *cell = xs + i * GDN_PT + 16 * wave + col;
            const float nrm = sqrtf(acc[a][v] + beta[i]), val = *cell;
            *cell = inverse ? val * nrm : val / nrm;
        }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < XQ; ++k) {
        const int e = tid + 256 * k, j = e / (GDN_PT / 4), q = e - j * (GDN_PT / 4);
        const long p = p0 + 4 * q;
        const gdn_f4 r = *(const gdn_f4 *)(xs + j * GDN_PT + 4 * q);
        if (VEC && p + 3 < P) *(gdn_f4 *)(on + (long)j * P + p) = r;
        else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (p + t < P) on[(long)j * P + p + t] = r[t];
        }
    }
}

// x, out: [n][c][p] (p = h*w, contiguous); gamma [c][c] and beta [c]: the EFFECTIVE parameters (after the bound / square / pedestal
// reparametrisation of GDN.py:80-89, which is a few kB of torch work per call)
LIC360_API int lic360_gdn(void *stream, const float *x, const float *gamma, const float *beta, float *out, int n, int c, long p, int inverse) {
    ARG_CHECK(x && gamma && beta && out && n > 0 && c > 0 && c % 16 == 0 && c <= 192 && p > 0 && n <= 65535);
    const dim3 grid((unsigned)((p + GDN_PT - 1) / GDN_PT), (unsigned)n);
    const size_t lds = ((size_t)c * GDN_PT + (size_t)c * GDN_GP) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = p % 4 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0;
#define GDN_LAUNCH(CT_)                                                                                              \
    do {                                                                                                             \
        if (vec) hipLaunchKernelGGL((k_gdn<CT_, true>), grid, dim3(256), lds, s, x, gamma, beta, out, c, p, inverse); \
        else hipLaunchKernelGGL((k_gdn<CT_, false>), grid, dim3(256), lds, s, x, gamma, beta, out, c, p, inverse);    \
    } while (0)
    switch (c / 16) {
        case 1: GDN_LAUNCH(1); break;   case 2: GDN_LAUNCH(2); break;   case 3: GDN_LAUNCH(3); break;   case 4: GDN_LAUNCH(4); break;
        case 6: GDN_LAUNCH(6); break;   case 8: GDN_LAUNCH(8); break;   case 12: GDN_LAUNCH(12); break;
        default: lic360_set_error("lic360_gdn: channel count %d is not one of 16, 32, 48, 64, 96, 128, 192", c); return 2;
    }
#undef GDN_LAUNCH
    LAUNCH_CHECK();
    return 0;
}

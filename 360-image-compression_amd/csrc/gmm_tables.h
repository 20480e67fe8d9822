// gmm_tables.h -- the per-symbol GMM CDF of the latent codec, shared by codec_fused.hip (table kernels) and
// cconv16_kernels.hip (last conv layer with the table build fused into its epilogue).
#pragma once
#include "lic360_exact_math.h"

// 9-entry GMM CDF of one symbol from the three nets' outputs (weights, sigma, mu; 3 components each)
__device__ __forceinline__ void gmm_cdf9(const float *lw_in, const float *ld_in, const float *lm, int *T) {
    float lw[3] = {lw_in[0], lw_in[1], lw_in[2]}, ld[3];
    lic360_softmax_inplace(lw, 3);
#pragma unroll
    for (int i = 0; i < 3; ++i) ld[i] = lic360_sigma_floor(ld_in[i], 1e-6f);
    float t[9];
    t[0] = 0.0f;
    t[8] = 65536.0f;
#pragma unroll
    for (int pt = 1; pt < 8; ++pt) t[pt] = (float)lic360_gmm_cdf_entry(pt, 3.5f, 65536.0f, lw, ld, lm, 3);
    lic360_cdf_fixup(t, 8, 0);
#pragma unroll
    for (int pt = 0; pt < 9; ++pt) T[pt] = (int)t[pt];
}


// N-way softmax CDF + monotonic fix-up (entropy_table_cuda.cu:24-76) for a COMPILE-TIME alphabet: the same operations in the same
// order as lic360_softmax_cdf + lic360_cdf_fixup(.., 1) of lic360_exact_math.h, with every array index static, so that logits,
// exponentials and the table live in registers.  (With a run-time alphabet the per-thread arrays are scratch memory: a table kernel
// of 32 positions took 49 us.)
template <int NSYM>
__device__ __forceinline__ void softmax_table_static(const float (&lg)[NSYM], float total, float (&T)[NSYM + 1]) {
    float tmp[NSYM];
    float m = lg[0];
#pragma unroll
    for (int i = 1; i < NSYM; ++i) if (m < lg[i]) m = lg[i];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NSYM; ++i) { tmp[i] = lic360_expf(lg[i] - m); s += tmp[i]; }
    T[0] = 0.0f;
    const float dp = total / s;
#pragma unroll
    for (int i = 0; i < NSYM - 1; ++i) {
        const float ts = T[i] + (float)(int)((double)(tmp[i] * dp) + 0.5);
        T[i + 1] = ts < total ? ts : total;
    }
    T[NSYM] = total;
    float bias = 0.0f, mval = 0.0f;
    int midx = 0;
#pragma unroll
    for (int i = 0; i < NSYM; ++i) {
        const float nxt = T[i + 1] + bias;
        if (nxt <= T[i]) bias += 1.0f;
        T[i + 1] += bias;
        if (T[i + 1] - T[i] > mval) { mval = T[i + 1] - T[i]; midx = i; }
    }
    if (bias > 0.0f) {
#pragma unroll
        for (int i = 0; i < NSYM; ++i) if (i >= midx) T[i + 1] -= bias;
    }
}

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


/* lic360_exact_math.h -- shared fp32 "exact math" contract for the LIC360 hot path.
 *
 * ONE source, compiled by gcc (CPU oracle, host code) and by hipcc (gfx950 kernels),
 * so that every float the GPU produces is reproduced bit for bit on the CPU.
 *
 * Why it exists: the reference evaluates expf/erff/logf through CUDA libdevice
 * (extension/entropy_gmm_table_cuda.cu:41,150; extension/entropy_gmm_cuda.cu:49-55;
 * extension/quant_cuda.cu:41; extension/dquant_cuda.cu:29) and lets nvcc contract
 * `a + b*c` into FMA.  Neither is reproducible without a CUDA device, and a 1-ulp
 * change flips ~0.4 % of the 16-bit CDF entries, hence the bitstream.  SURVEY.md
 * §8c therefore fixes the contract as:
 *   - every multiply-add the reference writes as `s = s + a*b` is ONE fmaf;
 *   - exp / erf / log are the routines below, built only from IEEE-754 binary32
 *     +, *, fmaf, floorf and integer bit moves (all correctly rounded on both
 *     targets; fp32 subnormals are preserved on gfx950 by hipcc's default mode);
 *   - translation units that include this header are built with -ffp-contract=off.
 *
 * Plain C99 / C++ compatible, no dependencies.
 */
#ifndef LIC360_EXACT_MATH_H
#define LIC360_EXACT_MATH_H

#include "lic360_exact_math_coeffs.h"

#if defined(__HIPCC__) || defined(__HIP__)
#define LIC360_HD __host__ __device__ static __forceinline__
#else
#define LIC360_HD static inline
#endif

LIC360_HD float lic360_bits2f(unsigned int u) {
    float f;
    __builtin_memcpy(&f, &u, sizeof(f));
    return f;
}
LIC360_HD unsigned int lic360_f2bits(float f) {
    unsigned int u;
    __builtin_memcpy(&u, &f, sizeof(u));
    return u;
}

/* exp(x), |rel err| ~ 1e-7.  x >= 88.73 saturates to +inf, x < -104 returns 0. */
LIC360_HD float lic360_expf(float x) {
    const float c[LIC360_EXP_DEG + 1] = LIC360_EXP_COEFFS;
    if (!(x < 88.72283f)) return lic360_bits2f(0x7f800000u); /* also NaN -> inf (never fed) */
    if (x < -104.0f) return 0.0f;
    float t = x * 0x1.715476p+0f;            /* x * log2(e) */
    float kf = __builtin_floorf(t + 0.5f);
    float r = __builtin_fmaf(kf, -0x1.62e400p-1f, x);   /* ln2 hi (exact product for |k|<2^11) */
    r = __builtin_fmaf(kf, -0x1.7f7d1cp-20f, r);        /* ln2 lo */
    float p = c[LIC360_EXP_DEG];
    for (int i = LIC360_EXP_DEG - 1; i >= 0; --i) p = __builtin_fmaf(p, r, c[i]);
    int k = (int)kf;
    int k1 = k / 2;          /* truncating division, same on both targets */
    int k2 = k - k1;
    float s1 = lic360_bits2f((unsigned int)(k1 + 127) << 23);
    float s2 = lic360_bits2f((unsigned int)(k2 + 127) << 23);
    return (p * s1) * s2;
}

/* erf(x), |abs err| < 1e-7 (piecewise degree-10 polynomials on [0,4), 8 pieces). */
LIC360_HD float lic360_erff(float x) {
    const float c[8][LIC360_ERF_DEG + 1] = LIC360_ERF_COEFFS;
    float a = __builtin_fabsf(x);
    float r;
    if (!(a < 4.0f)) {
        r = 1.0f;                               /* erf(4) rounds to 1.0f; NaN never fed */
    } else {
        int i = (int)(a * 2.0f);                /* 0..7 */
        float t = a - (0.5f * (float)i + 0.25f);
        float p = c[i][LIC360_ERF_DEG];
        for (int j = LIC360_ERF_DEG - 1; j >= 0; --j) p = __builtin_fmaf(p, t, c[i][j]);
        r = p;
    }
    return x < 0.0f ? -r : r;
}

/* natural log for x > 0 (normal or subnormal), |rel err| ~ 2e-7; x <= 0 -> -inf. */
LIC360_HD float lic360_logf(float x) {
    if (!(x > 0.0f)) return lic360_bits2f(0xff800000u);
    int e = 0;
    unsigned int u = lic360_f2bits(x);
    if (u < 0x00800000u) {                      /* subnormal: scale by 2^24 */
        x = x * 16777216.0f;
        u = lic360_f2bits(x);
        e = -24;
    }
    e += (int)(u >> 23) - 127;
    unsigned int m = (u & 0x007fffffu) | 0x3f800000u;
    float f = lic360_bits2f(m);                 /* [1,2) */
    if (f > 0x1.6a09e6p+0f) { f = f * 0.5f; e += 1; }   /* -> [sqrt(.5), sqrt(2)) */
    float s = (f - 1.0f) / (f + 1.0f);
    float z = s * s;
    float p = 0x1.745d18p-4f;                   /* 1/11 */
    p = __builtin_fmaf(p, z, 0x1.c71c72p-4f);   /* 1/9 */
    p = __builtin_fmaf(p, z, 0x1.24924ap-3f);   /* 1/7 */
    p = __builtin_fmaf(p, z, 0x1.99999ap-3f);   /* 1/5 */
    p = __builtin_fmaf(p, z, 0x1.555556p-2f);   /* 1/3 */
    p = __builtin_fmaf(p, z, 1.0f);
    float lm = 2.0f * s * p;
    float ef = (float)e;
    return __builtin_fmaf(ef, 0x1.62e400p-1f, __builtin_fmaf(ef, 0x1.7f7d1cp-20f, lm));
}

/* x*scale + bias exactly as Scale / TileInput evaluate it after nvcc's default FMA
 * contraction (extension/scale_cuda.cu:28, extension/tile_input_cuda.cu:38). */
LIC360_HD float lic360_affine(float x, float scale, float bias) {
    return __builtin_fmaf(x, scale, bias);
}

/* ---- CDF-table arithmetic (extension/entropy_gmm_table_cuda.cu:29-107,138-159) ---- */

/* 3..16-way softmax in place, evaluation order of entropy_gmm_table_weight_kernel. */
LIC360_HD void lic360_softmax_inplace(float *w, int n) {
    float m = w[0];
    for (int i = 1; i < n; ++i) if (m < w[i]) m = w[i];
    float s = 0.0f;
    for (int i = 0; i < n; ++i) { w[i] = lic360_expf(w[i] - m); s = s + w[i]; }
    for (int i = 0; i < n; ++i) w[i] = w[i] / s;
}

/* sigma floor of entropy_gmm_table_delta_kernel (:51-57). */
LIC360_HD float lic360_sigma_floor(float d, float beta) {
    return d < 0.0f ? beta : d + beta;
}

/* One interior CDF entry pt (1 <= pt <= nstep-1) for an ng-component mixture
 * (entropy_gmm_table_batch_forward_kernel :148-155).  Returns the integer count. */
LIC360_HD int lic360_gmm_cdf_entry(int pt, float bias, float total,
                                   const float *w, const float *sigma, const float *mu, int ng) {
    float v = (float)((double)((float)(pt - 1) - bias) + 0.5);
    float ps = 0.0f;
    for (int i = 0; i < ng; ++i) {
        float arg = (0x1.6a09e6p-1f * (v - mu[i])) / sigma[i];
        float f = (float)(0.5 + 0.5 * (double)lic360_erff(arg));
        ps = __builtin_fmaf(w[i], f, ps);
    }
    return (int)((double)(total * ps) + 0.5);
}

/* Monotonic fix-up.  variant 0 = entropy_gmm_table_check_kernel (:85-107),
 * variant 1 = entropy_table_forward_kernel (extension/entropy_table_cuda.cu:53-76,
 * which adds the running bias inside the comparison).  T has n+1 float entries. */
LIC360_HD void lic360_cdf_fixup(float *T, int n, int variant) {
    float bias = 0.0f, mval = 0.0f;
    int midx = 0;
    for (int i = 0; i < n; ++i) {
        float nxt = variant ? T[i + 1] + bias : T[i + 1];
        if (nxt <= T[i]) bias += 1.0f;
        T[i + 1] += bias;
        if (T[i + 1] - T[i] > mval) { mval = T[i + 1] - T[i]; midx = i; }
    }
    if (bias > 0.0f)
        for (int i = midx; i < n; ++i) T[i + 1] -= bias;
}

/* 49-way softmax CDF of entropy_table_soft_kernel (extension/entropy_table_cuda.cu:24-50).
 * logits[w] -> T[w+1]; tmp is caller scratch of w floats. */
LIC360_HD void lic360_softmax_cdf(const float *logits, float *T, float *tmp, int w, float total) {
    float m = logits[0];
    for (int i = 1; i < w; ++i) if (m < logits[i]) m = logits[i];
    float s = 0.0f;
    for (int i = 0; i < w; ++i) { tmp[i] = lic360_expf(logits[i] - m); s += tmp[i]; }
    T[0] = 0.0f;
    float dp = total / s;
    for (int i = 0; i < w - 1; ++i) {
        float ts = T[i] + (float)(int)((double)(tmp[i] * dp) + 0.5);
        T[i + 1] = ts < total ? ts : total;
    }
    T[w] = total;
}

/* ---- quantiser arithmetic (extension/quant_cuda.cu:35-77, extension/dquant_cuda.cu:24-47) ---- */

/* nearest-centre search; wq = [c0, e^{w1}, ...] increments.  Returns index, *top = value. */
LIC360_HD int lic360_quant_one(float x, const float *wq, int levels, float *top) {
    float tmp = x - wq[0];
    if (tmp < 0.0f) { *top = wq[0]; return 0; }
    int j = 1;
    for (; j < levels; ++j) {
        tmp -= wq[j];
        if (tmp < 0.0f) break;
    }
    if (j == levels) j--;
    if (tmp + tmp + wq[j] < 0.0f) { tmp = tmp + wq[j]; j--; }
    *top = x - tmp;
    return j;
}

#endif /* LIC360_EXACT_MATH_H */

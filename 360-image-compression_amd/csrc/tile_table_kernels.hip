// tile_table_kernels.hip -- plane gather/scatter/add (A11-A13), CDF-table builders (A14, A15),
// GMM likelihood (A16) and the scan-order tables (A8) for gfx950.
//
// The table builders are the per-symbol, bandwidth-bound part of the entropy path: one lane per
// symbol, softmax + sigma floor + 7 erf-CDF entries + monotonic fix-up fused in registers
// (the reference runs 4 kernels that round-trip the parameters through memory,
// extension/entropy_gmm_table_cuda.cu:161-191).  Arithmetic comes from lic360_exact_math.h.
#include "common.h"
#include "gmm_tables.h"
#include "lic360_exact_math.h"

#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// ---------------------------------------------------------------------------- A8 scan order (host)
LIC360_API int lic360_code_contex(int h, int w, int *idx, int *plane_idx) {
    ARG_CHECK(h > 0 && w > 0 && idx && plane_idx);
    // anti-diagonal order, row index ascending inside a diagonal (extension/code_contex_cuda.cu:19-31)
    int k = 0, stride = h * w;
    for (int s = 0; s < h + w - 1; ++s) {
        plane_idx[s] = k;
        int lo = s >= w ? s - w + 1 : 0, hi = s < h ? s : h - 1;
        for (int r = lo; r <= hi; ++r, ++k) {
            idx[k] = r;
            idx[k + stride] = s - r;
        }
    }
    plane_idx[h + w - 1] = k;
    return 0;
}
LIC360_API int lic360_plane_window(int psum, int ngroup, int h, int w, const int *plane_idx, int *start, int *len) {
    ARG_CHECK(plane_idx && start && len && ngroup > 0);
    if (psum < 0 || psum >= h + w + ngroup - 2) { *start = 0; *len = 0; return 0; }
    int la = psum >= ngroup ? psum - ngroup + 1 : 0;
    int lb = psum > h + w - 2 ? h + w - 2 : psum;
    *start = plane_idx[la];
    *len = plane_idx[lb + 1] - plane_idx[la];
    if (*len < 0) *len = 0;
    return 0;
}

// ---------------------------------------------------------------------------- A11-A13
// mode 0: out[i] (compact [N][len][cpn]) = in[...]; mode 1 (batch): slab placement of forward_batch_cuda
__global__ void k_tile_extract(const float *__restrict__ in, float *__restrict__ out, long num, const int *__restrict__ idx,
                               int start, int len, int HW, int H, int W, int C, int cpn, int psum, int batch, long slab_stride, long slab_inner) {
    GRID_STRIDE(i, num) {
        int ci = (int)(i % cpn), tl = (int)((i / cpn) % len);
        long tn = i / cpn / len;
        int th = idx[tl + start], tw = idx[tl + start + HW];
        int tc = psum - tw - th;
        float v = in[((tn * C + tc * cpn + ci) * H + th) * W + tw];
        if (batch) out[(i / slab_inner) * slab_stride + i % slab_inner] = v;
        else out[i] = v;
    }
}
__global__ void k_tile_input(const float *__restrict__ sym, float *__restrict__ out, long count, const int *__restrict__ idx,
                             int start, int len, int HW, int H, int W, int G, int psum, float bias, float scale, int rep, long stride_out) {
    GRID_STRIDE(i, count) {
        int tl = (int)(i % len);
        long tn = i / len;
        int th = idx[tl + start], tw = idx[tl + start + HW];
        int tc = psum - tw - th;
        long p = ((tn * G + tc) * H + th) * W + tw;
        float v = lic360_affine(sym[i], scale, bias);
        for (int j = 0; j < rep; ++j) out[p + j * stride_out] = v;
    }
}
__global__ void k_tile_add(float *__restrict__ y, const float *__restrict__ x, long count, const int *__restrict__ idx,
                           int start, int len, int HW, int H, int W, int C, int cpg, int psum, int N) {
    // plane position fastest (the reference walks the SAMPLE fastest, tile_add_cuda.cu:24-27: neighbouring threads a whole tensor apart -- every access
    // its own sector of another page: 45.9 us per plane at 32 images; an elementwise update, so the order is free)
    GRID_STRIDE(i, count) {
        int pb = (int)(i % len);
        long pp = i / len;
        int og = (int)(pp % cpg), pn = (int)(pp / cpg);
        int th = idx[pb + start], tw = idx[pb + start + HW];
        int tc = psum - th - tw;
        long o = (((long)pn * C + tc * cpg + og) * H + th) * W + tw;
        y[o] = y[o] + x[o];
    }
}

LIC360_API int lic360_tile_extract(void *stream, const float *x, float *out, int n, int c, int h, int w, int ngroup,
                                   const int *idx_dev, int start, int len, int psum) {
    ARG_CHECK(x && out && idx_dev && ngroup > 0 && c % ngroup == 0 && len >= 0);
    int cpn = c / ngroup;
    long num = (long)n * cpn * len;
    if (num == 0) return 0;
    hipLaunchKernelGGL(k_tile_extract, dim3(lic360_blocks(num)), dim3(256), 0, (hipStream_t)stream, x, out, num, idx_dev, start, len,
                       h * w, h, w, c, cpn, psum, 0, 0L, 1L);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_tile_extract_batch(void *stream, const float *x, float *out, int n, int c, int h, int w, int ngroup,
                                         const int *idx_dev, int start, int len, int psum) {
    ARG_CHECK(x && out && idx_dev && ngroup > 0 && c % ngroup == 0 && n % 3 == 0 && len >= 0);
    int cpn = c / ngroup, nout = n / 3;
    long num = (long)n * cpn * len;
    if (num == 0) return 0;
    hipLaunchKernelGGL(k_tile_extract, dim3(lic360_blocks(num)), dim3(256), 0, (hipStream_t)stream, x, out, num, idx_dev, start, len,
                       h * w, h, w, c, cpn, psum, 1, (long)cpn * h * w * nout, (long)len * cpn * nout);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_tile_input(void *stream, const float *sym, float *out, int n, int g, int h, int w, float bias, float scale, int rep,
                                 const int *idx_dev, int start, int len, int psum) {
    ARG_CHECK(sym && out && idx_dev && rep > 0 && len >= 0);
    long count = (long)n * len;
    if (count == 0) return 0;
    hipLaunchKernelGGL(k_tile_input, dim3(lic360_blocks(count)), dim3(256), 0, (hipStream_t)stream, sym, out, count, idx_dev, start, len,
                       h * w, h, w, g, psum, bias, scale, rep, (long)n * g * h * w);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_tile_add(void *stream, float *y, const float *x, int n, int c, int h, int w, int ngroup,
                               const int *idx_dev, int start, int len, int psum) {
    ARG_CHECK(y && x && idx_dev && ngroup > 0 && c % ngroup == 0 && len >= 0);
    int cpg = c / ngroup;
    long count = (long)n * cpg * len;
    if (count == 0) return 0;
    hipLaunchKernelGGL(k_tile_add, dim3(lic360_blocks(count)), dim3(256), 0, (hipStream_t)stream, y, x, count, idx_dev, start, len,
                       h * w, h, w, c, cpg, psum, n);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------- A14 GMM CDF table
// One lane per symbol.  w/d rewritten in place (softmax / sigma floor) as the reference does.
// CNG / CNSTEP > 0 fix the mixture size and the alphabet at compile time (3 and 8 for the latent model): every array then lives
// in registers; with run-time bounds the dynamically indexed arrays go to scratch memory.
template <int MAXG, int CNG, int CNSTEP>
__global__ void k_gmm_table(float *__restrict__ w, float *__restrict__ d, const float *__restrict__ m, float *__restrict__ out,
                            int tn, int ng_, int nstep_, float bias, float total, float beta) {
    const int ng = CNG ? CNG : ng_, nstep = CNSTEP ? CNSTEP : nstep_;
    GRID_STRIDE(n, tn) {
        float lw[CNG ? CNG : MAXG], ld[CNG ? CNG : MAXG], lm[CNG ? CNG : MAXG];
#pragma unroll
        for (int i = 0; i < ng; ++i) { lw[i] = w[n * ng + i]; ld[i] = d[n * ng + i]; lm[i] = m[n * ng + i]; }
        lic360_softmax_inplace(lw, ng);
#pragma unroll
        for (int i = 0; i < ng; ++i) ld[i] = lic360_sigma_floor(ld[i], beta);
#pragma unroll
        for (int i = 0; i < ng; ++i) { w[n * ng + i] = lw[i]; d[n * ng + i] = ld[i]; }
        float *T = out + n * (nstep + 1);
        // nstep <= 16 rows live in registers for the fix-up
        float t[CNSTEP ? CNSTEP + 1 : 17];
        t[0] = 0.0f;
        t[nstep] = (float)(int)total;
#pragma unroll
        for (int pt = 1; pt < nstep; ++pt) t[pt] = (float)lic360_gmm_cdf_entry(pt, bias, total, lw, ld, lm, ng);
        lic360_cdf_fixup(t, nstep, 0);
#pragma unroll
        for (int pt = 0; pt <= nstep; ++pt) T[pt] = t[pt];
    }
}
LIC360_API int lic360_gmm_table(void *stream, float *w, float *d, const float *m, float *out, int tn, int ng, int nstep,
                                float bias, float total, float beta) {
    ARG_CHECK(w && d && m && out && tn >= 0 && ng >= 1 && ng <= 16 && nstep >= 1 && nstep <= 16);
    if (tn == 0) return 0;
    if (ng == 3 && nstep == 8)
        hipLaunchKernelGGL((k_gmm_table<16, 3, 8>), dim3(lic360_blocks(tn)), dim3(256), 0, (hipStream_t)stream, w, d, m, out, tn, ng, nstep, bias, total, beta);
    else
        hipLaunchKernelGGL((k_gmm_table<16, 0, 0>), dim3(lic360_blocks(tn)), dim3(256), 0, (hipStream_t)stream, w, d, m, out, tn, ng, nstep, bias, total, beta);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------- A15 softmax CDF table (importance-map codec)
__global__ void k_entropy_table(const float *__restrict__ logits, float *__restrict__ out, int count, int nstep, float total) {
    GRID_STRIDE(n, count) {
        float tmp[64], T[65], lg[64];
        for (int i = 0; i < nstep; ++i) lg[i] = logits[n * nstep + i];
        lic360_softmax_cdf(lg, T, tmp, nstep, total);
        lic360_cdf_fixup(T, nstep, 1);
        for (int i = 0; i <= nstep; ++i) out[n * (nstep + 1) + i] = T[i];
    }
}
// the alphabet of the LIC360 importance nets at compile time (registers instead of scratch memory, see gmm_tables.h)
template <int NSTEP>
__global__ void k_entropy_table_static(const float *__restrict__ logits, float *__restrict__ out, int count, float total) {
    GRID_STRIDE(n, count) {
        float lg[NSTEP], T[NSTEP + 1];
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) lg[i] = logits[n * NSTEP + i];
        softmax_table_static<NSTEP>(lg, total, T);
#pragma unroll
        for (int i = 0; i <= NSTEP; ++i) out[n * (NSTEP + 1) + i] = T[i];
    }
}
LIC360_API int lic360_entropy_table(void *stream, const float *logits, float *out, int count, int nstep, float total) {
    ARG_CHECK(logits && out && count >= 0 && nstep >= 1 && nstep <= 64);
    if (count == 0) return 0;
    if (nstep == 49) hipLaunchKernelGGL(k_entropy_table_static<49>, dim3((count + 63) / 64), dim3(64), 0, (hipStream_t)stream, logits, out, count, total);
    else hipLaunchKernelGGL(k_entropy_table, dim3((count + 63) / 64), dim3(64), 0, (hipStream_t)stream, logits, out, count, nstep, total);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------- A16 GMM negative log-likelihood
__global__ void k_entropy_gmm(const float *__restrict__ weight, const float *__restrict__ delta, const float *__restrict__ mean,
                              const float *__restrict__ label, float *__restrict__ loss, float *__restrict__ wd, float *__restrict__ dd,
                              float *__restrict__ md, float *__restrict__ ld, int M, int ng) {
    const float s2 = 0x1.6a09e6p-1f, sp2 = 0x1.988454p-2f;
    GRID_STRIDE(n, M) {
        float sum_p = 0.0f, l = 0.0f;
        for (int i = 0; i < ng; ++i) {
            long k = n * ng + i;
            float xa = (float)((double)label[n] - 0.5 - (double)mean[k]);
            float xb = (float)((double)label[n] + 0.5 - (double)mean[k]);
            float id = (float)(1.0 / (double)delta[k]);
            float fa = (float)(0.5 + 0.5 * (double)lic360_erff(xa * id * s2));
            float fb = (float)(0.5 + 0.5 * (double)lic360_erff(xb * id * s2));
            float p = fb - fa;
            sum_p = __builtin_fmaf(weight[k], p, sum_p);
            float ga = sp2 * id * lic360_expf((float)(-0.5 * (double)xa * (double)xa * (double)id * (double)id));
            float gb = sp2 * id * lic360_expf((float)(-0.5 * (double)xb * (double)xb * (double)id * (double)id));
            l += (gb - ga) * weight[k];
            dd[k] = id * (-xb * gb + xa * ga) * weight[k];
            md[k] = (ga - gb) * weight[k];
            wd[k] = p;
        }
        loss[n] = -lic360_logf((float)((double)sum_p + 0.0000001));
        float ip = (float)(-1.0 / ((double)sum_p + 0.0000001));
        ld[n] = l * ip;
        for (int i = 0; i < ng; ++i) {
            long k = n * ng + i;
            dd[k] *= ip; md[k] *= ip; wd[k] *= ip;
        }
    }
}
__global__ void k_entropy_gmm_bwd(float *__restrict__ wd, float *__restrict__ dd, float *__restrict__ md, float *__restrict__ ld,
                                  const float *__restrict__ top, long total, int ng) {
    GRID_STRIDE(i, total) {       // extension/entropy_gmm_cuda.cu:94-106
        long pn = i / ng;
        if (i % ng == 0) ld[pn] *= top[pn];
        wd[i] *= top[pn]; dd[i] *= top[pn]; md[i] *= top[pn];
    }
}
LIC360_API int lic360_entropy_gmm(void *stream, const float *w, const float *d, const float *m, const float *label, float *loss,
                                  float *wd, float *dd, float *md, float *ld, int count, int ng) {
    ARG_CHECK(w && d && m && label && loss && wd && dd && md && ld && count >= 0 && ng >= 1);
    if (count == 0) return 0;
    hipLaunchKernelGGL(k_entropy_gmm, dim3(lic360_blocks(count)), dim3(256), 0, (hipStream_t)stream, w, d, m, label, loss, wd, dd, md, ld, count, ng);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_entropy_gmm_backward(void *stream, float *wd, float *dd, float *md, float *ld, const float *top_diff, int count, int ng) {
    ARG_CHECK(wd && dd && md && ld && top_diff && count >= 0 && ng >= 1);
    if (count == 0) return 0;
    long total = (long)count * ng;
    hipLaunchKernelGGL(k_entropy_gmm_bwd, dim3(lic360_blocks(total)), dim3(256), 0, (hipStream_t)stream, wd, dd, md, ld, top_diff, total, ng);
    LAUNCH_CHECK();
    return 0;
}

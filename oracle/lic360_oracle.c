/* lic360_oracle.c -- CPU ORACLE for the LIC360 hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a plain-C restatement of the reference's CUDA kernels and C++ coder,
 * function by function, each citing the reference file:line it follows (paths are
 * relative to /root/reference).  It is the checker for the HIP product path: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (360-image-compression_amd/) never links, imports or calls anything here.
 *
 * Pinning status:
 *   - arithmetic coder + bit I/O: PINNED against the reference's own
 *     extension/ArithmeticCoder.cpp + extension/BitIoStream.cpp compiled from where
 *     they lie into oracle/_ref/ (see oracle/Makefile) and against the committed
 *     fixtures tests/golden/ac_*.npz produced by oracle/gen_golden.py.
 *   - index/copy kernels (sphere, dtow, tile, context, imp_map, ...): pinned by
 *     independent numpy/torch-CPU restatements in tests/.
 *   - masked convolution: pinned to F.conv2d with the mask_constrain rule to 1e-4
 *     (summation order differs); the evaluation ORDER follows cconv_ec_cuda.cu
 *     literally with fmaf as the canonical multiply-add.
 *   - exp/erf/log: "parity unpinned" at the last ulp versus CUDA libdevice (no CUDA
 *     device or golden vector exists); the contract is csrc/lic360_exact_math.h,
 *     shared verbatim with the HIP kernels, checked against float64 scipy in tests/.
 *
 * Build: see oracle/Makefile (gcc -O2 -mavx2 -mfma -ffp-contract=off -fopenmp).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "../360-image-compression_amd/csrc/lic360_exact_math.h"
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------
 * A8  code_contex_opt::reshape           extension/code_contex_cuda.cu:11-32
 * idx[k]=ph, idx[k+HW]=pw in anti-diagonal scan order; plane_idx[pn] prefix, H+W entries.
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_code_contex(int H, int W, int *idx, int *plane_idx) {
    int stride = H * W;
    int pidx = 0, pn = 0;
    for (; pn < H + W - 1; pn++) {
        plane_idx[pn] = pidx;
        int ph = pn >= W ? pn - W + 1 : 0;
        for (; ph < H; ph++) {
            int pw = pn - ph;
            if (pw < 0) break;
            idx[pidx] = ph;
            idx[pidx + stride] = pw;
            pidx += 1;
        }
    }
    plane_idx[pn] = pidx;
}

/* plane window [la, lb] used by every plane-stepped op, e.g. cconv_dc_cuda.cu:374-376 */
static void plane_window(int psum, int G, int H, int W, const int *plane_idx, int *start, int *len) {
    int la = psum >= G ? psum - G + 1 : 0;
    int lb = psum > H + W - 2 ? H + W - 2 : psum;
    *start = plane_idx[la];
    *len = plane_idx[lb + 1] - plane_idx[la];
    if (*len < 0) *len = 0;
}
ORC_API int orc_plane_len(int psum, int G, int H, int W, const int *plane_idx) {
    int s, l;
    if (psum < 0 || psum >= H + W + G - 2) return 0;
    plane_window(psum, G, H, W, plane_idx, &s, &l);
    return l;
}

/* thread pool of the OpenMP loops below (bench.py's cpu_baseline leg times the oracle at the GPU's share of the host cores and at all of them) */
ORC_API int orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------
 * A9/A10 one output scalar of the masked convolution.
 * Literal restatement of cconv_ec_act_forward_kernel_batch, extension/cconv_ec_cuda.cu:268-315
 * (identical body in cconv_dc_cuda.cu:313-364): 128 virtual lanes, lane `tid` walks flat
 * indices tid, tid+128, ... < 25*cin, accumulating ONE running sum; then the fixed tree
 * p[i]+p[i+64]; +32; then shfl_down 16,8,4,2,1 inside a 32-wide warp (:299-309).
 * `sum = sum + x*w` is ONE fmaf.
 * ---------------------------------------------------------------------------------- */
static float conv_one(const float *input, const float *weight, int n, int o_global, int o_local,
                      int th, int tw, int C, int H, int W, int ksz, int group_in, int group_out,
                      int constrain) {
    const int blockSize = 128;
    float p[128];
    int inner = H * W, skernel = ksz * ksz, half = ksz / 2;
    int nblock = skernel * group_in;
    int tc = o_local / group_out;
    int psum = th + tw + tc;
    for (int tid = 0; tid < blockSize; ++tid) {
        float sum = 0.0f;
        for (int index = tid; index < nblock; index += blockSize) {
            int kw = index % ksz, kh = (index / ksz) % ksz, gid = index / ksz / ksz;
            int ph = th - half + kh, pw = tw - half + kw;
            if (ph >= H || ph < 0 || pw >= W || pw < 0) continue;
            int nchannel = constrain == 5 ? (psum - ph - pw) * group_in : (psum - ph - pw + 1) * group_in;
            if (nchannel > C) nchannel = C;
            if (nchannel > 0) {
                long wbase = ((long)o_global * C * ksz + kh) * ksz + kw;
                long dbase = ((long)n * C * H + ph) * W + pw;
                for (int ti = gid; ti < nchannel; ti += group_in)
                    sum = __builtin_fmaf(input[dbase + (long)ti * inner], weight[wbase + (long)ti * skernel], sum);
            }
        }
        p[tid] = sum;
    }
    for (int i = 0; i < 64; ++i) p[i] = p[i] + p[i + 64];
    for (int i = 0; i < 32; ++i) p[i] = p[i] + p[i + 32];
    for (int off = 16; off > 0; off >>= 1)
        for (int i = 0; i < off; ++i) p[i] = p[i] + p[i + off];
    return p[0];
}

/* EC: whole tensor.  nb = number of stacked nets (weight.size(0) in the *_batch forwards,
 * 1 otherwise); input [N,C,H,W] with N % nb == 0; weight [nb,nout,C,k,k]; bias/act [nb,nout].
 * act == NULL -> no PReLU.  EC PReLU: sum>0 ? sum : sum*a  (cconv_ec_cuda.cu:311-312). */
ORC_API void orc_cconv_ec(const float *input, const float *weight, const float *bias, const float *act,
                          float *output, int N, int C, int H, int W, int nout, int ngroup, int ksz,
                          int constrain, int nb) {
    int group_in = C / ngroup, group_out = nout / ngroup, npb = N / nb;
    long rows = (long)N * nout, HW = (long)H * W;
    /* one (sample, output channel) plane per iteration, planes dealt round-robin: neighbouring channels (chains of similar length) go to
     * different threads, every thread writes whole planes; no fork for a handful of outputs (`dynamic, 64` over
     * scalars made 256 threads slower than 16) */
#pragma omp parallel for schedule(static, 1) if (rows * HW >= 4096)
    for (long r = 0; r < rows; ++r) {
        int o = (int)(r % nout), n = (int)(r / nout);
        int nbatch = n / npb;
        int bid = nbatch * nout + o;
        for (int pos = 0; pos < H * W; ++pos) {
            float s = conv_one(input, weight, n, bid, o, pos / W, pos % W, C, H, W, ksz, group_in, group_out, constrain);
            s = s + bias[bid];
            if (act) s = s > 0 ? s : s * act[bid];
            output[r * HW + pos] = s;
        }
    }
}

/* DC: one plane `psum` (cconv_dc_cuda.cu:367-398).  Writes only the plane's outputs into the
 * persistent `output` [N,nout,H,W]; zero-fills it at psum==0 (:385).  DC PReLU: if(sum<0)
 * sum*=a (:360-362). */
ORC_API void orc_cconv_dc_plane(const float *input, const float *weight, const float *bias, const float *act,
                                float *output, int N, int C, int H, int W, int nout, int ngroup, int ksz,
                                int constrain, int nb, const int *idx, const int *plane_idx, int psum) {
    int group_in = C / ngroup, group_out = nout / ngroup, npb = N / nb;
    int mod = H + W + ngroup - 2, start, len;
    if (psum >= mod) return;
    plane_window(psum, ngroup, H, W, plane_idx, &start, &len);
    if (len <= 0) return;
    if (psum == 0) memset(output, 0, sizeof(float) * (size_t)N * nout * H * W);
    long total = (long)N * group_out * len;
#pragma omp parallel for schedule(static, 8) if (total >= 512)
    for (long b = 0; b < total; ++b) {
        int a = (int)(b % len), og = (int)((b / len) % group_out), n = (int)(b / ((long)len * group_out));
        int th = idx[a + start], tw = idx[a + start + H * W];
        int tc = psum - th - tw;
        int pout = tc * group_out + og;
        int nbatch = n / npb;
        int bid = nbatch * nout + pout;
        float s = conv_one(input, weight, n, bid, pout, th, tw, C, H, W, ksz, group_in, group_out, constrain);
        s = s + bias[bid];
        if (act) { if (s < 0) s = s * act[bid]; }
        output[(((long)n * nout + pout) * H + th) * W + tw] = s;
    }
}

/* ------------------------------------------------------------------------------------
 * A11 tile_extract                         extension/tile_extract_cuda.cu:31-45,48-98
 * label mode: gather plane psum into out[N][len][cpn]; returns len*N (the CPU count tensor).
 * non-label mode (:79-93): psum==0 zero-fills, else gathers plane psum-1.
 * ---------------------------------------------------------------------------------- */
static void tile_gather(const float *in, float *out, int N, int C, int H, int W, int cpn,
                        const int *idx, int start, int len, int psum) {
    long num = (long)N * cpn * len;
    for (long i = 0; i < num; ++i) {
        int ci = (int)(i % cpn), tl = (int)((i / cpn) % len), tn = (int)(i / cpn / len);
        int th = idx[tl + start], tw = idx[tl + start + H * W];
        int tc = psum - tw - th;
        out[i] = in[(((long)tn * C + tc * cpn + ci) * H + th) * W + tw];
    }
}
ORC_API int orc_tile_extract(const float *in, float *out, int N, int C, int H, int W, int ngroup, int label,
                             const int *idx, const int *plane_idx, int psum) {
    int cpn = C / ngroup, mod = H + W + ngroup - 2, start, len;
    if (label) {
        if (psum >= mod) return 0;
        plane_window(psum, ngroup, H, W, plane_idx, &start, &len);
        if (len > 0) tile_gather(in, out, N, C, H, W, cpn, idx, start, len, psum);
        return N * len;
    }
    if (psum == 0) { memset(out, 0, sizeof(float) * (size_t)N * cpn * H * W); return 0; }
    if (psum <= mod) {
        psum -= 1;
        plane_window(psum, ngroup, H, W, plane_idx, &start, &len);
        if (len > 0) tile_gather(in, out, N, C, H, W, cpn, idx, start, len, psum);
        return N * len;
    }
    return 0;
}
/* batch variant (:101-151): N = 3*nout stacked nets; slab pn placed at pn*cpn*H*W*nout. */
ORC_API int orc_tile_extract_batch(const float *in, float *out, int N, int C, int H, int W, int ngroup,
                                   const int *idx, const int *plane_idx, int psum) {
    int cpn = C / ngroup, mod = H + W + ngroup - 2, start, len, nout = N / 3;
    if (psum >= mod) return 0;
    plane_window(psum, ngroup, H, W, plane_idx, &start, &len);
    long num = (long)N * cpn * len;
    long stride = (long)cpn * H * W * nout, inner = (long)len * cpn * nout;
    for (long i = 0; i < num; ++i) {
        long ps = i % inner, pn = i / inner;
        int ci = (int)(i % cpn), tl = (int)((i / cpn) % len), tn = (int)(i / cpn / len);
        int th = idx[tl + start], tw = idx[tl + start + H * W];
        int tc = psum - tw - th;
        out[pn * stride + ps] = in[(((long)tn * C + tc * cpn + ci) * H + th) * W + tw];
    }
    return nout * len;
}

/* A12 tile_input                            extension/tile_input_cuda.cu:27-76
 * in: compact [N][len] symbols of plane psum-1; out [rep*N, G, H, W]. */
ORC_API void orc_tile_input(const float *in, float *out, int N, int G, int H, int W, float bias, float scale,
                            int rep, const int *idx, const int *plane_idx, int psum) {
    long stride_out = (long)N * G * H * W;
    int mod = H + W + G - 2, start, len;
    if (psum == 0) { memset(out, 0, sizeof(float) * (size_t)rep * stride_out); return; }
    if (psum > mod) return;
    psum -= 1;
    plane_window(psum, G, H, W, plane_idx, &start, &len);
    long count = (long)N * len;
    for (long i = 0; i < count; ++i) {
        int tl = (int)(i % len), tn = (int)(i / len);
        int th = idx[tl + start], tw = idx[tl + start + H * W];
        int tc = psum - tw - th;
        long pidx = (((long)tn * G + tc) * H + th) * W + tw;
        float tmp = lic360_affine(in[i], scale, bias);
        for (int j = 0; j < rep; ++j) out[pidx + j * stride_out] = tmp;
    }
}

/* A13 tile_add                              extension/tile_add_cuda.cu:22-60 (in place y += x on plane) */
ORC_API void orc_tile_add(float *y, const float *x, int N, int C, int H, int W, int ngroup,
                          const int *idx, const int *plane_idx, int psum) {
    int cpg = C / ngroup, start, len;
    if (psum >= H + W + ngroup - 2) return;
    plane_window(psum, ngroup, H, W, plane_idx, &start, &len);
    long count = (long)N * cpg * len;
    for (long i = 0; i < count; ++i) {
        int pn = (int)(i % N);
        long pp = i / N;
        int pb = (int)(pp % len), og = (int)(pp / len);
        int th = idx[pb + start], tw = idx[pb + start + H * W];
        int tc = psum - th - tw;
        long o = (((long)pn * C + tc * cpg + og) * H + th) * W + tw;
        y[o] = y[o] + x[o];
    }
}

/* ------------------------------------------------------------------------------------
 * A14 entropy_gmm_table                     extension/entropy_gmm_table_cuda.cu:109-135,161-191
 * Mutates weight/delta in place like the reference (:29-57).  out: float[tn][nstep+1].
 * ---------------------------------------------------------------------------------- */
static void gmm_rows(float *w, float *d, const float *m, float *out, int tn, int ng, int nstep,
                     float bias, float total, float beta) {
    for (int n = 0; n < tn; ++n) lic360_softmax_inplace(w + (long)n * ng, ng);
    for (long i = 0; i < (long)tn * ng; ++i) d[i] = lic360_sigma_floor(d[i], beta);
    int ntable = nstep + 1;
    for (int n = 0; n < tn; ++n) {
        float *T = out + (long)n * ntable;
        T[0] = 0.0f;
        T[ntable - 1] = (float)(int)total;
        for (int pt = 1; pt < ntable - 1; ++pt)
            T[pt] = (float)lic360_gmm_cdf_entry(pt, bias, total, w + (long)n * ng, d + (long)n * ng, m + (long)n * ng, ng);
        lic360_cdf_fixup(T, nstep, 0);
    }
}
ORC_API void orc_gmm_table(float *weight, float *delta, const float *mean, float *out, int tn, int ng,
                           int nstep, float bias, float total, float beta) {
    gmm_rows(weight, delta, mean, out, tn, ng, nstep, bias, total, beta);
}
/* batch layout: data = [w slab | sigma slab | mu slab], slab stride `stride` floats (:165,177-183) */
ORC_API void orc_gmm_table_batch(float *data, long stride, float *out, int tn, int ng, int nstep,
                                 float bias, float total, float beta) {
    if (tn > 0) gmm_rows(data, data + stride, data + 2 * stride, out, tn, ng, nstep, bias, total, beta);
}

/* elementwise views of the shared exp / erf (tests/independent_tables.py takes the transcendental VALUES from here and
 * restates everything around them -- summation order, divisions, double-precision rounding, clamp, fix-up -- on its own) */
ORC_API void orc_expf_v(const float *x, float *y, long n) { for (long i = 0; i < n; ++i) y[i] = lic360_expf(x[i]); }
ORC_API void orc_erff_v(const float *x, float *y, long n) { for (long i = 0; i < n; ++i) y[i] = lic360_erff(x[i]); }

/* A15 entropy_table                         extension/entropy_table_cuda.cu:24-96 */
ORC_API void orc_entropy_table(const float *data, float *out, int count, int nstep, float total) {
    float tmp[64];
    for (int n = 0; n < count; ++n) {
        float *T = out + (long)n * (nstep + 1);
        lic360_softmax_cdf(data + (long)n * nstep, T, tmp, nstep, total);
        lic360_cdf_fixup(T, nstep, 1);
    }
}

/* A16 entropy_gmm forward (+ analytic grads) extension/entropy_gmm_cuda.cu:36-68 */
ORC_API void orc_entropy_gmm(const float *weight, const float *delta, const float *mean, const float *label,
                             float *loss, float *wd, float *dd, float *md, float *ld, int M, int ng) {
    const float s2 = 0x1.6a09e6p-1f;             /* float(1/sqrt(2)) */
    const float sp2 = 0x1.988454p-2f;            /* float(1/sqrt(2*pi)) */
    for (int n = 0; n < M; ++n) {
        float sum_p = 0.0f;
        ld[n] = 0.0f;
        for (int i = 0; i < ng; ++i) {
            long k = (long)n * ng + i;
            float xa = (float)((double)label[n] - 0.5 - (double)mean[k]);
            float xb = (float)((double)label[n] + 0.5 - (double)mean[k]);
            float id = (float)(1.0 / (double)delta[k]);
            float fa = (float)(0.5 + 0.5 * (double)lic360_erff(xa * id * s2));
            float fb = (float)(0.5 + 0.5 * (double)lic360_erff(xb * id * s2));
            float p = fb - fa;
            sum_p = __builtin_fmaf(weight[k], p, sum_p);
            float ga = sp2 * id * lic360_expf((float)(-0.5 * (double)xa * (double)xa * (double)id * (double)id));
            float gb = sp2 * id * lic360_expf((float)(-0.5 * (double)xb * (double)xb * (double)id * (double)id));
            ld[n] += (gb - ga) * weight[k];
            dd[k] = id * (-xb * gb + xa * ga) * weight[k];
            md[k] = (ga - gb) * weight[k];
            wd[k] = p;
        }
        loss[n] = -lic360_logf((float)((double)sum_p + 0.0000001));
        float ip = (float)(-1.0 / ((double)sum_p + 0.0000001));
        ld[n] *= ip;
        for (int i = 0; i < ng; ++i) {
            long k = (long)n * ng + i;
            dd[k] *= ip; md[k] *= ip; wd[k] *= ip;
        }
    }
}

/* A17 context_reshape / contex_shift        extension/context_reshape_cuda.cu:30-39,
 *                                           extension/contex_shift_cuda.cu:36-62 */
ORC_API void orc_context_reshape(const float *in, float *out, int N, int C, int H, int W, int ngroup) {
    int cpg = C / ngroup;
    long inner = (long)H * W, total = (long)N * C * inner;
    for (long i = 0; i < total; ++i) {
        long pn = i / inner / C, pc = (i / inner) % C, ps = i % inner;
        long t = (pn * inner * C / cpg + pc / cpg * inner + ps) * cpg + pc % cpg;
        out[t] = in[i];
    }
}
/* inv==0: out[N,C,H+W+G-2,W] (zero-filled here; the reference leaves the rest uninitialised);
 * inv!=0: in is the skewed tensor of height Hin, out height Hin-W-G+2. */
ORC_API void orc_contex_shift(const float *in, float *out, int N, int C, int Hin, int W, int cpn, int inv) {
    int G = C / cpn;
    if (!inv) {
        int Hout = Hin + W + G - 2;
        memset(out, 0, sizeof(float) * (size_t)N * C * Hout * W);
        long total = (long)N * C * Hin * W;
        for (long i = 0; i < total; ++i) {
            int w = (int)(i % W), h = (int)((i / W) % Hin), c = (int)((i / W / Hin) % C), n = (int)(i / W / Hin / C);
            int ph = w + h + c / cpn;
            out[(((long)n * C + c) * Hout + ph) * W + w] = in[i];
        }
    } else {
        int Hout = Hin - W - G + 2;
        long total = (long)N * C * Hout * W;
        for (long i = 0; i < total; ++i) {
            int w = (int)(i % W), h = (int)((i / W) % Hout), c = (int)((i / W / Hout) % C), n = (int)(i / W / Hout / C);
            int ph = w + h + c / cpn;
            out[i] = in[(((long)n * C + c) * Hin + ph) * W + w];
        }
    }
}

/* ------------------------------------------------------------------------------------
 * A1-A3 sphere ops
 * ---------------------------------------------------------------------------------- */
/* sphere_pad_forward_kernel                 extension/sphere_pad_cuda.cu:29-46 */
ORC_API void orc_sphere_pad(const float *in, float *out, int NC, int H, int W, int pad) {
    int Ho = H + 2 * pad, Wo = W + 2 * pad;
    long total = (long)NC * Ho * Wo;
    for (long i = 0; i < total; ++i) {
        int pw = (int)(i % Wo), ph = (int)((i / Wo) % Ho);
        long pn = i / Wo / Ho;
        int th = ph - pad, tw = pw - pad;
        tw = (tw + W) % W;
        if (th < 0 || th >= H) { th = (2 * H - 1 - th) % H; tw = (2 * W - 1 - tw) % W; }
        out[i] = in[(pn * H + th) * W + tw];
    }
}
/* sphere_pad_forward_kernel_inplace          extension/sphere_pad_cuda.cu:48-65; Hp,Wp padded dims */
ORC_API void orc_sphere_pad_inplace(float *data, int NC, int Hp, int Wp, int pad) {
    int H = Hp - 2 * pad, W = Wp - 2 * pad;
    long total = (long)NC * Hp * Wp;
    for (long i = 0; i < total; ++i) {
        int pw = (int)(i % Wp), ph = (int)((i / Wp) % Hp);
        if (pw >= pad && pw < pad + W && ph >= pad && ph < pad + H) continue;
        long pn = i / Wp / Hp;
        int th = ph - pad, tw = pw - pad;
        tw = (tw + W) % W;
        if (th < 0 || th >= H) { th = (2 * H - 1 - th) % H; tw = (2 * W - 1 - tw) % W; }
        data[i] = data[(pn * Hp + th + pad) * Wp + tw + pad];
    }
}
/* sphere_trim_kernel                         extension/sphere_trim_cuda.cu:17-26 */
ORC_API void orc_sphere_trim(float *data, int NC, int H, int W, int pad) {
    long total = (long)NC * H * W;
    for (long i = 0; i < total; ++i) {
        int pw = (int)(i % W), ph = (int)((i / W) % H);
        if (ph < pad || ph >= H - pad || pw < pad || pw >= W - pad) data[i] = 0;
    }
}
/* sphere_cut_edge_forward_kernel             extension/sphere_cut_edge_cuda.cu:31-41 */
/* SpherePadOp.backward, not in place: the gradient of an interior cell is its own cell of top_diff plus the apron cells that were
 * copied from it -- wrap columns, then pole rows (mirrored), then the pole corner -- added in that order
 * (extension/sphere_pad_cuda.cu:107-136).  in_diff [NC][H][W], top_diff [NC][H+2p][W+2p]. */
ORC_API void orc_sphere_pad_backward(float *in_diff, const float *top_diff, int NC, int H, int W, int pad) {
    const int Ho = H + 2 * pad, Wo = W + 2 * pad;
    for (long index = 0; index < (long)NC * H * W; ++index) {
        const int pw = (int)(index % W), ph = (int)((index / W) % H);
        const long pn = index / W / H;
        int th = ph + pad, tw = pw + pad;
        float v = top_diff[(pn * Ho + th) * Wo + tw];
        if (pw < pad || pw >= W - pad) {
            tw = pw < pad ? pw + W + pad : pw - W + pad;
            v += top_diff[(pn * Ho + th) * Wo + tw];
        }
        if (ph < pad || ph >= H - pad) {
            th = ph < pad ? pad - ph - 1 : (2 * H - 1 - ph) + pad;
            tw = W - 1 - pw + pad;
            v += top_diff[(pn * Ho + th) * Wo + tw];
            if (pw < pad || pw >= W - pad) {
                tw = pw < pad ? pad - pw - 1 : 2 * W - pw - 1 + pad;
                v += top_diff[(pn * Ho + th) * Wo + tw];
            }
        }
        in_diff[index] = v;
    }
}
/* ... in place (inplace=true): the interior cells of the padded gradient accumulate their apron copies; the apron itself is left as it
 * is (extension/sphere_pad_cuda.cu:138-165).  diff [NC][Hp][Wp], Hp = H + 2p */
ORC_API void orc_sphere_pad_backward_inplace(float *diff, int NC, int Hp, int Wp, int pad) {
    const int H = Hp - 2 * pad, W = Wp - 2 * pad;
    for (long index = 0; index < (long)NC * H * W; ++index) {
        const int pw = (int)(index % W), ph = (int)((index / W) % H);
        const long pn = index / W / H;
        int th = ph + pad, tw = pw + pad;
        const long t = (pn * Hp + th) * Wp + tw;
        float v = diff[t];
        if (pw < pad || pw >= W - pad) {
            tw = pw < pad ? pw + W + pad : pw - W + pad;
            v += diff[(pn * Hp + th) * Wp + tw];
        }
        if (ph < pad || ph >= H - pad) {
            th = ph < pad ? pad - ph - 1 : (2 * H - 1 - ph) + pad;
            tw = W - 1 - pw + pad;
            v += diff[(pn * Hp + th) * Wp + tw];
            if (pw < pad || pw >= W - pad) {
                tw = pw < pad ? pad - pw - 1 : 2 * W - pw - 1 + pad;
                v += diff[(pn * Hp + th) * Wp + tw];
            }
        }
        diff[t] = v;                                       /* interior cells only: no cell read above is ever written */
    }
}
/* SphereCutEdgeOp.backward: zero apron around the gradient (extension/sphere_cut_edge_cuda.cu:63-78).  in_diff [NC][H][W] */
ORC_API void orc_sphere_cut_edge_backward(float *in_diff, const float *top_diff, int NC, int H, int W, int pad) {
    const int Ho = H - 2 * pad, Wo = W - 2 * pad;
    for (long index = 0; index < (long)NC * H * W; ++index) {
        const int pw = (int)(index % W), ph = (int)((index / W) % H);
        const long pn = index / W / H;
        in_diff[index] = (pw < pad || pw >= Wo + pad || ph < pad || ph >= Ho + pad) ? 0.0f : top_diff[(pn * Ho + ph - pad) * Wo + pw - pad];
    }
}
ORC_API void orc_sphere_cut_edge(const float *in, float *out, int NC, int H, int W, int pad) {
    int Ho = H - 2 * pad, Wo = W - 2 * pad;
    long total = (long)NC * Ho * Wo;
    for (long i = 0; i < total; ++i) {
        int pw = (int)(i % Wo), ph = (int)((i / Wo) % Ho);
        long pn = i / Wo / Ho;
        out[i] = in[(pn * H + ph + pad) * W + pw + pad];
    }
}
/* sphere_lat_scale_forward_kernel            extension/sphere_lat_scale_cuda.cu:31-38 */
ORC_API void orc_sphere_lat_scale(const float *in, const float *weight, float *out, int NC, int H, int W, int npart) {
    int hp = H / npart;
    long total = (long)NC * H * W;
    for (long i = 0; i < total; ++i) {
        int ph = (int)((i / W) % H) / hp;
        out[i] = in[i] * weight[ph];
    }
}

/* ------------------------------------------------------------------------------------
 * A4 imp_map forward (+mask)                 extension/imp_map_cuda.cu:79-110
 * A18 imp2mask, scale                        extension/imp2mask_cuda.cu:25-38, scale_cuda.cu:24-30
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_imp_map(const float *in, const float *imp, float *out, float *mask, int N, int C, int H, int W, int levels) {
    int cpl = C / levels;
    long inner = (long)H * W, total = (long)N * C * inner;
    for (long i = 0; i < total; ++i) {
        long ps = i % inner, pc = (i / inner) % C, pn = i / inner / C;
        int ch = (int)((double)(imp[pn * inner + ps] * (float)levels) + 0.00001) * cpl;
        out[i] = pc < ch ? in[i] : 0.0f;
        if (mask) mask[i] = pc < ch ? 1.0f : 0.0f;
    }
}
/* init_alpha_constrain_kernel + host post-processing, extension/imp_map_cuda.cu:27-71:
 * constrain[n,h] = rt*( |cos((0.5-(h+0.5)/H)*pi)| / max * sc + 1 - sc ).  fp32 steps as torch does them;
 * the cos itself is libdevice cosf in the reference -> compared with 1e-6 tolerance only. */
ORC_API void orc_imp_map_constrain(float *constrain, int N, int H, float rt, float sc) {
    float pi = (float)acos(-1.0);
    float mx = 0.0f;
    float *a = (float *)malloc(sizeof(float) * H);
    for (int h = 0; h < H; ++h) {
        float v = (float)cos((double)(float)((0.5 - ((double)h + 0.5) / (double)H) * (double)pi));
        a[h] = v < 0 ? -v : v;
        if (a[h] > mx) mx = a[h];
    }
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h) {
            float t = a[h] / mx;
            t = t * sc;
            t = t + 1.0f;
            t = t - sc;
            constrain[n * H + h] = rt * t;
        }
    free(a);
}
/* alpha_t of ImpMapOp (extension/imp_map_cuda.cu:27-36,49-52): alpha / (|cos| / max * sw + 1 - sw), fp32 steps as torch does them */
ORC_API void orc_imp_map_alpha(float *alpha_t, int H, float alpha, float sw) {
    float pi = (float)acos(-1.0), mx = 0.0f;
    for (int h = 0; h < H; ++h) {
        float v = (float)cos((double)(float)((0.5 - ((double)h + 0.5) / (double)H) * (double)pi));
        alpha_t[h] = v < 0 ? -v : v;
        if (alpha_t[h] > mx) mx = alpha_t[h];
    }
    for (int h = 0; h < H; ++h) {
        float t = alpha_t[h] / mx;
        t = t * sw;
        t = t + 1.0f;
        t = t - sw;
        alpha_t[h] = alpha / t;
    }
}
/* ImpMapOp.backward (extension/imp_map_cuda.cu:138-298): data_diff = top_diff under the importance mask (floor(imp*levels), no
 * epsilon here), imp_diff by one of four rules (imp_kernel 0..3 -> kernels v1..v4), channel sums in ascending order.
 * top_diff / data_diff [N][C][H][W]; imp, imp_diff [N][1][H][W]; sphere_constrain [N][H]; alpha_t [H] */
ORC_API void orc_imp_map_backward(const float *top_diff, const float *imp, const float *sphere_constrain, const float *alpha_t,
                                  float *data_diff, float *imp_diff, int N, int C, int H, int W, int levels, int imp_kernel, float gamma) {
    const int cpl = C / levels;
    const long inner = (long)H * W, total = (long)N * C * inner;
    for (long i = 0; i < total; ++i) {
        long ps = i % inner, pc = (i / inner) % C, pn = i / inner / C;
        int ch = (int)floor((double)(imp[pn * inner + ps] * (float)levels)) * cpl;
        data_diff[i] = pc < ch ? top_diff[i] : 0.0f;
    }
    for (long index = 0; index < (long)N * inner; ++index) {
        const long ps = index % inner, pn = index / inner;
        const int ph = (int)(ps / W);
        const float sc = sphere_constrain[index / W];
        const int ch = (int)((double)(imp[index] * (float)levels) + 0.00001) * cpl;
        if (imp_kernel == 3) {
            const float decay = sc < 0 ? 0.1f : 1.0f;
            long base = pn * C * inner + ps;
            float tmp = 0.0f, tmax = -10000.0f;
            int target = 0;
            for (int i = 0; i < C; ++i) {
                tmp = tmp + fabsf(top_diff[base]) - alpha_t[ph] * decay;
                base += inner;
                if (tmp > tmax) { tmax = tmp; target = i; }
            }
            imp_diff[index] = target < ch ? gamma : (target > ch ? -gamma : 0.0f);
            continue;
        }
        const int c0 = imp_kernel == 2 ? 0 : ch;
        long base = (pn * C + c0) * inner + ps;
        float diff = 0.0f;
        if (sc > 0) diff = imp_kernel == 0 ? alpha_t[ph] * (float)(C - ch) : alpha_t[ph];
        for (int i = c0; i < C; ++i) {
            diff -= fabsf(top_diff[base]);
            base += inner;
        }
        imp_diff[index] = diff;
    }
}
ORC_API void orc_imp2mask(const float *in, float *out, int N, int C, int H, int W, int cpn) {
    long inner = (long)H * W, total = (long)N * C * inner;
    for (long i = 0; i < total; ++i) {
        long ps = i % inner, pc = (i / inner) % C, pn = i / inner / C;
        int imp = (int)((double)in[pn * inner + ps] + 1e-5) * cpn;
        out[i] = pc < imp ? 1.0f : 0.0f;
    }
}
/* MaskConstrainOp.forward / .backward: zero the non-causal taps of a group-structured conv weight [nout][channel][sz][sz] in place
 * (extension/mask_constrain_cuda.cu:17-41; v5: tw + th + tc >= tn + sz - 1, v6: >) */
ORC_API void orc_mask_constrain(float *w, int nout, int channel, int sz, int ngroup, int constrain) {
    const int group_in = channel / ngroup, group_out = nout / ngroup;
    const long total = (long)nout * channel * sz * sz;
    for (long index = 0; index < total; ++index) {
        const int tw = (int)(index % sz), th = (int)((index / sz) % sz);
        const int tc = (int)((index / sz / sz) % channel) / group_in, tn = (int)(index / sz / sz / channel) / group_out;
        if (constrain == 5 ? (tw + th + tc >= tn + sz - 1) : (tw + th + tc > tn + sz - 1)) w[index] = 0.0f;
    }
}
ORC_API void orc_scale(const float *in, float *out, long n, float bias, float scale) {
    for (long i = 0; i < n; ++i) out[i] = lic360_affine(in[i], scale, bias);
}

/* ------------------------------------------------------------------------------------
 * A5 quant forward                           extension/quant_cuda.cu:35-86,136-169
 * weight_b [C,levels] -> centres increments; top = dequantised value, qidx = index as float,
 * count[C,levels] += -1 per hit (atomicAdd(count,-1.0), :56,74).
 * A7 dquant forward                          extension/dquant_cuda.cu:24-47
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_quant(const float *in, const float *weight_b, float *top, float *qidx, float *count,
                       int N, int C, int H, int W, int levels) {
    float *wq = (float *)malloc(sizeof(float) * C * levels);
    for (int i = 0; i < C * levels; ++i) wq[i] = (i % levels == 0) ? weight_b[i] : lic360_expf(weight_b[i]);
    memset(count, 0, sizeof(float) * C * levels);
    long inner = (long)H * W, total = (long)N * C * inner;
    for (long i = 0; i < total; ++i) {
        int pc = (int)((i / inner) % C);
        float t;
        int j = lic360_quant_one(in[i], wq + pc * levels, levels, &t);
        top[i] = t;
        if (qidx) qidx[i] = (float)j;
        count[pc * levels + j] += -1.0f;
    }
    free(wq);
}
/* QuantOp training side.
 * update_weight (extension/quant_cuda.cu:87-133): per channel, the trailing levels that received (almost) no samples share one
 * increment; an empty first level moves the first centre up; then the counts decay.  exp / log: lic360_exact_math.h (the reference
 * uses libdevice: unpinned at the last ulp, like the tables) */
ORC_API void orc_quant_update_weight(float *weight, float *ncount, int C, int levels, float weight_decay) {
    for (int i = 0; i < C; ++i) {
        int j = levels - 1;
        for (; j > 1; j--)
            if (ncount[i * levels + j] >= 1e-3f) break;
        float tmp = weight[i * levels + j] - lic360_logf((float)(levels - j));
        for (; j < levels; j++) weight[i * levels + j] = tmp;
        if (ncount[i * levels] < 1e-3f) {
            weight[i * levels] = weight[i * levels] + lic360_expf(weight[i * levels + 1]);
            tmp = lic360_logf((lic360_expf(weight[i * levels + 1]) + lic360_expf(weight[i * levels + 2])) / 2.0f);
            weight[i * levels + 1] = tmp;
            weight[i * levels + 2] = tmp;
        }
    }
    for (int i = 0; i < C * levels; ++i) ncount[i] = ncount[i] * weight_decay;
}
/* quant_backward (extension/quant_cuda.cu:135-262): weight_diff[c][j] = sum over the channel's elements with index >= j of
 * (top_data - bottom_data), times the level increment for j > 0 (the reference accumulates with float atomics in no fixed order:
 * here in double); data_diff = top_diff0 (straight through) + top_alpha * top_diff1 / beta when the index output has a gradient */
ORC_API void orc_quant_backward(const float *top_diff0, const float *top_diff1, const float *bottom_data, const float *top_data,
                                const float *qidx, const float *weight_b, float *data_diff, float *weight_diff,
                                int N, int C, int H, int W, int levels, float top_alpha) {
    float *wq = (float *)malloc(sizeof(float) * C * levels);
    double *acc = (double *)calloc((size_t)C * levels, sizeof(double));
    for (int i = 0; i < C * levels; ++i) wq[i] = (i % levels == 0) ? weight_b[i] : lic360_expf(weight_b[i]);
    long inner = (long)H * W, total = (long)N * C * inner;
    for (long i = 0; i < total; ++i) {
        int pc = (int)((i / inner) % C), q = (int)qidx[i];
        float d = top_data[i] - bottom_data[i];
        for (int j = 0; j <= q; ++j) acc[pc * levels + j] += (double)d;
    }
    for (int i = 0; i < C * levels; ++i) {
        float v = (float)acc[i];
        weight_diff[i] = (i % levels != 0) ? v * wq[i] : v;
    }
    for (long i = 0; i < total; ++i) {
        float g = top_diff0[i];
        if (top_diff1) {
            int tc = (int)((i / inner) % C), q = (int)qidx[i];
            const float *w = wq + tc * levels;
            float beta;
            if (top_data[i] < bottom_data[i]) beta = q < levels - 1 ? w[q + 1] : 10000.0f;
            else if (top_data[i] > bottom_data[i]) beta = q > 0 ? w[q] : 10000.0f;
            else if (q == 0) beta = w[q + 1];
            else if (q < levels - 1) beta = (float)(((double)w[q] + (double)w[q + 1]) / 2.0);
            else beta = w[q];
            if (beta < 0.001f) beta = 0.001f;
            g = g + top_alpha * top_diff1[i] / beta;
        }
        data_diff[i] = g;
    }
    free(wq);
    free(acc);
}
ORC_API void orc_dquant(const float *in, const float *mask, const float *weight_b, float *out,
                        int N, int C, int H, int W, int levels) {
    float *wc = (float *)malloc(sizeof(float) * C * levels);
    for (int c = 0; c < C; ++c) {
        wc[c * levels] = weight_b[c * levels];
        for (int i = 1; i < levels; ++i) wc[c * levels + i] = wc[c * levels + i - 1] + lic360_expf(weight_b[c * levels + i]);
    }
    long inner = (long)H * W, total = (long)N * C * inner;
    for (long i = 0; i < total; ++i) {
        int tc = (int)((i / inner) % C);
        int id = (int)((double)in[i] + 0.00001);
        out[i] = mask[i] > 0 ? wc[tc * levels + id] : wc[tc * levels];
    }
    free(wc);
}

/* A6 dtow / wtod                             extension/dtow_cuda.cu:38-75 */
ORC_API void orc_dtow(const float *in, float *out, int N, int C, int H, int W, int s, int d2w) {
    long total = (long)N * C * H * W;
    int p2 = s * s;
    int Co = d2w ? C / p2 : C * p2, Ho = d2w ? H * s : H / s, Wo = d2w ? W * s : W / s;
    for (long i = 0; i < total; ++i) {
        int tw = (int)(i % W), th = (int)((i / W) % H), tc = (int)((i / W / H) % C), tn = (int)(i / W / H / C);
        int pc, ph, pw;
        if (d2w) { pc = tc / p2; int rc = tc % p2; ph = th * s + rc / s; pw = tw * s + rc % s; }
        else { ph = th / s; pw = tw / s; pc = tc * p2 + (th % s) * s + tw % s; }
        out[(((long)tn * Co + pc) * Ho + ph) * Wo + pw] = in[i];
    }
}

/* ------------------------------------------------------------------------------------
 * A20 arithmetic coder + bit I/O              extension/ArithmeticCoder.cpp:15-170,
 *                                             extension/BitIoStream.cpp:12-71
 * Bit-by-bit restatement (STATE_SIZE = 32, extension/coder.h:20,34) on a memory buffer.
 * ---------------------------------------------------------------------------------- */
#define AC_MASK   0xFFFFFFFFull
#define AC_TOP    0x80000000ull
#define AC_SECOND 0x40000000ull
#define AC_MINR   ((1ull << 30) + 2)
#define AC_MAXR   (1ull << 32)

typedef struct {
    uint64_t low, high, code;
    unsigned long underflow;
    uint8_t *buf; size_t cap, len;       /* encoder output / decoder input */
    int cur, nbits;                      /* BitOutputStream currentByte/numBitsFilled or input state */
    size_t rpos; int eof;
    int error;
} orc_ac;

static void bit_write(orc_ac *a, int b) {               /* BitIoStream.cpp:52-66 */
    a->cur = (a->cur << 1) | b;
    a->nbits++;
    if (a->nbits == 8) {
        if (a->len == a->cap) { a->cap = a->cap ? a->cap * 2 : 4096; a->buf = (uint8_t *)realloc(a->buf, a->cap); }
        a->buf[a->len++] = (uint8_t)a->cur;
        a->cur = 0; a->nbits = 0;
    }
}
static int bit_read(orc_ac *a) {                         /* BitIoStream.cpp:19-34 + readCodeBit :131-136 */
    if (a->eof) return 0;
    if (a->nbits == 0) {
        if (a->rpos >= a->len) { a->eof = 1; return 0; }
        a->cur = a->buf[a->rpos++];
        a->nbits = 8;
    }
    a->nbits--;
    return (a->cur >> a->nbits) & 1;
}
static void ac_update(orc_ac *a, int enc, uint32_t symLow, uint32_t symHigh, uint32_t total) {  /* :34-69 */
    if (a->low >= a->high || (a->low & AC_MASK) != a->low || (a->high & AC_MASK) != a->high) { a->error = 1; return; }
    uint64_t range = a->high - a->low + 1;
    if (range < AC_MINR || range > AC_MAXR) { a->error = 2; return; }
    if (symLow == symHigh) { a->error = 3; return; }
    if (total > AC_MINR) { a->error = 4; return; }
    uint64_t newLow = a->low + symLow * range / total;
    uint64_t newHigh = a->low + symHigh * range / total - 1;
    a->low = newLow; a->high = newHigh;
    while (((a->low ^ a->high) & AC_TOP) == 0) {
        if (enc) {                                       /* ArithmeticEncoder::shift :157-164 */
            int bit = (int)(a->low >> 31);
            bit_write(a, bit);
            for (; a->underflow > 0; a->underflow--) bit_write(a, bit ^ 1);
        } else {                                         /* ArithmeticDecoder::shift :119-122 */
            a->code = ((a->code << 1) & AC_MASK) | (uint64_t)bit_read(a);
        }
        a->low = (a->low << 1) & AC_MASK;
        a->high = ((a->high << 1) & AC_MASK) | 1;
    }
    while ((a->low & ~a->high & AC_SECOND) != 0) {
        if (enc) a->underflow++;                         /* :166-170 */
        else a->code = (a->code & AC_TOP) | ((a->code << 1) & (AC_MASK >> 1)) | (uint64_t)bit_read(a);  /* :125-128 */
        a->low = (a->low << 1) & (AC_MASK >> 1);
        a->high = ((a->high << 1) & (AC_MASK >> 1)) | AC_TOP | 1;
    }
}

ORC_API orc_ac *orc_ac_enc_open(void) {
    orc_ac *a = (orc_ac *)calloc(1, sizeof(orc_ac));
    a->low = 0; a->high = AC_MASK;
    return a;
}
ORC_API orc_ac *orc_ac_dec_open(const uint8_t *bytes, size_t n) {
    orc_ac *a = (orc_ac *)calloc(1, sizeof(orc_ac));
    a->low = 0; a->high = AC_MASK;
    a->buf = (uint8_t *)malloc(n ? n : 1); memcpy(a->buf, bytes, n); a->len = n; a->cap = n;
    for (int i = 0; i < 32; i++) a->code = (a->code << 1) | (uint64_t)bit_read(a);   /* :72-79 */
    return a;
}
ORC_API void orc_ac_close(orc_ac *a) { if (a) { free(a->buf); free(a); } }
ORC_API int orc_ac_error(orc_ac *a) { return a->error; }

/* Coder::my_encoder_slice[_mask]             extension/coder.cpp:30-48,70-89
 * table int32 [num][ncode+1]; mask may be NULL (unmasked slice). */
ORC_API void orc_ac_encode_slice(orc_ac *a, const int *table, int ncode, const int *label, const float *mask, int num) {
    for (int i = 0; i < num; ++i) {
        if (mask && mask[i] < 0.5f) continue;
        const int *t = table + (long)i * (ncode + 1);
        uint32_t sym = (uint32_t)label[i];
        ac_update(a, 1, (uint32_t)t[sym], (uint32_t)t[sym + 1], (uint32_t)t[ncode]);
    }
}
/* end_encoder: finish() writes a single 1 bit, stream pads zeros   coder.h:22-26, ArithmeticCoder.cpp:152-154 */
ORC_API size_t orc_ac_enc_finish(orc_ac *a) {
    bit_write(a, 1);
    while (a->nbits != 0) bit_write(a, 0);
    return a->len;
}
ORC_API const uint8_t *orc_ac_bytes(orc_ac *a) { return a->buf; }

/* ArithmeticDecoder::read :82-116 + Coder::my_decoder_slice[_mask] coder.cpp:49-69,90-113.
 * out float[num]; masked entries get file_value. */
ORC_API void orc_ac_decode_slice(orc_ac *a, const int *table, int ncode, const float *mask, float file_value,
                                 float *out, int num) {
    for (int i = 0; i < num; ++i) {
        if (mask && mask[i] < 0.5f) { out[i] = file_value; continue; }
        const int *t = table + (long)i * (ncode + 1);
        uint32_t total = (uint32_t)t[ncode];
        uint64_t range = a->high - a->low + 1;
        uint64_t offset = a->code - a->low;
        uint64_t value = ((offset + 1) * total - 1) / range;
        if (value * range / total > offset || value >= total) { a->error = 5; return; }
        uint32_t start = 0, end = (uint32_t)ncode;
        while (end - start > 1) {
            uint32_t middle = (start + end) >> 1;
            if ((uint32_t)t[middle] > value) end = middle; else start = middle;
        }
        uint32_t sym = start;
        if (offset < (uint32_t)t[sym] * range / total || (uint32_t)t[sym + 1] * range / total <= offset) { a->error = 6; return; }
        ac_update(a, 0, (uint32_t)t[sym], (uint32_t)t[sym + 1], total);
        if (a->error) return;
        if (a->code < a->low || a->code > a->high) { a->error = 7; return; }
        out[i] = (float)sym;
    }
}

/* ------------------------------------------------------------------------------------
 * f3 viewport projection (ProjectsOp): 14 rectilinear viewports sampled from an ERP image.
 * Set-up: extension/projects.hpp:8-20, projects_cuda.cu:7-18 (view rays), :20-49 (Rodrigues), :50-67 (rays -> ERP coordinates),
 * :101-153 (init / update).  Everything in fp32 as the reference; sin / cos / asin / atan / sqrt are libm's here, libdevice's
 * there (unpinned at the last ulp, like the tables' exp / erf).  tf [14][h_out*w_out][2] = (x, y) in ERP pixels.
 * ---------------------------------------------------------------------------------- */
static void orc_mrod(const float *x, const float *y, const float *z, float *data) {
    for (int i = 0; i < 14; i++) {
        float *d = data + i * 9;
        for (int k = 0; k < 9; ++k) d[k] = 0.0f;
        float norm = sqrtf(x[i] * x[i] + y[i] * y[i] + z[i] * z[i]);
        if (norm == 0) { d[0] = 1.0f; d[4] = 1.0f; d[8] = 1.0f; continue; }
        float tx = x[i] / norm, ty = y[i] / norm, tz = z[i] / norm, c = cosf(norm), sn = sinf(norm);
        d[0] = c + (1 - c) * tx * tx;  d[1] = (1 - c) * tx * ty - sn * tz;  d[2] = (1 - c) * tx * tz + sn * ty;
        d[3] = (1 - c) * ty * tx + sn * tz;  d[4] = c + (1 - c) * ty * ty;  d[5] = (1 - c) * ty * tz - sn * tx;
        d[6] = (1 - c) * tz * tx - sn * ty;  d[7] = (1 - c) * tz * ty + sn * tx;  d[8] = c + (1 - c) * tz * tz;
    }
}
ORC_API void orc_projects_tf(float *tf, int h_out, int w_out, const float *theta, const float *phi, float fov, int height, int width) {
    const float pi = (float)acos(-1.0);
    float th[14], ph[14], xa[14], ya[14], za[14], r1[126], r2[126], r[126];
    for (int i = 0; i < 14; ++i) { th[i] = theta[i] * pi; ph[i] = phi[i] * pi; }
    const float fovr = fov * pi;
    const float hfov = fovr * h_out / w_out / 2, wfov = fovr / 2;
    const float c_x = (float)((w_out - 1) / 2.0), c_y = (float)((h_out - 1) / 2.0);
    const float pi_2 = pi / 2, wangle = pi_2 - wfov, hangle = pi_2 - hfov;
    const float w_stride = 2 * sinf(wfov) / sinf(wangle) / (w_out - 1), h_stride = 2 * sinf(hfov) / sinf(hangle) / (h_out - 1);
    for (int i = 0; i < 14; ++i) { xa[i] = 0; ya[i] = 0; za[i] = th[i]; }
    orc_mrod(xa, ya, za, r1);
    for (int i = 0; i < 14; ++i) { xa[i] = r1[i * 9 + 1] * (-ph[i]); ya[i] = r1[i * 9 + 4] * (-ph[i]); za[i] = r1[i * 9 + 7] * (-ph[i]); }
    orc_mrod(xa, ya, za, r2);
    for (int b = 0; b < 14; ++b)
        for (int m = 0; m < 3; ++m)
            for (int n = 0; n < 3; ++n) {
                float sum = 0;
                for (int j = 0; j < 3; ++j) sum += r2[b * 9 + m * 3 + j] * r1[b * 9 + j * 3 + n];
                r[b * 9 + m * 3 + n] = sum;
            }
    const float hx = (float)((width - 1) / 2.0), hy = (float)((height - 1) / 2.0);
    const int inner = h_out * w_out;
    for (int b = 0; b < 14; ++b)
        for (int i = 0; i < inner; ++i) {
            const int w = i % w_out, h = i / w_out;
            float x = 1.0f, y = (w - c_x) * w_stride, z = (h - c_y) * h_stride;
            float rr = sqrtf(x * x + y * y + z * z);
            float xa_ = x / rr, xb = y / rr, xc = -z / rr;
            const float *m = r + b * 9;
            float vx = xa_ * m[0] + xb * m[1] + xc * m[2], vy = xa_ * m[3] + xb * m[4] + xc * m[5], vz = xa_ * m[6] + xb * m[7] + xc * m[8];
            float lat = asinf(vz), t = atanf(vy / vx);
            if (vx <= 0) t = vy > 0 ? t + pi : t - pi;
            tf[((long)b * inner + i) * 2] = t / pi * hx + hx;
            tf[((long)b * inner + i) * 2 + 1] = -2 * lat / pi * hy + hy;
        }
}
/* ProjectsOp.forward: out [14][NC][h_out*w_out] (extension/projects_cuda.cu:181-213) */
ORC_API void orc_projects_forward(const float *in, const float *tf, float *out, int NC, int hs, int ws, int inner, int nearest) {
    for (long index = 0; index < (long)14 * NC * inner; ++index) {
        const int ps = (int)(index % inner), tn = (int)((index / inner) % NC), tb = (int)(index / inner / NC);
        const float fx = tf[((long)tb * inner + ps) * 2], fy = tf[((long)tb * inner + ps) * 2 + 1];
        const float *img = in + (long)tn * hs * ws;
        if (nearest) {
            int tw = (int)floor((double)fx + 0.5) % ws, th = (int)floor((double)fy + 0.5);
            th = th >= hs ? hs - 1 : th;
            out[index] = img[th * ws + tw];
        } else {
            int tw = (int)floorf(fx), th = (int)floorf(fy);
            int pw = (tw + 1) % ws, ph = th + 1 >= hs ? hs - 1 : th + 1;
            float tx = fx - tw, ty = fy - th, ntx = (float)(1. - tx), nty = (float)(1. - ty);
            out[index] = img[th * ws + tw] * ntx * nty + img[th * ws + pw] * tx * nty + img[ph * ws + tw] * ntx * ty + img[ph * ws + pw] * tx * ty;
        }
    }
}
/* ProjectsOp.backward: scatter of the viewport gradients and of their weights (extension/projects_cuda.cu:234-299); double accumulators
 * (the reference: float atomics in no fixed order) */
ORC_API void orc_projects_backward(const float *top_diff, const float *tf, float *in_diff, float *count, int NC, int hs, int ws, int inner, int nearest) {
    const long n_in = (long)NC * hs * ws;
    double *acc = (double *)calloc((size_t)n_in, sizeof(double)), *cnt = (double *)calloc((size_t)n_in, sizeof(double));
    for (long index = 0; index < (long)14 * NC * inner; ++index) {
        const int ps = (int)(index % inner), tn = (int)((index / inner) % NC), tb = (int)(index / inner / NC);
        const float fx = tf[((long)tb * inner + ps) * 2], fy = tf[((long)tb * inner + ps) * 2 + 1], g = top_diff[index];
        const long base = (long)tn * hs * ws;
        if (nearest) {
            int tw = (int)floor((double)fx + 0.5) % ws, th = (int)floor((double)fy + 0.5);
            th = th >= hs ? hs - 1 : th;
            acc[base + th * ws + tw] += g;
            cnt[base + th * ws + tw] += 1.0;
        } else {
            int tw = (int)floorf(fx), th = (int)floorf(fy);
            int pw = (tw + 1) % ws, ph = th + 1 >= hs ? hs - 1 : th + 1;
            float tx = fx - tw, ty = fy - th, ntx = (float)(1. - tx), nty = (float)(1. - ty);
            acc[base + th * ws + tw] += (double)(ntx * nty * g);  cnt[base + th * ws + tw] += (double)(ntx * nty);
            acc[base + th * ws + pw] += (double)(tx * nty * g);   cnt[base + th * ws + pw] += (double)(tx * nty);
            acc[base + ph * ws + tw] += (double)(ntx * ty * g);   cnt[base + ph * ws + tw] += (double)(ntx * ty);
            acc[base + ph * ws + pw] += (double)(tx * ty * g);    cnt[base + ph * ws + pw] += (double)(tx * ty);
        }
    }
    for (long i = 0; i < n_in; ++i) { in_diff[i] = (float)acc[i]; count[i] = (float)cnt[i]; }
    free(acc);
    free(cnt);
}

/* ------------------------------------------------------------------------------------
 * CppOp: ERP -> Craster parabolic projection, row by row (extension/CPP_cuda.cu:11-22,46-85).  Row th keeps ww = int((2 cos(2 t / 3) - 1) W
 * + 0.999) centred columns, t = 3 asin(0.5 - (th + 0.5) / H), resampled from the whole ERP row with linear interpolation in longitude;
 * the vertical weight `hf` is an int in the reference (so it is 0 or 1): kept.  out / mask [NC][H][W]
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_cpp_forward(const float *in, float *out, float *mask, int NC, int H, int W) {
    const float pi = (float)acos(-1);
    for (int th = 0; th < H; ++th) {
        const float t = 3 * asinf((float)(0.5 - (th + 0.5) / H));
        const int ww = (int)((2 * cosf(2 * t / 3) - 1) * W + 0.999), wstart = (W - ww) / 2, wend = wstart + ww;
        for (int tn = 0; tn < NC; ++tn)
            for (int tw = 0; tw < W; ++tw) {
                const long index = ((long)tn * H + th) * W + tw;
                if (mask) mask[index] = (tw < wstart || tw >= wend) ? 0.0f : 1.0f;
                if (tw < wstart || tw >= wend) { out[index] = 0; continue; }
                float phi = (float)((tw - wstart + 0.5) / ww);
                float qw = (float)(phi * W - 0.5), qh = (float)((0.5 - t / pi) * H - 0.5);
                qw = qw < 0 ? qw + W : qw;
                const int wa = (int)qw, wb = (wa + 1) % W;
                const float wf = wa + 1 - qw;
                if (qh < 0) {
                    const long pb = (long)tn * H * W;
                    out[index] = wf * in[pb + wa] + (1 - wf) * in[pb + wb];
                } else if (qh >= H) {
                    const long pb = ((long)tn * H + H - 1) * W;
                    out[index] = wf * in[pb + wa] + (1 - wf) * in[pb + wb];
                } else {
                    /* the reference addresses "the next row" of the flattened [NC*H][W] array even from the last row of a plane
                     * (it then reads the next plane's first row); kept, clamped to the array for the very last row */
                    const int ha = (int)qh, hf = (int)(ha + 1 - qh);
                    const long r0 = (long)tn * H + ha, r1 = r0 + 1 < (long)NC * H ? r0 + 1 : r0;
                    out[index] = wf * hf * in[r0 * W + wa] + (1 - wf) * hf * in[r0 * W + wb] + wf * (1 - hf) * in[r1 * W + wa] + (1 - wf) * (1 - hf) * in[r1 * W + wb];
                }
            }
    }
}

/* ------------------------------------------------------------------------------------
 * ViewportOp (extension/viewport.hpp:7-13, viewport_cuda.cu): one rectilinear viewport per sample, looking at (theta, phi) [radians],
 * field of view fov_deg; fp32 with libm.  Outputs as the reference returns them: view [n,c,ho,wo], rays0 [n,ho,wo,3] (camera frame),
 * rota [n,9], rays [n,ho,wo,3] (rotated), tf [n,ho,wo,2] = (longitude, latitude) of every viewport pixel in radians.
 * ---------------------------------------------------------------------------------- */
static void orc_vp_consts(float fov_deg, int ho, int wo, float *w_stride, float *h_stride, float *c_x, float *c_y, float *wangle) {
    const float pi = (float)acos(-1.0), fov = fov_deg / 180 * pi;
    const float hfov = fov * ho / wo / 2, wfov = fov / 2, pi_2 = pi / 2;
    *c_x = (float)(wo / 2.0);
    *c_y = (float)(ho / 2.0);
    *wangle = pi_2 - wfov;
    const float hangle = pi_2 - hfov;
    *w_stride = 2 * sinf(wfov) / sinf(*wangle) / wo;
    *h_stride = 2 * sinf(hfov) / sinf(hangle) / ho;
}
ORC_API void orc_viewport_rota(const float *theta_phi, float *r, int n) {
    for (int i = 0; i < n; ++i) {
        const float a11 = cosf(theta_phi[2 * i]), a12 = -sinf(theta_phi[2 * i]), a13 = 0, a21 = -a12, a22 = a11, a23 = 0, a31 = 0, a32 = 0, a33 = 1;
        const float c = cosf(theta_phi[2 * i + 1]), sn = sinf(theta_phi[2 * i + 1]);
        const float b11 = c + (1 - c) * a12 * a12, b12 = (1 - c) * a12 * a22, b13 = -sn * a22, b21 = (1 - c) * a12 * a22, b22 = c + (1 - c) * a22 * a22,
                    b23 = sn * a12, b31 = sn * a22, b32 = -sn * a12, b33 = c;
        float *o = r + 9 * i;
        o[0] = b11 * a11 + b12 * a21 + b13 * a31;  o[1] = b11 * a12 + b12 * a22 + b13 * a32;  o[2] = b11 * a13 + b12 * a23 + b13 * a33;
        o[3] = b21 * a11 + b22 * a21 + b23 * a31;  o[4] = b21 * a12 + b22 * a22 + b23 * a32;  o[5] = b21 * a13 + b22 * a23 + b23 * a33;
        o[6] = b31 * a11 + b32 * a21 + b33 * a31;  o[7] = b31 * a12 + b32 * a22 + b33 * a32;  o[8] = b31 * a13 + b32 * a23 + b33 * a33;
    }
}
ORC_API void orc_viewport_forward(const float *in, const float *theta_phi, float *out, float *rays0, float *rota, float *rays, float *tf,
                                  int N, int C, int H, int W, int ho, int wo, float fov_deg) {
    const float pi = (float)acos(-1.0);
    float w_stride, h_stride, c_x, c_y, wangle;
    orc_vp_consts(fov_deg, ho, wo, &w_stride, &h_stride, &c_x, &c_y, &wangle);
    orc_viewport_rota(theta_phi, rota, N);
    const int inner = ho * wo;
    const float hx = (float)W, hy = (float)H;
    for (long i = 0; i < (long)N * inner; ++i) {
        const int w = (int)(i % wo), h = (int)((i / wo) % ho), tb = (int)(i / inner);
        float x = 1.0f, y = (float)((w - c_x + 0.5) * w_stride), z = (float)((h - c_y + 0.5) * h_stride);
        float r = sqrtf(x * x + y * y + z * z);
        rays0[i * 3] = x / r;  rays0[i * 3 + 1] = y / r;  rays0[i * 3 + 2] = -z / r;
        const float *m = rota + 9 * tb;
        const float xa = rays0[i * 3], xb = rays0[i * 3 + 1], xc = rays0[i * 3 + 2];
        rays[i * 3] = xa * m[0] + xb * m[1] + xc * m[2];
        rays[i * 3 + 1] = xa * m[3] + xb * m[4] + xc * m[5];
        rays[i * 3 + 2] = xa * m[6] + xb * m[7] + xc * m[8];
        float lat = asinf(rays[i * 3 + 2]), tx = rays[i * 3], ty = rays[i * 3 + 1], t = atanf(ty / tx);
        if (tx <= 0) t = ty > 0 ? t + pi : t - pi;
        tf[i * 2] = (float)((0.5 * t / pi + 0.5) * hx - 0.5);
        tf[i * 2 + 1] = (float)((0.5 - lat / pi) * hy - 0.5);
    }
    for (long index = 0; index < (long)N * C * inner; ++index) {
        const int ps = (int)(index % inner);
        const long tbase = index / inner, tn = tbase / C;
        const float fx = tf[(tn * inner + ps) * 2], fy = tf[(tn * inner + ps) * 2 + 1];
        const int tw = (int)floorf(fx), th = (int)floorf(fy);
        const int ah = th > 0 ? th : 0, bh = th + 1 >= H ? H - 1 : th + 1, aw = (tw + W) % W, bw = (tw + 1) % W;
        const float tx = fx - tw, ty = fy - th, ntx = (float)(1. - tx), nty = (float)(1. - ty);
        const float *img = in + tbase * H * W;
        out[index] = img[ah * W + aw] * ntx * nty + img[ah * W + bw] * tx * nty + img[bh * W + aw] * ntx * ty + img[bh * W + bw] * tx * ty;
    }
    for (long i = 0; i < (long)N * inner; ++i) {
        tf[i * 2] = (float)(((tf[i * 2] + 0.5) / hx - 0.5) * pi * 2);
        tf[i * 2 + 1] = (float)((0.5 - (tf[i * 2 + 1] + 0.5) / hy) * pi);
    }
}
/* get_viewport_xy: where the direction (theta, phi) of each sample falls in that sample's current viewport, in viewport pixels
 * (extension/viewport_cuda.cu:232-289) */
ORC_API void orc_viewport_xy(const float *theta_phi_next, const float *rota, float *xy, int n, int ho, int wo, float fov_deg) {
    float w_stride, h_stride, c_x, c_y, wangle;
    orc_vp_consts(fov_deg, ho, wo, &w_stride, &h_stride, &c_x, &c_y, &wangle);
    const float rad = (float)(0.5 * wo * tan((double)wangle)), x_bias = (float)(0.5 * wo), y_bias = (float)(0.5 * ho);
    for (int i = 0; i < n; ++i) {
        const float *y = rota + 9 * i;
        const float ts = sinf(theta_phi_next[2 * i]), tc = cosf(theta_phi_next[2 * i]), fs = sinf(theta_phi_next[2 * i + 1]), fc = cosf(theta_phi_next[2 * i + 1]);
        const float xa = tc * fc, xb = ts * fc, xc = fs;
        const float tmp = xa * y[0] + xb * y[3] + xc * y[6], gamma = rad / tmp;
        xy[2 * i] = (float)(gamma * (xa * y[1] + xb * y[4] + xc * y[7]) - 0.5 + x_bias);
        xy[2 * i + 1] = (float)(-gamma * (xa * y[2] + xb * y[5] + xc * y[8]) - 0.5 + y_bias);
    }
}

/* ------------------------------------------------------------------------------------
 * f1: the pieces of the analysis / synthesis transforms that are plain torch in the reference
 * (test/model_zoo.py:8-105,145-170: nn.Conv2d, nn.PReLU, nn.Sigmoid; lic360_operator/GDN.py:66-100), restated in fp32 so
 * that the -m gpu tests of lic360_models.py have a CPU oracle and not only a torch expression.  The library convolution's
 * summation order is not specified (cuDNN there, MIOpen here): this restatement fixes ONE order -- bias first, then input
 * channel major, kernel row, kernel column, one fmaf per term -- and the tests compare within 1e-4.
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_conv2d(const float *x, const float *w, const float *b, float *out, int N, int Cin, int H, int W, int Cout,
                        int k, int stride, int pad) {
    int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    long total = (long)N * Cout * Ho;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < total; ++r) {
        int ho = (int)(r % Ho), co = (int)((r / Ho) % Cout), n = (int)(r / ((long)Ho * Cout));
        for (int wo = 0; wo < Wo; ++wo) {
            float s = b ? b[co] : 0.0f;
            for (int ci = 0; ci < Cin; ++ci)
                for (int kh = 0; kh < k; ++kh) {
                    int hi = ho * stride - pad + kh;
                    if (hi < 0 || hi >= H) continue;
                    for (int kw = 0; kw < k; ++kw) {
                        int wi = wo * stride - pad + kw;
                        if (wi < 0 || wi >= W) continue;
                        s = fmaf(x[(((long)n * Cin + ci) * H + hi) * W + wi], w[(((long)co * Cin + ci) * k + kh) * k + kw], s);
                    }
                }
            out[(((long)n * Cout + co) * Ho + ho) * Wo + wo] = s;
        }
    }
}
/* nn.PReLU(C): y = x > 0 ? x : a[c] x */
ORC_API void orc_prelu(const float *x, const float *a, float *out, int N, int C, long HW) {
    long total = (long)N * C * HW;
    for (long i = 0; i < total; ++i) {
        int c = (int)((i / HW) % C);
        out[i] = x[i] > 0 ? x[i] : a[c] * x[i];
    }
}
/* GDN.py:88-97 on the EFFECTIVE parameters (the caller applies the lower bounds and the pedestal, GDN.py:81-86):
 * norm[c] = sqrt(beta[c] + sum_j gamma[c][j] x[j]^2)  (j ascending, fmaf);  y = x / norm, or x * norm (inverse) */
ORC_API void orc_gdn(const float *x, const float *gamma, const float *beta, float *out, int N, int C, long HW, int inverse) {
#pragma omp parallel for schedule(static)
    for (long p = 0; p < (long)N * HW; ++p) {
        long n = p / HW, q = p % HW;
        for (int c = 0; c < C; ++c) {
            float s = beta[c];
            for (int j = 0; j < C; ++j) {
                float v = x[(n * C + j) * HW + q];
                s = fmaf(gamma[(long)c * C + j], v * v, s);
            }
            float nr = sqrtf(s), xv = x[(n * C + c) * HW + q];
            out[(n * C + c) * HW + q] = inverse ? xv * nr : xv / nr;
        }
    }
}

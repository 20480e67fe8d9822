// pointwise_kernels.hip -- HBM-streaming ops of the LIC360 hot path for gfx950:
// sphere pad/trim/cut/lat-scale (A1-A3), imp_map / imp2mask / scale (A4, A18), quant / dquant
// (A5, A7), dtow (A6), context_reshape / contex_shift (A17).
//
// All are pure gather/scatter/elementwise work: one element per lane, consecutive lanes on
// consecutive W addresses of whichever side is contiguous, grid-stride over the tensor.  The
// sphere apron kernels launch over apron cells only (the reference launches one thread per
// tensor element and lets 97 % of them exit, extension/sphere_pad_cuda.cu:53).
#include "common.h"
#include "lic360_exact_math.h"
#include <cstring>
#include <cstdlib>
#include <cmath>

static thread_local char g_err[512] = "";
void lic360_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
LIC360_API const char *lic360_last_error(void) { return g_err; }
LIC360_API int lic360_version(void) { return 100; }

#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// ---------------------------------------------------------------------------- sphere pad
// source coordinate of padded cell (ph,pw): longitude wrap, pole rows reflected + mirrored
// (extension/sphere_pad_cuda.cu:33-43)
__device__ __forceinline__ void sphere_src(int ph, int pw, int H, int W, int pad, int &th, int &tw) {
    th = ph - pad;
    tw = pw - pad;
    tw = (tw + W) % W;
    if (th < 0 || th >= H) {
        th = (2 * H - 1 - th) % H;
        tw = (2 * W - 1 - tw) % W;
    }
}

__global__ void k_sphere_pad(const float *__restrict__ in, float *__restrict__ out, long total, int H, int W, int Ho, int Wo, int pad) {
    GRID_STRIDE(i, total) {
        int pw = (int)(i % Wo), ph = (int)((i / Wo) % Ho);
        long pn = i / Wo / Ho;
        int th, tw;
        sphere_src(ph, pw, H, W, pad, th, tw);
        out[i] = in[(pn * H + th) * W + tw];
    }
}

// apron-only enumeration of a padded [Hp,Wp] plane: first 2*pad full rows (top then bottom),
// then the 2*pad side columns of the H interior rows.  cells per plane = 2*pad*Wp + 2*pad*H.
__device__ __forceinline__ void apron_cell(int a, int Hp, int Wp, int pad, int &ph, int &pw) {
    int H = Hp - 2 * pad;
    int nrow = 2 * pad * Wp;
    if (a < nrow) {
        int r = a / Wp;
        pw = a % Wp;
        ph = r < pad ? r : Hp - 2 * pad + r;
    } else {
        a -= nrow;
        int r = a / (2 * pad), c = a % (2 * pad);
        ph = pad + r;
        pw = c < pad ? c : Wp - 2 * pad + c;
        (void)H;
    }
}

__global__ void k_sphere_pad_inplace(float *__restrict__ x, long total, int per_plane, int Hp, int Wp, int pad) {
    int H = Hp - 2 * pad, W = Wp - 2 * pad;
    GRID_STRIDE(i, total) {
        long pn = i / per_plane;
        int a = (int)(i % per_plane), ph, pw, th, tw;
        apron_cell(a, Hp, Wp, pad, ph, pw);
        sphere_src(ph, pw, H, W, pad, th, tw);
        x[(pn * Hp + ph) * Wp + pw] = x[(pn * Hp + th + pad) * Wp + tw + pad];   // source is always interior
    }
}

__global__ void k_sphere_trim(float *__restrict__ x, long total, int per_plane, int Hp, int Wp, int pad) {
    GRID_STRIDE(i, total) {
        long pn = i / per_plane;
        int a = (int)(i % per_plane), ph, pw;
        apron_cell(a, Hp, Wp, pad, ph, pw);
        x[(pn * Hp + ph) * Wp + pw] = 0.0f;
    }
}

__global__ void k_sphere_cut_edge(const float *__restrict__ in, float *__restrict__ out, long total, int H, int W, int Ho, int Wo, int pad) {
    GRID_STRIDE(i, total) {
        int pw = (int)(i % Wo), ph = (int)((i / Wo) % Ho);
        long pn = i / Wo / Ho;
        out[i] = in[(pn * H + ph + pad) * W + pw + pad];
    }
}

__global__ void k_sphere_lat_scale(const float *__restrict__ in, const float *__restrict__ weight, float *__restrict__ out, long total, int H, int W, int hp) {
    GRID_STRIDE(i, total) {
        int ph = (int)((i / W) % H) / hp;
        out[i] = in[i] * weight[ph];
    }
}

LIC360_API int lic360_sphere_pad(void *stream, const float *x, float *out, int nc, int h, int w, int pad) {
    ARG_CHECK(x && out && nc > 0 && h > 0 && w > 0 && pad >= 0 && pad <= h && pad <= w);
    int Ho = h + 2 * pad, Wo = w + 2 * pad;
    long total = (long)nc * Ho * Wo;
    hipLaunchKernelGGL(k_sphere_pad, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, x, out, total, h, w, Ho, Wo, pad);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_sphere_pad_inplace(void *stream, float *x, int nc, int hp, int wp, int pad) {
    ARG_CHECK(x && nc > 0 && pad >= 0 && hp > 2 * pad && wp > 2 * pad && pad <= hp - 2 * pad && pad <= wp - 2 * pad);
    if (pad == 0) return 0;
    int per_plane = 2 * pad * wp + 2 * pad * (hp - 2 * pad);
    long total = (long)nc * per_plane;
    hipLaunchKernelGGL(k_sphere_pad_inplace, dim3(lic360_blocks(total, 2)), dim3(256), 0, (hipStream_t)stream, x, total, per_plane, hp, wp, pad);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_sphere_trim(void *stream, float *x, int nc, int h, int w, int pad) {
    ARG_CHECK(x && nc > 0 && pad >= 0 && h >= 2 * pad && w >= 2 * pad);
    if (pad == 0) return 0;
    if (h == 2 * pad || w == 2 * pad) {   // everything is apron
        HIP_TRY(hipMemsetAsync(x, 0, sizeof(float) * (size_t)nc * h * w, (hipStream_t)stream));
        return 0;
    }
    int per_plane = 2 * pad * w + 2 * pad * (h - 2 * pad);
    long total = (long)nc * per_plane;
    hipLaunchKernelGGL(k_sphere_trim, dim3(lic360_blocks(total, 2)), dim3(256), 0, (hipStream_t)stream, x, total, per_plane, h, w, pad);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_sphere_cut_edge(void *stream, const float *x, float *out, int nc, int h, int w, int pad) {
    ARG_CHECK(x && out && nc > 0 && pad >= 0 && h > 2 * pad && w > 2 * pad);
    int Ho = h - 2 * pad, Wo = w - 2 * pad;
    long total = (long)nc * Ho * Wo;
    hipLaunchKernelGGL(k_sphere_cut_edge, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, x, out, total, h, w, Ho, Wo, pad);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_sphere_lat_scale(void *stream, const float *x, const float *weight, float *out, int nc, int h, int w, int npart) {
    ARG_CHECK(x && weight && out && nc > 0 && npart > 0 && h % npart == 0);
    long total = (long)nc * h * w;
    hipLaunchKernelGGL(k_sphere_lat_scale, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, x, weight, out, total, h, w, h / npart);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------- imp_map / imp2mask / scale
__global__ void k_imp_map(const float *__restrict__ in, const float *__restrict__ imp, float *__restrict__ out, float *__restrict__ mask,
                          long total, long inner, int C, int levels, int cpl) {
    GRID_STRIDE(i, total) {
        long ps = i % inner, pc = (i / inner) % C, pn = i / inner / C;
        int ch = (int)((double)(imp[pn * inner + ps] * (float)levels) + 0.00001) * cpl;   // imp_map_cuda.cu:87
        bool keep = pc < ch;
        out[i] = keep ? in[i] : 0.0f;
        if (mask) mask[i] = keep ? 1.0f : 0.0f;
    }
}
__global__ void k_imp2mask(const float *__restrict__ in, float *__restrict__ out, long total, long inner, int C, int cpn) {
    GRID_STRIDE(i, total) {
        long ps = i % inner, pc = (i / inner) % C, pn = i / inner / C;
        int imp = (int)((double)in[pn * inner + ps] + 1e-5) * cpn;                      // imp2mask_cuda.cu:31
        out[i] = pc < imp ? 1.0f : 0.0f;
    }
}
__global__ void k_scale(const float *__restrict__ in, float *__restrict__ out, long total, float bias, float scale) {
    GRID_STRIDE(i, total) out[i] = lic360_affine(in[i], scale, bias);
}

LIC360_API int lic360_imp_map(void *stream, const float *x, const float *imp, float *out, float *mask, int n, int c, int h, int w, int levels) {
    ARG_CHECK(x && imp && out && n > 0 && c > 0 && levels > 0 && c % levels == 0);
    long inner = (long)h * w, total = (long)n * c * inner;
    hipLaunchKernelGGL(k_imp_map, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, x, imp, out, mask, total, inner, c, levels, c / levels);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_imp_map_constrain(void *stream, float *constrain, int n, int h, float rt, float sc) {
    // extension/imp_map_cuda.cu:27-71: |cos((0.5-(h+0.5)/H)*pi)| / max, then rt*(a*sc + 1 - sc); tiny, computed on the host.
    ARG_CHECK(constrain && n > 0 && h > 0);
    float *buf = (float *)malloc(sizeof(float) * (size_t)n * h);
    float pi = (float)acos(-1.0), mx = 0.0f;
    for (int i = 0; i < h; ++i) {
        float v = (float)cos((double)(float)((0.5 - ((double)i + 0.5) / (double)h) * (double)pi));
        buf[i] = v < 0 ? -v : v;
        if (buf[i] > mx) mx = buf[i];
    }
    for (int i = 0; i < h; ++i) {
        float t = buf[i] / mx;
        t = t * sc; t = t + 1.0f; t = t - sc;
        buf[i] = rt * t;
    }
    for (int j = 1; j < n; ++j) memcpy(buf + (size_t)j * h, buf, sizeof(float) * h);
    hipError_t e = hipMemcpyAsync(constrain, buf, sizeof(float) * (size_t)n * h, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    free(buf);
    HIP_TRY(e);
    return 0;
}
LIC360_API int lic360_imp2mask(void *stream, const float *x, float *out, int n, int c, int h, int w, int cpn) {
    ARG_CHECK(x && out && n > 0 && c > 0 && cpn > 0);
    long inner = (long)h * w, total = (long)n * c * inner;
    hipLaunchKernelGGL(k_imp2mask, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, x, out, total, inner, c, cpn);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_scale(void *stream, const float *x, float *out, long count, float bias, float scale) {
    ARG_CHECK(x && out && count >= 0);
    if (count == 0) return 0;
    hipLaunchKernelGGL(k_scale, dim3(lic360_blocks(count, 4)), dim3(256), 0, (hipStream_t)stream, x, out, count, bias, scale);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------- quant / dquant
__global__ void k_quant_weight(const float *__restrict__ wb, float *__restrict__ wq, int total, int levels) {
    GRID_STRIDE(i, total) wq[i] = (i % levels == 0) ? wb[i] : lic360_expf(wb[i]);        // quant_cuda.cu:35-43
}
__global__ void k_quant(const float *__restrict__ in, const float *__restrict__ wq, float *__restrict__ top, float *__restrict__ qidx,
                        float *__restrict__ count, long total, long inner, int C, int levels) {
    const int lane = threadIdx.x & 63;
    GRID_STRIDE(i, total) {
        int pc = (int)((i / inner) % C);
        float t;
        int j = lic360_quant_one(in[i], wq + pc * levels, levels, &t);
        top[i] = t;
        if (qidx) qidx[i] = (float)j;
        // count[pc][j] -= 1 (quant_cuda.cu:56,74).  The sums are integer-valued and < 2^24, so any grouping is exact: one
        // atomic per (wave, channel, level) instead of one per element (12.6 M atomics on 1536 counters otherwise).
        unsigned long long todo = __ballot(1);
        while (todo) {
            const int lead = __ffsll((long long)todo) - 1;
            const int lpc = __builtin_amdgcn_readlane(pc, lead);
            const unsigned long long same = __ballot(pc == lpc) & todo;
            for (int b = 0; b < levels; ++b) {
                const unsigned long long m = __ballot(pc == lpc && j == b) & todo;
                if (m && lane == lead) atomicAdd(count + lpc * levels + b, -(float)__popcll(m));
            }
            todo &= ~same;
        }
    }
}
// one workgroup per (n, channel) slab: level histogram in LDS, one global atomic per (slab, level) -- 32x fewer colliding
// atomics than per wave at 32 images (the sums are integer-valued, so any grouping is exact)
__global__ __launch_bounds__(256) void k_quant_slab(const float *__restrict__ in, const float *__restrict__ wq, float *__restrict__ top,
                                                    float *__restrict__ qidx, float *__restrict__ count, long inner, int C, int levels) {
    __shared__ int hist[64];
    const int pc = blockIdx.x % C, lane = threadIdx.x & 63;
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    const long base = (long)blockIdx.x * inner;
    for (long e0 = 0; e0 < inner; e0 += 256) {
        const long e = e0 + threadIdx.x;
        const bool live = e < inner;
        int j = -1;
        if (live) {
            float t;
            j = lic360_quant_one(in[base + e], wq + pc * levels, levels, &t);
            top[base + e] = t;
            if (qidx) qidx[base + e] = (float)j;
        }
        for (int b = 0; b < levels; ++b) {
            const unsigned long long m = __ballot(j == b);
            if (m && lane == 0) atomicAdd(&hist[b], __popcll(m));
        }
    }
    __syncthreads();
    if (threadIdx.x < levels && hist[threadIdx.x]) atomicAdd(count + pc * levels + threadIdx.x, -(float)hist[threadIdx.x]);   // quant_cuda.cu:56,74
}
__global__ void k_dquant_weight(const float *__restrict__ wb, float *__restrict__ wc, int C, int levels) {
    GRID_STRIDE(c, C) {
        wc[c * levels] = wb[c * levels];
        for (int i = 1; i < levels; ++i) wc[c * levels + i] = wc[c * levels + i - 1] + lic360_expf(wb[c * levels + i]);   // dquant_cuda.cu:24-32
    }
}
__global__ void k_dquant(const float *__restrict__ in, const float *__restrict__ mask, const float *__restrict__ wc, float *__restrict__ out,
                         long total, long inner, int C, int levels) {
    GRID_STRIDE(i, total) {
        int tc = (int)((i / inner) % C);
        int id = (int)((double)in[i] + 0.00001);
        out[i] = mask[i] > 0 ? wc[tc * levels + id] : wc[tc * levels];
    }
}
LIC360_API int lic360_quant(void *stream, const float *x, const float *weight_b, float *wq, float *top, float *qidx, float *count,
                            int n, int c, int h, int w, int levels) {
    ARG_CHECK(x && weight_b && wq && top && count && n > 0 && c > 0 && levels > 0);
    long inner = (long)h * w, total = (long)n * c * inner;
    HIP_TRY(hipMemsetAsync(count, 0, sizeof(float) * (size_t)c * levels, (hipStream_t)stream));
    hipLaunchKernelGGL(k_quant_weight, dim3(lic360_blocks(c * levels)), dim3(256), 0, (hipStream_t)stream, weight_b, wq, c * levels, levels);
    if (levels <= 64 && (long)n * c < (1l << 30))
        hipLaunchKernelGGL(k_quant_slab, dim3((unsigned)(n * c)), dim3(256), 0, (hipStream_t)stream, x, wq, top, qidx, count, inner, c, levels);
    else
        hipLaunchKernelGGL(k_quant, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, x, wq, top, qidx, count, total, inner, c, levels);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_dquant(void *stream, const float *x, const float *mask, const float *weight_b, float *wc, float *out,
                             int n, int c, int h, int w, int levels) {
    ARG_CHECK(x && mask && weight_b && wc && out && n > 0 && c > 0 && levels > 0);
    long inner = (long)h * w, total = (long)n * c * inner;
    hipLaunchKernelGGL(k_dquant_weight, dim3(lic360_blocks(c)), dim3(256), 0, (hipStream_t)stream, weight_b, wc, c, levels);
    hipLaunchKernelGGL(k_dquant, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, x, mask, wc, out, total, inner, c, levels);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------- dtow / wtod
// Indexed by the OUTPUT element so that stores are coalesced; loads hit `stride` interleaved rows.
__global__ void k_dtow(const float *__restrict__ in, float *__restrict__ out, long total, int C, int H, int W, int s) {
    // in [N,C,H,W] -> out [N,C/s^2,H*s,W*s]; out(pc,ph,pw) = in(pc*s^2 + (ph%s)*s + pw%s, ph/s, pw/s)   (dtow_cuda.cu:38-56)
    int Co = C / (s * s), Ho = H * s, Wo = W * s;
    GRID_STRIDE(i, total) {
        int pw = (int)(i % Wo), ph = (int)((i / Wo) % Ho), pc = (int)((i / Wo / Ho) % Co);
        long tn = i / Wo / Ho / Co;
        int tc = pc * s * s + (ph % s) * s + pw % s;
        out[i] = in[((tn * C + tc) * H + ph / s) * W + pw / s];
    }
}
__global__ void k_wtod(const float *__restrict__ in, float *__restrict__ out, long total, int C, int H, int W, int s) {
    // in [N,C,H,W] -> out [N,C*s^2,H/s,W/s]; out(pc,ph,pw) = in(pc/s^2, ph*s + (pc%s^2)/s, pw*s + pc%s)  (dtow_cuda.cu:58-75)
    int p2 = s * s, Co = C * p2, Ho = H / s, Wo = W / s;
    GRID_STRIDE(i, total) {
        int pw = (int)(i % Wo), ph = (int)((i / Wo) % Ho), pc = (int)((i / Wo / Ho) % Co);
        long tn = i / Wo / Ho / Co;
        int tc = pc / p2, rc = pc % p2;
        out[i] = in[((tn * C + tc) * H + ph * s + rc / s) * W + pw * s + rc % s];
    }
}
LIC360_API int lic360_dtow(void *stream, const float *x, float *out, int n, int c, int h, int w, int stride, int d2w) {
    ARG_CHECK(x && out && n > 0 && stride > 0);
    if (d2w) ARG_CHECK(c % (stride * stride) == 0);
    else ARG_CHECK(h % stride == 0 && w % stride == 0);
    long total = (long)n * c * h * w;
    if (d2w) hipLaunchKernelGGL(k_dtow, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, x, out, total, c, h, w, stride);
    else hipLaunchKernelGGL(k_wtod, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, x, out, total, c, h, w, stride);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------- context_reshape / contex_shift
__global__ void k_context_reshape(const float *__restrict__ in, float *__restrict__ out, long total, long inner, int C, int cpg, int inverse) {
    // [N,G*cpg,H,W] <-> [N*G*H*W, cpg]   (context_reshape_cuda.cu:30-39 / :63-72)
    GRID_STRIDE(i, total) {
        long pn = i / inner / C, pc = (i / inner) % C, ps = i % inner;
        long t = (pn * inner * C / cpg + pc / cpg * inner + ps) * cpg + pc % cpg;
        if (inverse) out[i] = in[t];
        else out[t] = in[i];
    }
}
__global__ void k_contex_shift(const float *__restrict__ in, float *__restrict__ out, long total, int C, int Hs, int Hn, int W, int cpn, int inv) {
    // Hn = un-skewed height, Hs = skewed height; element i enumerates the un-skewed tensor (contex_shift_cuda.cu:36-62)
    GRID_STRIDE(i, total) {
        int w = (int)(i % W), h = (int)((i / W) % Hn), c = (int)((i / W / Hn) % C);
        long n = i / W / Hn / C;
        int ph = w + h + c / cpn;
        long p = ((n * C + c) * Hs + ph) * W + w;
        if (inv) out[i] = in[p];
        else out[p] = in[i];
    }
}
LIC360_API int lic360_context_reshape(void *stream, const float *x, float *out, int n, int c, int h, int w, int ngroup, int inverse) {
    ARG_CHECK(x && out && n > 0 && ngroup > 0 && c % ngroup == 0);
    long inner = (long)h * w, total = (long)n * c * inner;
    hipLaunchKernelGGL(k_context_reshape, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, x, out, total, inner, c, c / ngroup, inverse);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_contex_shift(void *stream, const float *x, float *out, int n, int c, int hin, int w, int cpn, int inv) {
    ARG_CHECK(x && out && n > 0 && cpn > 0 && c % cpn == 0);
    int G = c / cpn;
    int Hn = inv ? hin - w - G + 2 : hin, Hs = inv ? hin : hin + w + G - 2;
    ARG_CHECK(Hn > 0);
    long total = (long)n * c * Hn * w;
    if (!inv) HIP_TRY(hipMemsetAsync(out, 0, sizeof(float) * (size_t)n * c * Hs * w, (hipStream_t)stream));
    hipLaunchKernelGGL(k_contex_shift, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, x, out, total, c, Hs, Hn, w, cpn, inv);
    LAUNCH_CHECK();
    return 0;
}

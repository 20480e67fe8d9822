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
    // closed forms of  tw = (tw + W) % W;  th = (2H-1-th) % H;  tw = (2W-1-tw) % W  for pad <= min(H, W) (every entry point checks it):
    // no integer division per cell
    th = ph - pad;
    tw = pw - pad;
    if (tw < 0) tw += W;
    if (tw >= W) tw -= W;
    if (th < 0 || th >= H) {
        th = th < 0 ? -1 - th : 2 * H - 1 - th;
        tw = W - 1 - tw;
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

// Plane-per-workgroup forms (blockIdx.x = plane chunk, blockIdx.y = plane): 32-bit index arithmetic only -- the grid-stride forms
// above spend most of their time in 64-bit div/mod -- and 16-byte accesses along W where rows are 16-byte aligned.
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_sphere_pad_plane(const float *__restrict__ in, float *__restrict__ out, int H, int W, int Ho, int Wo, int pad) {
    const float *ip = in + (long)blockIdx.y * H * W;
    float *op = out + (long)blockIdx.y * Ho * Wo;
    const int cells = Ho * Wo;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < cells; i += gridDim.x * 256) {
        const int ph = i / Wo, pw = i - ph * Wo;
        int th, tw;
        sphere_src(ph, pw, H, W, pad, th, tw);
        op[i] = ip[th * W + tw];
    }
}
__global__ __launch_bounds__(256) void k_sphere_pad_inplace_plane(float *__restrict__ x, int per_plane, int Hp, int Wp, int pad) {
    float *xp = x + (long)blockIdx.y * Hp * Wp;
    const int H = Hp - 2 * pad, W = Wp - 2 * pad;
    for (int a = blockIdx.x * 256 + threadIdx.x; a < per_plane; a += gridDim.x * 256) {
        int ph, pw, th, tw;
        apron_cell(a, Hp, Wp, pad, ph, pw);
        sphere_src(ph, pw, H, W, pad, th, tw);
        xp[ph * Wp + pw] = xp[(th + pad) * Wp + tw + pad];                      // source is always interior
    }
}
// zero the apron: the 2*pad full rows as 16-byte stores (Wp % 4 == 0 keeps every row 16-byte aligned), the side columns per row
__global__ __launch_bounds__(256) void k_sphere_trim_plane(float *__restrict__ x, int Hp, int Wp, int pad, int vec) {
    float *xp = x + (long)blockIdx.y * Hp * Wp;
    const int H = Hp - 2 * pad;
    if (vec) {
        const int q = Wp / 4, nq = 2 * pad * q;
        const f4 z = {0.f, 0.f, 0.f, 0.f};
        for (int i = threadIdx.x; i < nq; i += 256) {
            const int r = i / q, c = i - r * q, ph = r < pad ? r : Hp - 2 * pad + r;
            *(f4 *)(xp + ph * Wp + 4 * c) = z;
        }
    } else {
        for (int i = threadIdx.x; i < 2 * pad * Wp; i += 256) {
            const int r = i / Wp, c = i - r * Wp, ph = r < pad ? r : Hp - 2 * pad + r;
            xp[ph * Wp + c] = 0.0f;
        }
    }
    for (int i = threadIdx.x; i < 2 * pad * H; i += 256) {
        const int r = i / (2 * pad), c = i - r * 2 * pad;
        xp[(pad + r) * Wp + (c < pad ? c : Wp - 2 * pad + c)] = 0.0f;
    }
}
__global__ __launch_bounds__(256) void k_sphere_cut_edge_plane(const float *__restrict__ in, float *__restrict__ out, int H, int W, int Ho, int Wo, int pad) {
    const float *ip = in + (long)blockIdx.y * H * W + pad * W + pad;
    float *op = out + (long)blockIdx.y * Ho * Wo;
    const int cells = Ho * Wo;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < cells; i += gridDim.x * 256) {
        const int ph = i / Wo, pw = i - ph * Wo;
        op[i] = ip[ph * W + pw];
    }
}
static inline unsigned plane_chunks(long cells, int nc) {          // enough workgroups per plane to fill the chip when planes are few
    long per = (cells + 255) / 256, want = (256L * 16 + nc - 1) / nc;
    long g = per < want ? per : want;
    return (unsigned)(g < 1 ? 1 : g);
}

LIC360_API int lic360_sphere_pad(void *stream, const float *x, float *out, int nc, int h, int w, int pad) {
    ARG_CHECK(x && out && nc > 0 && h > 0 && w > 0 && pad >= 0 && pad <= h && pad <= w);
    int Ho = h + 2 * pad, Wo = w + 2 * pad;
    if (nc <= 65535 && (long)Ho * Wo < (1l << 30)) {
        hipLaunchKernelGGL(k_sphere_pad_plane, dim3(plane_chunks((long)Ho * Wo, nc), nc), dim3(256), 0, (hipStream_t)stream, x, out, h, w, Ho, Wo, pad);
        LAUNCH_CHECK();
        return 0;
    }
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
    if (nc <= 65535 && (long)hp * wp < (1l << 30)) {
        hipLaunchKernelGGL(k_sphere_pad_inplace_plane, dim3(plane_chunks(per_plane, nc), nc), dim3(256), 0, (hipStream_t)stream, x, per_plane, hp, wp, pad);
        LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_sphere_pad_inplace, dim3(lic360_blocks(total, 2)), dim3(256), 0, (hipStream_t)stream, x, total, per_plane, hp, wp, pad);
    LAUNCH_CHECK();
    return 0;
}
// SphereTrim immediately followed by the in-place SpherePad of the same width (test/model_zoo.py:83-84, 90-91, 160-161: `y = self.trim(y); y = self.pad2(y)`):
// the refresh writes EVERY apron cell the trim zeroed (k_sphere_pad_inplace* enumerate the whole apron), from interior cells the trim does not touch -- the
// pair is the refresh alone, one pass over the apron's sectors instead of two (147 + 51 -> 147 us per 32 maps of 192 x 260 x 516).
LIC360_API int lic360_sphere_trim_pad_inplace(void *stream, float *x, int nc, int hp, int wp, int pad) {
    return lic360_sphere_pad_inplace(stream, x, nc, hp, wp, pad);
}
// the apron of dst <- the sphere-wrapped interior of src (same geometry; src == dst is lic360_sphere_pad_inplace): what `x + SphereTrim(y)`
// leaves in the apron of a block's output when x's apron had been refreshed (test/model_zoo.py:56-62) -- used behind lic360_sconv3x3
__global__ __launch_bounds__(256) void k_sphere_apron_from_plane(const float *__restrict__ src, float *__restrict__ dst, int per_plane, int Hp, int Wp, int pad) {
    const float *sp = src + (long)blockIdx.y * Hp * Wp;
    float *dp = dst + (long)blockIdx.y * Hp * Wp;
    const int H = Hp - 2 * pad, W = Wp - 2 * pad;
    for (int a = blockIdx.x * 256 + threadIdx.x; a < per_plane; a += gridDim.x * 256) {
        int ph, pw, th, tw;
        apron_cell(a, Hp, Wp, pad, ph, pw);
        sphere_src(ph, pw, H, W, pad, th, tw);
        dp[ph * Wp + pw] = sp[(th + pad) * Wp + tw + pad];
    }
}
LIC360_API int lic360_sphere_apron_from(void *stream, const float *src, float *dst, int nc, int hp, int wp, int pad) {
    ARG_CHECK(src && dst && nc > 0 && nc <= 65535 && pad >= 1 && hp > 2 * pad && wp > 2 * pad && pad <= hp - 2 * pad && pad <= wp - 2 * pad && (long)hp * wp < (1l << 30));
    const int per_plane = 2 * pad * wp + 2 * pad * (hp - 2 * pad);
    hipLaunchKernelGGL(k_sphere_apron_from_plane, dim3(plane_chunks(per_plane, nc), nc), dim3(256), 0, (hipStream_t)stream, src, dst, per_plane, hp, wp, pad);
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
    if (nc <= 65535 && (long)h * w < (1l << 30)) {
        const int vec = (w % 4 == 0) && (((uintptr_t)x & 15) == 0);
        hipLaunchKernelGGL(k_sphere_trim_plane, dim3(1, nc), dim3(256), 0, (hipStream_t)stream, x, h, w, pad, vec);
        LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_sphere_trim, dim3(lic360_blocks(total, 2)), dim3(256), 0, (hipStream_t)stream, x, total, per_plane, h, w, pad);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_sphere_cut_edge(void *stream, const float *x, float *out, int nc, int h, int w, int pad) {
    ARG_CHECK(x && out && nc > 0 && pad >= 0 && h > 2 * pad && w > 2 * pad);
    int Ho = h - 2 * pad, Wo = w - 2 * pad;
    long total = (long)nc * Ho * Wo;
    if (nc <= 65535 && (long)h * w < (1l << 30)) {
        hipLaunchKernelGGL(k_sphere_cut_edge_plane, dim3(plane_chunks((long)Ho * Wo, nc), nc), dim3(256), 0, (hipStream_t)stream, x, out, h, w, Ho, Wo, pad);
        LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_sphere_cut_edge, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, x, out, total, h, w, Ho, Wo, pad);
    LAUNCH_CHECK();
    return 0;
}
// ---- training-side gradients of the sphere ops.  One output element per thread; the reference's order of additions.
// STRIDED: the interior of a padded tensor (in-place form), pitch = w + 2 pad
template <bool INPLACE>
__global__ void k_sphere_pad_backward(float *__restrict__ in_diff, const float *top_diff, long total, int H, int W, int pad) {
    const int Ho = H + 2 * pad, Wo = W + 2 * pad;
    GRID_STRIDE(i, total) {
        const int pw = (int)(i % W), ph = (int)((i / W) % H);
        const long pn = i / W / H;
        int th = ph + pad, tw = pw + pad;
        const long t = (pn * Ho + th) * Wo + tw;
        float v = top_diff[t];
        const bool edge_w = pw < pad || pw >= W - pad;
        if (edge_w) {
            tw = pw < pad ? pw + W + pad : pw - W + pad;
            v += top_diff[(pn * Ho + th) * Wo + tw];
        }
        if (ph < pad || ph >= H - pad) {
            th = ph < pad ? pad - ph - 1 : (2 * H - 1 - ph) + pad;
            tw = W - 1 - pw + pad;
            v += top_diff[(pn * Ho + th) * Wo + tw];
            if (edge_w) {
                tw = pw < pad ? pad - pw - 1 : 2 * W - pw - 1 + pad;
                v += top_diff[(pn * Ho + th) * Wo + tw];
            }
        }
        if constexpr (INPLACE) in_diff[t] = v;              // interior cell of the same buffer: apron cells are only ever read
        else in_diff[i] = v;
    }
}
LIC360_API int lic360_sphere_pad_backward(void *stream, float *in_diff, const float *top_diff, int nc, int h, int w, int pad) {
    ARG_CHECK(in_diff && top_diff && nc > 0 && pad >= 0 && h >= pad && w >= pad && h > 0 && w > 0);
    const long total = (long)nc * h * w;
    hipLaunchKernelGGL(k_sphere_pad_backward<false>, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, in_diff, top_diff, total, h, w, pad);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_sphere_pad_backward_inplace(void *stream, float *diff, int nc, int hp, int wp, int pad) {
    ARG_CHECK(diff && nc > 0 && pad >= 0 && hp > 2 * pad && wp > 2 * pad && hp - 2 * pad >= pad && wp - 2 * pad >= pad);
    const int h = hp - 2 * pad, w = wp - 2 * pad;
    const long total = (long)nc * h * w;
    hipLaunchKernelGGL(k_sphere_pad_backward<true>, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, diff, diff, total, h, w, pad);
    LAUNCH_CHECK();
    return 0;
}
__global__ void k_sphere_cut_edge_backward(float *__restrict__ in_diff, const float *__restrict__ top_diff, long total, int H, int W, int pad) {
    const int Ho = H - 2 * pad, Wo = W - 2 * pad;
    GRID_STRIDE(i, total) {
        const int pw = (int)(i % W), ph = (int)((i / W) % H);
        const long pn = i / W / H;
        in_diff[i] = (pw < pad || pw >= Wo + pad || ph < pad || ph >= Ho + pad) ? 0.0f : top_diff[(pn * Ho + ph - pad) * Wo + pw - pad];
    }
}
LIC360_API int lic360_sphere_cut_edge_backward(void *stream, float *in_diff, const float *top_diff, int nc, int h, int w, int pad) {
    ARG_CHECK(in_diff && top_diff && nc > 0 && pad >= 0 && h > 2 * pad && w > 2 * pad);
    const long total = (long)nc * h * w;
    hipLaunchKernelGGL(k_sphere_cut_edge_backward, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, in_diff, top_diff, total, h, w, pad);
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
LIC360_API int lic360_imp_map_alpha(void *stream, float *alpha_t, int h, float alpha, float sw) {
    ARG_CHECK(alpha_t && h > 0);
    float *buf = (float *)malloc(sizeof(float) * (size_t)h);
    float pi = (float)acos(-1.0), mx = 0.0f;
    for (int i = 0; i < h; ++i) {
        float v = (float)cos((double)(float)((0.5 - ((double)i + 0.5) / (double)h) * (double)pi));
        buf[i] = v < 0 ? -v : v;
        if (buf[i] > mx) mx = buf[i];
    }
    for (int i = 0; i < h; ++i) {
        float t = buf[i] / mx;
        t = t * sw; t = t + 1.0f; t = t - sw;
        buf[i] = alpha / t;
    }
    hipError_t e = hipMemcpyAsync(alpha_t, buf, sizeof(float) * (size_t)h, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    free(buf);
    HIP_TRY(e);
    return 0;
}
// ImpMapOp.backward (training side)
__global__ void k_imp_map_backward_data(const float *__restrict__ top_diff, const float *__restrict__ imp, float *__restrict__ data_diff, long total,
                                        long inner, int C, int levels, int cpl) {
    GRID_STRIDE(i, total) {
        const long ps = i % inner, pn = i / inner / C;
        const int pc = (int)((i / inner) % C);
        const int ch = (int)floor((double)(imp[pn * inner + ps] * (float)levels)) * cpl;       // no epsilon here (imp_map_cuda.cu:147)
        data_diff[i] = pc < ch ? top_diff[i] : 0.0f;
    }
}
__global__ void k_imp_map_backward_imp(const float *__restrict__ top_diff, const float *__restrict__ imp, const float *__restrict__ sphere_constrain,
                                       const float *__restrict__ alpha_t, float *__restrict__ imp_diff, long count, long inner, int C, int W,
                                       int levels, int cpl, int imp_kernel, float gamma) {
    GRID_STRIDE(index, count) {
        const long ps = index % inner, pn = index / inner;
        const int ph = (int)(ps / W);
        const float sc = sphere_constrain[index / W];
        const int ch = (int)((double)(imp[index] * (float)levels) + 0.00001) * cpl;
        if (imp_kernel == 3) {                                                                 // v4: sign of (arg max of the running gain) - level
            const float decay = sc < 0 ? 0.1f : 1.0f, cost = alpha_t[ph];
            long base = pn * C * inner + ps;
            float tmp = 0.0f, tmax = -10000.0f;
            int target = 0;
            for (int i = 0; i < C; ++i) {
                tmp = tmp + fabsf(top_diff[base]) - cost * decay;
                base += inner;
                if (tmp > tmax) { tmax = tmp; target = i; }
            }
            imp_diff[index] = target < ch ? gamma : (target > ch ? -gamma : 0.0f);
        } else {                                                                               // v1 / v2 / v3
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
}
LIC360_API int lic360_imp_map_backward(void *stream, const float *top_diff, const float *imp, const float *sphere_constrain, const float *alpha_t,
                                       float *data_diff, float *imp_diff, int n, int c, int h, int w, int levels, int imp_kernel, float gamma) {
    ARG_CHECK(top_diff && imp && sphere_constrain && alpha_t && data_diff && imp_diff && n > 0 && c > 0 && h > 0 && w > 0 && levels > 0 && c % levels == 0 &&
              imp_kernel >= 0);
    const long inner = (long)h * w, total = (long)n * c * inner, count = (long)n * inner;
    hipLaunchKernelGGL(k_imp_map_backward_data, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, top_diff, imp, data_diff, total, inner, c,
                       levels, c / levels);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_imp_map_backward_imp, dim3(lic360_blocks(count)), dim3(256), 0, (hipStream_t)stream, top_diff, imp, sphere_constrain, alpha_t,
                       imp_diff, count, inner, c, w, levels, c / levels, imp_kernel > 3 ? 0 : imp_kernel, gamma);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_imp2mask(void *stream, const float *x, float *out, int n, int c, int h, int w, int cpn) {
    ARG_CHECK(x && out && n > 0 && c > 0 && cpn > 0);
    long inner = (long)h * w, total = (long)n * c * inner;
    hipLaunchKernelGGL(k_imp2mask, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, x, out, total, inner, c, cpn);
    LAUNCH_CHECK();
    return 0;
}
// MaskConstrainOp (training-side helper of MaskConv2, extension/mask_constrain_cuda.cu:17-41): in place, tiny
__global__ void k_mask_constrain(float *__restrict__ w, long total, int channel, int sz, int group_in, int group_out, int strict) {
    GRID_STRIDE(i, total) {
        const int tw = (int)(i % sz), th = (int)((i / sz) % sz);
        const int tc = (int)((i / sz / sz) % channel) / group_in, tn = (int)(i / sz / sz / channel) / group_out;
        const int lhs = tw + th + tc, rhs = tn + sz - 1;
        if (strict ? lhs > rhs : lhs >= rhs) w[i] = 0.0f;
    }
}
LIC360_API int lic360_mask_constrain(void *stream, float *weight, int nout, int channel, int ksz, int ngroup, int constrain) {
    ARG_CHECK(weight && nout > 0 && channel > 0 && ksz > 0 && ngroup > 0 && nout % ngroup == 0 && channel % ngroup == 0 && (constrain == 5 || constrain == 6));
    const long total = (long)nout * channel * ksz * ksz;
    hipLaunchKernelGGL(k_mask_constrain, dim3(lic360_blocks(total)), dim3(256), 0, (hipStream_t)stream, weight, total, channel, ksz, channel / ngroup,
                       nout / ngroup, constrain == 6 ? 1 : 0);
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
// (also clears the level counts: same size, one launch less than a memset of its own)
__global__ void k_quant_weight(const float *__restrict__ wb, float *__restrict__ wq, float *__restrict__ count, int total, int levels) {
    GRID_STRIDE(i, total) {
        wq[i] = (i % levels == 0) ? wb[i] : lic360_expf(wb[i]);                           // quant_cuda.cu:35-43
        count[i] = 0.0f;
    }
}
// lic360_quant_one for levels <= 8 with the level increments in registers and no data-dependent loop: the subtractions stop at
// the first negative remainder exactly as the loop's `break` does, so every float operation is the same one
__device__ __forceinline__ int quant_one8(float x, const float (&w)[8], int levels, float *top) {
    float tmp = x - w[0];
    if (tmp < 0.0f) { *top = w[0]; return 0; }
    int j = levels - 1;
    float wj = w[0];                                                      // levels == 1: the loop below never runs
    bool done = false;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        if (k < levels && !done) {
            tmp -= w[k];
            wj = w[k];
            if (tmp < 0.0f) { done = true; j = k; }
        }
    }
    if (tmp + tmp + wj < 0.0f) { tmp = tmp + wj; j--; }
    *top = x - tmp;
    return j;
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
// plane (n, c) on blockIdx.y: the channel's centre table sits in registers, 16 bytes per lane along the plane
__global__ __launch_bounds__(256) void k_dquant_plane4(const float *__restrict__ in, const float *__restrict__ mask, const float *__restrict__ wc,
                                                       float *__restrict__ out, int inner4, int C, int levels) {
    const int tc = blockIdx.y % C;
    const float *w = wc + tc * levels;
    const float w0 = w[0];
    const long base = (long)blockIdx.y * inner4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < inner4; i += gridDim.x * 256) {
        const f4 v = ((const f4 *)in)[base + i], m = ((const f4 *)mask)[base + i];
        f4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = m[k] > 0 ? w[(int)((double)v[k] + 0.00001)] : w0;                  // dquant_cuda.cu:34-47
        ((f4 *)out)[base + i] = o;
    }
}
// slab (n, c) per workgroup, 16 bytes per lane; per-thread level counts in 8-bit fields of one 64-bit register (levels <= 8,
// at most 255 elements per thread between flushes), one LDS atomic per (thread, flush) and one global atomic per (slab, level)
__global__ __launch_bounds__(256) void k_quant_slab4(const float *__restrict__ in, const float *__restrict__ wq, float *__restrict__ top,
                                                     float *__restrict__ qidx, float *__restrict__ count, int inner4, int C, int levels, int N, int nper) {
    // workgroup (channel pc, sample group ng) walks the slabs (n, pc), n = ng * nper ..: the level counts are per channel, so the
    // samples of a group share one flush and one set of global atomics
    __shared__ int hist[8];
    const int pc = blockIdx.x % C, ng = blockIdx.x / C;
    const int n_lo = ng * nper, n_hi = n_lo + nper < N ? n_lo + nper : N;
    if (threadIdx.x < 8) hist[threadIdx.x] = 0;
    __syncthreads();
    float wl[8];                                                          // the channel's level increments (uniform address: scalar loads)
#pragma unroll
    for (int k = 0; k < 8; ++k) wl[k] = k < levels ? wq[pc * levels + k] : 0.0f;
    unsigned long long pk = 0;
    int since = 0;
    // per-thread 8-bit fields -> two registers of 16-bit fields, summed over the wave with 6 xor-shuffles, then 8 LDS atomics
    // per wave (64 x 255 < 2^16)
    auto flush = [&]() {
        unsigned long long lo = 0, hi = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            lo |= ((pk >> (8 * b)) & 0xffull) << (16 * b);
            hi |= ((pk >> (8 * (b + 4))) & 0xffull) << (16 * b);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            lo += __shfl_xor(lo, m);
            hi += __shfl_xor(hi, m);
        }
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int c0 = (int)((lo >> (16 * b)) & 0xffffull), c1 = (int)((hi >> (16 * b)) & 0xffffull);
                if (c0) atomicAdd(&hist[b], c0);
                if (c1) atomicAdd(&hist[b + 4], c1);
            }
        }
        pk = 0;
        since = 0;
    };
    for (int n = n_lo; n < n_hi; ++n)
    for (int i0 = 0; i0 < inner4; i0 += 256) {                            // uniform trip count: flush() shuffles across the wave
        const long base = ((long)n * C + pc) * inner4;
        const int i = i0 + threadIdx.x;
        if (i < inner4) {
            const f4 v = ((const f4 *)in)[base + i];
            f4 t, q;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float tt;
                const int j = quant_one8(v[k], wl, levels, &tt);
                t[k] = tt;
                q[k] = (float)j;
                pk += 1ull << (8 * j);
            }
            ((f4 *)top)[base + i] = t;
            if (qidx) ((f4 *)qidx)[base + i] = q;
        }
        since += 4;
        if (since > 250) flush();
    }
    flush();
    __syncthreads();
    if (threadIdx.x < levels && hist[threadIdx.x]) atomicAdd(count + pc * levels + threadIdx.x, -(float)hist[threadIdx.x]);   // quant_cuda.cu:56,74
}
LIC360_API int lic360_quant(void *stream, const float *x, const float *weight_b, float *wq, float *top, float *qidx, float *count,
                            int n, int c, int h, int w, int levels) {
    ARG_CHECK(x && weight_b && wq && top && count && n > 0 && c > 0 && levels > 0);
    long inner = (long)h * w, total = (long)n * c * inner;
    hipLaunchKernelGGL(k_quant_weight, dim3(lic360_blocks(c * levels)), dim3(256), 0, (hipStream_t)stream, weight_b, wq, count, c * levels, levels);
    const bool al16 = inner % 4 == 0 && (((uintptr_t)x | (uintptr_t)top | (uintptr_t)qidx) & 15) == 0;
    if (levels <= 8 && al16 && (long)n * c < (1l << 30) && inner / 4 < (1l << 30))
    {
        const int nper = 1;                                               // (several samples per workgroup were measured: slower)
        const int ngroups = (n + nper - 1) / nper;
        hipLaunchKernelGGL(k_quant_slab4, dim3((unsigned)(c * ngroups)), dim3(256), 0, (hipStream_t)stream, x, wq, top, qidx, count, (int)(inner / 4), c,
                           levels, n, nper);
    }
    else if (levels <= 64 && (long)n * c < (1l << 30))
        hipLaunchKernelGGL(k_quant_slab, dim3((unsigned)(n * c)), dim3(256), 0, (hipStream_t)stream, x, wq, top, qidx, count, inner, c, levels);
    else
        hipLaunchKernelGGL(k_quant, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, x, wq, top, qidx, count, total, inner, c, levels);
    LAUNCH_CHECK();
    return 0;
}
// ---- QuantOp training side
__global__ void k_quant_update_weight(float *__restrict__ weight, float *__restrict__ ncount, int C, int levels, float weight_decay) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    float *w = weight + (long)i * levels, *cnt = ncount + (long)i * levels;
    int j = levels - 1;
    for (; j > 1; j--)
        if (cnt[j] >= 1e-3f) break;
    float tmp = w[j] - lic360_logf((float)(levels - j));
    for (; j < levels; j++) w[j] = tmp;
    if (cnt[0] < 1e-3f) {
        w[0] = w[0] + lic360_expf(w[1]);
        tmp = lic360_logf((lic360_expf(w[1]) + lic360_expf(w[2])) / 2.0f);
        w[1] = tmp;
        w[2] = tmp;
    }
    for (j = 0; j < levels; ++j) cnt[j] = cnt[j] * weight_decay;
}
LIC360_API int lic360_quant_update_weight(void *stream, float *weight, float *ncount, int c, int levels, float weight_decay) {
    ARG_CHECK(weight && ncount && c > 0 && levels >= 3);
    hipLaunchKernelGGL(k_quant_update_weight, dim3(lic360_blocks(c)), dim3(256), 0, (hipStream_t)stream, weight, ncount, c, levels, weight_decay);
    LAUNCH_CHECK();
    return 0;
}
// workgroup (channel, level j): sum of (top - bottom) over the channel's elements with index >= j, in a fixed order (thread-strided
// partial sums, then a tree over the workgroup): reproducible, unlike the reference's float atomics
__global__ __launch_bounds__(256) void k_quant_weight_diff(const float *__restrict__ bottom, const float *__restrict__ top, const float *__restrict__ qidx,
                                                           const float *__restrict__ wq, float *__restrict__ weight_diff, int N, int C, long inner, int levels) {
    __shared__ float red[256];
    const int pc = blockIdx.x / levels, j = blockIdx.x % levels;
    float acc = 0.0f;
    for (int n = 0; n < N; ++n) {
        const long base = ((long)n * C + pc) * inner;
        for (long i = threadIdx.x; i < inner; i += 256)
            if ((int)qidx[base + i] >= j) acc += top[base + i] - bottom[base + i];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) weight_diff[pc * levels + j] = j ? red[0] * wq[pc * levels + j] : red[0];
}
// levels <= 8: one workgroup per channel, one pass: per-thread bucket sums S_q of (top - bottom) in registers, a tree over the workgroup,
// then weight_diff[j] = S_j + S_{j+1} + ... (from the top level down)
__global__ __launch_bounds__(256) void k_quant_weight_diff8(const float *__restrict__ bottom, const float *__restrict__ top, const float *__restrict__ qidx,
                                                            const float *__restrict__ wq, float *__restrict__ weight_diff, int N, int C, long inner, int levels) {
    __shared__ float red[8][256];
    const int pc = blockIdx.x;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < N; ++n) {
        const long base = ((long)n * C + pc) * inner;
        for (long i = threadIdx.x; i < inner; i += 256) {
            const int q = (int)qidx[base + i];
            const float d = top[base + i] - bottom[base + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += q == k ? d : 0.0f;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int k = 0; k < 8; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float run = 0.0f;
        for (int j = levels - 1; j >= 0; --j) {
            run += red[j][0];
            weight_diff[pc * levels + j] = j ? run * wq[pc * levels + j] : run;
        }
    }
}
__global__ void k_quant_data_diff(const float *__restrict__ top_diff0, const float *__restrict__ top_diff1, const float *__restrict__ bottom,
                                  const float *__restrict__ top, const float *__restrict__ qidx, const float *__restrict__ wq, float *__restrict__ data_diff,
                                  long total, long inner, int C, int levels, float alpha) {
    GRID_STRIDE(i, total) {
        float g = top_diff0[i];
        if (top_diff1) {
            const int tc = (int)((i / inner) % C), q = (int)qidx[i];
            const float *w = wq + tc * levels;
            float beta;
            if (top[i] < bottom[i]) beta = q < levels - 1 ? w[q + 1] : 10000.0f;
            else if (top[i] > bottom[i]) beta = q > 0 ? w[q] : 10000.0f;
            else if (q == 0) beta = w[q + 1];
            else if (q < levels - 1) beta = (float)(((double)w[q] + (double)w[q + 1]) / 2.0);
            else beta = w[q];
            if (beta < 0.001f) beta = 0.001f;
            g = g + alpha * top_diff1[i] / beta;
        }
        data_diff[i] = g;
    }
}
LIC360_API int lic360_quant_backward(void *stream, const float *top_diff0, const float *top_diff1, const float *bottom_data, const float *top_data,
                                     const float *qidx, const float *wq, float *data_diff, float *weight_diff, int n, int c, int h, int w, int levels,
                                     float top_alpha) {
    ARG_CHECK(top_diff0 && bottom_data && top_data && qidx && wq && data_diff && weight_diff && n > 0 && c > 0 && h > 0 && w > 0 && levels > 0 &&
              (long)c * levels < (1l << 30));
    const long inner = (long)h * w, total = (long)n * c * inner;
    if (levels <= 8)
        hipLaunchKernelGGL(k_quant_weight_diff8, dim3((unsigned)c), dim3(256), 0, (hipStream_t)stream, bottom_data, top_data, qidx, wq, weight_diff, n, c, inner,
                           levels);
    else
        hipLaunchKernelGGL(k_quant_weight_diff, dim3((unsigned)(c * levels)), dim3(256), 0, (hipStream_t)stream, bottom_data, top_data, qidx, wq, weight_diff, n,
                           c, inner, levels);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_quant_data_diff, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, top_diff0, top_diff1, bottom_data, top_data, qidx, wq,
                       data_diff, total, inner, c, levels, top_alpha);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_dquant(void *stream, const float *x, const float *mask, const float *weight_b, float *wc, float *out,
                             int n, int c, int h, int w, int levels) {
    ARG_CHECK(x && mask && weight_b && wc && out && n > 0 && c > 0 && levels > 0);
    long inner = (long)h * w, total = (long)n * c * inner;
    hipLaunchKernelGGL(k_dquant_weight, dim3(lic360_blocks(c)), dim3(256), 0, (hipStream_t)stream, weight_b, wc, c, levels);
    if (inner % 4 == 0 && (((uintptr_t)x | (uintptr_t)mask | (uintptr_t)out) & 15) == 0 && (long)n * c <= 65535 && inner / 4 < (1l << 30)) {
        hipLaunchKernelGGL(k_dquant_plane4, dim3(plane_chunks(inner / 4, n * c), n * c), dim3(256), 0, (hipStream_t)stream, x, mask, wc, out,
                           (int)(inner / 4), c, levels);
        LAUNCH_CHECK();
        return 0;
    }
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
// stride 2, 16 bytes per lane on both sides.  d2w: lane (row h, quad q) of output channel co reads 4 columns of the input
// channels 4co..4co+3 and writes 8 columns of output rows 2h, 2h+1 (channels 4co, 4co+1 interleaved / 4co+2, 4co+3).
// wtod is the inverse: 8 columns of input rows 2h, 2h+1 -> 4 columns of 4 output channels.
template <bool D2W>
__global__ __launch_bounds__(256) void k_dtow2(const float *__restrict__ in, float *__restrict__ out, int Hs, int Ws4) {
    // Hs x (4 Ws4): the SMALL plane (d2w: input plane; wtod: output plane); blockIdx.y = n * Co + co with Co small-side channel groups
    const long small = (long)blockIdx.y * 4 * Hs * Ws4 * 4, big = (long)blockIdx.y * 4 * Hs * Ws4 * 4;
    const int cells = Hs * Ws4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < cells; i += gridDim.x * 256) {
        const int h = i / Ws4, q = i - h * Ws4;
        const long so = small + (long)h * Ws4 * 4 + 4 * q, ps = (long)Hs * Ws4 * 4;           // + k * ps for channel k of the group
        const long bo = big + (long)(2 * h) * Ws4 * 8 + 8 * q, br = (long)Ws4 * 8;             // + br for row 2h + 1
        if constexpr (D2W) {
            const f4 a = *(const f4 *)(in + so), b = *(const f4 *)(in + so + ps), c = *(const f4 *)(in + so + 2 * ps), d = *(const f4 *)(in + so + 3 * ps);
            *(f4 *)(out + bo) = (f4){a[0], b[0], a[1], b[1]};
            *(f4 *)(out + bo + 4) = (f4){a[2], b[2], a[3], b[3]};
            *(f4 *)(out + bo + br) = (f4){c[0], d[0], c[1], d[1]};
            *(f4 *)(out + bo + br + 4) = (f4){c[2], d[2], c[3], d[3]};
        } else {
            const f4 r0 = *(const f4 *)(in + bo), r1 = *(const f4 *)(in + bo + 4), r2 = *(const f4 *)(in + bo + br), r3 = *(const f4 *)(in + bo + br + 4);
            *(f4 *)(out + so) = (f4){r0[0], r0[2], r1[0], r1[2]};
            *(f4 *)(out + so + ps) = (f4){r0[1], r0[3], r1[1], r1[3]};
            *(f4 *)(out + so + 2 * ps) = (f4){r2[0], r2[2], r3[0], r3[2]};
            *(f4 *)(out + so + 3 * ps) = (f4){r2[1], r2[3], r3[1], r3[3]};
        }
    }
}
LIC360_API int lic360_dtow(void *stream, const float *x, float *out, int n, int c, int h, int w, int stride, int d2w) {
    ARG_CHECK(x && out && n > 0 && stride > 0);
    if (d2w) ARG_CHECK(c % (stride * stride) == 0);
    else ARG_CHECK(h % stride == 0 && w % stride == 0);
    long total = (long)n * c * h * w;
    if (stride == 2 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
        // small-side plane: d2w -> the input plane [h, w] (w % 4 == 0); wtod -> the output plane [h/2, w/2] (w % 8 == 0)
        const int Hs = d2w ? h : h / 2, Ws = d2w ? w : w / 2, groups = d2w ? n * (c / 4) : n * c;
        if (Ws % 4 == 0 && groups <= 65535 && (long)Hs * Ws < (1l << 28)) {
            const dim3 grid(plane_chunks((long)Hs * (Ws / 4), groups), groups);
            if (d2w) hipLaunchKernelGGL(k_dtow2<true>, grid, dim3(256), 0, (hipStream_t)stream, x, out, Hs, Ws / 4);
            else hipLaunchKernelGGL(k_dtow2<false>, grid, dim3(256), 0, (hipStream_t)stream, x, out, Hs, Ws / 4);
            LAUNCH_CHECK();
            return 0;
        }
    }
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

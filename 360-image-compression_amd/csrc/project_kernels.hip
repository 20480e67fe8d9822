// project_kernels.hip -- ProjectsOp: 14 rectilinear viewports of an ERP image, the sampling stage of the viewport
// quality metrics (VPSNR / VSSIM of test/lic360_demo.py:424-441).  Reference: extension/projects.hpp:8-20, projects_cuda.cu.
//   * The sampling coordinates tf [14][h_out*w_out][2] depend on the viewport geometry and the ERP size only; they are computed once
//     per (op, ERP size) on the host, in fp32 with libm where the reference runs libdevice (projects_cuda.cu:7-67,101-153), and
//     uploaded.
//   * forward = bilinear (or nearest) gather, one output element per thread, viewport-major output [14][n*c][h_out*w_out];
//     backward = scatter-add of the gradients and of the interpolation weights (float atomics, as the reference).
#include "common.h"
#include <cmath>
#include <vector>

#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

namespace {
struct Mat3 { float m[9]; };
// Rodrigues rotation about the axis (x, y, z) / |(x, y, z)| by the angle |(x, y, z)| (projects_cuda.cu:20-49)
Mat3 rodrigues(float x, float y, float z) {
    Mat3 r{};
    const float a = sqrtf(x * x + y * y + z * z);
    if (a == 0) { r.m[0] = r.m[4] = r.m[8] = 1.0f; return r; }
    const float ux = x / a, uy = y / a, uz = z / a, c = cosf(a), s = sinf(a);
    r.m[0] = c + (1 - c) * ux * ux;       r.m[1] = (1 - c) * ux * uy - s * uz;  r.m[2] = (1 - c) * ux * uz + s * uy;
    r.m[3] = (1 - c) * uy * ux + s * uz;  r.m[4] = c + (1 - c) * uy * uy;       r.m[5] = (1 - c) * uy * uz - s * ux;
    r.m[6] = (1 - c) * uz * ux - s * uy;  r.m[7] = (1 - c) * uz * uy + s * ux;  r.m[8] = c + (1 - c) * uz * uz;
    return r;
}
}  // namespace

LIC360_API int lic360_projects_tf(void *stream, float *tf_dev, int h_out, int w_out, const float *theta14, const float *phi14, float fov,
                                  int height, int width) {
    ARG_CHECK(tf_dev && theta14 && phi14 && h_out > 1 && w_out > 1 && height > 0 && width > 0);
    const float pi = (float)acos(-1.0), fovr = fov * pi;
    const float hfov = fovr * h_out / w_out / 2, wfov = fovr / 2, half_pi = pi / 2;
    const float cx = (float)((w_out - 1) / 2.0), cy = (float)((h_out - 1) / 2.0);
    const float wstep = 2 * sinf(wfov) / sinf(half_pi - wfov) / (w_out - 1), hstep = 2 * sinf(hfov) / sinf(half_pi - hfov) / (h_out - 1);
    const float hx = (float)((width - 1) / 2.0), hy = (float)((height - 1) / 2.0);
    const int inner = h_out * w_out;
    std::vector<float> tf((size_t)14 * inner * 2);
    for (int v = 0; v < 14; ++v) {
        // yaw about z by theta, then pitch by -phi about the yawed y axis (the second column of the yaw matrix); R = pitch * yaw
        const Mat3 yaw = rodrigues(0.0f, 0.0f, theta14[v] * pi);
        const float ph = -(phi14[v] * pi);
        const Mat3 pitch = rodrigues(yaw.m[1] * ph, yaw.m[4] * ph, yaw.m[7] * ph);
        Mat3 R;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                float sum = 0;
                for (int j = 0; j < 3; ++j) sum += pitch.m[a * 3 + j] * yaw.m[j * 3 + b];
                R.m[a * 3 + b] = sum;
            }
        for (int i = 0; i < inner; ++i) {
            const int w = i % w_out, h = i / w_out;
            const float y = (w - cx) * wstep, z = (h - cy) * hstep, len = sqrtf(1.0f * 1.0f + y * y + z * z);
            const float rx = 1.0f / len, ry = y / len, rz = -z / len;                          // the pixel's ray in the camera frame
            const float vx = rx * R.m[0] + ry * R.m[1] + rz * R.m[2], vy = rx * R.m[3] + ry * R.m[4] + rz * R.m[5];
            const float vz = rx * R.m[6] + ry * R.m[7] + rz * R.m[8];
            const float lat = asinf(vz);
            float lon = atanf(vy / vx);
            if (vx <= 0) lon = vy > 0 ? lon + pi : lon - pi;
            tf[((size_t)v * inner + i) * 2] = lon / pi * hx + hx;
            tf[((size_t)v * inner + i) * 2 + 1] = -2 * lat / pi * hy + hy;
        }
    }
    hipError_t e = hipMemcpyAsync(tf_dev, tf.data(), tf.size() * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    HIP_TRY(e);
    return 0;
}

template <bool NEAREST>
__global__ void k_projects_forward(const float *__restrict__ in, const float *__restrict__ tf, float *__restrict__ out, long total, int NC, int hs, int ws,
                                   int inner) {
    GRID_STRIDE(index, total) {
        const int ps = (int)(index % inner), tn = (int)((index / inner) % NC), tb = (int)(index / inner / NC);
        const float2 f = ((const float2 *)tf)[(long)tb * inner + ps];
        const float *img = in + (long)tn * hs * ws;
        if constexpr (NEAREST) {
            const int tw = (int)floor((double)f.x + 0.5) % ws;
            int th = (int)floor((double)f.y + 0.5);
            th = th >= hs ? hs - 1 : th;
            out[index] = img[th * ws + tw];
        } else {
            const int tw = (int)floorf(f.x), th = (int)floorf(f.y);
            const int pw = (tw + 1) % ws, ph = th + 1 >= hs ? hs - 1 : th + 1;
            const float tx = f.x - tw, ty = f.y - th, ntx = (float)(1. - tx), nty = (float)(1. - ty);
            out[index] = img[th * ws + tw] * ntx * nty + img[th * ws + pw] * tx * nty + img[ph * ws + tw] * ntx * ty + img[ph * ws + pw] * tx * ty;
        }
    }
}
LIC360_API int lic360_projects_forward(void *stream, const float *x, const float *tf, float *out, int nc, int h, int w, int h_out, int w_out, int nearest) {
    ARG_CHECK(x && tf && out && nc > 0 && h > 0 && w > 0 && h_out > 0 && w_out > 0);
    const int inner = h_out * w_out;
    const long total = (long)14 * nc * inner;
    if (nearest) hipLaunchKernelGGL(k_projects_forward<true>, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, x, tf, out, total, nc, h, w, inner);
    else hipLaunchKernelGGL(k_projects_forward<false>, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, x, tf, out, total, nc, h, w, inner);
    LAUNCH_CHECK();
    return 0;
}

template <bool NEAREST>
__global__ void k_projects_backward(const float *__restrict__ top_diff, const float *__restrict__ tf, float *in_diff, float *count, long total, int NC,
                                    int hs, int ws, int inner) {
    GRID_STRIDE(index, total) {
        const int ps = (int)(index % inner), tn = (int)((index / inner) % NC), tb = (int)(index / inner / NC);
        const float2 f = ((const float2 *)tf)[(long)tb * inner + ps];
        const float g = top_diff[index];
        const long base = (long)tn * hs * ws;
        if constexpr (NEAREST) {
            const int tw = (int)floor((double)f.x + 0.5) % ws;
            int th = (int)floor((double)f.y + 0.5);
            th = th >= hs ? hs - 1 : th;
            atomicAdd(in_diff + base + th * ws + tw, g);
            atomicAdd(count + base + th * ws + tw, 1.0f);
        } else {
            const int tw = (int)floorf(f.x), th = (int)floorf(f.y);
            const int pw = (tw + 1) % ws, ph = th + 1 >= hs ? hs - 1 : th + 1;
            const float tx = f.x - tw, ty = f.y - th, ntx = (float)(1. - tx), nty = (float)(1. - ty);
            atomicAdd(in_diff + base + th * ws + tw, ntx * nty * g);  atomicAdd(count + base + th * ws + tw, ntx * nty);
            atomicAdd(in_diff + base + th * ws + pw, tx * nty * g);   atomicAdd(count + base + th * ws + pw, tx * nty);
            atomicAdd(in_diff + base + ph * ws + tw, ntx * ty * g);   atomicAdd(count + base + ph * ws + tw, ntx * ty);
            atomicAdd(in_diff + base + ph * ws + pw, tx * ty * g);    atomicAdd(count + base + ph * ws + pw, tx * ty);
        }
    }
}
LIC360_API int lic360_projects_backward(void *stream, const float *top_diff, const float *tf, float *in_diff, float *count, int nc, int h, int w,
                                        int h_out, int w_out, int nearest) {
    ARG_CHECK(top_diff && tf && in_diff && count && nc > 0 && h > 0 && w > 0 && h_out > 0 && w_out > 0);
    const int inner = h_out * w_out;
    const long total = (long)14 * nc * inner, n_in = (long)nc * h * w;
    HIP_TRY(hipMemsetAsync(in_diff, 0, sizeof(float) * (size_t)n_in, (hipStream_t)stream));
    HIP_TRY(hipMemsetAsync(count, 0, sizeof(float) * (size_t)n_in, (hipStream_t)stream));
    if (nearest) hipLaunchKernelGGL(k_projects_backward<true>, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, top_diff, tf, in_diff, count, total, nc, h, w, inner);
    else hipLaunchKernelGGL(k_projects_backward<false>, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, top_diff, tf, in_diff, count, total, nc, h, w, inner);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------- CppOp
// ERP -> Craster parabolic projection (extension/CPP.hpp:6-11, CPP_cuda.cu:11-22,46-104): row th keeps ww(th) centred columns, resampled
// from the whole ERP row; per-row constants on the host (libm where the reference runs libdevice), one output element per thread.
__global__ void k_cpp_forward(const float *__restrict__ in, float *__restrict__ out, float *__restrict__ mask, const int *__restrict__ ws,
                              const float *__restrict__ theta, long total, int NC, int H, int W, float pi) {
    GRID_STRIDE(index, total) {
        const int tw = (int)(index % W), th = (int)((index / W) % H);
        const long tn = index / W / H;
        const int wstart = ws[2 * th], ww = ws[2 * th + 1], wend = wstart + ww;
        const bool outside = tw < wstart || tw >= wend;
        if (mask) mask[index] = outside ? 0.0f : 1.0f;
        if (outside) { out[index] = 0.0f; continue; }
        const float t = theta[th];
        const float phi = (float)((tw - wstart + 0.5) / ww);
        float qw = (float)(phi * W - 0.5);
        const float qh = (float)((0.5 - t / pi) * H - 0.5);
        qw = qw < 0 ? qw + W : qw;
        const int wa = (int)qw, wb = (wa + 1) % W;
        const float wf = wa + 1 - qw;
        if (qh < 0) {
            const long pb = tn * H * W;
            out[index] = wf * in[pb + wa] + (1 - wf) * in[pb + wb];
        } else if (qh >= H) {
            const long pb = (tn * H + H - 1) * W;
            out[index] = wf * in[pb + wa] + (1 - wf) * in[pb + wb];
        } else {
            const int ha = (int)qh, hf = (int)(ha + 1 - qh);                 // an int in the reference: 0 or 1
            const long r0 = tn * H + ha, r1 = r0 + 1 < (long)NC * H ? r0 + 1 : r0;   // "next row" of the flattened array, as the reference
            out[index] = wf * hf * in[r0 * W + wa] + (1 - wf) * hf * in[r0 * W + wb] + wf * (1 - hf) * in[r1 * W + wa] + (1 - wf) * (1 - hf) * in[r1 * W + wb];
        }
    }
}
// ws_dev [h][2] ints and theta_dev [h] floats are scratch the caller owns (filled here for this height / width)
LIC360_API int lic360_cpp_rows(void *stream, int *ws_dev, float *theta_dev, int h, int w) {
    ARG_CHECK(ws_dev && theta_dev && h > 0 && w > 0);
    std::vector<int> ws((size_t)2 * h);
    std::vector<float> th((size_t)h);
    for (int i = 0; i < h; ++i) {
        const float t = 3 * asinf((float)(0.5 - (i + 0.5) / h));
        const int ww = (int)((2 * cosf(2 * t / 3) - 1) * w + 0.999);
        th[i] = t;
        ws[2 * i] = (w - ww) / 2;
        ws[2 * i + 1] = ww;
    }
    hipError_t e = hipMemcpyAsync(ws_dev, ws.data(), ws.size() * sizeof(int), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipMemcpyAsync(theta_dev, th.data(), th.size() * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    HIP_TRY(e);
    return 0;
}
LIC360_API int lic360_cpp_forward(void *stream, const float *x, float *out, float *mask, const int *ws_dev, const float *theta_dev, int nc, int h, int w) {
    ARG_CHECK(x && out && ws_dev && theta_dev && nc > 0 && h > 0 && w > 0);
    const long total = (long)nc * h * w;
    hipLaunchKernelGGL(k_cpp_forward, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, x, out, mask, ws_dev, theta_dev, total, nc, h, w,
                       (float)acos(-1));
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------- ViewportOp
// One rectilinear viewport per sample, looking at (theta, phi) given per call as a device tensor (extension/viewport.hpp:7-13,
// viewport_cuda.cu).  Per-sample trigonometry on the device (the angles live there); the pinhole constants on the host.
struct VpConsts { float w_stride, h_stride, c_x, c_y, wangle; };
static VpConsts vp_consts(float fov_deg, int ho, int wo) {
    const float pi = (float)acos(-1.0), fov = fov_deg / 180 * pi, hfov = fov * ho / wo / 2, wfov = fov / 2, half = pi / 2;
    VpConsts k;
    k.c_x = (float)(wo / 2.0);
    k.c_y = (float)(ho / 2.0);
    k.wangle = half - wfov;
    k.w_stride = 2 * sinf(wfov) / sinf(k.wangle) / wo;
    k.h_stride = 2 * sinf(hfov) / sinf(half - hfov) / ho;
    return k;
}
// yaw by theta about z, then pitch by phi about the yawed y axis, written out (viewport_cuda.cu:23-58)
__global__ void k_vp_rota(const float *__restrict__ theta_phi, float *__restrict__ r, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a11 = cosf(theta_phi[2 * i]), a12 = -sinf(theta_phi[2 * i]), a21 = -a12, a22 = a11;
    const float c = cosf(theta_phi[2 * i + 1]), s = sinf(theta_phi[2 * i + 1]);
    const float b11 = c + (1 - c) * a12 * a12, b12 = (1 - c) * a12 * a22, b13 = -s * a22, b21 = b12, b22 = c + (1 - c) * a22 * a22, b23 = s * a12;
    const float b31 = s * a22, b32 = -s * a12, b33 = c;
    float *o = r + 9 * i;
    o[0] = b11 * a11 + b12 * a21 + b13 * 0.0f;  o[1] = b11 * a12 + b12 * a22 + b13 * 0.0f;  o[2] = b11 * 0.0f + b12 * 0.0f + b13 * 1.0f;
    o[3] = b21 * a11 + b22 * a21 + b23 * 0.0f;  o[4] = b21 * a12 + b22 * a22 + b23 * 0.0f;  o[5] = b21 * 0.0f + b22 * 0.0f + b23 * 1.0f;
    o[6] = b31 * a11 + b32 * a21 + b33 * 0.0f;  o[7] = b31 * a12 + b32 * a22 + b33 * 0.0f;  o[8] = b31 * 0.0f + b32 * 0.0f + b33 * 1.0f;
}
__global__ void k_vp_rays(float *__restrict__ rays0, const float *__restrict__ rota, float *__restrict__ rays, float *__restrict__ tf, long count,
                          int ho, int wo, VpConsts k, float hx, float hy, float pi) {
    GRID_STRIDE(i, count) {
        const int w = (int)(i % wo), h = (int)((i / wo) % ho), tb = (int)(i / ((long)ho * wo));
        const float x = 1.0f, y = (float)((w - k.c_x + 0.5) * k.w_stride), z = (float)((h - k.c_y + 0.5) * k.h_stride);
        const float r = sqrtf(x * x + y * y + z * z);
        const float xa = x / r, xb = y / r, xc = -z / r;
        rays0[i * 3] = xa;  rays0[i * 3 + 1] = xb;  rays0[i * 3 + 2] = xc;
        const float *m = rota + 9 * tb;
        const float vx = xa * m[0] + xb * m[1] + xc * m[2], vy = xa * m[3] + xb * m[4] + xc * m[5], vz = xa * m[6] + xb * m[7] + xc * m[8];
        rays[i * 3] = vx;  rays[i * 3 + 1] = vy;  rays[i * 3 + 2] = vz;
        const float lat = asinf(vz);
        float t = atanf(vy / vx);
        if (vx <= 0) t = vy > 0 ? t + pi : t - pi;
        tf[i * 2] = (float)((0.5 * t / pi + 0.5) * hx - 0.5);
        tf[i * 2 + 1] = (float)((0.5 - lat / pi) * hy - 0.5);
    }
}
__global__ void k_vp_sample(const float *__restrict__ in, const float *__restrict__ tf, float *__restrict__ out, long total, int C, int hs, int ws, int inner) {
    GRID_STRIDE(index, total) {
        const int ps = (int)(index % inner);
        const long tbase = index / inner, tn = tbase / C;
        const float2 f = ((const float2 *)tf)[tn * inner + ps];
        const int tw = (int)floorf(f.x), th = (int)floorf(f.y);
        const int ah = th > 0 ? th : 0, bh = th + 1 >= hs ? hs - 1 : th + 1, aw = (tw + ws) % ws, bw = (tw + 1) % ws;
        const float tx = f.x - tw, ty = f.y - th, ntx = (float)(1. - tx), nty = (float)(1. - ty);
        const float *img = in + tbase * hs * ws;
        out[index] = img[ah * ws + aw] * ntx * nty + img[ah * ws + bw] * tx * nty + img[bh * ws + aw] * ntx * ty + img[bh * ws + bw] * tx * ty;
    }
}
__global__ void k_vp_angles(float *__restrict__ tf, long count, float hx, float hy, float pi) {
    GRID_STRIDE(i, count) {
        tf[i * 2] = (float)(((tf[i * 2] + 0.5) / hx - 0.5) * pi * 2);
        tf[i * 2 + 1] = (float)((0.5 - (tf[i * 2 + 1] + 0.5) / hy) * pi);
    }
}
LIC360_API int lic360_viewport_rota(void *stream, const float *theta_phi, float *rota, int n) {
    ARG_CHECK(theta_phi && rota && n > 0);
    hipLaunchKernelGGL(k_vp_rota, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, theta_phi, rota, n);
    LAUNCH_CHECK();
    return 0;
}
// out [n,c,ho,wo]; rays0, rays [n,ho,wo,3]; rota [n,9]; tf [n,ho,wo,2]: on return the (longitude, latitude) of every viewport pixel
LIC360_API int lic360_viewport_forward(void *stream, const float *x, const float *theta_phi, float *out, float *rays0, float *rota, float *rays, float *tf,
                                       int n, int c, int h, int w, int ho, int wo, float fov_deg) {
    ARG_CHECK(x && theta_phi && out && rays0 && rota && rays && tf && n > 0 && c > 0 && h > 0 && w > 0 && ho > 0 && wo > 0);
    hipStream_t s = (hipStream_t)stream;
    const float pi = (float)acos(-1.0);
    const VpConsts k = vp_consts(fov_deg, ho, wo);
    const long count = (long)n * ho * wo, total = count * c;
    hipLaunchKernelGGL(k_vp_rota, dim3((n + 63) / 64), dim3(64), 0, s, theta_phi, rota, n);
    hipLaunchKernelGGL(k_vp_rays, dim3(lic360_blocks(count)), dim3(256), 0, s, rays0, rota, rays, tf, count, ho, wo, k, (float)w, (float)h, pi);
    hipLaunchKernelGGL(k_vp_sample, dim3(lic360_blocks(total, 4)), dim3(256), 0, s, x, tf, out, total, c, h, w, ho * wo);
    hipLaunchKernelGGL(k_vp_angles, dim3(lic360_blocks(count)), dim3(256), 0, s, tf, count, (float)w, (float)h, pi);
    LAUNCH_CHECK();
    return 0;
}
__global__ void k_vp_xy(const float *__restrict__ next, const float *__restrict__ rota, float *__restrict__ xy, int n, float rad, float x_bias, float y_bias) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *y = rota + 9 * i;
    const float ts = sinf(next[2 * i]), tc = cosf(next[2 * i]), fs = sinf(next[2 * i + 1]), fc = cosf(next[2 * i + 1]);
    const float xa = tc * fc, xb = ts * fc, xc = fs;
    const float gamma = rad / (xa * y[0] + xb * y[3] + xc * y[6]);
    xy[2 * i] = (float)(gamma * (xa * y[1] + xb * y[4] + xc * y[7]) - 0.5 + x_bias);
    xy[2 * i + 1] = (float)(-gamma * (xa * y[2] + xb * y[5] + xc * y[8]) - 0.5 + y_bias);
}
LIC360_API int lic360_viewport_xy(void *stream, const float *theta_phi_next, const float *rota, float *xy, int n, int ho, int wo, float fov_deg) {
    ARG_CHECK(theta_phi_next && rota && xy && n > 0 && ho > 0 && wo > 0);
    const VpConsts k = vp_consts(fov_deg, ho, wo);
    hipLaunchKernelGGL(k_vp_xy, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, theta_phi_next, rota, xy, n, (float)(0.5 * wo * tan((double)k.wangle)),
                       (float)(0.5 * wo), (float)(0.5 * ho));
    LAUNCH_CHECK();
    return 0;
}

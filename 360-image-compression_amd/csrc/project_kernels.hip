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

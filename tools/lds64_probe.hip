// lds64_probe.hip -- one-off probe: does ds_read_b64 at a 4-byte-aligned (odd dword) LDS address work on gfx950, and at what cost?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(float *out, long *cyc, int off, int iters) {
    __shared__ float buf[4096];
    int l = threadIdx.x;
    for (int i = l; i < 4096; i += 64) buf[i] = (float)i;
    __syncthreads();
    float acc0 = 0, acc1 = 0;
    long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            f32x2 v;
            const float *p = buf + ((l + off + j * 66 + it) & 2047);
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(unsigned long)(__attribute__((address_space(3))) const float *)p));
            acc0 += v[0]; acc1 += v[1];
        }
    }
    long t1 = clock64();
    out[l] = acc0; out[64 + l] = acc1;
    if (l == 0) cyc[0] = (t1 - t0);
}
int main() {
    float *o; long *c;
    (void)hipMalloc(&o, 1024); (void)hipMalloc(&c, 8);
    for (int off = 0; off < 2; ++off) {
        hipLaunchKernelGGL(k, 1, 64, 0, 0, o, c, off, 1);
        float h[128]; (void)hipMemcpy(h, o, 512, hipMemcpyDeviceToHost);
        // expected: sum over j of (idx, idx+1)
        int bad = 0;
        for (int l = 0; l < 64; ++l) { float e0 = 0, e1 = 0; for (int j = 0; j < 16; ++j) { int idx = (l + off + j * 66) & 2047; e0 += idx; e1 += idx + 1; } bad += (h[l] != e0) + (h[64 + l] != e1); }
        hipLaunchKernelGGL(k, 1, 64, 0, 0, o, c, off, 1000);
        long hc; (void)hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
        printf("lane-parity offset %d: mismatches %d/128, cycles per ds_read_b64 (dependent wait) %.1f, err=%d\n", off, bad, hc / 16000.0, (int)hipGetLastError());
    }
    return 0;
}

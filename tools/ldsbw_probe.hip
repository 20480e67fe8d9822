// ldsbw_probe.hip -- LDS read throughput on gfx950 for lane-consecutive (stride 1 dword, possibly 4-byte aligned only) reads
// of width b32/b64/b96/b128, 12 waves per CU on all CUs, plus the shader clock actually sustained.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define LDSA(p) ((unsigned)(unsigned long)(__attribute__((address_space(3))) const float *)(p))
template <int W, int STRIDE>
__global__ __launch_bounds__(768) void k(float *out, long *cyc, int iters, int off) {
    __shared__ float buf[8192];
    int t = threadIdx.x, l = t & 63;
    for (int i = t; i < 8192; i += 768) buf[i] = (float)i;
    __syncthreads();
    float acc = 0;
    unsigned a0 = LDSA(buf) + 4u * (unsigned)(l * STRIDE + off + (t >> 6) * 72);
    long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned a = a0 + 4u * (unsigned)(j * 288 + ((it & 3) * 1152));
            if constexpr (W == 1) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a)); asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); acc += v; }
            if constexpr (W == 2) { f2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a)); asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); acc += v[0] + v[1]; }
            if constexpr (W == 3) { f3 v; asm volatile("ds_read_b96 %0, %1" : "=v"(v) : "v"(a)); asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); acc += v[0] + v[2]; }
            if constexpr (W == 4) { f4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); acc += v[0] + v[3]; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long t1 = clock64();
    out[blockIdx.x * 768 + t] = acc;
    if (t == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int W, int STRIDE>
void run(const char *name, float *o, long *c, int off) {
    const int iters = 20000;
    hipLaunchKernelGGL((k<W, STRIDE>), 256, 768, 0, 0, o, c, 10, off);
    (void)hipDeviceSynchronize();
    auto w0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL((k<W, STRIDE>), 256, 768, 0, 0, o, c, iters, off);
    (void)hipDeviceSynchronize();
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
    long hc; (void)hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    double ninst = 12.0 * 8 * iters;                         // wave-level LDS instructions per CU
    printf("%-28s off %d: %.2f shader clocks / wave-instr / CU  (%.1f B/clk/CU), clock64 rate %.0f MHz, wall %.0f us, err %d\n", name, off, hc / ninst,
           64.0 * 4 * W * ninst / hc, hc / us, us, (int)hipGetLastError());
}
int main() {
    float *o; long *c;
    (void)hipMalloc(&o, 256 * 768 * 4); (void)hipMalloc(&c, 8);
    for (int off = 0; off < 2; ++off) {
        run<1, 1>("b32 lane-stride 1 dword", o, c, off);
        run<2, 1>("b64 lane-stride 1 dword", o, c, off);
        run<3, 1>("b96 lane-stride 1 dword", o, c, off);
        run<4, 1>("b128 lane-stride 1 dword", o, c, off);
    }
    run<2, 2>("b64 lane-stride 2 dwords", o, c, 0);
    run<4, 4>("b128 lane-stride 4 dwords", o, c, 0);
    return 0;
}

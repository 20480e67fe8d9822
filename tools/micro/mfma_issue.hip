// Micro-benchmark (GPU box): how many cycles does v_mfma_f32_16x16x4_f32 cost a SIMD when ONE wave (or two) issues it, alone or
// with scalar / vector-memory instructions in between?  Prints cycles per MFMA for a few instruction mixes.
//   hipcc --offload-arch=gfx950 -O3 mfma_issue.hip -o /tmp/mfma_issue && /tmp/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define REP6(x) x x x x x x
template <int MODE>
__global__ __launch_bounds__(512) void k(const float *__restrict__ src, float *__restrict__ out, unsigned long long *cyc, int iters) {
    __shared__ float ballast[30 * 1024];                       // one workgroup per CU
    if (iters < 0) ballast[threadIdx.x] = 1.f;
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0;
    float a = src[threadIdx.x], b = src[threadIdx.x + 64];
    const float *p = src + (threadIdx.x & 63);
    float l0 = 0, l1 = 0;
    unsigned s0 = 1, s1 = 2;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) {                             // six MFMAs back to back
            asm volatile(
                "v_mfma_f32_16x16x4_f32 %0, %6, %7, %0\n v_mfma_f32_16x16x4_f32 %1, %6, %7, %1\n v_mfma_f32_16x16x4_f32 %2, %6, %7, %2\n"
                "v_mfma_f32_16x16x4_f32 %3, %6, %7, %3\n v_mfma_f32_16x16x4_f32 %4, %6, %7, %4\n v_mfma_f32_16x16x4_f32 %5, %6, %7, %5\n"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5) : "v"(a), "v"(b));
        } else if constexpr (MODE == 1) {                      // six MFMAs, then nine scalar instructions (the production K loop's shape)
            asm volatile(
                "v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n v_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n v_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n"
                "v_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n v_mfma_f32_16x16x4_f32 %4, %8, %9, %4\n v_mfma_f32_16x16x4_f32 %5, %8, %9, %5\n"
                "s_add_u32 %6, %6, 3\n s_addc_u32 %7, %7, 0\n s_add_u32 %6, %6, 3\n s_addc_u32 %7, %7, 0\n s_add_u32 %6, %6, 3\n s_addc_u32 %7, %7, 0\n"
                "s_add_u32 %6, %6, 3\n s_addc_u32 %7, %7, 0\n s_mul_i32 %6, %6, 3\n"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+s"(s0), "+s"(s1) : "v"(a), "v"(b) : "scc");
        } else if constexpr (MODE == 2) {                      // the same nine scalar instructions spread between the MFMAs
            asm volatile(
                "v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n s_add_u32 %6, %6, 3\n s_addc_u32 %7, %7, 0\n"
                "v_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n s_add_u32 %6, %6, 3\n s_addc_u32 %7, %7, 0\n"
                "v_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n s_add_u32 %6, %6, 3\n s_addc_u32 %7, %7, 0\n"
                "v_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n s_add_u32 %6, %6, 3\n s_addc_u32 %7, %7, 0\n"
                "v_mfma_f32_16x16x4_f32 %4, %8, %9, %4\n s_mul_i32 %6, %6, 3\n"
                "v_mfma_f32_16x16x4_f32 %5, %8, %9, %5\n"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+s"(s0), "+s"(s1) : "v"(a), "v"(b) : "scc");
        } else if constexpr (MODE == 3) {                      // six MFMAs + two global loads (L2 hits) consumed one iteration later
            asm volatile(
                "s_waitcnt vmcnt(0)\n"
                "v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n v_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n v_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n"
                "v_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n v_mfma_f32_16x16x4_f32 %4, %8, %9, %4\n v_mfma_f32_16x16x4_f32 %5, %8, %9, %5\n"
                "global_load_dword %6, %10, off\n global_load_dword %7, %10, off offset:256\n"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(l0), "+v"(l1) : "v"(a), "v"(b), "v"(p));
        } else if constexpr (MODE == 4) {                      // the A operand changes every two MFMAs, B alternates (register-read pattern of the K loop)
            asm volatile(
                "v_mfma_f32_16x16x4_f32 %0, %6, %7, %0\n v_mfma_f32_16x16x4_f32 %1, %6, %8, %1\n v_mfma_f32_16x16x4_f32 %2, %9, %7, %2\n"
                "v_mfma_f32_16x16x4_f32 %3, %9, %8, %3\n v_mfma_f32_16x16x4_f32 %4, %8, %7, %4\n v_mfma_f32_16x16x4_f32 %5, %8, %6, %5\n"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5) : "v"(a), "v"(b), "v"(l0), "v"(l1));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    f4 s = c0 + c1 + c2 + c3 + c4 + c5;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w + l0 + l1 + (float)s0 + (iters < 0 ? ballast[threadIdx.x ^ 1] : 0.f);
}
template <int MODE>
static void run(const char *what, int threads, float *src, float *out, unsigned long long *cyc) {
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, src, out, cyc, 100);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, src, out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) { mean += (double)h[b * 8 + w]; ++n; }
    mean /= n;
    const int waves_per_simd = threads / 256;
    fflush(stdout); printf("%-62s %d wave/SIMD: %6.1f ns per MFMA and SIMD  (memtime ticks per MFMA and wave %.2f)\n", what, waves_per_simd,
           ms * 1e6 / (6.0 * iters * waves_per_simd), mean / (6.0 * iters));
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *src, *out; unsigned long long *cyc;
    hipMalloc(&src, 1 << 20); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    hipMemset(src, 0, 1 << 20);
    for (int threads : {256, 512}) {
        run<0>("6 MFMA", threads, src, out, cyc);
        run<4>("6 MFMA, operands alternating", threads, src, out, cyc);
        run<1>("6 MFMA then 9 SALU", threads, src, out, cyc);
        run<2>("6 MFMA with 9 SALU in between", threads, src, out, cyc);
        run<3>("6 MFMA + 2 global loads + waitcnt", threads, src, out, cyc);
    }
    return 0;
}

// Micro-benchmark (GPU box): what does the vector-memory path of ONE CU sustain for the operand fetch patterns of the decode-order
// convolution, 8 waves per CU (two per SIMD), data resident in the XCD's L2 or beyond it (Infinity Cache)?
//   P0  global_load_dword      256 B contiguous per wave instruction            (one weight tile)
//   P1  global_load_dwordx2    4 planes x 128 B per wave instruction            (the B operand of k_cconv16dc: 4 channel planes x 16 lanes x 2 rows)
//   P2  global_load_dwordx4    1 KB contiguous                                  (four weight tiles)
//   P3  global_load_lds_dwordx4 1 KB contiguous -> LDS                          (LDS-DMA of packed weights)
//   P4  global_load_lds_dwordx4 4 planes x 256 B -> LDS                         (LDS-DMA of a B operand for both row halves)
//   P5  global_load_dwordx4    4 planes x 256 B                                 (the same gather into registers)
// Prints bytes per clock and CU (2.4 GHz nominal) and ns per wave instruction and CU.
//   hipcc --offload-arch=gfx950 -O3 vmem_rate.hip -o /tmp/vmem_rate && /tmp/vmem_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define PLANE 17408                                       // floats between channel planes (not a multiple of the line size)
template <int P>
__global__ __launch_bounds__(512) void k(const float *__restrict__ src, float *__restrict__ out, long window_floats, int iters) {
    __shared__ float lds[36 * 1024];                          // 144 KB: one workgroup per CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xcd = blockIdx.x & 7;
    // every XCD walks a window of its own (window_floats), every wave a pseudo-random sequence of 4 KB-aligned spots inside it
    const float *base = src + (long)xcd * window_floats;
    unsigned pos = __builtin_amdgcn_readfirstlane((blockIdx.x * 8 + wave) * 2654435761u);
    const unsigned gmask = (unsigned)(window_floats / 1024) / 2 - 1;       // (power of two; the upper half leaves room for the 4 planes)
    float acc = 0.f;
    float *my = lds + wave * 4096;                            // 16 KB of LDS per wave: 16 DMA slots
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            pos = pos * 1664525u + 1013904223u;
            const long spot = (long)((pos >> 8) & gmask) * 1024;               // 4 KB granules
            const float *p = base + spot;
            if constexpr (P == 0) acc += p[lane];
            else if constexpr (P == 1) { const f2 v = *(const f2 *)(p + (lane >> 4) * PLANE + (lane & 15) * 2 + 2); acc += v.x + v.y; }
            else if constexpr (P == 2) { const f4 v = *(const f4 *)(p + lane * 4); acc += v.x + v.y + v.z + v.w; }
            else if constexpr (P == 3) __builtin_amdgcn_global_load_lds(p + lane * 4, (lds_void *)(my + j * 256), 16, 0, 0);
            else if constexpr (P == 4) __builtin_amdgcn_global_load_lds(p + (lane >> 4) * PLANE + (lane & 15) * 4 + 2, (lds_void *)(my + j * 256), 16, 0, 0);
            else { const f4 v = *(const f4 *)(p + (lane >> 4) * PLANE + (lane & 15) * 4); acc += v.x + v.y + v.z + v.w; }
        }
        if constexpr (P == 3 || P == 4) { __builtin_amdgcn_s_waitcnt(0x0f70); acc += my[lane + (i & 7) * 256]; }   // vmcnt(0)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int P>
static void run(const char *what, int bytes_per_instr, const float *src, float *out, long window_floats) {
    const int iters = 400;
    hipLaunchKernelGGL(k<P>, dim3(256), dim3(512), 0, 0, src, out, window_floats, 20);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<P>, dim3(256), dim3(512), 0, 0, src, out, window_floats, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = 8.0 * iters * 8, ns = ms * 1e6;
    printf("  %-58s %6.1f ns per wave instruction and CU  %6.1f B/clk/CU  (%5.2f TB/s chip)\n", what, ns / instr_per_cu,
           instr_per_cu * bytes_per_instr / (ns * 2.4), 256.0 * instr_per_cu * bytes_per_instr / ns * 1e-3);
    fflush(stdout);
}
int main() {
    const long total = 8L * 16 * 1024 * 1024;                 // floats: 8 windows of up to 64 MB
    float *src, *out;
    if (hipMalloc(&src, total * 4) != hipSuccess || hipMalloc(&out, 256 * 512 * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(src, 0, total * 4);
    for (long win : {256L * 1024, 4L * 1024 * 1024}) {         // 1 MB per XCD (L2-resident), 16 MB per XCD (128 MB: Infinity Cache)
        printf("window %ld MB per XCD\n", win * 4 >> 20);
        run<0>("P0 global_load_dword   256 B contiguous", 256, src, out, win);
        run<1>("P1 global_load_dwordx2 4 planes x 128 B", 512, src, out, win);
        run<2>("P2 global_load_dwordx4 1 KB contiguous", 1024, src, out, win);
        run<5>("P5 global_load_dwordx4 4 planes x 256 B", 1024, src, out, win);
        run<3>("P3 global_load_lds_dwordx4 1 KB contiguous", 1024, src, out, win);
        run<4>("P4 global_load_lds_dwordx4 4 planes x 256 B (+8 B skew)", 1024, src, out, win);
    }
    return 0;
}

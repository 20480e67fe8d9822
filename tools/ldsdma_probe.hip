// ldsdma_probe.hip -- one-off probe: semantics of __builtin_amdgcn_global_load_lds on gfx950
// (per-lane global address, wave-uniform LDS base, destination = base + lane*size), dword and dwordx4 forms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef OFF
#define OFF 0
#endif
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const float *src, const int *idx, float *out, float *out4) {
    __shared__ float buf[256];
    __shared__ float buf4[1024];
    int l = threadIdx.x;
    // each lane fetches src[idx[l]] ; lands at buf[64*wave + lane]
    const float *g = src + idx[l];
    __builtin_amdgcn_global_load_lds(g, (lds_void *)(buf + (l & ~63)), 4, 0, 0);
    const float *g4 = src + 4 * (63 - (l & 63)) + 1024 * (l >> 6) + OFF;
    __builtin_amdgcn_global_load_lds(g4, (lds_void *)(buf4 + 4 * (l & ~63)), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) lgkmcnt(0) expcnt(0)
    __syncthreads();
    out[l] = buf[l];
    for (int i = 0; i < 4; ++i) out4[l * 4 + i] = buf4[l * 4 + i];
}
int main() {
    const int N = 4096 + 16;
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = i * 0.5f;
    std::vector<int> hi(256);
    for (int i = 0; i < 256; ++i) hi[i] = (i * 37 + 11) % N;
    float *s, *o, *o4; int *ix;
    (void)hipMalloc(&s, N * 4); (void)hipMalloc(&o, 1024); (void)hipMalloc(&o4, 4096); (void)hipMalloc(&ix, 1024);
    (void)hipMemcpy(s, h.data(), N * 4, hipMemcpyHostToDevice); (void)hipMemcpy(ix, hi.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 1, 256, 0, 0, s, ix, o, o4);
    std::vector<float> r(256), r4(1024);
    (void)hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(r4.data(), o4, 4096, hipMemcpyDeviceToHost);
    int bad = 0, bad4 = 0;
    for (int i = 0; i < 256; ++i) bad += r[i] != h[hi[i]];
    for (int l = 0; l < 256; ++l) for (int i = 0; i < 4; ++i) bad4 += r4[l * 4 + i] != h[4 * (63 - (l & 63)) + 1024 * (l >> 6) + i + OFF];
    printf("global_load_lds dword mismatches %d/256, dwordx4 mismatches %d/1024 (hipGetLastError=%d)\n", bad, bad4, (int)hipGetLastError());
    return 0;
}

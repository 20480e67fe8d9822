// What do large kernel arguments cost?  (round 5: a decode launch whose argument struct grew from 124 bytes to 1.8 KB made the three-stream step 4.6 % slower although the
// kernel never read the new bytes.)  K threads, one stream each, launch N kernels of ~T us with an argument struct of S bytes back to back; us per launch and stream.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/kernarg_size.hip -o /tmp/kernarg_size -lpthread && /tmp/kernarg_size
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
template <int S> struct Args { int spin; float *out; char pad[S - 16]; };
template <int S> __global__ void k(Args<S> a) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)a.spin) {}
    if (a.out && threadIdx.x == 0 && blockIdx.x == 0) a.out[0] = (float)a.pad[S - 17];
}
template <int S> static double run(int nthreads, int n, int spin, int blocks) {
    std::vector<std::thread> th;
    std::vector<hipStream_t> st(nthreads);
    for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < nthreads; ++t) th.emplace_back([&, t] {
        Args<S> a; a.spin = spin; a.out = nullptr;
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k<S>, dim3(blocks), dim3(256), 0, st[t], a);
        (void)hipStreamSynchronize(st[t]);
    });
    for (auto &x : th) x.join();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    for (auto &s : st) (void)hipStreamDestroy(s);
    return us / n;
}
template <int S> static void row(int spin, int blocks) {
    run<S>(1, 200, spin, blocks);
    const double a = run<S>(1, 4000, spin, blocks), b = run<S>(3, 4000, spin, blocks);
    printf("args %5d B   spin %6d ticks x %3d blocks:   1 stream %7.2f us per launch   3 streams (3 threads) %7.2f us per launch and stream\n", S, spin, blocks, a, b);
    fflush(stdout);
}
int main() {
    for (int spin : {0, 1000}) for (int blocks : {1, 256}) {          // s_memtime ticks at 100 MHz: 1000 ticks = 10 us
        row<64>(spin, blocks); row<128>(spin, blocks); row<192>(spin, blocks); row<256>(spin, blocks); row<272>(spin, blocks); row<320>(spin, blocks);
        row<512>(spin, blocks); row<1024>(spin, blocks); row<2048>(spin, blocks); row<4000>(spin, blocks);
    }
    return 0;
}

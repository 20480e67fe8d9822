// What do kernel-argument bytes cost under the bench's launch pattern?  (companion of kernarg_size.hip, which launches from one host thread per stream.)
// ONE host thread enqueues N kernels round-robin into K streams -- 256 workgroups of 768 threads that spin T us and READ every argument word (so that the
// scalar loads of the whole struct are really issued) -- with an argument struct of S bytes; wall time per launch and stream, and the host's enqueue time.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/kernarg_streams.hip -o /tmp/kernarg_streams && GPU_MAX_HW_QUEUES=8 /tmp/kernarg_streams
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int S> struct Args { int spin; int touch; float *out; int pad[(S - 16) / 4]; };
template <int S> __global__ __launch_bounds__(768) void k(Args<S> a) {
    int acc = 0;
    if (a.touch)
#pragma unroll
        for (int i = 0; i < (S - 16) / 4; ++i) acc += a.pad[i];          // uniform: s_load of the whole struct
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)a.spin) {}
    if (a.out && threadIdx.x == 0 && blockIdx.x == 0) a.out[0] = (float)acc;
}
template <int S> static void run(int K, int n, int spin, int touch) {
    std::vector<hipStream_t> st(K);
    for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    Args<S> a{};
    a.spin = spin; a.touch = touch; a.out = nullptr;
    for (int i = 0; i < 50; ++i) for (int t = 0; t < K; ++t) hipLaunchKernelGGL(k<S>, dim3(256), dim3(768), 0, st[t], a);
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) for (int t = 0; t < K; ++t) hipLaunchKernelGGL(k<S>, dim3(256), dim3(768), 0, st[t], a);
    const auto t1 = std::chrono::steady_clock::now();
    (void)hipDeviceSynchronize();
    const auto t2 = std::chrono::steady_clock::now();
    const double enq = std::chrono::duration<double, std::micro>(t1 - t0).count() / (n * K), wall = std::chrono::duration<double, std::micro>(t2 - t0).count() / n;
    printf("streams %d  args %5d B  touch %d  spin %5.1f us:  wall per launch and stream %7.2f us   host enqueue %5.2f us per launch\n", K, S, touch, spin / 100.0, wall, enq);
    fflush(stdout);
    for (auto &s : st) (void)hipStreamDestroy(s);
}
int main() {
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    printf("GPU_MAX_HW_QUEUES=%s HIP_FORCE_DEV_KERNARG=%s\n", q ? q : "(default 4)", getenv("HIP_FORCE_DEV_KERNARG") ? getenv("HIP_FORCE_DEV_KERNARG") : "(unset)");
    for (int spin : {0, 2000})                                              // s_memtime ticks at 100 MHz: 2000 ticks = 20 us, a decode-order launch of one image
        for (int K : {1, 3, 6})
            for (int touch : {0, 1}) {
                run<96>(K, 2000, spin, touch); run<352>(K, 2000, spin, touch); run<1808>(K, 2000, spin, touch);
            }
    return 0;
}

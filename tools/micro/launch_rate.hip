// Micro-benchmark (development aid): aggregate kernel dispatch rate of the GPU with S streams, one host thread per stream,
// each stream a chain of dependent (in-order) tiny kernels.  Is a frame of ~18 launches bound by dispatch?
// build: hipcc --offload-arch=gfx950 -O3 -pthread -o launch_rate launch_rate.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void k_tiny(int *p) { if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = 1; }
__global__ void k_wide(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }
int main()
{
    const int K = 4000;
    float *buf; hipMalloc(&buf, 8 << 20);            // 2 M floats: 6 streams x 262144 fit
    for (int wide = 0; wide < 2; wide++)
        for (int S : {1, 2, 3, 4, 6}) {
            std::vector<hipStream_t> st(S);
            for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            auto run = [&](int n) {
                std::vector<std::thread> th;
                for (int t = 0; t < S; t++) th.emplace_back([&, t] {
                    hipSetDevice(0);
                    for (int i = 0; i < n; i++) {
                        if (wide) hipLaunchKernelGGL(k_wide, dim3(1024), dim3(256), 0, st[t], buf + t * 262144, 262144);
                        else hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st[t], (int *)nullptr);
                    }
                    hipStreamSynchronize(st[t]);
                });
                for (auto &x : th) x.join();
            };
            run(200);
            auto t0 = std::chrono::steady_clock::now();
            run(K);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("%s kernels, %d streams: %.2f us per kernel per stream, %.2f us per kernel overall (%.0f k kernels/s)\n",
                   wide ? "1024-WG" : "1-WG", S, us / K, us / K / S, 1e3 * K * S / us);
            for (auto &s : st) hipStreamDestroy(s);
        }
    return 0;
}

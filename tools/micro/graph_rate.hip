// Micro-benchmark (development aid; VERDICT r2 #7a "measure, don't argue"): a frame is a chain of ~18 short DEPENDENT kernels.
// Does a captured hipGraph run that chain faster than 18 stream launches?  Three forms of the same chain of K kernels (each a
// 64-workgroup kernel that touches 64 KB, like the frame's scans and histograms), on one stream:
//   stream      K hipLaunchKernelGGL per chain (what the library's enqueue threads do)
//   graph       the chain captured once (stream capture), one hipGraphLaunch per chain
//   graph+set   the same with hipGraphExecKernelNodeSetParams on 7 of the K nodes before every launch (a frame's uniforms,
//               grids that follow the adaptive share, ring events: what the library would have to update per frame)
// Reported: GPU time per chain (HIP events around 200 back-to-back chains, so the host is never the limit: the stream is kept
// full) and host time per chain (how long the enqueuing thread is busy).
// build: hipcc --offload-arch=gfx950 -O3 -o graph_rate graph_rate.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
// spin: the kernel additionally waits ~spin x 64 clocks, so that a chain of them is bound by the GPU (kernel time + the gap between
// dependent kernels), not by the host thread that launches them
__global__ void k_step(float *p, int n, float a, int spin)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < spin; k++) __builtin_amdgcn_s_sleep(1);
    if (i < n) p[i] = p[i] * a + 1.0f;
}
int main()
{
    float *buf; CK(hipMalloc(&buf, 1 << 20));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n = 16384, CH = 200;
    for (int spin : {0, 150}) for (int K : {9, 18}) {
        printf("-- kernels of %s\n", spin ? "~4 us (GPU-bound chain)" : "~0 us (host-bound chain)");
        // ---- stream launches
        auto chain_stream = [&] { for (int k = 0; k < K; k++) hipLaunchKernelGGL(k_step, dim3(64), dim3(256), 0, st, buf, n, 1.0f + 1e-6f * k, spin); };
        for (int i = 0; i < 20; i++) chain_stream();
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < CH; i++) chain_stream();
        CK(hipEventRecord(e1, st));
        const double host_s = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / CH;
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("K=%2d stream     : GPU %.1f us per chain (%.2f us per kernel), host %.1f us per chain\n", K, ms * 1e3 / CH, ms * 1e3 / CH / K, host_s);
        // ---- captured graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        chain_stream();
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
        std::vector<hipGraphNode_t> nodes(nn); CK(hipGraphGetNodes(g, nodes.data(), &nn));
        for (int i = 0; i < 20; i++) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        for (int upd = 0; upd < 2; upd++) {
            float a = 1.0f; int nv = n; float *bp = buf;
            int sp = spin;
            void *args[4] = { &bp, &nv, &a, &sp };
            t0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < CH; i++) {
                if (upd) for (size_t k = 0; k < nn && k < 7; k++) {
                    hipKernelNodeParams p; memset(&p, 0, sizeof p);
                    a = 1.0f + 1e-6f * (float)(i + (int)k);
                    p.func = (void *)k_step; p.gridDim = dim3(64); p.blockDim = dim3(256); p.kernelParams = args;
                    CK(hipGraphExecKernelNodeSetParams(ge, nodes[k], &p));
                }
                CK(hipGraphLaunch(ge, st));
            }
            CK(hipEventRecord(e1, st));
            const double host_g = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / CH;
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("K=%2d graph%s: GPU %.1f us per chain (%.2f us per kernel), host %.1f us per chain\n", K, upd ? "+set7 " : "      ", ms * 1e3 / CH,
                   ms * 1e3 / CH / K, host_g);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}

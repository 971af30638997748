// Micro-benchmark (development aid): throughput of global atomicAdd on gfx950 as a function of how many distinct words the
// atomics land on -- the figure that decides whether a radix histogram can be accumulated with global atomics by the kernel
// that PRODUCES the keys (no separate histogram launch) or must stay an LDS histogram + one row per chunk.
//   mode 0: non-returning agent-scope atomicAdd      mode 1: returning agent-scope atomicAdd
//   mode 2: non-returning workgroup-scope atomicAdd (performed in the issuing XCD's L2; coherent inside one XCD only)
// build: hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void k_atomics(uint32_t *tab, uint32_t words, uint32_t per_thread, uint32_t *sink)
{
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per_thread; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t w = (x >> 8) % words;
        if (MODE == 0) atomicAdd(&tab[w], 1u);
        else if (MODE == 1) acc += atomicAdd(&tab[w], 1u);
        else __hip_atomic_fetch_add(&tab[w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (MODE == 1 && acc == 0xFFFFFFFFu) *sink = acc;
}

int main()
{
    uint32_t *tab, *sink;
    hipMalloc(&tab, 64u << 20); hipMalloc(&sink, 4);
    hipMemset(tab, 0, 64u << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t grid = 2048, per = 64;                    // 33.5 M atomics per launch
    for (int mode = 0; mode < 3; mode++)
        for (uint32_t words : {1u, 256u, 4096u, 65536u, 1u << 20, 1u << 24}) {
            auto launch = [&] {
                if (mode == 0) hipLaunchKernelGGL(k_atomics<0>, dim3(grid), dim3(256), 0, 0, tab, words, per, sink);
                else if (mode == 1) hipLaunchKernelGGL(k_atomics<1>, dim3(grid), dim3(256), 0, 0, tab, words, per, sink);
                else hipLaunchKernelGGL(k_atomics<2>, dim3(grid), dim3(256), 0, 0, tab, words, per, sink);
            };
            if (words == 1u && mode != 2) {                       // one word serialises at ~11 ns: keep that case short
                hipLaunchKernelGGL(k_atomics<0>, dim3(8), dim3(256), 0, 0, tab, 1u, 8u, sink);
            }
            const uint32_t g = (words == 1u) ? 64u : grid;
            auto launch_g = [&](uint32_t gg) {
                if (mode == 0) hipLaunchKernelGGL(k_atomics<0>, dim3(gg), dim3(256), 0, 0, tab, words, per, sink);
                else if (mode == 1) hipLaunchKernelGGL(k_atomics<1>, dim3(gg), dim3(256), 0, 0, tab, words, per, sink);
                else hipLaunchKernelGGL(k_atomics<2>, dim3(gg), dim3(256), 0, 0, tab, words, per, sink);
            };
            (void)launch;
            launch_g(g);
            hipEventRecord(e0, 0);
            launch_g(g);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)g * 256 * per;
            printf("mode %d (%s)  words %9u : %8.3f ms for %.1f M atomics = %8.2f G atomics/s\n", mode,
                   mode == 0 ? "agent, no return" : mode == 1 ? "agent, returning" : "workgroup scope", words, ms, n / 1e6, n / ms / 1e6);
        }
    return 0;
}

// tools/micro/cumask_probe.hip -- does a stream created with hipExtStreamCreateWithCUMask run its kernels on the masked compute units only?
// A kernel of 4096 workgroups x 256 threads of pure VALU work is timed on a plain stream and on streams whose masks keep 128, 64 and 32 of
// the 256 CUs (bit i of the mask = CU i; two layouts: the first n bits, and every (256 / n)-th bit).  If the masks are honoured the time
// grows like 256 / n.   hipcc --offload-arch=gfx950 -O3 tools/micro/cumask_probe.hip -o tools/micro/cumask_probe && tools/micro/cumask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
__global__ void burn(float *out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; i++) { a = a * b + 0.5f; b = b * 0.99999f + 1e-6f; }
    if (a == 123.456f) out[0] = a + b;
}
static float run(hipStream_t st, float *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    burn<<<4096, 256, 0, st>>>(d, 2000);
    hipEventRecord(e0, st);
    for (int k = 0; k < 5; k++) burn<<<4096, 256, 0, st>>>(d, 2000);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / 5.0f * 1e3f;
}
int main()
{
    float *d; hipMalloc(&d, 64);
    hipStream_t s0; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    printf("plain stream: %.1f us\n", run(s0, d));
    for (int n = 128; n >= 32; n /= 2)
        for (int layout = 0; layout < 2; layout++) {
            uint32_t m[8]; memset(m, 0, sizeof m);
            for (int i = 0; i < 256; i++) { const bool on = layout == 0 ? i < n : (i % (256 / n)) == 0; if (on) m[i / 32] |= 1u << (i % 32); }
            hipStream_t s; const hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m);
            if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed: %s\n", hipGetErrorString(e)); continue; }
            uint32_t back[8] = { 0 }; const hipError_t e2 = hipExtStreamGetCUMask(s, 8, back);
            int bits = 0; for (int i = 0; i < 8; i++) bits += __builtin_popcount(back[i]);
            printf("%3d CUs (%s): %.1f us   [GetCUMask: %s, %d bits]\n", n, layout ? "strided" : "first n", run(s, d), hipGetErrorString(e2), bits);
            hipStreamDestroy(s);
        }
    return 0;
}

// Micro-benchmark (development aid): issue rate of v_fma_f32, v_pk_fma_f32, v_exp_f32, v_cndmask on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; prints wave-instructions per cycle per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITER 4096
template <int MODE> __global__ __launch_bounds__(64) void k(float *out, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const f2 s2 = {s, s};
    for (int i = 0; i < ITER; i++) {
        if (MODE == 0) {
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                         "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 1) {
            asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                         "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(s2));
        } else if (MODE == 2) {
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 3) {
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "vcc");
        } else if (MODE == 4) {
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(s2));
        } else {
            asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.y + p3.y + p4.x + p5.x + p6.y + p7.y;
}
template <int MODE> void run(const char *name, float *out, int wavesPerSimd)
{
    const int blocks = 256 * 4 * wavesPerSimd;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 1.0f);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)ITER * 8 * wavesPerSimd;       // wave-instructions per SIMD
    printf("%-22s waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, wavesPerSimd, ms, ms * 1e-3 * 2.4e9 / instr);
}
int main()
{
    float *out; hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    for (int w : {1, 2, 8}) {
        run<0>("v_fma_f32", out, w); run<1>("v_pk_fma_f32", out, w); run<2>("v_exp_f32", out, w);
        run<3>("v_cmp+v_cndmask", out, w); run<4>("v_pk_mul/add_f32", out, w); run<5>("v_mul/add_f32", out, w);
    }
    return 0;
}

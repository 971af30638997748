// gs_ply.hip -- processPlyBuffer (reference index.js:600-745) on the GPU for the 6 M - 20 M row files:
//
//   host      header parse + property resolution (gs_ply_plan, gs_host.cpp): same error messages, same order
//   H2D       the raw vertex block, once (n x row_bytes; 248 B/row in the INRIA layout)
//   k_ply_keys   one thread per row: importance = exp(s0)exp(s1)exp(s2) * sigmoid(opacity) in f64 (gs_ply.h, bit-equal
//                to the host converter), f32-rounded like the reference's Float32Array, as a descending radix key
//   radix x4     stable 8-bit LSD passes over (key, row) records (gs_prims): ties keep file order = the reference's
//                stable comparator sort
//   k_ply_rows   one thread per OUTPUT row: gather the source row, emit the 32-byte .splat row
//
// The rows stay in HBM and go straight into the pack kernel (gs_load_ply), or are copied back for processPlyBuffer
// (gs_ply_to_splat_gpu).  The kernels are byte-gather / f64-ALU work on a one-time path; no LDS tiling is needed.
#include "gs_internal.h"
#include "gs_ply.h"

namespace {

__global__ __launch_bounds__(GS_BLOCK) void k_ply_keys(const uint8_t *__restrict__ data, gsm::PlyLayout L, uint32_t n,
                                                       uint2 *__restrict__ kv, uint32_t *__restrict__ flags)
{
    for (uint32_t i = blockIdx.x * GS_BLOCK + threadIdx.x; i < n; i += gridDim.x * GS_BLOCK) {
        const float imp = gsm::ply_importance(data + (size_t)i * L.row_bytes, L);
        if (imp != imp) atomicOr(flags, 1u);                         // NaN importance: the comparator sort's order is
        kv[i] = make_uint2(gsm::ply_order_key(imp), i);              // engine-defined; the host path handles it
    }
}

__global__ __launch_bounds__(GS_BLOCK) void k_ply_rows(const uint8_t *__restrict__ data, gsm::PlyLayout L, uint32_t n,
                                                       const uint2 *__restrict__ kv, uint4 *__restrict__ rows)
{
    for (uint32_t j = blockIdx.x * GS_BLOCK + threadIdx.x; j < n; j += gridDim.x * GS_BLOCK) {
        const uint32_t r = kv ? kv[j].y : j;                         // no scale_0: importance is all zero -> file order
        uint32_t w[8];
        gsm::ply_row(data + (size_t)r * L.row_bytes, L, w);
        rows[2 * (size_t)j] = make_uint4(w[0], w[1], w[2], w[3]);
        rows[2 * (size_t)j + 1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

}  // namespace

// rows_out: n x 32 B device buffer (caller-allocated).  *had_nan is set when an importance came out NaN (the caller then
// uses the host converter, whose order for that case is the one the tests pin).
int gs_ply_rows_device(gs_ctx *ctx, const uint8_t *host_data, const gsm::PlyLayout &L, size_t n, uint4 *rows_out, bool *had_nan)
{
    *had_nan = false;
    if (!n) return GS_OK;
    if (n > 0x7FFFFFF0ull) { snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "more than 2^31 PLY rows"); return GS_E_BADARG; }
    hipStream_t st = ctx->stream;
    uint8_t *data = nullptr; uint2 *kv_a = nullptr, *kv_b = nullptr; uint32_t *small = nullptr;
    const size_t raw = n * (size_t)L.row_bytes;
    int rc = GS_OK;
    auto cleanup = [&]() { (void)hipFree(data); (void)hipFree(kv_a); (void)hipFree(kv_b); (void)hipFree(small); };
#define PLY_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) { snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "%s failed: %s", #call, hipGetErrorString(_e)); \
                           cleanup(); return _e == hipErrorOutOfMemory ? GS_E_OOM : GS_E_HIP; } } while (0)
    PLY_HIP(hipMalloc(&data, raw + 8));
    PLY_HIP(hipMalloc(&small, 16));
    PLY_HIP(hipMemcpyAsync(data, host_data, raw, hipMemcpyHostToDevice, st));
    const uint32_t n32 = (uint32_t)n;
    const uint32_t host_small[2] = { n32, 0u };                      // [0] = n for the radix kernels, [1] = NaN flag
    PLY_HIP(hipMemcpyAsync(small, host_small, 8, hipMemcpyHostToDevice, st));
    uint32_t g = gs_div_up(n, GS_BLOCK); if (g > 16384) g = 16384;
    const uint2 *order = nullptr;
    if (L.has_scale) {
        PLY_HIP(hipMalloc(&kv_a, n * sizeof(uint2)));
        PLY_HIP(hipMalloc(&kv_b, n * sizeof(uint2)));
        rc = gs_ensure_radix_scratch(ctx, n);
        if (rc != GS_OK) { cleanup(); return rc; }
        hipLaunchKernelGGL(k_ply_keys, dim3(g), dim3(GS_BLOCK), 0, st, data, L, n32, kv_a, small + 1);
        for (int pass = 0; pass < 4 && rc == GS_OK; pass++)
            rc = gs_launch_radix_pass(ctx, (pass & 1) ? kv_b : kv_a, GS_RADIX_PACKED, (pass & 1) ? kv_a : kv_b, GS_RADIX_PACKED, small, n32, n32, 8 * pass, 8);
        if (rc != GS_OK) { cleanup(); return rc; }
        order = kv_a;                                                // 4 passes: back in kv_a
    }
    hipLaunchKernelGGL(k_ply_rows, dim3(g), dim3(GS_BLOCK), 0, st, data, L, n32, order, rows_out);
    PLY_HIP(hipGetLastError());
    uint32_t back[2] = { 0, 0 };
    PLY_HIP(hipMemcpyAsync(back, small, 8, hipMemcpyDeviceToHost, st));
    PLY_HIP(hipStreamSynchronize(st));
#undef PLY_HIP
    *had_nan = back[1] != 0;
    cleanup();
    return GS_OK;
}

// gs_pack.hip -- pushDataBuffer's pack loop (index.js:343-402) as one HIP kernel: 32 B .splat row in,
// one interleaved 32 B record (16 B centre/scale + 16 B covariance/colour) + 16 B sort row out, one thread per splat, all 16-byte
// coalesced accesses.  The covariance is built in f64 in three.js' operation order (gs_device_math.h) so the
// int16 quantisation, including the parseInt exponent-form quirk, is bit-identical to the reference.
#include "gs_internal.h"

namespace {

__global__ __launch_bounds__(GS_BLOCK) void k_pack(const uint4 *__restrict__ rows, uint32_t nrows, const double *__restrict__ pow10tab,
                                                   uint4 *__restrict__ splat, float4 *__restrict__ sort_rows, float *__restrict__ bound_r)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nrows; i += gridDim.x * blockDim.x) {
        const uint4 a = rows[2 * i], b = rows[2 * i + 1];
        const uint32_t w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
        gsm::PackOut o;
        gsm::pack_row(w, pow10tab, o);
        splat[2 * i] = make_uint4(__float_as_uint(o.cs[0]), __float_as_uint(o.cs[1]), __float_as_uint(o.cs[2]), __float_as_uint(o.cs[3]));
        splat[2 * i + 1] = make_uint4(o.cc[0], o.cc[1], o.cc[2], o.cc[3]);
        sort_rows[i] = make_float4(o.sort_row[0], o.sort_row[1], o.sort_row[2], o.sort_row[3]);
        // largest eigenvalue of the dequantised covariance <= largest absolute row sum <= 3 max|Sigma_ij| = 3 * 32767 * cs[3]
        const float mx = o.cs[3] * 32767.0f;
        bound_r[i] = (mx == mx && mx < 3.0e37f) ? sqrtf(3.0f * mx) * 1.001f : INFINITY;
    }
}

}  // namespace

int gs_launch_pack(gs_ctx *ctx, const uint4 *rows_dev, size_t first, size_t nrows)
{
    if (!nrows) return GS_OK;
    uint32_t g = gs_div_up(nrows, GS_BLOCK); if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_pack, dim3(g), dim3(GS_BLOCK), 0, ctx->stream, rows_dev, (uint32_t)nrows, ctx->pow10tab,
                       ctx->splat + 2 * first, ctx->sort_rows + first, ctx->bound_r + first);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

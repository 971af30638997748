// gs_ply.h -- processPlyBuffer's per-row arithmetic (reference index.js:653-742), shared by the host converter
// (gs_host.cpp) and the HIP converter (gs_ply.hip) so that both produce the same bytes by construction.
//
// Everything here is IEEE f64 built from + - * / sqrt only (all correctly rounded on the host and on gfx950; the
// library is compiled with -ffp-contract=off), including Math.exp, which is restated below exactly as the
// reference's JavaScript engine evaluates it.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GS_PLY_HD __host__ __device__ inline
#else
#define GS_PLY_HD inline
#endif

namespace gsm {

// Math.exp.  THIRD-PARTY arithmetic that is not under /root/reference: the reference calls Math.exp (index.js:659-662,
// 700-702, 722), which V8 (node 12 / Chromium) implements as base::ieee754::exp = Sun fdlibm 5.3 e_exp.c plus one
// special case (exp(1) returns the constant E).  <1 ulp, but not correctly rounded: libm's exp() differs from it in the
// last bit for ~10 % of arguments, which would leak into the f32 scales and the importance order.  Restated from the
// published algorithm: x = k*ln2 + r (two-word ln2), degree-5 minimax polynomial of r*(e^r+1)/(e^r-1), scale by 2^k.
// Pinned bit for bit by tests/golden/math_exp.bin (8200 arguments evaluated by Math.exp under node).
GS_PLY_HD double js_exp(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, inv_ln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05;
    const double P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    union { double d; uint64_t u; } c; c.d = x;
    const bool neg = (c.u >> 63) != 0;
    const uint32_t hx = (uint32_t)(c.u >> 32) & 0x7fffffffu;       // high word of |x|
    if (hx >= 0x40862E42u) {                                        // |x| >= 709.78...
        if (hx >= 0x7ff00000u) {
            if (c.u & 0x000fffffffffffffull) return x + x;          // NaN
            return neg ? 0.0 : x;                                   // exp(-inf) = 0, exp(+inf) = +inf
        }
        if (x > 7.09782712893383973096e+02) { c.u = 0x7ff0000000000000ull; return c.d; }   // overflow -> +inf
        if (x < -7.45133219101941108420e+02) return 0.0;            // underflow
    }
    double hi = 0.0, lo = 0.0;
    int k = 0;
    if (hx > 0x3fd62e42u) {                                         // |x| > 0.5 ln2
        if (hx < 0x3FF0A2B2u) {                                     // and |x| < 1.5 ln2
            if (x == 1.0) return 2.718281828459045;                 // V8's special case
            hi = neg ? x + ln2_hi : x - ln2_hi;
            lo = neg ? -ln2_lo : ln2_lo;
            k = neg ? -1 : 1;
        } else {
            k = (int)(inv_ln2 * x + (neg ? -0.5 : 0.5));
            const double t = (double)k;
            hi = x - t * ln2_hi;                                    // t*ln2_hi is exact
            lo = t * ln2_lo;
        }
        x = hi - lo;
    } else if (hx < 0x3e300000u) {                                  // |x| < 2^-28
        return 1.0 + x;
    }
    const double t = x * x;
    const double p = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * p) / (p - 2.0) - x);
    c.d = 1.0 - ((lo - (x * p) / (2.0 - p)) - hi);
    if (k >= -1021) { c.u += (uint64_t)(int64_t)k << 52; return c.d; }             // * 2^k through the exponent field
    c.u += (uint64_t)(int64_t)(k + 1000) << 52;
    return c.d * 9.33263618503218878990e-302;                       // * 2^-1000: gradual underflow
}

// Uint8ClampedArray element store (index.js:671-676 views): clamp to [0,255], round half to even, NaN -> 0
GS_PLY_HD uint8_t clamped_u8(double v)
{
    if (!(v > 0)) return 0;
    if (v >= 255) return 255;
    const double f = floor(v), d = v - f;
    if (d > 0.5) return (uint8_t)(f + 1);
    if (d < 0.5) return (uint8_t)f;
    return (uint8_t)((((int)f) & 1) ? f + 1 : f);
}

// property types of the reference's TYPE_MAP (index.js:613-621); anything else is read with getInt8 (index.js:628)
enum PlyType { PLY_F64 = 0, PLY_I32, PLY_U32, PLY_F32, PLY_I16, PLY_U16, PLY_U8, PLY_I8 };

// the properties processPlyBuffer reads, resolved once from the header
enum PlyProp { PP_X = 0, PP_Y, PP_Z, PP_S0, PP_S1, PP_S2, PP_R0, PP_R1, PP_R2, PP_R3, PP_C0, PP_C1, PP_C2, PP_OPACITY, PP_COUNT };
struct PlyLayout {
    uint32_t row_bytes;
    uint32_t offset[PP_COUNT];
    uint8_t type[PP_COUNT];
    uint8_t has_scale;       // scale_0 present: gaussian rows (scale/rot read), else points with default scale/rotation
    uint8_t has_dc;          // f_dc_0 present: SH DC colour, else red/green/blue
    uint8_t has_opacity;
};

// DataView.get*(offset, littleEndian = true) at any byte alignment
GS_PLY_HD double ply_read(const uint8_t *p, int type)
{
    switch (type) {
    case PLY_F64: { uint64_t u = 0; for (int i = 7; i >= 0; i--) u = (u << 8) | p[i]; union { uint64_t u; double d; } c; c.u = u; return c.d; }
    case PLY_I32: { const uint32_t u = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); return (double)(int32_t)u; }
    case PLY_U32: { const uint32_t u = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); return (double)u; }
    case PLY_F32: { union { uint32_t u; float f; } c; c.u = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); return (double)c.f; }
    case PLY_I16: return (double)(int16_t)(uint16_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8));
    case PLY_U16: return (double)(uint16_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8));
    case PLY_U8: return (double)p[0];
    default: return (double)(int8_t)p[0];
    }
}
GS_PLY_HD double ply_attr(const uint8_t *row, const PlyLayout &L, int prop) { return ply_read(row + L.offset[prop], L.type[prop]); }

// sizeList[row] (index.js:657-663): exp(s0)*exp(s1)*exp(s2) * sigmoid(opacity), stored in a Float32Array.
// Only called when has_scale (the list stays 0 otherwise, index.js:656).
GS_PLY_HD float ply_importance(const uint8_t *row, const PlyLayout &L)
{
    const double size = js_exp(ply_attr(row, L, PP_S0)) * js_exp(ply_attr(row, L, PP_S1)) * js_exp(ply_attr(row, L, PP_S2));
    const double opacity = 1 / (1 + js_exp(-ply_attr(row, L, PP_OPACITY)));
    return (float)(size * opacity);
}

// one 32-byte .splat row (index.js:680-742): position f32 x3, scale f32 x3, RGBA u8, rotation u8 x4
GS_PLY_HD void ply_row(const uint8_t *row, const PlyLayout &L, uint32_t out[8])
{
    float f[6];
    uint8_t b[8];
    if (L.has_scale) {
        const double q0 = ply_attr(row, L, PP_R0), q1 = ply_attr(row, L, PP_R1), q2 = ply_attr(row, L, PP_R2), q3 = ply_attr(row, L, PP_R3);
        const double qlen = sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
        b[4] = clamped_u8((q0 / qlen) * 128 + 128); b[5] = clamped_u8((q1 / qlen) * 128 + 128);
        b[6] = clamped_u8((q2 / qlen) * 128 + 128); b[7] = clamped_u8((q3 / qlen) * 128 + 128);
        f[3] = (float)js_exp(ply_attr(row, L, PP_S0)); f[4] = (float)js_exp(ply_attr(row, L, PP_S1)); f[5] = (float)js_exp(ply_attr(row, L, PP_S2));
    } else {
        f[3] = f[4] = f[5] = (float)0.01;
        b[4] = 255; b[5] = b[6] = b[7] = 0;
    }
    f[0] = (float)ply_attr(row, L, PP_X); f[1] = (float)ply_attr(row, L, PP_Y); f[2] = (float)ply_attr(row, L, PP_Z);
    if (L.has_dc) {
        const double SH_C0 = 0.28209479177387814;
        for (int k = 0; k < 3; k++) b[k] = clamped_u8((0.5 + SH_C0 * ply_attr(row, L, PP_C0 + k)) * 255);
    } else {
        for (int k = 0; k < 3; k++) b[k] = clamped_u8(ply_attr(row, L, PP_C0 + k));
    }
    b[3] = L.has_opacity ? clamped_u8((1 / (1 + js_exp(-ply_attr(row, L, PP_OPACITY)))) * 255) : 255;
    for (int k = 0; k < 6; k++) { union { float f; uint32_t u; } c; c.f = f[k]; out[k] = c.u; }
    out[6] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
    out[7] = (uint32_t)b[4] | ((uint32_t)b[5] << 8) | ((uint32_t)b[6] << 16) | ((uint32_t)b[7] << 24);
}

// Descending order on a non-negative, non-NaN f32 key as an ascending u32 radix key (ties keep the input order in a
// stable sort = the comparator sort `sizeList[a] - sizeList[b]`, index.js:668, which returns 0 for equal keys).
GS_PLY_HD uint32_t ply_order_key(float importance)
{
    union { float f; uint32_t u; } c; c.f = importance;
    const uint32_t ordered = (c.u >> 31) ? ~c.u : (c.u | 0x80000000u);
    return ~ordered;
}

}  // namespace gsm

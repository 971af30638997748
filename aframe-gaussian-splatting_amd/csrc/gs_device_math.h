// gs_device_math.h -- per-splat arithmetic of the hot path, written once and
// compiled for the GPU (hipcc, __device__) -- the only place the product runs
// it -- and, for tests/host_check.cpp only, for the host so the formulas can be
// compared with the oracle without a GPU.
//
// Contracts:
//  * sort key arithmetic is IEEE f64, left-to-right, NO fused multiply-add:
//    the whole library is built with -ffp-contract=off and this header never
//    calls fma() on the f64 path.
//  * projection is fp32 with a fixed operation order, again un-fused, so that
//    the projected record is a deterministic function of its inputs and the
//    coverage decision |p|^2 <= 4 can be reproduced exactly by the checker.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define GS_HD __host__ __device__ __forceinline__
#else
#define GS_HD static inline
#endif

namespace gsm {

// ---------------------------------------------------------------- XCD-aware chunk order (see gs_internal.h)
// virtual workgroup index v (grid a multiple of 8, grid-strided over [0, 8*ceil(nchunks/8))) -> chunk: XCD v % 8 takes the
// (v % 8)-th contiguous eighth of the chunks.  Every chunk is produced exactly once; padding slots return false.
GS_HD bool xcd_chunk(uint32_t v, uint32_t nchunks, uint32_t &chunk)
{
    const uint32_t per = (nchunks + 7u) >> 3;
    chunk = (v & 7u) * per + (v >> 3);
    return (v >> 3) < per && chunk < nchunks;
}

// ---------------------------------------------------------------- sort key (index.js:507-561)

// ECMAScript ToInt32, the `|0` at index.js:561.
GS_HD int32_t js_toint32(double d)
{
    // in range: plain truncation (one conversion instruction); everything else (|d| >= 2^31, NaN, Inf) modular, below
    if (d > -2147483649.0 && d < 2147483648.0) return (int32_t)d;
    union { double d; uint64_t u; } c; c.d = d;
    const int e = (int)((c.u >> 52) & 0x7FF);
    if (e == 0x7FF) return 0;                       // NaN, +-Inf -> 0
    if (e < 1023) return 0;                         // |d| < 1 -> trunc = +-0
    const int sh = e - 1075;                        // value = mant * 2^sh
    uint64_t mant = (c.u & 0xFFFFFFFFFFFFFull) | (1ull << 52);
    uint32_t lo;
    if (sh >= 32) lo = 0;
    else if (sh >= 0) lo = (uint32_t)(mant << sh);
    else lo = (uint32_t)(mant >> (-sh));            // -sh <= 52
    if (c.u >> 63) lo = 0u - lo;
    return (int32_t)lo;
}

// view-space depth of worker row (x,y,z,*)  (index.js:519-523): ((v0*x + v1*y) + v2*z) + v3 in f64.
// The uniform arrives widened to f64 (exact): as kernel arguments the doubles sit in scalar registers, where the
// float -> double conversions of 20 uniform values would otherwise occupy 40 vector registers per thread.
GS_HD double view_depth(const double view[4], float x, float y, float z)
{
    return ((view[0] * (double)x + view[1] * (double)y) + view[2] * (double)z) + view[3];
}
GS_HD double view_depth(const float view[4], float x, float y, float z)
{
    const double v[4] = { (double)view[0], (double)view[1], (double)view[2], (double)view[3] };
    return view_depth(v, x, y, z);
}

// box cutout (index.js:492-500, 526-545); c = column-major object->unit-box matrix
GS_HD bool in_cutout(const double *c, float xf, float yf, float zf)
{
    const double x = xf, y = -(double)yf, z = zf;
    const double w = 1.0 / (((c[3] * x + c[7] * y) + c[11] * z) + c[15]);
    const double q0 = (((c[0] * x + c[4] * y) + c[8] * z) + c[12]) * w;
    const double q1 = (((c[1] * x + c[5] * y) + c[9] * z) + c[13]) * w;
    const double q2 = (((c[2] * x + c[6] * y) + c[10] * z) + c[14]) * w;
    return !(q0 < -0.5 || q0 > 0.5 || q1 < -0.5 || q1 > 0.5 || q2 < -0.5 || q2 > 0.5);
}
// the same test for a matrix whose last row is (0, 0, 0, 1) -- checked by the caller --: for finite positions w is exactly 1
// (0 * x is a zero, the sum of zeros and 1 is 1, 1 / 1 is 1) and q * 1 is q: the division and its three products are left out
GS_HD bool in_cutout_affine(const double *c, float xf, float yf, float zf)
{
    const float fin = (xf - xf) + (yf - yf) + (zf - zf);           // 0 iff all three are finite (inf - inf, nan - nan: NaN)
    if (!(fin == 0.0f)) return in_cutout(c, xf, yf, zf);
    const double x = xf, y = -(double)yf, z = zf;
    const double q0 = ((c[0] * x + c[4] * y) + c[8] * z) + c[12];
    const double q1 = ((c[1] * x + c[5] * y) + c[9] * z) + c[13];
    const double q2 = ((c[2] * x + c[6] * y) + c[10] * z) + c[14];
    return !(q0 < -0.5 || q0 > 0.5 || q1 < -0.5 || q1 > 0.5 || q2 < -0.5 || q2 > 0.5);
}
GS_HD bool in_cutout(const float *c, float xf, float yf, float zf)
{
    double cd[16];
    for (int i = 0; i < 16; i++) cd[i] = (double)c[i];
    return in_cutout(cd, xf, yf, zf);
}

// keep test (index.js:548)
GS_HD bool sort_keep(double depth, float size, bool inside)
{
    return depth < 0 && (double)size > -0.0001 * depth && inside;
}

// bucket of a stored (f32-rounded) depth (index.js:558-561); returns -1 for the buckets the
// reference's typed-array writes silently drop (<0 or >65535).
GS_HD int32_t sort_bucket(float depth_f32, double min_depth, double depth_inv)
{
    const int32_t b = js_toint32(((double)depth_f32 - min_depth) * depth_inv);
    return (b >= 0 && b < 65536) ? b : -1;
}

// order-preserving f64 <-> u64 for atomicMin/atomicMax
GS_HD uint64_t f64_to_ordered(double d)
{
    union { double d; uint64_t u; } c; c.d = d;
    return (c.u >> 63) ? ~c.u : (c.u | 0x8000000000000000ull);
}
GS_HD double ordered_to_f64(uint64_t u)
{
    union { double d; uint64_t u; } c;
    c.u = (u >> 63) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u;
    return c.d;
}

// ---------------------------------------------------------------- pack (index.js:343-402)

// JS parseInt(Number) for the value range the pack loop produces.  `pow10tab` holds, for E = 6..330 and
// D = 1..9, the double nearest to D*10^-E at index (E-6)*9 + (D-1) (built on the host with strtod).
// For 0 < |v| < 1e-6 the JS string is in exponent form and parseInt yields its leading digit, which is
// the D of the largest table entry <= |v| (docs/LAB_NOTES.md "Sort bit-exactness notes").
#define GS_POW10_EMIN 6
#define GS_POW10_EMAX 330
GS_HD int32_t js_parse_int(double v, const double *pow10tab)
{
    if (!(v == v)) return 0;                                    // NaN -> Int16Array stores 0
    const double a = fabs(v);
    if (a == 0.0 || a > 1.0e300) return 0;                      // +-Infinity -> NaN -> 0
    if (a >= pow10tab[0]) {                                     // >= 1e-6: "d.ddd" form -> truncation
        return (int32_t)(int16_t)(int64_t)v;                    // Int16Array store is modular
    }
    int E = GS_POW10_EMIN + 1;
    while (E < GS_POW10_EMAX && !(pow10tab[(E - GS_POW10_EMIN) * 9] <= a)) E++;
    int D = 1;
    while (D < 9 && pow10tab[(E - GS_POW10_EMIN) * 9 + D] <= a) D++;
    return v < 0 ? -D : D;
}

struct PackOut {
    float cs[4];        // centerAndScaleData texel (index.js:378-382)
    uint32_t cc[4];     // covAndColorData texel (index.js:384-394)
    float sort_row[4];  // worker matrices elements 12..15 (index.js:396-401)
    float sigma[6];     // f32 of the 6 covariance entries (worker row elements 0,1,2,5,6,10)
};

GS_HD void pack_row(const uint32_t w[8] /* the 32-byte .splat row as 8 LE words */, const double *pow10tab, PackOut &o)
{
    union { uint32_t u; float f; } cv;
    cv.u = w[0]; const float px = cv.f; cv.u = w[1]; const float py = cv.f; cv.u = w[2]; const float pz = cv.f;
    cv.u = w[3]; const double sx = cv.f; cv.u = w[4]; const double sy = cv.f; cv.u = w[5]; const double sz = cv.f;
    const uint32_t rgba = w[6], q = w[7];
    const int b0 = (int)(q & 0xFF), b1 = (int)((q >> 8) & 0xFF), b2 = (int)((q >> 16) & 0xFF), b3 = (int)(q >> 24);
    // THREE.Quaternion(x = b1, y = b2, z = -b3, w = b0), not normalised (index.js:344-349)
    const double qx = (double)(b1 - 128) / 128.0, qy = (double)(b2 - 128) / 128.0;
    const double qz = -(double)(b3 - 128) / 128.0, qw = (double)(b0 - 128) / 128.0;
    const double cx = px, cy = py, cz = -(double)pz;                               // index.js:350-354

    // Matrix4.makeRotationFromQuaternion (three.js compose with unit scale)
    const double x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
    const double xx = qx * x2, xy = qx * y2, xz = qx * z2;
    const double yy = qy * y2, yz = qy * z2, zz = qz * z2;
    const double wx = qw * x2, wy = qw * y2, wz = qw * z2;
    const double r00 = 1 - (yy + zz), r10 = xy + wz, r20 = xz - wy;   // column 0 (te0,te1,te2)
    const double r01 = xy - wz, r11 = 1 - (xx + zz), r21 = yz + wx;   // column 1 (te4,te5,te6)
    const double r02 = xz + wy, r12 = yz - wx, r22 = 1 - (xx + yy);   // column 2 (te8,te9,te10)
    // transpose(); scale(s): M = R^T * diag(s).  M[row][col]:
    const double m00 = r00 * sx, m01 = r10 * sy, m02 = r20 * sz;
    const double m10 = r01 * sx, m11 = r11 * sy, m12 = r21 * sz;
    const double m20 = r02 * sx, m21 = r12 * sy, m22 = r22 * sz;
    // mtx_t = M; mtx = M^T; premultiply: S = M * M^T, element(row,col) = sum_k M[row][k]*M[col][k],
    // summed left to right with a trailing "+ a_r3*b_3c" term that is exactly 0*0 (three.js multiplyMatrices)
#define GS_DOT3(a0, a1, a2, b0, b1, b2) ((((a0) * (b0) + (a1) * (b1)) + (a2) * (b2)) + 0.0 * 0.0)
    const double e0 = GS_DOT3(m00, m01, m02, m00, m01, m02);    // elements[0]  (row0,col0)
    const double e1 = GS_DOT3(m10, m11, m12, m00, m01, m02);    // elements[1]  (row1,col0)
    const double e2 = GS_DOT3(m20, m21, m22, m00, m01, m02);    // elements[2]  (row2,col0)
    const double e5 = GS_DOT3(m10, m11, m12, m10, m11, m12);    // elements[5]  (row1,col1)
    const double e6 = GS_DOT3(m20, m21, m22, m10, m11, m12);    // elements[6]  (row2,col1)
    const double e10 = GS_DOT3(m20, m21, m22, m20, m21, m22);   // elements[10] (row2,col2)
#undef GS_DOT3
    const double e[6] = { e0, e1, e2, e5, e6, e10 };
    double max_value = 0.0;                                                         // index.js:370-376
    for (int j = 0; j < 6; j++) if (fabs(e[j]) > max_value) max_value = fabs(e[j]);
    o.cs[0] = (float)cx; o.cs[1] = (float)cy; o.cs[2] = (float)cz; o.cs[3] = (float)(max_value / 32767.0);
    uint32_t h[6];
    for (int j = 0; j < 6; j++) {
        h[j] = (uint32_t)(uint16_t)(int16_t)js_parse_int(e[j] * 32767.0 / max_value, pow10tab);   // index.js:386
        o.sigma[j] = (float)e[j];
    }
    o.cc[0] = h[0] | (h[1] << 16); o.cc[1] = h[2] | (h[3] << 16); o.cc[2] = h[4] | (h[5] << 16); o.cc[3] = rgba;
    double mx = sx > sy ? sx : sy; if (sz > mx) mx = sz;                             // Math.max(scale.x,y,z)
    if (sx != sx || sy != sy || sz != sz) mx = sx + sy + sz;                         // NaN propagates
    o.sort_row[0] = (float)cx; o.sort_row[1] = (float)cy; o.sort_row[2] = (float)cz;
    o.sort_row[3] = (float)(mx * (double)(rgba >> 24) / 255.0);                      // index.js:397
}

// ---------------------------------------------------------------- project (index.js:92-164)

struct Projected {      // 32-byte record consumed by the blend kernel
    float cx, cy;       // centre, device pixels, GL orientation (y up, origin bottom-left)
    float ax, ay;       // a = v2/|v2|^2 : vPosition.x = dot(d, a)
    float bx, by;       // b = v1/|v1|^2 : vPosition.y = dot(d, b)
    uint32_t rgba;      // colour bytes as stored (index.js:151-157)
    float alpha;        // rgba>>24 / 255
};

struct ProjExtra { float v1x, v1y, v2x, v2y, zndc; };

// Returns false when the vertex shader would emit nothing for this splat (frustum cull index.js:110-115,
// far-plane clip, or NaN axes).  mv/P are the f32 uniforms, column-major.
GS_HD bool project_splat(const float cs[4], const uint32_t cc[4], const float *mv, const float *P, float focal,
                         float vw, float vh, Projected &o, ProjExtra &x)
{
    const float cx = cs[0], cy = cs[1], cz = cs[2], scl = cs[3];
    const float camx = ((mv[0] * cx + mv[4] * cy) + mv[8] * cz) + mv[12];
    const float camy = ((mv[1] * cx + mv[5] * cy) + mv[9] * cz) + mv[13];
    const float camz = ((mv[2] * cx + mv[6] * cy) + mv[10] * cz) + mv[14];
    const float camw = ((mv[3] * cx + mv[7] * cy) + mv[11] * cz) + mv[15];
    const float px = ((P[0] * camx + P[4] * camy) + P[8] * camz) + P[12] * camw;
    const float py = ((P[1] * camx + P[5] * camy) + P[9] * camz) + P[13] * camw;
    const float pz = ((P[2] * camx + P[6] * camy) + P[10] * camz) + P[14] * camw;
    const float pw = ((P[3] * camx + P[7] * camy) + P[11] * camz) + P[15] * camw;
    const float bounds = 1.2f * pw;
    if (pz < -pw || px < -bounds || px > bounds || py < -bounds || py > bounds) return false;
    if (!(pw > 0.0f)) return false;

    const float m11 = (float)(int16_t)(cc[0] & 0xFFFF) * scl, m12 = (float)(int16_t)(cc[0] >> 16) * scl;
    const float m13 = (float)(int16_t)(cc[1] & 0xFFFF) * scl, m22 = (float)(int16_t)(cc[1] >> 16) * scl;
    const float m23 = (float)(int16_t)(cc[2] & 0xFFFF) * scl, m33 = (float)(int16_t)(cc[2] >> 16) * scl;

    const float j00 = focal / camz, j02 = -(focal * camx) / (camz * camz);
    const float j11 = -focal / camz, j12 = (focal * camy) / (camz * camz);
    const float M00 = j00 * mv[0] + j02 * mv[2], M01 = j00 * mv[4] + j02 * mv[6], M02 = j00 * mv[8] + j02 * mv[10];
    const float M10 = j11 * mv[1] + j12 * mv[2], M11 = j11 * mv[5] + j12 * mv[6], M12 = j11 * mv[9] + j12 * mv[10];
    const float t0 = (m11 * M00 + m12 * M01) + m13 * M02;
    const float t1 = (m12 * M00 + m22 * M01) + m23 * M02;
    const float t2 = (m13 * M00 + m23 * M01) + m33 * M02;
    const float u0 = (m11 * M10 + m12 * M11) + m13 * M12;
    const float u1 = (m12 * M10 + m22 * M11) + m23 * M12;
    const float u2 = (m13 * M10 + m23 * M11) + m33 * M12;
    const float cov00 = (M00 * t0 + M01 * t1) + M02 * t2;
    const float cov01 = (M10 * t0 + M11 * t1) + M12 * t2;
    const float cov11 = (M10 * u0 + M11 * u1) + M12 * u2;

    const float d1 = cov00 + 0.3f, od = cov01, d2 = cov11 + 0.3f;
    const float mid = 0.5f * (d1 + d2);
    const float hd = (d1 - d2) / 2.0f;
    const float radius = sqrtf(hd * hd + od * od);
    const float l1 = mid + radius;
    const float l2 = fmaxf(mid - radius, 0.1f);
    const float dvx0 = od, dvy0 = l1 - d1;
    const float len = sqrtf(dvx0 * dvx0 + dvy0 * dvy0);
    if (!(len > 0.0f) || !(len <= 3.402823466e+38f) || !(fabsf(l1) <= 3.402823466e+38f)) return false;
    const float dvx = dvx0 / len, dvy = dvy0 / len;
    const float s1 = fminf(sqrtf(2.0f * l1), 1024.0f), s2 = fminf(sqrtf(2.0f * l2), 1024.0f);
    x.v1x = s1 * dvx; x.v1y = s1 * dvy;
    x.v2x = s2 * dvy; x.v2y = s2 * -dvx;
    const float ndcx = px / pw, ndcy = py / pw;
    x.zndc = pz / pw;
    if (x.zndc > 1.0f) return false;
    o.cx = (ndcx * 0.5f + 0.5f) * vw;
    o.cy = (ndcy * 0.5f + 0.5f) * vh;
    const float n1 = x.v1x * x.v1x + x.v1y * x.v1y, n2 = x.v2x * x.v2x + x.v2y * x.v2y;
    o.ax = x.v2x / n2; o.ay = x.v2y / n2;
    o.bx = x.v1x / n1; o.by = x.v1y / n1;
    o.rgba = cc[3];
    o.alpha = (float)(cc[3] >> 24) / 255.0f;
    return true;
}

// |p|^2 of the interpolated vPosition at pixel-centre offset (dx,dy) (index.js:158-163, 171).
// fmaf here is deliberate and mirrored by the checker.
GS_HD float frag_power(float dx, float dy, float ax, float ay, float bx, float by)
{
    const float ppx = fmaf(dx, ax, dy * ay);
    const float ppy = fmaf(dx, bx, dy * by);
    return fmaf(ppx, ppx, ppy * ppy);
}

// Conservative pixel bounding box (GL orientation) of the |p|<=2 ellipse: half extents
// 2*sqrt(v1x^2+v2x^2), 2*sqrt(v1y^2+v2y^2) plus a safety pad against fp32 rounding.
GS_HD void splat_pixel_bounds(const Projected &p, const ProjExtra &x, float &xmin, float &xmax, float &ymin, float &ymax)
{
    const float hw = 2.0f * sqrtf(x.v1x * x.v1x + x.v2x * x.v2x);
    const float hh = 2.0f * sqrtf(x.v1y * x.v1y + x.v2y * x.v2y);
    const float padx = 0.01f + 1.0e-4f * hw, pady = 0.01f + 1.0e-4f * hh;
    // pixel i (centre i+0.5) can be covered iff  cx-hw-pad <= i+0.5 <= cx+hw+pad
    xmin = ceilf(p.cx - hw - padx - 0.5f); xmax = floorf(p.cx + hw + padx - 0.5f);
    ymin = ceilf(p.cy - hh - pady - 0.5f); ymax = floorf(p.cy + hh + pady - 0.5f);
}


// ---------------------------------------------------------------- exact tile coverage of the |p|<=2 ellipse
// The ellipse is the conic  q(dx,dy) = A dx^2 + 2B dx dy + C dy^2 <= 4  with  A = |(ax,bx)|^2, B = ax*ay + bx*by,
// C = |(ay,by)|^2 and discriminant D = AC - B^2 = (ax*by - ay*bx)^2 (Lagrange identity: no cancellation).
// For a horizontal band dy in [ya,yb] the covered x-interval is [xl(clamp(dy_l)), xr(clamp(dy_r))] where
// xr/xl are the right/left roots of q = 4 and dy_r = -B*hw/C (= -dy_l) is the height of the rightmost point
// (xr is concave, xl convex).  A convex shape meets each tile row in one contiguous run of tiles, so per-row
// ranges give EXACTLY the set of tiles the ellipse touches instead of its bounding rectangle.
struct EllipseRows { float invA, B, D4A, D, dy_r, pad; };   // D4A = 4*A

// The tile ranges only have to be (a) a superset of the true coverage -- guaranteed by `pad` -- and (b) the SAME
// in the counting pass (k_project) and the writing pass (k_emit), which run this very code on the same stored
// record.  They are not part of the pixel parity contract, so the 1-ulp hardware reciprocal / square root
// (v_rcp_f32 / v_sqrt_f32, one instruction each) replace the ~15-instruction IEEE sequences here.
#if defined(__HIP_DEVICE_COMPILE__)
GS_HD float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
GS_HD float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
GS_HD float fast_rcp(float x) { return 1.0f / x; }
GS_HD float fast_sqrt(float x) { return sqrtf(x); }
#endif

GS_HD void ellipse_rows_setup(const Projected &p, EllipseRows &e)
{
    const float A = p.ax * p.ax + p.bx * p.bx;
    e.B = p.ax * p.ay + p.bx * p.by;
    const float C = p.ay * p.ay + p.by * p.by;
    const float cr = p.ax * p.by - p.ay * p.bx;
    e.D = cr * cr;
    e.invA = fast_rcp(A);
    e.D4A = 4.0f * A;
    const float hw = 2.0f * fast_sqrt(C * fast_rcp(e.D));
    e.dy_r = -(e.B * hw) * fast_rcp(C);
    e.pad = 0.02f + 2.0e-4f * hw;
}

// x-interval (relative to the splat centre) covered inside the band dy in [ya, yb]
GS_HD void ellipse_band_xrange(const EllipseRows &e, float ya, float yb, float &xmin, float &xmax)
{
    const float dr = fminf(fmaxf(e.dy_r, ya), yb), dl = fminf(fmaxf(-e.dy_r, ya), yb);
    xmax = (-e.B * dr + fast_sqrt(fmaxf(e.D4A - e.D * dr * dr, 0.0f))) * e.invA;
    xmin = (-e.B * dl - fast_sqrt(fmaxf(e.D4A - e.D * dl * dl, 0.0f))) * e.invA;
}

// Tiles of image tile-row `ty` (rows 16*ty .. 16*ty+15, top-down) that the splat can touch inside the strip
// [x0,x1): first tile column (strip-local) and count (0 = none).  H = viewport height.
GS_HD void splat_tile_row(const Projected &p, const EllipseRows &e, int ty, int H, int x0, int x1, uint32_t &tx0, uint32_t &n)
{
    const int r_lo = ty * 16, r_hi = (ty * 16 + 15 < H - 1) ? ty * 16 + 15 : H - 1;
    // GL pixel-centre y of image row r is (H-1-r) + 0.5
    const float ya = ((float)(H - 1 - r_hi) + 0.5f) - p.cy - e.pad, yb = ((float)(H - 1 - r_lo) + 0.5f) - p.cy + e.pad;
    float xmin, xmax;
    ellipse_band_xrange(e, ya, yb, xmin, xmax);
    const float fi0 = fmaxf(ceilf(p.cx + xmin - e.pad - 0.5f), (float)x0);
    const float fi1 = fminf(floorf(p.cx + xmax + e.pad - 0.5f), (float)(x1 - 1));
    if (!(fi0 <= fi1)) { tx0 = 0; n = 0; return; }
    const uint32_t a = (uint32_t)((int)fi0 - x0) >> 4, b = (uint32_t)((int)fi1 - x0) >> 4;
    tx0 = a; n = b - a + 1;
}

}  // namespace gsm

/*
 * gs_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement of the hot path of
 * quadjr/aframe-gaussian-splatting (reference file: index.js).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  The product library (libgs_splat_hip.so) never links, loads or
 * calls anything here and has no CPU fallback.
 *
 * Parity status
 *   sort / pack / ply / camera : PINNED -- checked bit-for-bit against golden
 *       vectors produced by executing the reference's own JavaScript under
 *       node (oracle/gen_golden.js -> tests/golden/, tests/test_oracle_golden.py).
 *   project / raster / blend   : PINNED against the reference's own GLSL as Mesa
 *       llvmpipe executes it -- oracle/gen_golden_gl.js captures the shaders,
 *       material state, textures, order and uniforms from the reference
 *       component under node, oracle/gl_ref.c draws them (headless GL through
 *       the DRI swrast interface), tests/test_gl_pin.py compares: equal
 *       fragment counts, max 1 LSB against the float-framebuffer image.  This
 *       file restates index.js:77-181 in fp32 with a fixed operation order; at
 *       sizes a software rasteriser cannot draw it is the pixel oracle.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; no fast-math: the
 * sort contract needs un-fused IEEE f64).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define GSO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ sort */

/* ECMAScript ToInt32 (the `|0` at index.js:561). */
static int32_t js_toint32(double d)
{
    if (!isfinite(d)) return 0;
    double t = trunc(d);
    double m = fmod(t, 4294967296.0);
    if (m < 0) m += 4294967296.0;
    return (int32_t)(uint32_t)m;
}

/*
 * sortSplats, index.js:507-570.  `m` points at element 12 of the first worker
 * row; `stride` is the row stride in floats (16 for the worker's matrices, 4
 * for a packed xyz+size array).  Returns V (= validCount); `out` must hold n
 * entries.  Output is the reference's Uint32Array(V) including its behaviour
 * for out-of-range buckets (typed-array writes at invalid indices are dropped:
 * such splats are missing and the tail of the array stays 0).
 */
GSO_API size_t gso_sort(const float *m, size_t n, size_t stride, const float view[4],
                        const float *cutout /* 16 or NULL */, uint32_t *out)
{
    const double threshold = -0.0001;                       /* index.js:509 */
    double maxDepth = -INFINITY, minDepth = INFINITY;       /* index.js:511-512 */
    float *depthList = (float *)malloc(sizeof(float) * (n ? n : 1));
    int32_t *validIndexList = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    size_t validCount = 0;
    const double v0 = view[0], v1 = view[1], v2 = view[2], v3 = view[3];

    for (size_t i = 0; i < n; i++) {
        const double x = m[i * stride + 0], y = m[i * stride + 1], z = m[i * stride + 2];
        const double depth = ((v0 * x + v1 * y) + v2 * z) + v3;     /* index.js:519-523 */
        int cutoutArea = 1;
        if (cutout) {                                               /* index.js:526-545 */
            const double yn = -y;
            const double w = 1.0 / (((cutout[3] * x + cutout[7] * yn) + cutout[11] * z) + cutout[15]);
            const double q0 = (((cutout[0] * x + cutout[4] * yn) + cutout[8] * z) + cutout[12]) * w;
            const double q1 = (((cutout[1] * x + cutout[5] * yn) + cutout[9] * z) + cutout[13]) * w;
            const double q2 = (((cutout[2] * x + cutout[6] * yn) + cutout[10] * z) + cutout[14]) * w;
            if (q0 < -0.5 || q0 > 0.5 || q1 < -0.5 || q1 > 0.5 || q2 < -0.5 || q2 > 0.5) cutoutArea = 0;
        }
        if (depth < 0 && (double)m[i * stride + 3] > threshold * depth && cutoutArea) {   /* index.js:548 */
            depthList[validCount] = (float)depth;
            validIndexList[validCount] = (int32_t)i;
            validCount++;
            if (depth > maxDepth) maxDepth = depth;
            if (depth < minDepth) minDepth = depth;
        }
    }

    const double depthInv = (256 * 256 - 1) / (maxDepth - minDepth);  /* index.js:558 */
    uint32_t *counts0 = (uint32_t *)calloc(65536, sizeof(uint32_t));
    uint32_t *starts0 = (uint32_t *)calloc(65536, sizeof(uint32_t));
    int32_t *sizeList = (int32_t *)depthList;                       /* aliases, index.js:514 */
    for (size_t i = 0; i < validCount; i++) {
        const int32_t b = js_toint32(((double)depthList[i] - minDepth) * depthInv);
        sizeList[i] = b;
        if (b >= 0 && b < 65536) counts0[b]++;                      /* OOB typed-array write is a no-op */
    }
    for (int i = 1; i < 65536; i++) starts0[i] = starts0[i - 1] + counts0[i - 1];
    memset(out, 0, sizeof(uint32_t) * validCount);
    for (size_t i = 0; i < validCount; i++) {
        const int32_t b = sizeList[i];
        if (b >= 0 && b < 65536) out[starts0[b]++] = (uint32_t)validIndexList[i];
    }
    free(counts0); free(starts0); free(depthList); free(validIndexList);
    return validCount;
}

/* ------------------------------------------------------------------ pack */

/* JS parseInt(Number): Number -> shortest round-trip string -> leading integer.
 * For |v| >= 1e-6 that is truncation; below it the string is in exponent form
 * ("3.2767e-8") and parseInt returns the leading digit (index.js:386). */
static int32_t js_parse_int(double v)
{
    if (isnan(v) || isinf(v)) return 0;          /* NaN -> Int16Array stores 0 */
    double a = fabs(v);
    if (a == 0) return 0;
    if (a >= 1e-6) {
        double t = trunc(v);
        if (a >= 1e21) {                          /* exponent form "1.2e+21" -> leading digit */
            char buf[40];
            for (int p = 0; p < 17; p++) { snprintf(buf, sizeof buf, "%.*e", p, a); if (strtod(buf, NULL) == a) break; }
            return (int32_t)((v < 0 ? -1 : 1) * (buf[0] - '0'));
        }
        return (int32_t)(int16_t)(int64_t)t;     /* Int16Array store: modular */
    }
    char buf[40];
    for (int p = 0; p < 17; p++) {               /* shortest digits that round-trip */
        snprintf(buf, sizeof buf, "%.*e", p, a);
        if (strtod(buf, NULL) == a) break;
    }
    return (v < 0 ? -1 : 1) * (buf[0] - '0');
}

/*
 * pushDataBuffer pack loop, index.js:343-402, with the three.js Matrix4 /
 * Quaternion closed forms evaluated in f64 in the same order.
 * rows: n x 32 B .splat rows.  Outputs (any may be NULL):
 *   center_scale n x 4 f32, cov_color n x 4 u32, matrices n x 16 f32.
 */
GSO_API void gso_pack(const uint8_t *rows, size_t n, float *center_scale, uint32_t *cov_color, float *matrices)
{
    for (size_t i = 0; i < n; i++) {
        const uint8_t *u = rows + 32 * i;
        float f[6]; memcpy(f, u, 24);
        const double qx = (u[28 + 1] - 128) / 128.0, qy = (u[28 + 2] - 128) / 128.0;
        const double qz = -(u[28 + 3] - 128) / 128.0, qw = (u[28 + 0] - 128) / 128.0;
        const double cx = f[0], cy = f[1], cz = -(double)f[2];
        const double sx = f[3], sy = f[4], sz = f[5];

        /* makeRotationFromQuaternion */
        const double x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
        const double xx = qx * x2, xy = qx * y2, xz = qx * z2;
        const double yy = qy * y2, yz = qy * z2, zz = qz * z2;
        const double wx = qw * x2, wy = qw * y2, wz = qw * z2;
        double te[16] = { 1 - (yy + zz), xy + wz, xz - wy, 0,
                          xy - wz, 1 - (xx + zz), yz + wx, 0,
                          xz + wy, yz - wx, 1 - (xx + yy), 0,
                          0, 0, 0, 1 };
        double t;
#define SWAP(a, b) t = te[a]; te[a] = te[b]; te[b] = t
        SWAP(1, 4); SWAP(2, 8); SWAP(6, 9); SWAP(3, 12); SWAP(7, 13); SWAP(11, 14);      /* transpose */
        te[0] *= sx; te[4] *= sy; te[8] *= sz; te[1] *= sx; te[5] *= sy; te[9] *= sz;      /* scale */
        te[2] *= sx; te[6] *= sy; te[10] *= sz; te[3] *= sx; te[7] *= sy; te[11] *= sz;
        double a[16]; memcpy(a, te, sizeof a);                                             /* mtx_t = clone */
        SWAP(1, 4); SWAP(2, 8); SWAP(6, 9); SWAP(3, 12); SWAP(7, 13); SWAP(11, 14);      /* transpose */
#undef SWAP
        double r[16];                                                                       /* premultiply: a * te */
        for (int row = 0; row < 4; row++)
            for (int col = 0; col < 4; col++)
                r[col * 4 + row] = a[row] * te[col * 4] + a[4 + row] * te[col * 4 + 1]
                                 + a[8 + row] * te[col * 4 + 2] + a[12 + row] * te[col * 4 + 3];
        r[12] = cx; r[13] = cy; r[14] = cz;                                                 /* setPosition */

        static const int ci[6] = { 0, 1, 2, 5, 6, 10 };
        double max_value = 0.0;
        for (int j = 0; j < 6; j++) if (fabs(r[ci[j]]) > max_value) max_value = fabs(r[ci[j]]);

        if (center_scale) {
            center_scale[4 * i + 0] = (float)cx; center_scale[4 * i + 1] = (float)cy;
            center_scale[4 * i + 2] = (float)cz; center_scale[4 * i + 3] = (float)(max_value / 32767.0);
        }
        if (cov_color) {
            int16_t q[6];
            for (int j = 0; j < 6; j++) q[j] = (int16_t)js_parse_int(r[ci[j]] * 32767.0 / max_value);
            uint32_t *o = cov_color + 4 * i;
            o[0] = (uint16_t)q[0] | ((uint32_t)(uint16_t)q[1] << 16);
            o[1] = (uint16_t)q[2] | ((uint32_t)(uint16_t)q[3] << 16);
            o[2] = (uint16_t)q[4] | ((uint32_t)(uint16_t)q[5] << 16);
            o[3] = u[24] | ((uint32_t)u[25] << 8) | ((uint32_t)u[26] << 16) | ((uint32_t)u[27] << 24);
        }
        if (matrices) {
            double mx = sx > sy ? sx : sy; if (sz > mx) mx = sz;          /* Math.max; NaN ignored here */
            if (isnan(sx) || isnan(sy) || isnan(sz)) mx = NAN;
            r[15] = mx * u[27] / 255.0;                                   /* index.js:397 */
            for (int j = 0; j < 16; j++) matrices[16 * i + j] = (float)r[j];
        }
    }
}

/* ------------------------------------------------------------------ camera (index.js:438-487) */

static void m4_mul(const double *a, const double *b, double *o)   /* three.js multiplyMatrices, o = a*b */
{
    double r[16];
    for (int row = 0; row < 4; row++)
        for (int col = 0; col < 4; col++)
            r[col * 4 + row] = a[row] * b[col * 4] + a[4 + row] * b[col * 4 + 1]
                             + a[8 + row] * b[col * 4 + 2] + a[12 + row] * b[col * 4 + 3];
    memcpy(o, r, sizeof r);
}

static void m4_invert(const double *te, double *o)                /* three.js Matrix4.invert */
{
    const double n11 = te[0], n21 = te[1], n31 = te[2], n41 = te[3], n12 = te[4], n22 = te[5], n32 = te[6], n42 = te[7],
                 n13 = te[8], n23 = te[9], n33 = te[10], n43 = te[11], n14 = te[12], n24 = te[13], n34 = te[14], n44 = te[15];
    const double t11 = n23 * n34 * n42 - n24 * n33 * n42 + n24 * n32 * n43 - n22 * n34 * n43 - n23 * n32 * n44 + n22 * n33 * n44;
    const double t12 = n14 * n33 * n42 - n13 * n34 * n42 - n14 * n32 * n43 + n12 * n34 * n43 + n13 * n32 * n44 - n12 * n33 * n44;
    const double t13 = n13 * n24 * n42 - n14 * n23 * n42 + n14 * n22 * n43 - n12 * n24 * n43 - n13 * n22 * n44 + n12 * n23 * n44;
    const double t14 = n14 * n23 * n32 - n13 * n24 * n32 - n14 * n22 * n33 + n12 * n24 * n33 + n13 * n22 * n34 - n12 * n23 * n34;
    const double det = n11 * t11 + n21 * t12 + n31 * t13 + n41 * t14;
    if (det == 0) { memset(o, 0, 16 * sizeof(double)); return; }
    const double d = 1 / det;
    double r[16];
    r[0] = t11 * d;
    r[1] = (n24 * n33 * n41 - n23 * n34 * n41 - n24 * n31 * n43 + n21 * n34 * n43 + n23 * n31 * n44 - n21 * n33 * n44) * d;
    r[2] = (n22 * n34 * n41 - n24 * n32 * n41 + n24 * n31 * n42 - n21 * n34 * n42 - n22 * n31 * n44 + n21 * n32 * n44) * d;
    r[3] = (n23 * n32 * n41 - n22 * n33 * n41 - n23 * n31 * n42 + n21 * n33 * n42 + n22 * n31 * n43 - n21 * n32 * n43) * d;
    r[4] = t12 * d;
    r[5] = (n13 * n34 * n41 - n14 * n33 * n41 + n14 * n31 * n43 - n11 * n34 * n43 - n13 * n31 * n44 + n11 * n33 * n44) * d;
    r[6] = (n14 * n32 * n41 - n12 * n34 * n41 - n14 * n31 * n42 + n11 * n34 * n42 + n12 * n31 * n44 - n11 * n32 * n44) * d;
    r[7] = (n12 * n33 * n41 - n13 * n32 * n41 + n13 * n31 * n42 - n11 * n33 * n42 - n12 * n31 * n43 + n11 * n32 * n43) * d;
    r[8] = t13 * d;
    r[9] = (n14 * n23 * n41 - n13 * n24 * n41 - n14 * n21 * n43 + n11 * n24 * n43 + n13 * n21 * n44 - n11 * n23 * n44) * d;
    r[10] = (n12 * n24 * n41 - n14 * n22 * n41 + n14 * n21 * n42 - n11 * n24 * n42 - n12 * n21 * n44 + n11 * n22 * n44) * d;
    r[11] = (n13 * n22 * n41 - n12 * n23 * n41 - n13 * n21 * n42 + n11 * n23 * n42 + n12 * n21 * n43 - n11 * n22 * n43) * d;
    r[12] = t14 * d;
    r[13] = (n13 * n24 * n31 - n14 * n23 * n31 + n14 * n21 * n33 - n11 * n24 * n33 - n13 * n21 * n34 + n11 * n23 * n34) * d;
    r[14] = (n14 * n22 * n31 - n12 * n24 * n31 - n14 * n21 * n32 + n11 * n24 * n32 + n12 * n21 * n34 - n11 * n22 * n34) * d;
    r[15] = (n12 * n23 * n31 - n13 * n22 * n31 + n13 * n21 * n32 - n11 * n23 * n32 - n12 * n21 * n33 + n11 * n22 * n33) * d;
    memcpy(o, r, sizeof r);
}

static void flip_s(double *e) { e[1] *= -1.0; e[4] *= -1.0; e[6] *= -1.0; e[9] *= -1.0; e[13] *= -1.0; }

/* getModelViewMatrix, index.js:467-487 */
GSO_API void gso_model_view(const double cam_world[16], const double obj_world[16], double out[16])
{
    double vm[16], m[16];
    memcpy(vm, cam_world, sizeof vm); flip_s(vm);
    m4_invert(obj_world, m); flip_s(m);
    m4_mul(m, vm, m);
    m4_invert(m, out);
}

/* getProjectionMatrix, index.js:456-466 */
GSO_API void gso_projection(const double proj[16], double out[16])
{
    memcpy(out, proj, 16 * sizeof(double));
    out[4] *= -1; out[5] *= -1; out[6] *= -1; out[7] *= -1;
}

/* tick, index.js:441-448: view row (f32) and worldToCutout (f32) */
GSO_API void gso_tick(const double cam_world[16], const double obj_world[16], const double *cutout_world,
                      float view[4], float cutout[16])
{
    double mv[16]; gso_model_view(cam_world, obj_world, mv);
    view[0] = (float)mv[2]; view[1] = (float)mv[6]; view[2] = (float)mv[10]; view[3] = (float)mv[14];
    if (cutout_world && cutout) {
        double w[16]; m4_invert(cutout_world, w); m4_mul(w, obj_world, w);
        for (int i = 0; i < 16; i++) cutout[i] = (float)w[i];
    }
}

/* onBeforeRender focal, index.js:191 */
GSO_API double gso_focal(const double gs_proj[16], double viewport_h) { return (viewport_h / 2.0) * fabs(gs_proj[5]); }

/* ------------------------------------------------------------------ project (index.js:92-164) */

typedef struct {
    int32_t visible;
    float cx, cy;          /* centre in device pixels, y up, origin bottom-left */
    float ax, ay, bx, by;  /* a = v2/|v2|^2, b = v1/|v1|^2 : vPosition = (d.a, d.b) */
    float v1x, v1y, v2x, v2y;
    float zndc;
    float r, g, b, alpha;
} gso_proj_t;

GSO_API void gso_project(const float *center_scale, const uint32_t *cov_color, uint32_t idx, const float mv[16],
                         const float P[16], float focal, float vw, float vh, gso_proj_t *o)
{
    memset(o, 0, sizeof *o);
    const float *cs = center_scale + 4 * (size_t)idx;
    const float cx = cs[0], cy = cs[1], cz = cs[2], scl = cs[3];
    /* camspace = gsModelViewMatrix * vec4(center,1)     index.js:106-108 */
    const float camx = ((mv[0] * cx + mv[4] * cy) + mv[8] * cz) + mv[12];
    const float camy = ((mv[1] * cx + mv[5] * cy) + mv[9] * cz) + mv[13];
    const float camz = ((mv[2] * cx + mv[6] * cy) + mv[10] * cz) + mv[14];
    const float camw = ((mv[3] * cx + mv[7] * cy) + mv[11] * cz) + mv[15];
    const float px = ((P[0] * camx + P[4] * camy) + P[8] * camz) + P[12] * camw;
    const float py = ((P[1] * camx + P[5] * camy) + P[9] * camz) + P[13] * camw;
    const float pz = ((P[2] * camx + P[6] * camy) + P[10] * camz) + P[14] * camw;
    const float pw = ((P[3] * camx + P[7] * camy) + P[11] * camz) + P[15] * camw;
    const float bounds = 1.2f * pw;                                   /* index.js:110-115 */
    if (pz < -pw || px < -bounds || px > bounds || py < -bounds || py > bounds) return;
    if (!(pw > 0.0f)) return;                /* w<=0 survivors are NaN positions: never rasterised */

    const uint32_t *cc = cov_color + 4 * (size_t)idx;                 /* index.js:92-99, 117-125 */
    const float m11 = (float)(int16_t)(cc[0] & 0xFFFF) * scl, m12 = (float)(int16_t)(cc[0] >> 16) * scl;
    const float m13 = (float)(int16_t)(cc[1] & 0xFFFF) * scl, m22 = (float)(int16_t)(cc[1] >> 16) * scl;
    const float m23 = (float)(int16_t)(cc[2] & 0xFFFF) * scl, m33 = (float)(int16_t)(cc[2] >> 16) * scl;

    /* J_true rows (index.js:127-131) */
    const float j00 = focal / camz, j02 = -(focal * camx) / (camz * camz);
    const float j11 = -focal / camz, j12 = (focal * camy) / (camz * camz);
    /* M = J_true * A, A = mat3(gsModelViewMatrix), A[r][c] = mv[c*4+r] */
    const float M00 = j00 * mv[0] + j02 * mv[2], M01 = j00 * mv[4] + j02 * mv[6], M02 = j00 * mv[8] + j02 * mv[10];
    const float M10 = j11 * mv[1] + j12 * mv[2], M11 = j11 * mv[5] + j12 * mv[6], M12 = j11 * mv[9] + j12 * mv[10];
    /* cov = M * Vrk * M^T  (index.js:133-135) */
    const float t0 = (m11 * M00 + m12 * M01) + m13 * M02;
    const float t1 = (m12 * M00 + m22 * M01) + m23 * M02;
    const float t2 = (m13 * M00 + m23 * M01) + m33 * M02;
    const float u0 = (m11 * M10 + m12 * M11) + m13 * M12;
    const float u1 = (m12 * M10 + m22 * M11) + m23 * M12;
    const float u2 = (m13 * M10 + m23 * M11) + m33 * M12;
    const float cov00 = (M00 * t0 + M01 * t1) + M02 * t2;
    const float cov01 = (M10 * t0 + M11 * t1) + M12 * t2;
    const float cov11 = (M10 * u0 + M11 * u1) + M12 * u2;

    const float d1 = cov00 + 0.3f, od = cov01, d2 = cov11 + 0.3f;     /* index.js:139-146 */
    const float mid = 0.5f * (d1 + d2);
    const float hd = (d1 - d2) / 2.0f;
    const float radius = sqrtf(hd * hd + od * od);
    const float l1 = mid + radius;
    const float l2 = fmaxf(mid - radius, 0.1f);
    const float dvx0 = od, dvy0 = l1 - d1;                            /* index.js:147 */
    const float len = sqrtf(dvx0 * dvx0 + dvy0 * dvy0);
    if (!(len > 0.0f) || !isfinite(len) || !isfinite(l1)) return;     /* normalize(0) / non-finite: never rasterised */
    const float dvx = dvx0 / len, dvy = dvy0 / len;
    const float s1 = fminf(sqrtf(2.0f * l1), 1024.0f), s2 = fminf(sqrtf(2.0f * l2), 1024.0f);
    o->v1x = s1 * dvx; o->v1y = s1 * dvy;                             /* index.js:148-149 */
    o->v2x = s2 * dvy; o->v2y = s2 * -dvx;

    const float ndcx = px / pw, ndcy = py / pw;                        /* index.js:137 */
    o->zndc = pz / pw;
    if (o->zndc > 1.0f) return;                                       /* far-plane clip of the whole quad */
    o->cx = (ndcx * 0.5f + 0.5f) * vw;
    o->cy = (ndcy * 0.5f + 0.5f) * vh;
    const float n1 = o->v1x * o->v1x + o->v1y * o->v1y, n2 = o->v2x * o->v2x + o->v2y * o->v2y;
    o->ax = o->v2x / n2; o->ay = o->v2y / n2;                         /* vPosition.x = d.v2/|v2|^2 */
    o->bx = o->v1x / n1; o->by = o->v1y / n1;                         /* vPosition.y = d.v1/|v1|^2 */
    o->r = (float)(cc[3] & 0xFF) / 255.0f; o->g = (float)((cc[3] >> 8) & 0xFF) / 255.0f;   /* index.js:151-157 */
    o->b = (float)((cc[3] >> 16) & 0xFF) / 255.0f; o->alpha = (float)(cc[3] >> 24) / 255.0f;
    o->visible = 1;
}

/* ------------------------------------------------------------------ raster + shade + blend (index.js:52-66, 158-181)
 * Back-to-front "over" in the submitted (sorted) order, fp32 accumulators, one
 * final rounding to RGBA8.  Output rows are top-down (row 0 = top of the
 * image); x0..x1 selects a column strip (multi-GPU tests).  `frags` counts the
 * reference-equivalent splat-fragments (|p|^2 <= 4, index.js:171-172). */
GSO_API int gso_render_scene(const float *center_scale, const uint32_t *cov_color, const uint32_t *sorted, size_t V,
                             const float mv[16], const float P[16], float focal, int W, int H, int x0, int x1,
                             const float bg[4], const float *scene_depth /* W*H window depth, row 0 = top, or NULL */,
                             const uint8_t *scene_rgba /* W*H*4, row 0 = top, or NULL */,
                             float *out_f32, uint8_t *out_u8, uint64_t *frags)
{
    const int SW = x1 - x0;
    float *fb = (float *)malloc(sizeof(float) * 4 * (size_t)SW * H);
    if (!fb) return -1;
    for (int r = 0; r < H; r++)
        for (int i = 0; i < SW; i++) {
            float *d = fb + 4 * ((size_t)r * SW + i);
            if (scene_rgba) for (int k = 0; k < 4; k++) d[k] = (float)scene_rgba[4 * ((size_t)r * W + x0 + i) + k] / 255.0f;
            else memcpy(d, bg, 4 * sizeof(float));
        }
    uint64_t nf = 0;
    for (size_t s = 0; s < V; s++) {
        gso_proj_t p;
        gso_project(center_scale, cov_color, sorted[s], mv, P, focal, (float)W, (float)H, &p);
        if (!p.visible) continue;
        const float hw = 2.0f * sqrtf(p.v1x * p.v1x + p.v2x * p.v2x) + 2.0f;
        const float hh = 2.0f * sqrtf(p.v1y * p.v1y + p.v2y * p.v2y) + 2.0f;
        double lo = floor((double)p.cx - hw), hi = ceil((double)p.cx + hw);
        int ix0 = lo < x0 ? x0 : (lo > x1 ? x1 : (int)lo), ix1 = hi > x1 - 1 ? x1 - 1 : (hi < x0 - 1 ? x0 - 1 : (int)hi);
        lo = floor((double)p.cy - hh); hi = ceil((double)p.cy + hh);
        int iy0 = lo < 0 ? 0 : (lo > H ? H : (int)lo), iy1 = hi > H - 1 ? H - 1 : (hi < -1 ? -1 : (int)hi);
        /* depthTest: true, depthWrite: false (index.js:179-180): every fragment of the quad carries the same window
         * depth zndc*0.5+0.5 and survives iff it is <= the opaque scene's depth (LEQUAL) */
        const float zwin = p.zndc * 0.5f + 0.5f;
        for (int j = iy0; j <= iy1; j++) {
            const float dy = ((float)j + 0.5f) - p.cy;
            for (int i = ix0; i <= ix1; i++) {
                if (scene_depth && !(zwin <= scene_depth[(size_t)(H - 1 - j) * W + i])) continue;
                const float dx = ((float)i + 0.5f) - p.cx;
                const float ppx = fmaf(dx, p.ax, dy * p.ay);
                const float ppy = fmaf(dx, p.bx, dy * p.by);
                const float q = fmaf(ppx, ppx, ppy * ppy);               /* -A, index.js:171 */
                if (q > 4.0f) continue;                                   /* discard, index.js:172 */
                const float B = expf(-q) * p.alpha;                      /* index.js:173 */
                float *d = fb + 4 * ((size_t)(H - 1 - j) * SW + (i - x0));
                const float om = 1.0f - B;                                /* blend state index.js:177-181 */
                d[0] = p.r * B + d[0] * om; d[1] = p.g * B + d[1] * om; d[2] = p.b * B + d[2] * om;
                d[3] = B + d[3] * om;
                nf++;
            }
        }
    }
    if (out_f32) memcpy(out_f32, fb, sizeof(float) * 4 * (size_t)SW * H);
    if (out_u8)
        for (size_t i = 0; i < 4 * (size_t)SW * H; i++) {
            float v = fb[i]; v = v < 0 ? 0 : (v > 1 ? 1 : v);
            out_u8[i] = (uint8_t)(v * 255.0f + 0.5f);
        }
    if (frags) *frags = nf;
    free(fb);
    return 0;
}

GSO_API int gso_render(const float *center_scale, const uint32_t *cov_color, const uint32_t *sorted, size_t V,
                       const float mv[16], const float P[16], float focal, int W, int H, int x0, int x1,
                       const float bg[4], float *out_f32, uint8_t *out_u8, uint64_t *frags)
{
    return gso_render_scene(center_scale, cov_color, sorted, V, mv, P, focal, W, H, x0, x1, bg, NULL, NULL, out_f32, out_u8, frags);
}

/* ------------------------------------------------------------------ PLY -> .splat rows (index.js:600-745) */

enum { T_F64, T_I32, T_U32, T_F32, T_I16, T_U16, T_U8, T_I8 };
typedef struct { char name[64]; int type; size_t off; } prop_t;
/* Math.exp as the reference's engine evaluates it.  THIRD-PARTY arithmetic, absent from /root/reference: V8
 * (node 12.22.9 here, V8 7.8; any Chromium of the A-Frame 1.4 era) implements Math.exp with base::ieee754::exp, which
 * is Sun's fdlibm 5.3 e_exp.c: argument reduction x = k*ln2 + r with a two-word ln2, the degree-5 minimax polynomial
 * for r*(exp(r)+1)/(exp(r)-1), then scaling by 2^k; V8 adds one special case, exp(1) = E.  It is accurate to <1 ulp but NOT correctly rounded, so libm's
 * exp() is not a substitute where bits matter.  Restated from the published algorithm; pinned bit for bit against
 * tests/golden/math_exp.bin (8200 values produced by Math.exp under node in this container). */
GSO_API double gso_js_exp(double x)
{
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        inv_ln2 = 1.44269504088896338700e+00,
                        P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    uint64_t bits;
    memcpy(&bits, &x, 8);
    const int neg = (int)(bits >> 63);
    const uint32_t hx = (uint32_t)(bits >> 32) & 0x7fffffffu;          /* high word of |x| */
    double hi = 0.0, lo = 0.0;
    int k = 0;
    if (hx >= 0x40862E42u) {                                           /* |x| >= 709.78 */
        if (hx >= 0x7ff00000u) {
            if ((bits & 0x000fffffffffffffull) != 0) return x + x;     /* NaN */
            return neg ? 0.0 : x;                                      /* exp(-inf) = 0, exp(+inf) = +inf */
        }
        if (x > 7.09782712893383973096e+02) return HUGE_VAL;           /* overflow */
        if (x < -7.45133219101941108420e+02) return 0.0;               /* underflow */
    }
    if (hx > 0x3fd62e42u) {                                            /* |x| > 0.5 ln2 */
        if (hx < 0x3FF0A2B2u) {                                        /* and |x| < 1.5 ln2 */
            if (x == 1.0) return 2.718281828459045;                    /* V8 returns the constant E here (the formula is 1 ulp off) */
            hi = neg ? x + ln2_hi : x - ln2_hi;
            lo = neg ? -ln2_lo : ln2_lo;
            k = neg ? -1 : 1;
        } else {
            k = (int)(inv_ln2 * x + (neg ? -0.5 : 0.5));
            const double t = (double)k;
            hi = x - t * ln2_hi;                                       /* t*ln2_hi is exact */
            lo = t * ln2_lo;
        }
        x = hi - lo;
    } else if (hx < 0x3e300000u) {                                     /* |x| < 2^-28 */
        return 1.0 + x;
    }
    const double t = x * x;
    const double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    /* y * 2^k */
    uint64_t yb;
    memcpy(&yb, &y, 8);
    if (k >= -1021) {
        yb += (uint64_t)(int64_t)k << 52;
        memcpy(&y, &yb, 8);
        return y;
    }
    yb += (uint64_t)(int64_t)(k + 1000) << 52;
    memcpy(&y, &yb, 8);
    return y * 9.33263618503218878990e-302;                            /* 2^-1000 */
}

static const int TSIZE[] = { 8, 4, 4, 4, 2, 2, 1, 1 };

static double rd(const uint8_t *p, int type)
{
    switch (type) {
    case T_F64: { double v; memcpy(&v, p, 8); return v; }
    case T_I32: { int32_t v; memcpy(&v, p, 4); return v; }
    case T_U32: { uint32_t v; memcpy(&v, p, 4); return v; }
    case T_F32: { float v; memcpy(&v, p, 4); return v; }
    case T_I16: { int16_t v; memcpy(&v, p, 2); return v; }
    case T_U16: { uint16_t v; memcpy(&v, p, 2); return v; }
    case T_U8: return *p;
    default: return (int8_t)*p;
    }
}

static uint8_t clamp_u8(double v)     /* Uint8ClampedArray store: clamp, round half to even; NaN -> 0 */
{
    if (!(v > 0)) return 0;
    if (v >= 255) return 255;
    double f = floor(v), d = v - f;
    if (d > 0.5) return (uint8_t)(f + 1);
    if (d < 0.5) return (uint8_t)f;
    return (uint8_t)(((int)f & 1) ? f + 1 : f);
}

typedef struct { float key; uint32_t idx; } imp_t;
static void merge_sort_desc(imp_t *a, imp_t *tmp, size_t n)     /* stable, descending by key */
{
    if (n < 2) return;
    size_t h = n / 2;
    merge_sort_desc(a, tmp, h); merge_sort_desc(a + h, tmp, n - h);
    size_t i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = ((double)a[j].key - (double)a[i].key > 0) ? a[j++] : a[i++];
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, n * sizeof *a);
}

/* Returns 0 and *out_n (rows written if out!=NULL) or <0 with a message in err:
 *  -1 "Unable to read .ply file header"  -2 "<prop> not found"  -3 truncated/bad vertex count */
GSO_API int gso_ply_to_splat(const uint8_t *buf, size_t len, uint8_t *out, size_t *out_n, char *err, size_t errlen)
{
    size_t hl = len < 10240 ? len : 10240;
    const char *END = "end_header\n";
    long hend = -1;
    for (size_t i = 0; i + 11 <= hl; i++) if (!memcmp(buf + i, END, 11)) { hend = (long)i; break; }
    if (hend < 0) { snprintf(err, errlen, "Unable to read .ply file header"); return -1; }
    /* /element vertex (\d+)\n/ over the decoded 10 KiB */
    long vcount = -1;
    { const char *K = "element vertex ";
      for (size_t i = 0; i + 15 < hl && vcount < 0; i++) if (!memcmp(buf + i, K, 15)) {
          size_t j = i + 15; long v = 0; int nd = 0;
          while (j < hl && buf[j] >= '0' && buf[j] <= '9') { v = v * 10 + (buf[j] - '0'); j++; nd++; }
          if (nd && j < hl && buf[j] == '\n') vcount = v;
      } }
    if (vcount < 0) { snprintf(err, errlen, "Unable to read .ply file header"); return -3; }
    prop_t props[256]; int np = 0; size_t row = 0;
    { size_t ls = 0;
      while (ls < (size_t)hend) {
          size_t le = ls; while (le < (size_t)hend && buf[le] != '\n') le++;
          if (le - ls >= 9 && !memcmp(buf + ls, "property ", 9)) {
              char line[256]; size_t L = le - ls < 255 ? le - ls : 255; memcpy(line, buf + ls, L); line[L] = 0;
              char *parts[3] = { 0, 0, 0 }; int k = 0; char *s = line;      /* prop.split(" ") -> [p,type,name] */
              while (k < 3) { parts[k++] = s; char *sp = strchr(s, ' '); if (!sp) break; *sp = 0; s = sp + 1; }
              const char *ty = parts[1] ? parts[1] : "", *nm = parts[2] ? parts[2] : "undefined";
              int t = T_I8;
              if (!strcmp(ty, "double")) t = T_F64; else if (!strcmp(ty, "int")) t = T_I32; else if (!strcmp(ty, "uint")) t = T_U32;
              else if (!strcmp(ty, "float")) t = T_F32; else if (!strcmp(ty, "short")) t = T_I16; else if (!strcmp(ty, "ushort")) t = T_U16;
              else if (!strcmp(ty, "uchar")) t = T_U8;
              int slot = -1; for (int q = 0; q < np; q++) if (!strcmp(props[q].name, nm)) slot = q;
              if (slot < 0 && np < 256) slot = np++;
              if (slot >= 0) { snprintf(props[slot].name, sizeof props[slot].name, "%s", nm); props[slot].type = t; props[slot].off = row; }
              row += TSIZE[t];
          }
          ls = le + 1;
      } }
    const uint8_t *data = buf + hend + 11; size_t dlen = len - (size_t)hend - 11;
#define FIND(nm) ({ int _s = -1; for (int _q = 0; _q < np; _q++) if (!strcmp(props[_q].name, nm)) _s = _q; _s; })
#define NEED(var, nm) int var = FIND(nm); if (var < 0) { snprintf(err, errlen, "%s not found", nm); return -2; }
#define AT(r, s) rd(data + (size_t)(r) * row + props[s].off, props[s].type)
    size_t n = (size_t)vcount;
    const int has_scale = FIND("scale_0") >= 0;
    imp_t *imp = (imp_t *)malloc(sizeof(imp_t) * (n ? n : 1)), *tmp = (imp_t *)malloc(sizeof(imp_t) * (n ? n : 1));
    int rc = 0;
    if (n && row * n > dlen) { snprintf(err, errlen, "Offset is outside the bounds of the DataView"); rc = -3; goto done; }
    for (size_t r = 0; r < n; r++) { imp[r].idx = (uint32_t)r; imp[r].key = 0; }
    if (has_scale && n) {                                                   /* index.js:653-664 */
        int s0 = FIND("scale_0");
        int s1 = FIND("scale_1"); if (s1 < 0) { snprintf(err, errlen, "scale_1 not found"); rc = -2; goto done; }
        int s2 = FIND("scale_2"); if (s2 < 0) { snprintf(err, errlen, "scale_2 not found"); rc = -2; goto done; }
        int op = FIND("opacity"); if (op < 0) { snprintf(err, errlen, "opacity not found"); rc = -2; goto done; }
        for (size_t r = 0; r < n; r++) {
            const double size = gso_js_exp(AT(r, s0)) * gso_js_exp(AT(r, s1)) * gso_js_exp(AT(r, s2));
            const double opacity = 1 / (1 + gso_js_exp(-AT(r, op)));
            imp[r].key = (float)(size * opacity);
        }
    }
    merge_sort_desc(imp, tmp, n);                                            /* index.js:668 */
    if (out_n) *out_n = n;
    if (out && n) {
        int r0 = -1, r1 = -1, r2 = -1, r3 = -1, s0 = -1, s1 = -1, s2 = -1;
        if (has_scale) {
            const char *nm[7] = { "rot_0", "rot_1", "rot_2", "rot_3", "scale_0", "scale_1", "scale_2" }; int *dst[7] = { &r0, &r1, &r2, &r3, &s0, &s1, &s2 };
            for (int k = 0; k < 7; k++) { *dst[k] = FIND(nm[k]); if (*dst[k] < 0) { snprintf(err, errlen, "%s not found", nm[k]); rc = -2; goto done; } }
        }
        int px = FIND("x"), py = FIND("y"), pz = FIND("z");
        if (px < 0) { snprintf(err, errlen, "x not found"); rc = -2; goto done; }
        if (py < 0) { snprintf(err, errlen, "y not found"); rc = -2; goto done; }
        if (pz < 0) { snprintf(err, errlen, "z not found"); rc = -2; goto done; }
        int dc0 = FIND("f_dc_0"), dc1 = -1, dc2 = -1, cr = -1, cg = -1, cb = -1;
        if (dc0 >= 0) {
            dc1 = FIND("f_dc_1"); if (dc1 < 0) { snprintf(err, errlen, "f_dc_1 not found"); rc = -2; goto done; }
            dc2 = FIND("f_dc_2"); if (dc2 < 0) { snprintf(err, errlen, "f_dc_2 not found"); rc = -2; goto done; }
        } else {
            cr = FIND("red"); if (cr < 0) { snprintf(err, errlen, "red not found"); rc = -2; goto done; }
            cg = FIND("green"); if (cg < 0) { snprintf(err, errlen, "green not found"); rc = -2; goto done; }
            cb = FIND("blue"); if (cb < 0) { snprintf(err, errlen, "blue not found"); rc = -2; goto done; }
        }
        int op = FIND("opacity");
        for (size_t j = 0; j < n; j++) {                                    /* index.js:680-742 */
            const size_t r = imp[j].idx; uint8_t *o = out + 32 * j; float f[6];
            if (has_scale) {
                const double q0 = AT(r, r0), q1 = AT(r, r1), q2 = AT(r, r2), q3 = AT(r, r3);
                const double qlen = sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
                o[28] = clamp_u8((q0 / qlen) * 128 + 128); o[29] = clamp_u8((q1 / qlen) * 128 + 128);
                o[30] = clamp_u8((q2 / qlen) * 128 + 128); o[31] = clamp_u8((q3 / qlen) * 128 + 128);
                f[3] = (float)gso_js_exp(AT(r, s0)); f[4] = (float)gso_js_exp(AT(r, s1)); f[5] = (float)gso_js_exp(AT(r, s2));
            } else {
                f[3] = f[4] = f[5] = (float)0.01; o[28] = 255; o[29] = o[30] = o[31] = 0;
            }
            f[0] = (float)AT(r, px); f[1] = (float)AT(r, py); f[2] = (float)AT(r, pz);
            memcpy(o, f, 24);
            if (dc0 >= 0) {
                const double SH_C0 = 0.28209479177387814;
                o[24] = clamp_u8((0.5 + SH_C0 * AT(r, dc0)) * 255); o[25] = clamp_u8((0.5 + SH_C0 * AT(r, dc1)) * 255);
                o[26] = clamp_u8((0.5 + SH_C0 * AT(r, dc2)) * 255);
            } else { o[24] = clamp_u8(AT(r, cr)); o[25] = clamp_u8(AT(r, cg)); o[26] = clamp_u8(AT(r, cb)); }
            o[27] = op >= 0 ? clamp_u8((1 / (1 + gso_js_exp(-AT(r, op)))) * 255) : 255;
        }
    }
done:
    free(imp); free(tmp);
    return rc;
}

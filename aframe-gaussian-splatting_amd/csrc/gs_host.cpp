// gs_host.cpp -- host-side pieces of the boundary that are not on the per-frame GPU path:
//   * the uniform producers (tick / getModelViewMatrix / getProjectionMatrix, index.js:438-487): a dozen 4x4
//     f64 operations per frame, in three.js' operation order so the f32 uniforms match the reference's;
//   * processPlyBuffer (index.js:600-745): the one-time .ply -> .splat row conversion.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#include "../../include/gs_splat.h"

namespace {

// three.js Matrix4.multiplyMatrices: o = a * b, column-major, each element summed left to right
void mat_mul(const double *a, const double *b, double *o)
{
    double r[16];
    for (int c = 0; c < 4; c++)
        for (int rw = 0; rw < 4; rw++)
            r[c * 4 + rw] = a[rw] * b[c * 4] + a[4 + rw] * b[c * 4 + 1] + a[8 + rw] * b[c * 4 + 2] + a[12 + rw] * b[c * 4 + 3];
    memcpy(o, r, sizeof r);
}

// three.js Matrix4.invert (cofactor expansion in three.js' term order; det == 0 -> zero matrix)
void mat_inv(const double *m, double *o)
{
    const double n11 = m[0], n21 = m[1], n31 = m[2], n41 = m[3], n12 = m[4], n22 = m[5], n32 = m[6], n42 = m[7];
    const double n13 = m[8], n23 = m[9], n33 = m[10], n43 = m[11], n14 = m[12], n24 = m[13], n34 = m[14], n44 = m[15];
    const double t11 = n23 * n34 * n42 - n24 * n33 * n42 + n24 * n32 * n43 - n22 * n34 * n43 - n23 * n32 * n44 + n22 * n33 * n44;
    const double t12 = n14 * n33 * n42 - n13 * n34 * n42 - n14 * n32 * n43 + n12 * n34 * n43 + n13 * n32 * n44 - n12 * n33 * n44;
    const double t13 = n13 * n24 * n42 - n14 * n23 * n42 + n14 * n22 * n43 - n12 * n24 * n43 - n13 * n22 * n44 + n12 * n23 * n44;
    const double t14 = n14 * n23 * n32 - n13 * n24 * n32 - n14 * n22 * n33 + n12 * n24 * n33 + n13 * n22 * n34 - n12 * n23 * n34;
    const double det = n11 * t11 + n21 * t12 + n31 * t13 + n41 * t14;
    if (det == 0) { for (int i = 0; i < 16; i++) o[i] = 0; return; }
    const double s = 1 / det;
    double r[16];
    r[0] = t11 * s;
    r[1] = (n24 * n33 * n41 - n23 * n34 * n41 - n24 * n31 * n43 + n21 * n34 * n43 + n23 * n31 * n44 - n21 * n33 * n44) * s;
    r[2] = (n22 * n34 * n41 - n24 * n32 * n41 + n24 * n31 * n42 - n21 * n34 * n42 - n22 * n31 * n44 + n21 * n32 * n44) * s;
    r[3] = (n23 * n32 * n41 - n22 * n33 * n41 - n23 * n31 * n42 + n21 * n33 * n42 + n22 * n31 * n43 - n21 * n32 * n43) * s;
    r[4] = t12 * s;
    r[5] = (n13 * n34 * n41 - n14 * n33 * n41 + n14 * n31 * n43 - n11 * n34 * n43 - n13 * n31 * n44 + n11 * n33 * n44) * s;
    r[6] = (n14 * n32 * n41 - n12 * n34 * n41 - n14 * n31 * n42 + n11 * n34 * n42 + n12 * n31 * n44 - n11 * n32 * n44) * s;
    r[7] = (n12 * n33 * n41 - n13 * n32 * n41 + n13 * n31 * n42 - n11 * n33 * n42 - n12 * n31 * n43 + n11 * n32 * n43) * s;
    r[8] = t13 * s;
    r[9] = (n14 * n23 * n41 - n13 * n24 * n41 - n14 * n21 * n43 + n11 * n24 * n43 + n13 * n21 * n44 - n11 * n23 * n44) * s;
    r[10] = (n12 * n24 * n41 - n14 * n22 * n41 + n14 * n21 * n42 - n11 * n24 * n42 - n12 * n21 * n44 + n11 * n22 * n44) * s;
    r[11] = (n13 * n22 * n41 - n12 * n23 * n41 - n13 * n21 * n42 + n11 * n23 * n42 + n12 * n21 * n43 - n11 * n22 * n43) * s;
    r[12] = t14 * s;
    r[13] = (n13 * n24 * n31 - n14 * n23 * n31 + n14 * n21 * n33 - n11 * n24 * n33 - n13 * n21 * n34 + n11 * n23 * n34) * s;
    r[14] = (n14 * n22 * n31 - n12 * n24 * n31 - n14 * n21 * n32 + n11 * n24 * n32 + n12 * n21 * n34 - n11 * n22 * n34) * s;
    r[15] = (n12 * n23 * n31 - n13 * n22 * n31 + n13 * n21 * n32 - n11 * n23 * n32 - n12 * n21 * n33 + n11 * n22 * n33) * s;
    memcpy(o, r, sizeof r);
}

// conjugation by S = diag(1,-1,1,1) as the reference spells it (index.js:472-476, 479-483)
void flip_y(double *e) { e[1] *= -1.0; e[4] *= -1.0; e[6] *= -1.0; e[9] *= -1.0; e[13] *= -1.0; }

// ---------------------------------------------------------------- PLY

enum PType { P_F64, P_I32, P_U32, P_F32, P_I16, P_U16, P_U8, P_I8 };
const size_t kSize[] = { 8, 4, 4, 4, 2, 2, 1, 1 };

struct Prop { std::string name; PType type; size_t offset; };

struct Header {
    std::vector<Prop> props;
    size_t row_bytes = 0, data_start = 0, vertex_count = 0;
    const Prop *find(const char *name) const
    {
        const Prop *hit = nullptr;
        for (const Prop &p : props) if (p.name == name) hit = &p;     // later duplicates win (JS object assignment)
        return hit;
    }
};

PType parse_type(const std::string &t)       // TYPE_MAP, anything else reads as getInt8 (index.js:613-628)
{
    if (t == "double") return P_F64; if (t == "int") return P_I32; if (t == "uint") return P_U32; if (t == "float") return P_F32;
    if (t == "short") return P_I16; if (t == "ushort") return P_U16; if (t == "uchar") return P_U8;
    return P_I8;
}

double read_le(const uint8_t *p, PType t)
{
    switch (t) {
    case P_F64: { double v; memcpy(&v, p, 8); return v; }
    case P_I32: { int32_t v; memcpy(&v, p, 4); return v; }
    case P_U32: { uint32_t v; memcpy(&v, p, 4); return v; }
    case P_F32: { float v; memcpy(&v, p, 4); return v; }
    case P_I16: { int16_t v; memcpy(&v, p, 2); return v; }
    case P_U16: { uint16_t v; memcpy(&v, p, 2); return v; }
    case P_U8: return *p;
    default: return (int8_t)*p;
    }
}

// Uint8ClampedArray element store: clamp to [0,255], round half to even, NaN -> 0
uint8_t to_clamped_u8(double v)
{
    if (!(v > 0)) return 0;
    if (v >= 255) return 255;
    const double f = floor(v), d = v - f;
    if (d > 0.5) return (uint8_t)(f + 1);
    if (d < 0.5) return (uint8_t)f;
    return (uint8_t)((((int)f) & 1) ? f + 1 : f);
}

int fail(char *err, size_t errlen, int code, const char *fmt, const char *arg = "")
{
    if (err && errlen) snprintf(err, errlen, fmt, arg);
    return code;
}

int parse_header(const uint8_t *buf, size_t len, Header &h, char *err, size_t errlen)
{
    const size_t hl = std::min<size_t>(len, 10240);                  // "10KB ought to be enough for a header"
    const std::string head((const char *)buf, hl);
    const size_t end = head.find("end_header\n");
    if (end == std::string::npos) return fail(err, errlen, GS_E_PLY_HEADER, "Unable to read .ply file header");
    // /element vertex (\d+)\n/ over the decoded 10 KiB
    bool have_count = false;
    for (size_t at = head.find("element vertex "); at != std::string::npos && !have_count; at = head.find("element vertex ", at + 1)) {
        size_t j = at + 15, v = 0, nd = 0;
        while (j < hl && head[j] >= '0' && head[j] <= '9') { v = v * 10 + (size_t)(head[j] - '0'); j++; nd++; }
        if (nd && j < hl && head[j] == '\n') { h.vertex_count = v; have_count = true; }
    }
    if (!have_count) return fail(err, errlen, GS_E_PLY_HEADER, "Unable to read .ply file header");
    size_t ls = 0;
    while (ls < end) {
        size_t le = head.find('\n', ls);
        if (le == std::string::npos || le > end) le = end;
        const std::string line = head.substr(ls, le - ls);
        if (line.compare(0, 9, "property ") == 0) {                  // const [p, type, name] = prop.split(" ")
            std::vector<std::string> parts;
            size_t s = 0;
            while (parts.size() < 3) {
                const size_t sp = line.find(' ', s);
                parts.push_back(line.substr(s, sp == std::string::npos ? std::string::npos : sp - s));
                if (sp == std::string::npos) break;
                s = sp + 1;
            }
            const PType t = parse_type(parts.size() > 1 ? parts[1] : "");
            h.props.push_back({ parts.size() > 2 ? parts[2] : "undefined", t, h.row_bytes });
            h.row_bytes += kSize[t];
        }
        ls = le + 1;
    }
    h.data_start = end + 11;
    return GS_OK;
}

}  // namespace

extern "C" {

GS_API void gs_model_view_matrix(const double cam_world[16], const double obj_world[16], double out[16])
{
    double view[16], m[16];
    memcpy(view, cam_world, sizeof view); flip_y(view);              // viewMatrix = camera.matrixWorld, flipped
    mat_inv(obj_world, m); flip_y(m);                                // mtx = object.matrixWorld^-1, flipped
    mat_mul(m, view, m);                                             // mtx.multiply(viewMatrix)
    mat_inv(m, out);                                                 // mtx.invert()
}

GS_API void gs_projection_matrix(const double proj[16], double out[16])
{
    memcpy(out, proj, 16 * sizeof(double));
    out[4] *= -1; out[5] *= -1; out[6] *= -1; out[7] *= -1;
}

GS_API void gs_tick_uniforms(const double cam_world[16], const double obj_world[16], const double *cutout_world, float view[4],
                             float cutout[16])
{
    double mv[16];
    gs_model_view_matrix(cam_world, obj_world, mv);
    view[0] = (float)mv[2]; view[1] = (float)mv[6]; view[2] = (float)mv[10]; view[3] = (float)mv[14];
    if (cutout_world && cutout) {
        double w[16];
        mat_inv(cutout_world, w);                                    // worldToCutout.copy(cutout.matrixWorld).invert()
        mat_mul(w, obj_world, w);                                    // .multiply(object.matrixWorld)
        for (int i = 0; i < 16; i++) cutout[i] = (float)w[i];
    }
}

GS_API double gs_focal(const double gs_proj[16], double viewport_h) { return (viewport_h / 2.0) * fabs(gs_proj[5]); }

GS_API void gs_scaled_size(int css_w, int css_h, double ratio, int *out_w, int *out_h)
{
    // renderer.setPixelRatio / xr.setFramebufferScaleFactor are only applied when the property is > 0
    // (index.js:10-15); three.js sizes the drawing buffer as floor(css * pixelRatio).
    if (ratio > 0) { css_w = (int)floor(css_w * ratio); css_h = (int)floor(css_h * ratio); }
    if (out_w) *out_w = css_w;
    if (out_h) *out_h = css_h;
}

GS_API int gs_ply_to_splat(const void *bytes, size_t nbytes, void *out_rows, size_t *out_nrows, char *err, size_t errlen)
{
    if (!bytes || !out_nrows) return fail(err, errlen, GS_E_BADARG, "gs_ply_to_splat: NULL argument");
    const uint8_t *buf = (const uint8_t *)bytes;
    Header h;
    int rc = parse_header(buf, nbytes, h, err, errlen);
    if (rc != GS_OK) return rc;
    const size_t n = h.vertex_count;
    const uint8_t *data = buf + h.data_start;
    if (n && h.row_bytes * n > nbytes - h.data_start)
        return fail(err, errlen, GS_E_PLY_DATA, "Offset is outside the bounds of the DataView");
    auto at = [&](size_t row, const Prop *p) { return read_le(data + row * h.row_bytes + p->offset, p->type); };
#define NEED(var, nm) const Prop *var = h.find(nm); if (!var) return fail(err, errlen, GS_E_PLY_PROP, "%s not found", nm)

    // importance = exp(s0)*exp(s1)*exp(s2) * sigmoid(opacity), stored f32; 0 when there is no scale_0 (index.js:653-664)
    const Prop *scale0 = h.find("scale_0");
    std::vector<float> importance(n, 0.0f);
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = (uint32_t)i;
    if (scale0 && n) {
        NEED(scale1, "scale_1"); NEED(scale2, "scale_2"); NEED(opac, "opacity");
        for (size_t i = 0; i < n; i++) {
            const double size = exp(at(i, scale0)) * exp(at(i, scale1)) * exp(at(i, scale2));
            const double opacity = 1 / (1 + exp(-at(i, opac)));
            importance[i] = (float)(size * opacity);
        }
    }
    // sizeIndex.sort((b, a) => sizeList[a] - sizeList[b]): descending, stable (index.js:668)
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t b, uint32_t a) { return (double)importance[a] - (double)importance[b] < 0; });
    *out_nrows = n;
    if (!out_rows || !n) return GS_OK;

    const Prop *rot[4] = { nullptr, nullptr, nullptr, nullptr }, *sc[3] = { nullptr, nullptr, nullptr };
    if (scale0) {
        static const char *rn[4] = { "rot_0", "rot_1", "rot_2", "rot_3" }, *sn[3] = { "scale_0", "scale_1", "scale_2" };
        for (int k = 0; k < 4; k++) { rot[k] = h.find(rn[k]); if (!rot[k]) return fail(err, errlen, GS_E_PLY_PROP, "%s not found", rn[k]); }
        for (int k = 0; k < 3; k++) { sc[k] = h.find(sn[k]); if (!sc[k]) return fail(err, errlen, GS_E_PLY_PROP, "%s not found", sn[k]); }
    }
    NEED(px, "x"); NEED(py, "y"); NEED(pz, "z");
    const Prop *dc[3] = { h.find("f_dc_0"), nullptr, nullptr }, *col[3] = { nullptr, nullptr, nullptr };
    if (dc[0]) {
        dc[1] = h.find("f_dc_1"); if (!dc[1]) return fail(err, errlen, GS_E_PLY_PROP, "%s not found", "f_dc_1");
        dc[2] = h.find("f_dc_2"); if (!dc[2]) return fail(err, errlen, GS_E_PLY_PROP, "%s not found", "f_dc_2");
    } else {
        static const char *cn[3] = { "red", "green", "blue" };
        for (int k = 0; k < 3; k++) { col[k] = h.find(cn[k]); if (!col[k]) return fail(err, errlen, GS_E_PLY_PROP, "%s not found", cn[k]); }
    }
    const Prop *opac = h.find("opacity");
#undef NEED
    uint8_t *out = (uint8_t *)out_rows;
    for (size_t j = 0; j < n; j++) {                                  // index.js:680-742
        const size_t r = order[j];
        uint8_t *o = out + 32 * j;
        float f[6];
        if (scale0) {
            const double q0 = at(r, rot[0]), q1 = at(r, rot[1]), q2 = at(r, rot[2]), q3 = at(r, rot[3]);
            const double qlen = sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
            o[28] = to_clamped_u8((q0 / qlen) * 128 + 128); o[29] = to_clamped_u8((q1 / qlen) * 128 + 128);
            o[30] = to_clamped_u8((q2 / qlen) * 128 + 128); o[31] = to_clamped_u8((q3 / qlen) * 128 + 128);
            for (int k = 0; k < 3; k++) f[3 + k] = (float)exp(at(r, sc[k]));
        } else {
            f[3] = f[4] = f[5] = (float)0.01;
            o[28] = 255; o[29] = o[30] = o[31] = 0;
        }
        f[0] = (float)at(r, px); f[1] = (float)at(r, py); f[2] = (float)at(r, pz);
        memcpy(o, f, 24);
        if (dc[0]) {
            const double SH_C0 = 0.28209479177387814;
            for (int k = 0; k < 3; k++) o[24 + k] = to_clamped_u8((0.5 + SH_C0 * at(r, dc[k])) * 255);
        } else {
            for (int k = 0; k < 3; k++) o[24 + k] = to_clamped_u8(at(r, col[k]));
        }
        o[27] = opac ? to_clamped_u8((1 / (1 + exp(-at(r, opac)))) * 255) : 255;
    }
    return GS_OK;
}

}  // extern "C"

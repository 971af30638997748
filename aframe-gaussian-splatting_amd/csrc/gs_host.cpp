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
#include "gs_ply.h"

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

typedef gsm::PlyType PType;
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
    if (t == "double") return gsm::PLY_F64; if (t == "int") return gsm::PLY_I32; if (t == "uint") return gsm::PLY_U32;
    if (t == "float") return gsm::PLY_F32; if (t == "short") return gsm::PLY_I16; if (t == "ushort") return gsm::PLY_U16;
    if (t == "uchar") return gsm::PLY_U8;
    return gsm::PLY_I8;
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

// Header parse + resolution of every property processPlyBuffer reads, with the reference's error messages in the
// reference's order (index.js:606-607 header, :643 "<prop> not found").  Shared by the host converter below and the
// HIP converter (gs_ply.hip).
static int ply_plan(const void *bytes, size_t nbytes, gsm::PlyLayout *layout, size_t *nrows, size_t *data_start, char *err, size_t errlen)
{
    const uint8_t *buf = (const uint8_t *)bytes;
    Header h;
    int rc = parse_header(buf, nbytes, h, err, errlen);
    if (rc != GS_OK) return rc;
    const size_t n = h.vertex_count;
    if (n && h.row_bytes * n > nbytes - h.data_start)
        return fail(err, errlen, GS_E_PLY_DATA, "Offset is outside the bounds of the DataView");
    gsm::PlyLayout L;
    memset(&L, 0, sizeof L);
    L.row_bytes = (uint32_t)h.row_bytes;
    auto need = [&](int slot, const char *nm) -> bool {
        const Prop *p = h.find(nm);
        if (!p) { fail(err, errlen, GS_E_PLY_PROP, "%s not found", nm); return false; }
        L.offset[slot] = (uint32_t)p->offset; L.type[slot] = (uint8_t)p->type;
        return true;
    };
    L.has_scale = h.find("scale_0") != nullptr;
    L.has_dc = h.find("f_dc_0") != nullptr;
    L.has_opacity = h.find("opacity") != nullptr;
    *nrows = n; *data_start = h.data_start; *layout = L;
    // importance pass (index.js:656-664) touches scale_0..2 and opacity; it only runs when there are rows
    if (L.has_scale && n) {
        if (!need(gsm::PP_S0, "scale_0") || !need(gsm::PP_S1, "scale_1") || !need(gsm::PP_S2, "scale_2") || !need(gsm::PP_OPACITY, "opacity"))
            return GS_E_PLY_PROP;
    }
    if (!n) return GS_OK;
    // row pass (index.js:680-742)
    if (L.has_scale) {
        if (!need(gsm::PP_R0, "rot_0") || !need(gsm::PP_R1, "rot_1") || !need(gsm::PP_R2, "rot_2") || !need(gsm::PP_R3, "rot_3")) return GS_E_PLY_PROP;
        if (!need(gsm::PP_S0, "scale_0") || !need(gsm::PP_S1, "scale_1") || !need(gsm::PP_S2, "scale_2")) return GS_E_PLY_PROP;
    }
    if (!need(gsm::PP_X, "x") || !need(gsm::PP_Y, "y") || !need(gsm::PP_Z, "z")) return GS_E_PLY_PROP;
    if (L.has_dc) { if (!need(gsm::PP_C0, "f_dc_0") || !need(gsm::PP_C1, "f_dc_1") || !need(gsm::PP_C2, "f_dc_2")) return GS_E_PLY_PROP; }
    else { if (!need(gsm::PP_C0, "red") || !need(gsm::PP_C1, "green") || !need(gsm::PP_C2, "blue")) return GS_E_PLY_PROP; }
    if (L.has_opacity && !need(gsm::PP_OPACITY, "opacity")) return GS_E_PLY_PROP;
    *layout = L;
    return GS_OK;
}

// no C++ exception crosses the C ABI: the header strings and the order arrays below are the only host allocations
int gs_ply_plan(const void *bytes, size_t nbytes, gsm::PlyLayout *layout, size_t *nrows, size_t *data_start, char *err, size_t errlen)
{
    try { return ply_plan(bytes, nbytes, layout, nrows, data_start, err, errlen); }
    catch (...) { return fail(err, errlen, GS_E_OOM, "out of host memory while reading the .ply header"); }
}

static int ply_to_splat(const void *bytes, size_t nbytes, void *out_rows, size_t *out_nrows, char *err, size_t errlen)
{
    if (!bytes || !out_nrows) return fail(err, errlen, GS_E_BADARG, "gs_ply_to_splat: NULL argument");
    gsm::PlyLayout L;
    size_t n = 0, data_start = 0;
    // the size query (out_rows == NULL) stops where the reference would have thrown so far: header and importance pass
    int rc = gs_ply_plan(bytes, nbytes, &L, &n, &data_start, err, errlen);
    *out_nrows = n;
    if (rc != GS_OK) return rc;
    if (!out_rows || !n) return GS_OK;
    const uint8_t *data = (const uint8_t *)bytes + data_start;

    // importance = exp(s0)*exp(s1)*exp(s2) * sigmoid(opacity), stored f32; 0 when there is no scale_0 (index.js:653-664)
    std::vector<float> importance(n, 0.0f);
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = (uint32_t)i;
    if (L.has_scale)
        for (size_t i = 0; i < n; i++) importance[i] = gsm::ply_importance(data + i * L.row_bytes, L);
    // sizeIndex.sort((b, a) => sizeList[a] - sizeList[b]): descending, stable (index.js:668)
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t b, uint32_t a) { return (double)importance[a] - (double)importance[b] < 0; });
    uint32_t *out = (uint32_t *)out_rows;
    for (size_t j = 0; j < n; j++) {                                  // index.js:680-742
        uint32_t w[8];
        gsm::ply_row(data + (size_t)order[j] * L.row_bytes, L, w);
        memcpy(out + 8 * j, w, 32);
    }
    return GS_OK;
}

GS_API int gs_ply_to_splat(const void *bytes, size_t nbytes, void *out_rows, size_t *out_nrows, char *err, size_t errlen)
{
    try { return ply_to_splat(bytes, nbytes, out_rows, out_nrows, err, errlen); }
    catch (...) { return fail(err, errlen, GS_E_OOM, "out of host memory while converting the .ply"); }
}

}  // extern "C"

// TEST INFRASTRUCTURE.  Compiles the product's per-splat arithmetic header
// (csrc/gs_device_math.h) for the HOST so its formulas can be compared with the
// oracle and the golden vectors in the CPU-only test tier.  Nothing here is
// part of, linked into or reachable from the product library.
#include <string.h>
#include <vector>
#include "gs_device_math.h"
#include "gs_ply.h"
#include "gs_host_tables.h"

extern "C" {

__attribute__((visibility("default")))
void hc_pack(const uint8_t *rows, size_t n, float *cs, uint32_t *cc, float *sort_rows)
{
    std::vector<double> tab(GS_POW10_ENTRIES);
    gs_build_pow10_table(tab.data());
    for (size_t i = 0; i < n; i++) {
        uint32_t w[8]; memcpy(w, rows + 32 * i, 32);
        gsm::PackOut o; gsm::pack_row(w, tab.data(), o);
        memcpy(cs + 4 * i, o.cs, 16); memcpy(cc + 4 * i, o.cc, 16); memcpy(sort_rows + 4 * i, o.sort_row, 16);
    }
}

// the sort restated with the product's key functions + a trivially correct stable sort
__attribute__((visibility("default")))
size_t hc_sort(const float *rows4, size_t n, const float *view, const float *cutout, uint32_t *out)
{
    std::vector<float> depth(n); std::vector<uint32_t> idx; idx.reserve(n);
    double mn = INFINITY, mx = -INFINITY;
    for (size_t i = 0; i < n; i++) {
        const float *r = rows4 + 4 * i;
        const double d = gsm::view_depth(view, r[0], r[1], r[2]);
        const bool inside = cutout ? gsm::in_cutout(cutout, r[0], r[1], r[2]) : true;
        if (gsm::sort_keep(d, r[3], inside)) { depth[i] = (float)d; idx.push_back((uint32_t)i); if (d > mx) mx = d; if (d < mn) mn = d; }
    }
    // round-trip the extrema through the ordered-u64 encoding the GPU atomics use
    mn = gsm::ordered_to_f64(gsm::f64_to_ordered(mn)); mx = gsm::ordered_to_f64(gsm::f64_to_ordered(mx));
    const double inv = 65535.0 / (mx - mn);
    std::vector<std::vector<uint32_t>> bins(65536);
    for (uint32_t i : idx) { int32_t b = gsm::sort_bucket(depth[i], mn, inv); if (b >= 0) bins[b].push_back(i); }
    size_t k = 0;
    for (auto &b : bins) for (uint32_t i : b) out[k++] = i;
    for (; k < idx.size(); k++) out[k] = 0;
    return idx.size();
}

__attribute__((visibility("default")))
int hc_project(const float *cs, const uint32_t *cc, uint32_t idx, const float *mv, const float *P, float focal,
               float vw, float vh, float *out16)
{
    gsm::Projected p; gsm::ProjExtra x;
    memset(&p, 0, sizeof p); memset(&x, 0, sizeof x);
    const bool vis = gsm::project_splat(cs + 4 * idx, cc + 4 * idx, mv, P, focal, vw, vh, p, x);
    if (!vis) return 0;
    out16[0] = p.cx; out16[1] = p.cy; out16[2] = p.ax; out16[3] = p.ay; out16[4] = p.bx; out16[5] = p.by;
    out16[6] = x.v1x; out16[7] = x.v1y; out16[8] = x.v2x; out16[9] = x.v2y; out16[10] = x.zndc; out16[11] = p.alpha;
    memcpy(out16 + 12, &p.rgba, 4);
    float b[4]; gsm::splat_pixel_bounds(p, x, b[0], b[1], b[2], b[3]);
    // ints as floats for transport
    out16[13] = b[0]; out16[14] = b[1]; out16[15] = b[2]; out16[16] = b[3];
    return 1;
}


// exact per-tile-row coverage: returns rows written; out[3*k] = ty, tx0, n.  Also the AABB rect in rect[4].
__attribute__((visibility("default")))
int hc_tile_rows(const float *rec8, int W, int H, int x0, int x1, int *out, int max_rows, int *rect)
{
    gsm::Projected p; memcpy(&p, rec8, 32);
    gsm::ProjExtra x; memcpy(&x, rec8 + 8, 5 * sizeof(float));
    float xmin, xmax, ymin, ymax;
    gsm::splat_pixel_bounds(p, x, xmin, xmax, ymin, ymax);
    const float fy0 = fmaxf(ymin, 0.0f), fy1 = fminf(ymax, (float)(H - 1));
    const float fx0 = fmaxf(xmin, (float)x0), fx1 = fminf(xmax, (float)(x1 - 1));
    if (!(fx0 <= fx1 && fy0 <= fy1)) return 0;
    const int r0 = H - 1 - (int)fy1, r1 = H - 1 - (int)fy0;
    rect[0] = ((int)fx0 - x0) / 16; rect[1] = r0 / 16; rect[2] = ((int)fx1 - x0) / 16; rect[3] = r1 / 16;
    gsm::EllipseRows e; gsm::ellipse_rows_setup(p, e);
    int k = 0;
    for (int ty = r0 / 16; ty <= r1 / 16 && k < max_rows; ty++) {
        uint32_t tx0, n; gsm::splat_tile_row(p, e, ty, H, x0, x1, tx0, n);
        out[3 * k] = ty; out[3 * k + 1] = (int)tx0; out[3 * k + 2] = (int)n; k++;
    }
    return k;
}

__attribute__((visibility("default")))
float hc_frag_power(float dx, float dy, float ax, float ay, float bx, float by) { return gsm::frag_power(dx, dy, ax, ay, bx, by); }

__attribute__((visibility("default")))
int32_t hc_toint32(double d) { return gsm::js_toint32(d); }

__attribute__((visibility("default")))
int hc_xcd_chunk(uint32_t v, uint32_t nchunks, uint32_t *chunk) { return gsm::xcd_chunk(v, nchunks, *chunk) ? 1 : 0; }

// positions (x, y, z) x n against an affine cut-out matrix: how many differ between the general test and the affine path
__attribute__((visibility("default")))
size_t hc_cutout_affine_mismatches(const float *pos3, size_t n, const double *c16)
{
    size_t bad = 0;
    for (size_t i = 0; i < n; i++)
        bad += gsm::in_cutout(c16, pos3[3 * i], pos3[3 * i + 1], pos3[3 * i + 2]) != gsm::in_cutout_affine(c16, pos3[3 * i], pos3[3 * i + 1], pos3[3 * i + 2]);
    return bad;
}

__attribute__((visibility("default")))
void hc_js_exp(const double *x, size_t n, double *out) { for (size_t i = 0; i < n; i++) out[i] = gsm::js_exp(x[i]); }
}

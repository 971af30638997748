/*
 * gs_splat_napi.c -- raw-C N-API addon over the C ABI (include/gs_splat.h).
 *
 * This is the binding a maintainer of the reference would add so that the component's hot path
 * (Worker sort + WebGL draw, index.js:77-207, 438-598) runs on an MI355X and hands an RGBA framebuffer
 * back to JavaScript.  It is deliberately thin: every function converts arguments, calls one gs_* entry
 * point and converts the result; failures become thrown JS Errors carrying gs_last_error() (for the two
 * PLY cases that is the reference's own message, index.js:606-607, 643).
 *
 * Build (no node-gyp): see aframe-gaussian-splatting_amd/build.py:build_addon().
 */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gs_splat.h"

#define NAPI_OK(call)                                                                   \
    do { if ((call) != napi_ok) { napi_throw_error(env, NULL, "N-API call failed: " #call); return NULL; } } while (0)

static napi_value throw_gs(napi_env env, gs_ctx *ctx, int rc)
{
    char code[16];
    const char *msg = gs_last_error(ctx);
    snprintf(code, sizeof code, "GS%d", rc);
    napi_throw_error(env, code, (msg && *msg) ? msg : "gs_splat call failed");
    return NULL;
}

/* What a JS handle points to.  destroy() frees the context at once (HBM, streams, worker threads) and leaves the shell for
 * the garbage collector; `busy` is set while an asynchronous call (sortAsync / renderAsync) owns the context: a gs_ctx is
 * single-caller, like the reference's single-flight worker (sortReady, index.js:206, 220, 439-440). */
/* Frames queued with GS_RENDER_ASYNC are written by the GPU (a copy behind the frame's kernels) until the next sync: the
 * handle holds a reference to each such frame until then, so that the garbage collector cannot finalize -- and, for
 * allocFrame's page-locked memory, free -- a buffer a copy is still in flight to. */
typedef struct gs_held { napi_ref *refs; size_t n, cap; } gs_held;
static int held_add(napi_env env, gs_held *h, napi_value v)
{
    if (h->n == h->cap) {
        const size_t cap = h->cap ? h->cap * 2 : 64;
        napi_ref *r = (napi_ref *)realloc(h->refs, cap * sizeof *r);
        if (!r) return 0;
        h->refs = r; h->cap = cap;
    }
    if (napi_create_reference(env, v, 1, &h->refs[h->n]) != napi_ok) return 0;
    h->n++;
    return 1;
}
static void held_release(napi_env env, gs_held *h)               /* the GPU is done with them: after a sync, a clear, a destroy */
{
    for (size_t i = 0; i < h->n; i++) napi_delete_reference(env, h->refs[i]);
    h->n = 0;
}
static void held_free(napi_env env, gs_held *h) { held_release(env, h); free(h->refs); h->refs = NULL; h->cap = 0; }

/* The index list a sort hands back (the worker's reply {sortedIndexes}, index.js:587-596) lands in page-locked memory owned by the handle
 * and wrapped ONCE in an external ArrayBuffer: a tick costs no allocation and no copy besides the device-to-host one (round 5: a
 * malloc, the copy into a fresh ArrayBuffer and its zero fill were a third of tick() + render() at 1 M splats).  The Uint32Array a
 * sort returns is therefore overwritten by the next sort of the same context -- the reference's consumer copies the reply into its
 * instanced attribute at once (index.js:201-203).  The buffer grows with the scene; an outgrown one stays alive as long as JavaScript
 * holds an array over it (its finalizer frees the page-locked memory). */
typedef struct gs_handle { gs_ctx *ctx; int busy; gs_held held; uint32_t *idx; size_t idx_cap; napi_ref idx_ab; } gs_handle;

static void ctx_finalize(napi_env env, void *data, void *hint)
{
    (void)hint;
    gs_handle *h = (gs_handle *)data;
    if (!h) return;
    if (h->ctx) gs_destroy(h->ctx);                              /* (drains the device first) */
    if (h->idx_ab) napi_delete_reference(env, h->idx_ab);         /* (the ArrayBuffer's finalizer frees the memory) */
    held_free(env, &h->held);
    free(h);
}

/* argument helpers ------------------------------------------------------------------------------- */

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv, size_t *got)
{
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok) return 0;
    for (size_t i = argc; i < want; i++) napi_get_undefined(env, &argv[i]);
    if (got) *got = argc;
    return 1;
}

static gs_handle *get_handle(napi_env env, napi_value v)
{
    void *p = NULL;
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok || t != napi_external || napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "expected a context handle returned by create()");
        return NULL;
    }
    return (gs_handle *)p;
}

static gs_ctx *get_ctx(napi_env env, napi_value v)
{
    gs_handle *h = get_handle(env, v);
    if (!h) return NULL;
    if (!h->ctx) { napi_throw_error(env, "GS_DESTROYED", "the context has been destroyed"); return NULL; }
    if (h->busy) { napi_throw_error(env, "GS_BUSY", "an asynchronous sort or render of this context is in flight"); return NULL; }
    return h->ctx;
}

/* bytes of an ArrayBuffer / TypedArray / DataView / Buffer; returns 0 if v is none of those */
static int get_bytes(napi_env env, napi_value v, void **data, size_t *len)
{
    bool is = false;
    if (napi_is_arraybuffer(env, v, &is) == napi_ok && is) return napi_get_arraybuffer_info(env, v, data, len) == napi_ok;
    if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
        napi_typedarray_type ty; size_t n; napi_value ab; size_t off;
        if (napi_get_typedarray_info(env, v, &ty, &n, data, &ab, &off) != napi_ok) return 0;
        static const size_t esz[] = { 1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8 };
        *len = n * esz[ty];
        return 1;
    }
    if (napi_is_buffer(env, v, &is) == napi_ok && is) return napi_get_buffer_info(env, v, data, len) == napi_ok;
    if (napi_is_dataview(env, v, &is) == napi_ok && is) {
        napi_value ab; size_t off;
        return napi_get_dataview_info(env, v, len, data, &ab, &off) == napi_ok;
    }
    return 0;
}

static int is_nullish(napi_env env, napi_value v)
{
    napi_valuetype t;
    return napi_typeof(env, v, &t) != napi_ok || t == napi_undefined || t == napi_null;
}

/* array-like of numbers -> doubles (accepts Array, Float32Array, Float64Array, THREE.Matrix4.elements) */
static int get_doubles(napi_env env, napi_value v, double *out, uint32_t n)
{
    bool is = false;
    if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
        napi_typedarray_type ty; size_t len; void *data; napi_value ab; size_t off;
        if (napi_get_typedarray_info(env, v, &ty, &len, &data, &ab, &off) != napi_ok || len < n) return 0;
        if (ty == napi_float32_array) { for (uint32_t i = 0; i < n; i++) out[i] = ((float *)data)[i]; return 1; }
        if (ty == napi_float64_array) { for (uint32_t i = 0; i < n; i++) out[i] = ((double *)data)[i]; return 1; }
        return 0;
    }
    for (uint32_t i = 0; i < n; i++) {
        napi_value e;
        if (napi_get_element(env, v, i, &e) != napi_ok || napi_get_value_double(env, e, &out[i]) != napi_ok) return 0;
    }
    return 1;
}

static int get_floats(napi_env env, napi_value v, float *out, uint32_t n)
{
    double d[16];
    if (n > 16 || !get_doubles(env, v, d, n)) return 0;
    for (uint32_t i = 0; i < n; i++) out[i] = (float)d[i];
    return 1;
}

static napi_value make_f64_array(napi_env env, const double *v, size_t n)
{
    napi_value ab, ta; void *data;
    NAPI_OK(napi_create_arraybuffer(env, n * 8, &data, &ab));
    memcpy(data, v, n * 8);
    NAPI_OK(napi_create_typedarray(env, napi_float64_array, n, ab, 0, &ta));
    return ta;
}

static int get_named_double(napi_env env, napi_value obj, const char *name, double *out, int required)
{
    napi_value v; bool has = false;
    if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return !required;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok) return 0;
    if (is_nullish(env, v)) return !required;
    return napi_get_value_double(env, v, out) == napi_ok;
}

/* functions ---------------------------------------------------------------------------------------- */

static napi_value fn_create(napi_env env, napi_callback_info info)      /* create(device = 0) -> handle */
{
    napi_value argv[1]; int32_t dev = 0;
    if (!get_args(env, info, 1, argv, NULL)) return NULL;
    if (!is_nullish(env, argv[0])) NAPI_OK(napi_get_value_int32(env, argv[0], &dev));
    gs_ctx *ctx = NULL;
    int rc = gs_create(dev, &ctx);
    if (rc != GS_OK) return throw_gs(env, NULL, rc);
    gs_handle *hd = (gs_handle *)calloc(1, sizeof *hd);
    if (!hd) { gs_destroy(ctx); napi_throw_error(env, NULL, "out of memory"); return NULL; }
    hd->ctx = ctx;
    napi_value h;
    if (napi_create_external(env, hd, ctx_finalize, NULL, &h) != napi_ok) { gs_destroy(ctx); free(hd); napi_throw_error(env, NULL, "napi_create_external"); return NULL; }
    return h;
}

static napi_value fn_clear(napi_env env, napi_callback_info info)       /* worker {method:"clear"} */
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    int rc = gs_clear(ctx);                                      /* (drains every lane) */
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    held_release(env, &get_handle(env, argv[0])->held);
    return NULL;
}

static napi_value push_common(napi_env env, napi_callback_info info, int which)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    void *data; size_t len;
    if (!get_bytes(env, argv[1], &data, &len)) { napi_throw_type_error(env, NULL, "expected an ArrayBuffer or TypedArray"); return NULL; }
    const size_t row = which == 0 ? 32 : 64;
    size_t n = len / row;
    if (!is_nullish(env, argv[2])) {                       /* pushDataBuffer(buffer, vertexCount) */
        int64_t want = 0;
        NAPI_OK(napi_get_value_int64(env, argv[2], &want));
        if (want < 0) want = 0;
        if ((size_t)want < n) n = (size_t)want;
    }
    int rc = which == 0 ? gs_push_splat(ctx, data, n) : (which == 1 ? gs_push_matrices(ctx, (const float *)data, n) : gs_load_ply(ctx, data, len));
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    napi_value r;
    NAPI_OK(napi_create_double(env, (double)gs_count(ctx), &r));
    return r;
}
static napi_value fn_push_splat(napi_env env, napi_callback_info info) { return push_common(env, info, 0); }
static napi_value fn_push_matrices(napi_env env, napi_callback_info info) { return push_common(env, info, 1); }
static napi_value fn_load_ply(napi_env env, napi_callback_info info) { return push_common(env, info, 2); }

static napi_value fn_ply_to_splat(napi_env env, napi_callback_info info)  /* processPlyBuffer(inputBuffer) -> ArrayBuffer */
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv, NULL)) return NULL;
    void *data; size_t len;
    if (!get_bytes(env, argv[0], &data, &len)) { napi_throw_type_error(env, NULL, "expected an ArrayBuffer or TypedArray"); return NULL; }
    size_t n = 0; char err[256] = "";
    int rc = gs_ply_to_splat(data, len, NULL, &n, err, sizeof err);
    if (rc != GS_OK) { napi_throw_error(env, NULL, err); return NULL; }
    napi_value ab; void *out;
    NAPI_OK(napi_create_arraybuffer(env, n * 32, &out, &ab));
    rc = gs_ply_to_splat(data, len, out, &n, err, sizeof err);
    if (rc != GS_OK) { napi_throw_error(env, NULL, err); return NULL; }
    return ab;
}

/* plyToSplatGpu(h, inputBuffer) -> ArrayBuffer: processPlyBuffer converted on the context's GPU (same bytes) */
static napi_value fn_ply_to_splat_gpu(napi_env env, napi_callback_info info)
{
    napi_value argv[2];
    if (!get_args(env, info, 2, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    void *data; size_t len;
    if (!get_bytes(env, argv[1], &data, &len)) { napi_throw_type_error(env, NULL, "expected an ArrayBuffer or TypedArray"); return NULL; }
    size_t n = 0;
    int rc = gs_ply_to_splat_gpu(ctx, data, len, NULL, &n);
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    napi_value ab; void *out;
    NAPI_OK(napi_create_arraybuffer(env, n * 32, &out, &ab));
    if (n) { rc = gs_ply_to_splat_gpu(ctx, data, len, out, &n); if (rc != GS_OK) return throw_gs(env, ctx, rc); }
    return ab;
}

static napi_value fn_count(napi_env env, napi_callback_info info)
{
    napi_value argv[1], r;
    if (!get_args(env, info, 1, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    NAPI_OK(napi_create_double(env, (double)gs_count(ctx), &r));
    return r;
}

/* `view` as tick posts it (view.buffer: an ArrayBuffer holding 4 floats, index.js:441-442, 453) or any array of 4 numbers; the raw
 * bytes are taken only from an ArrayBuffer or a Float32Array -- a Float64Array view goes through the element-wise conversion */
static int get_sort_args(napi_env env, napi_value vview, napi_value vcut, float view[4], float cut[16], const float **cutp)
{
    bool is_ab = false, is_ta = false;
    void *vb = NULL; size_t vlen = 0;
    int raw = 0;
    if (napi_is_arraybuffer(env, vview, &is_ab) == napi_ok && is_ab) raw = napi_get_arraybuffer_info(env, vview, &vb, &vlen) == napi_ok && vlen >= 16;
    else if (napi_is_typedarray(env, vview, &is_ta) == napi_ok && is_ta) {
        napi_typedarray_type ty; size_t n; napi_value ab; size_t off;
        raw = napi_get_typedarray_info(env, vview, &ty, &n, &vb, &ab, &off) == napi_ok && ty == napi_float32_array && n >= 4;
    }
    if (raw) memcpy(view, vb, 16);
    else if (!get_floats(env, vview, view, 4)) { napi_throw_type_error(env, NULL, "view: expected 4 floats"); return 0; }
    *cutp = NULL;
    if (!is_nullish(env, vcut)) {
        if (!get_floats(env, vcut, cut, 16)) { napi_throw_type_error(env, NULL, "cutout: expected 16 floats"); return 0; }
        *cutp = cut;
    }
    return 1;
}

static void idx_host_finalize(napi_env env, void *data, void *hint) { (void)env; (void)hint; gs_host_free(data); }

/* the handle's page-locked index buffer for the splats resident now, as its external ArrayBuffer */
static int index_buffer(napi_env env, gs_handle *h, napi_value *ab)
{
    size_t cap = gs_count(h->ctx); if (cap < 1) cap = 1;
    if (h->idx && h->idx_cap >= cap && h->idx_ab && napi_get_reference_value(env, h->idx_ab, ab) == napi_ok && *ab) return 1;
    if (h->idx_ab) { napi_delete_reference(env, h->idx_ab); h->idx_ab = NULL; }    /* (outgrown: JavaScript's arrays keep it alive) */
    cap += cap / 4;                                               /* (a scene that is still streaming in: not every push a new buffer) */
    uint32_t *p = (uint32_t *)gs_host_alloc(cap * 4);
    if (!p) { napi_throw_error(env, NULL, "page-locked allocation for the index list failed"); return 0; }
    if (napi_create_external_arraybuffer(env, p, cap * 4, idx_host_finalize, NULL, ab) != napi_ok) { gs_host_free(p); napi_throw_error(env, NULL, "external arraybuffer"); return 0; }
    if (napi_create_reference(env, *ab, 1, &h->idx_ab) != napi_ok) { h->idx_ab = NULL; napi_throw_error(env, NULL, "reference"); return 0; }
    h->idx = p; h->idx_cap = cap;
    return 1;
}

/* sort(h, view[4], cutout[16] | undefined, wantIndexes = true) -> Uint32Array (worker reply `sortedIndexes`) */
static napi_value fn_sort(napi_env env, napi_callback_info info)
{
    napi_value argv[4];
    if (!get_args(env, info, 4, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    float view[4], cut[16];
    const float *cutp = NULL;
    if (!get_sort_args(env, argv[1], argv[2], view, cut, &cutp)) return NULL;
    bool want = true;
    if (!is_nullish(env, argv[3])) NAPI_OK(napi_get_value_bool(env, argv[3], &want));
    if (!want) {
        int rc = gs_sort(ctx, view, cutp, NULL, NULL);
        if (rc != GS_OK) return throw_gs(env, ctx, rc);
        return NULL;
    }
    gs_handle *h = get_handle(env, argv[0]);
    napi_value ab;
    if (!index_buffer(env, h, &ab)) return NULL;
    uint32_t n = 0;
    int rc = gs_sort(ctx, view, cutp, h->idx, &n);
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    napi_value ta;
    NAPI_OK(napi_create_typedarray(env, napi_uint32_array, n, ab, 0, &ta));
    return ta;
}

static int fill_params(napi_env env, napi_value o, gs_render_params *p)
{
    napi_value v; double d;
    memset(p, 0, sizeof *p);
    if (napi_get_named_property(env, o, "modelView", &v) != napi_ok || !get_floats(env, v, p->model_view, 16)) return 0;
    if (napi_get_named_property(env, o, "projection", &v) != napi_ok || !get_floats(env, v, p->projection, 16)) return 0;
    if (!get_named_double(env, o, "width", &d, 1)) return 0;
    p->fb_width = (int32_t)d;
    if (!get_named_double(env, o, "height", &d, 1)) return 0;
    p->fb_height = (int32_t)d;
    p->x0 = 0; p->x1 = p->fb_width;
    d = 0;
    if (!get_named_double(env, o, "x0", &d, 0)) return 0;
    p->x0 = (int32_t)d;
    d = p->fb_width;
    if (!get_named_double(env, o, "x1", &d, 0)) return 0;
    p->x1 = (int32_t)d;
    d = 0;
    if (!get_named_double(env, o, "focal", &d, 0)) return 0;
    p->focal = (float)d;
    d = 0;
    if (!get_named_double(env, o, "flags", &d, 0)) return 0;
    p->flags = (uint32_t)d;
    p->background[0] = p->background[1] = p->background[2] = 0.0f; p->background[3] = 1.0f;
    bool has = false;
    if (napi_has_named_property(env, o, "background", &has) == napi_ok && has) {
        if (napi_get_named_property(env, o, "background", &v) != napi_ok) return 0;
        if (!is_nullish(env, v) && !get_floats(env, v, p->background, 4)) return 0;
    }
    return 1;
}

/* render(h, {modelView, projection, width, height, x0?, x1?, focal?, background?, flags?}) -> Uint8Array RGBA, row 0 = top */
static napi_value fn_render(napi_env env, napi_callback_info info)
{
    napi_value argv[2];
    if (!get_args(env, info, 2, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    gs_render_params p;
    if (!fill_params(env, argv[1], &p)) { napi_throw_type_error(env, NULL, "render: bad parameter object"); return NULL; }
    if (p.x1 <= p.x0 || p.fb_height <= 0) { napi_throw_range_error(env, NULL, "render: empty strip"); return NULL; }
    const size_t bytes = (size_t)(p.x1 - p.x0) * (size_t)p.fb_height * 4;
    napi_value ab, ta; void *out;
    NAPI_OK(napi_create_arraybuffer(env, bytes, &out, &ab));
    int rc = gs_render(ctx, &p, (uint8_t *)out, 0);
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    NAPI_OK(napi_create_typedarray(env, napi_uint8_array, bytes, ab, 0, &ta));
    return ta;
}

/* page-locked framebuffers handed to JavaScript as external ArrayBuffers: gs_render copies into them at PCIe speed and the
 * same memory is reused frame after frame (no per-call allocation, no extra copy) */
static void frame_finalize(napi_env env, void *data, void *hint) { (void)env; (void)hint; gs_host_free(data); }

/* allocFrame(width, height) -> Uint8Array(width * height * 4) over page-locked memory */
static napi_value fn_alloc_frame(napi_env env, napi_callback_info info)
{
    napi_value argv[2]; int32_t w = 0, h = 0;
    if (!get_args(env, info, 2, argv, NULL)) return NULL;
    NAPI_OK(napi_get_value_int32(env, argv[0], &w)); NAPI_OK(napi_get_value_int32(env, argv[1], &h));
    if (w <= 0 || h <= 0) { napi_throw_range_error(env, NULL, "allocFrame: bad size"); return NULL; }
    const size_t bytes = (size_t)w * (size_t)h * 4;
    void *p = gs_host_alloc(bytes);
    if (!p) { napi_throw_error(env, NULL, "allocFrame: page-locked allocation failed"); return NULL; }
    napi_value ab, ta;
    if (napi_create_external_arraybuffer(env, p, bytes, frame_finalize, NULL, &ab) != napi_ok) { gs_host_free(p); napi_throw_error(env, NULL, "external arraybuffer"); return NULL; }
    NAPI_OK(napi_create_typedarray(env, napi_uint8_array, bytes, ab, 0, &ta));
    return ta;
}

/* renderInto(h, params, frameUint8Array) -> frame: the same draw as render(), into a caller-owned buffer (allocFrame) */
static napi_value fn_render_into(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    gs_render_params p;
    if (!fill_params(env, argv[1], &p)) { napi_throw_type_error(env, NULL, "render: bad parameter object"); return NULL; }
    if (p.x1 <= p.x0 || p.fb_height <= 0) { napi_throw_range_error(env, NULL, "render: empty strip"); return NULL; }
    void *out; size_t len;
    if (!get_bytes(env, argv[2], &out, &len) || len < (size_t)(p.x1 - p.x0) * (size_t)p.fb_height * 4) {
        napi_throw_range_error(env, NULL, "renderInto: frame buffer missing or too small"); return NULL;
    }
    if ((p.flags & GS_RENDER_ASYNC) && !held_add(env, &get_handle(env, argv[0])->held, argv[2])) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
    int rc = gs_render(ctx, &p, (uint8_t *)out, 0);
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    return argv[2];
}

/* ---- asynchronous calls (napi_async_work on the libuv pool): tick's fire-and-forget sort (index.js:438-455, reply handler
 * 201-207) and the draw.  One in flight per context; the promise settles on the JS thread. ------------------------------ */
typedef struct gs_async {
    napi_async_work work; napi_deferred deferred; napi_ref keep;     /* keep: the handle (and the frame) stay alive meanwhile */
    gs_handle *h; int kind;                                          /* 0 = sort, 1 = render */
    float view[4], cut[16]; int has_cut;
    uint32_t *idx; uint32_t n; size_t cap;
    gs_render_params p; uint8_t *frame; napi_ref frame_ref;
    int rc; char err[512];
} gs_async;

static void async_execute(napi_env env, void *data)
{
    (void)env;
    gs_async *a = (gs_async *)data;
    if (a->kind == 0) a->rc = gs_sort(a->h->ctx, a->view, a->has_cut ? a->cut : NULL, a->idx, &a->n);
    else a->rc = gs_render(a->h->ctx, &a->p, a->frame, 0);
    if (a->rc != GS_OK) { const char *m = gs_last_error(a->h->ctx); snprintf(a->err, sizeof a->err, "%s", (m && *m) ? m : "gs_splat call failed"); }
}

static void idx_finalize(napi_env env, void *data, void *hint) { (void)env; (void)hint; free(data); }

static void async_complete(napi_env env, napi_status status, void *data)
{
    gs_async *a = (gs_async *)data;
    napi_value result = NULL;
    a->h->busy = 0;
    if (status == napi_ok && a->rc == GS_OK) {
        if (a->kind == 0) {                                       /* the worker's reply: {sortedIndexes}, moved not copied */
            napi_value ab;
            if (napi_create_external_arraybuffer(env, a->idx, (size_t)a->cap * 4, idx_finalize, NULL, &ab) == napi_ok) {
                a->idx = NULL;
                napi_create_typedarray(env, napi_uint32_array, a->n, ab, 0, &result);
            }
        } else napi_get_reference_value(env, a->frame_ref, &result);
    }
    if (result) napi_resolve_deferred(env, a->deferred, result);
    else {
        napi_value msg, err;
        napi_create_string_utf8(env, a->rc != GS_OK ? a->err : "asynchronous call failed", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &err);
        napi_reject_deferred(env, a->deferred, err);
    }
    if (a->keep) napi_delete_reference(env, a->keep);
    if (a->frame_ref) napi_delete_reference(env, a->frame_ref);
    napi_delete_async_work(env, a->work);
    free(a->idx);
    free(a);
}

static napi_value async_start(napi_env env, gs_async *a, napi_value handle, const char *name)
{
    napi_value promise, rname;
    if (napi_create_promise(env, &a->deferred, &promise) != napi_ok || napi_create_reference(env, handle, 1, &a->keep) != napi_ok ||
        napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &rname) != napi_ok ||
        napi_create_async_work(env, NULL, rname, async_execute, async_complete, a, &a->work) != napi_ok) {
        free(a->idx); free(a); napi_throw_error(env, NULL, "could not start the asynchronous call"); return NULL;
    }
    a->h->busy = 1;
    if (napi_queue_async_work(env, a->work) != napi_ok) { a->h->busy = 0; napi_throw_error(env, NULL, "could not queue the asynchronous call"); return NULL; }
    return promise;
}

/* sortAsync(h, view, cutout | undefined) -> Promise<Uint32Array> */
static napi_value fn_sort_async(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    gs_async *a = (gs_async *)calloc(1, sizeof *a);
    if (!a) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
    const float *cutp = NULL;
    if (!get_sort_args(env, argv[1], argv[2], a->view, a->cut, &cutp)) { free(a); return NULL; }
    a->has_cut = cutp != NULL; a->kind = 0; a->h = get_handle(env, argv[0]);
    a->cap = gs_count(ctx); if (a->cap < 1) a->cap = 1;
    a->idx = (uint32_t *)malloc(a->cap * 4);
    if (!a->idx) { free(a); napi_throw_error(env, NULL, "out of memory"); return NULL; }
    return async_start(env, a, argv[0], "gs_sort");
}

/* The reference's rhythm without a second thread (gs_sort_begin / gs_sort_poll, index.js:201-207, 438-455): sortBegin posts the sort
 * and returns; render() / renderInto() keep drawing from the last completed order; sortPoll(h, wait, wantIndexes) returns null while
 * the sort runs, else installs the new order and returns it (Uint32Array; `true` if wantIndexes is false). */
static napi_value fn_sort_begin(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    float view[4], cut[16];
    const float *cutp = NULL;
    if (!get_sort_args(env, argv[1], argv[2], view, cut, &cutp)) return NULL;
    const int rc = gs_sort_begin(ctx, view, cutp);
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    return NULL;
}

static napi_value fn_sort_poll(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    bool wait = false, want = true;
    if (!is_nullish(env, argv[1])) NAPI_OK(napi_get_value_bool(env, argv[1], &wait));
    if (!is_nullish(env, argv[2])) NAPI_OK(napi_get_value_bool(env, argv[2], &want));
    gs_handle *h = get_handle(env, argv[0]);
    napi_value ab = NULL, res;
    if (want && !index_buffer(env, h, &ab)) return NULL;
    uint32_t n = 0; int done = 0;
    const int rc = gs_sort_poll(ctx, wait ? 1 : 0, want ? h->idx : NULL, &n, &done);
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    if (!done) { NAPI_OK(napi_get_null(env, &res)); return res; }
    if (!want) { NAPI_OK(napi_get_boolean(env, true, &res)); return res; }
    NAPI_OK(napi_create_typedarray(env, napi_uint32_array, n, ab, 0, &res));
    return res;
}

/* renderAsync(h, params, frameUint8Array) -> Promise<frame> */
static napi_value fn_render_async(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    gs_async *a = (gs_async *)calloc(1, sizeof *a);
    if (!a) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
    if (!fill_params(env, argv[1], &a->p) || a->p.x1 <= a->p.x0 || a->p.fb_height <= 0) { free(a); napi_throw_type_error(env, NULL, "renderAsync: bad parameter object"); return NULL; }
    void *out; size_t len;
    if (!get_bytes(env, argv[2], &out, &len) || len < (size_t)(a->p.x1 - a->p.x0) * (size_t)a->p.fb_height * 4) {
        free(a); napi_throw_range_error(env, NULL, "renderAsync: frame buffer missing or too small"); return NULL;
    }
    a->frame = (uint8_t *)out; a->kind = 1; a->h = get_handle(env, argv[0]);
    if (napi_create_reference(env, argv[2], 1, &a->frame_ref) != napi_ok) { free(a); napi_throw_error(env, NULL, "reference"); return NULL; }
    return async_start(env, a, argv[0], "gs_render");
}

/* ---- several GPUs: one node process (or worker) per GPU, each with its own context; see gs_splat.h ------------------- */

/* partition([width0(, width1)], world) -> [{view, x0, x1, owner}, ...] */
static napi_value fn_partition(napi_env env, napi_callback_info info)
{
    napi_value argv[2], arr;
    if (!get_args(env, info, 2, argv, NULL)) return NULL;
    uint32_t nv = 0; int32_t world = 1; int widths[2] = { 0, 0 };
    NAPI_OK(napi_get_array_length(env, argv[0], &nv));
    if (nv < 1 || nv > 2) { napi_throw_range_error(env, NULL, "partition: one or two views"); return NULL; }
    for (uint32_t i = 0; i < nv; i++) { napi_value e; NAPI_OK(napi_get_element(env, argv[0], i, &e)); NAPI_OK(napi_get_value_int32(env, e, &widths[i])); }
    NAPI_OK(napi_get_value_int32(env, argv[1], &world));
    gs_piece pcs[128];
    const int n = gs_partition((int)nv, widths, world, pcs, 128);
    if (n < 0) { napi_throw_range_error(env, NULL, "partition: bad widths or world size"); return NULL; }
    NAPI_OK(napi_create_array_with_length(env, (size_t)n, &arr));
    for (int i = 0; i < n; i++) {
        napi_value o, v;
        NAPI_OK(napi_create_object(env, &o));
        NAPI_OK(napi_create_int32(env, pcs[i].view, &v)); NAPI_OK(napi_set_named_property(env, o, "view", v));
        NAPI_OK(napi_create_int32(env, pcs[i].x0, &v)); NAPI_OK(napi_set_named_property(env, o, "x0", v));
        NAPI_OK(napi_create_int32(env, pcs[i].x1, &v)); NAPI_OK(napi_set_named_property(env, o, "x1", v));
        NAPI_OK(napi_create_int32(env, pcs[i].owner, &v)); NAPI_OK(napi_set_named_property(env, o, "owner", v));
        NAPI_OK(napi_set_element(env, arr, (uint32_t)i, o));
    }
    return arr;
}

/* commUniqueId(h) -> ArrayBuffer(128): rank 0 creates it, the launcher hands it to every rank (IPC message, file, ...) */
static napi_value fn_comm_unique_id(napi_env env, napi_callback_info info)
{
    napi_value argv[1], ab; void *out;
    if (!get_args(env, info, 1, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    NAPI_OK(napi_create_arraybuffer(env, GS_COMM_ID_BYTES, &out, &ab));
    int rc = gs_comm_unique_id(ctx, out);
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    return ab;
}

/* commInit(h, id, rank, world): collective -- returns when every rank has called it */
static napi_value fn_comm_init(napi_env env, napi_callback_info info)
{
    napi_value argv[4]; int32_t rank = 0, world = 1;
    if (!get_args(env, info, 4, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    void *id; size_t len;
    if (!get_bytes(env, argv[1], &id, &len) || len < GS_COMM_ID_BYTES) { napi_throw_type_error(env, NULL, "commInit: id must hold 128 bytes"); return NULL; }
    NAPI_OK(napi_get_value_int32(env, argv[2], &rank)); NAPI_OK(napi_get_value_int32(env, argv[3], &world));
    int rc = gs_comm_init(ctx, id, rank, world);
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    return NULL;
}

static int fill_views(napi_env env, napi_value arr, gs_render_params views[2], uint32_t *nv)
{
    bool is_arr = false;
    if (napi_is_array(env, arr, &is_arr) != napi_ok) return 0;
    if (!is_arr) { *nv = 1; return fill_params(env, arr, &views[0]); }
    if (napi_get_array_length(env, arr, nv) != napi_ok || *nv < 1 || *nv > 2) return 0;
    for (uint32_t i = 0; i < *nv; i++) { napi_value e; if (napi_get_element(env, arr, i, &e) != napi_ok || !fill_params(env, e, &views[i])) return 0; }
    return 1;
}

/* sortGathered(h, view, cutout | undefined, views): the sort of the frame renderGathered(views) draws (this rank's strip) */
static napi_value fn_sort_gathered(napi_env env, napi_callback_info info)
{
    napi_value argv[4];
    if (!get_args(env, info, 4, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    float view[4], cut[16]; const float *cutp = NULL;
    if (!get_sort_args(env, argv[1], argv[2], view, cut, &cutp)) return NULL;
    gs_render_params views[2]; uint32_t nv = 0;
    if (!fill_views(env, argv[3], views, &nv)) { napi_throw_type_error(env, NULL, "sortGathered: bad view parameters"); return NULL; }
    int rc = gs_sort_gathered(ctx, view, cutp, views, (int)nv);
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    return NULL;
}

/* renderGathered(h, views, root = 0, flags = 0): every rank; the image(s) are assembled on the root (readGathered) */
static napi_value fn_render_gathered(napi_env env, napi_callback_info info)
{
    napi_value argv[4]; int32_t root = 0; uint32_t flags = 0;
    if (!get_args(env, info, 4, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    gs_render_params views[2]; uint32_t nv = 0;
    if (!fill_views(env, argv[1], views, &nv)) { napi_throw_type_error(env, NULL, "renderGathered: bad view parameters"); return NULL; }
    if (!is_nullish(env, argv[2])) NAPI_OK(napi_get_value_int32(env, argv[2], &root));
    if (!is_nullish(env, argv[3])) NAPI_OK(napi_get_value_uint32(env, argv[3], &flags));
    int rc = gs_render_gathered(ctx, views, (int)nv, root, NULL, flags);
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    return NULL;
}

/* readGathered(h, view, frameUint8Array) -> frame (root only; after sync() for asynchronous frames) */
static napi_value fn_read_gathered(napi_env env, napi_callback_info info)
{
    napi_value argv[3]; int32_t view = 0;
    if (!get_args(env, info, 3, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    NAPI_OK(napi_get_value_int32(env, argv[1], &view));
    void *out; size_t len;
    if (!get_bytes(env, argv[2], &out, &len)) { napi_throw_type_error(env, NULL, "readGathered: frame buffer missing"); return NULL; }
    int w = 0, h = 0;
    int rc = gs_gathered_size(ctx, view, &w, &h);
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    if (len < (size_t)w * (size_t)h * 4) { napi_throw_range_error(env, NULL, "readGathered: frame buffer too small"); return NULL; }
    rc = gs_read_gathered(ctx, view, (uint8_t *)out, 0);
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    return argv[2];
}

static napi_value fn_sync(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    int rc = gs_sync(ctx);
    /* GS_OK / GS_E_RETRY: every lane's stream has been waited for, every queued frame has reached its buffer.  Any other status may have
     * left gs_sync early (a lane's failure is returned before the other lanes' streams are synchronised): a copy into one of the held
     * buffers can still be in flight, so the references stay until the next sync that completes, a clear or the destroy (which drain) */
    if (rc == GS_OK || rc == GS_E_RETRY) held_release(env, &get_handle(env, argv[0])->held);
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    return NULL;
}

/* setScene(h, depthFloat32Array | null, rgbaUint8Array | null, width, height): the opaque scene the splats are depth-tested
 * against (depthTest: true, index.js:179) and blended over */
static napi_value fn_set_scene(napi_env env, napi_callback_info info)
{
    napi_value argv[5];
    if (!get_args(env, info, 5, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    void *d = NULL, *c = NULL; size_t dl = 0, cl = 0; int32_t w = 0, h = 0;
    if (!is_nullish(env, argv[1]) && !get_bytes(env, argv[1], &d, &dl)) { napi_throw_type_error(env, NULL, "depth: expected a Float32Array"); return NULL; }
    if (!is_nullish(env, argv[2]) && !get_bytes(env, argv[2], &c, &cl)) { napi_throw_type_error(env, NULL, "rgba: expected a Uint8Array"); return NULL; }
    if (!is_nullish(env, argv[3])) NAPI_OK(napi_get_value_int32(env, argv[3], &w));
    if (!is_nullish(env, argv[4])) NAPI_OK(napi_get_value_int32(env, argv[4], &h));
    if ((d && dl < (size_t)w * h * 4) || (c && cl < (size_t)w * h * 4)) { napi_throw_range_error(env, NULL, "setScene: buffer smaller than width*height"); return NULL; }
    int rc = gs_set_scene(ctx, (const float *)d, (const uint8_t *)c, w, h);
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    return NULL;
}

static napi_value fn_stats(napi_env env, napi_callback_info info)
{
    napi_value argv[1], o, v;
    if (!get_args(env, info, 1, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    gs_stats s;
    int rc = gs_get_stats(ctx, &s);
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    NAPI_OK(napi_create_object(env, &o));
#define PUT(name, val) do { NAPI_OK(napi_create_double(env, (double)(val), &v)); NAPI_OK(napi_set_named_property(env, o, name, v)); } while (0)
    PUT("nSplats", s.n_splats); PUT("nSorted", s.n_sorted); PUT("nVisible", s.n_visible); PUT("nPairs", s.n_pairs);
    PUT("nFrags", s.n_frags); PUT("nTiles", s.n_tiles); PUT("msSort", s.ms_sort); PUT("msProject", s.ms_project);
    PUT("msBin", s.ms_bin); PUT("msBlend", s.ms_blend); PUT("msRender", s.ms_render);
    PUT("accFrames", s.acc_frames); PUT("unsatTiles", s.unsat_tiles); PUT("nearPermille", s.near_permille); PUT("sortRecords", s.sort_records);
    PUT("retriedFrames", s.retried_frames); PUT("specSorts", s.spec_sorts); PUT("specMisses", s.spec_misses); PUT("needSplats", s.need_splats); PUT("sortMode", s.sort_mode); PUT("subtile", s.subtile);
#undef PUT
    return o;
}

static napi_value fn_set_option(napi_env env, napi_callback_info info)
{
    napi_value argv[3]; int32_t opt; int64_t val;
    if (!get_args(env, info, 3, argv, NULL)) return NULL;
    gs_ctx *ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    NAPI_OK(napi_get_value_int32(env, argv[1], &opt));
    NAPI_OK(napi_get_value_int64(env, argv[2], &val));
    int rc = gs_set_option(ctx, opt, val);
    if (rc != GS_OK) return throw_gs(env, ctx, rc);
    return NULL;
}

/* uniform producers (host helpers) */
static napi_value fn_model_view(napi_env env, napi_callback_info info)   /* getModelViewMatrix */
{
    napi_value argv[2]; double a[16], b[16], o[16];
    if (!get_args(env, info, 2, argv, NULL)) return NULL;
    if (!get_doubles(env, argv[0], a, 16) || !get_doubles(env, argv[1], b, 16)) { napi_throw_type_error(env, NULL, "expected two 16-element matrices"); return NULL; }
    gs_model_view_matrix(a, b, o);
    return make_f64_array(env, o, 16);
}

static napi_value fn_projection(napi_env env, napi_callback_info info)   /* getProjectionMatrix */
{
    napi_value argv[1]; double a[16], o[16];
    if (!get_args(env, info, 1, argv, NULL)) return NULL;
    if (!get_doubles(env, argv[0], a, 16)) { napi_throw_type_error(env, NULL, "expected a 16-element matrix"); return NULL; }
    gs_projection_matrix(a, o);
    return make_f64_array(env, o, 16);
}

static napi_value fn_tick(napi_env env, napi_callback_info info)         /* tick -> {view: Float32Array(4), cutout: Float32Array(16)|undefined} */
{
    napi_value argv[3]; double a[16], b[16], c[16];
    if (!get_args(env, info, 3, argv, NULL)) return NULL;
    if (!get_doubles(env, argv[0], a, 16) || !get_doubles(env, argv[1], b, 16)) { napi_throw_type_error(env, NULL, "expected two 16-element matrices"); return NULL; }
    const int has_cut = !is_nullish(env, argv[2]);
    if (has_cut && !get_doubles(env, argv[2], c, 16)) { napi_throw_type_error(env, NULL, "cutout: expected a 16-element matrix"); return NULL; }
    float view[4], cut[16];
    gs_tick_uniforms(a, b, has_cut ? c : NULL, view, cut);
    napi_value o, ab, ta; void *data;
    NAPI_OK(napi_create_object(env, &o));
    NAPI_OK(napi_create_arraybuffer(env, 16, &data, &ab)); memcpy(data, view, 16);
    NAPI_OK(napi_create_typedarray(env, napi_float32_array, 4, ab, 0, &ta));
    NAPI_OK(napi_set_named_property(env, o, "view", ta));
    if (has_cut) {
        NAPI_OK(napi_create_arraybuffer(env, 64, &data, &ab)); memcpy(data, cut, 64);
        NAPI_OK(napi_create_typedarray(env, napi_float32_array, 16, ab, 0, &ta));
        NAPI_OK(napi_set_named_property(env, o, "cutout", ta));
    }
    return o;
}

static napi_value fn_scaled_size(napi_env env, napi_callback_info info)
{
    napi_value argv[3], arr, v; int32_t w, h, ow, oh; double r;
    if (!get_args(env, info, 3, argv, NULL)) return NULL;
    NAPI_OK(napi_get_value_int32(env, argv[0], &w)); NAPI_OK(napi_get_value_int32(env, argv[1], &h));
    NAPI_OK(napi_get_value_double(env, argv[2], &r));
    gs_scaled_size(w, h, r, &ow, &oh);
    NAPI_OK(napi_create_array_with_length(env, 2, &arr));
    NAPI_OK(napi_create_int32(env, ow, &v)); NAPI_OK(napi_set_element(env, arr, 0, v));
    NAPI_OK(napi_create_int32(env, oh, &v)); NAPI_OK(napi_set_element(env, arr, 1, v));
    return arr;
}

/* ---- one Node.js process, several GPUs (gs_create_multi): the consumer `north_star` names is ONE JavaScript thread ----------
 * createMulti([dev, ...]) -> handle;  multiPushSplat / multiLoadPly / multiClear / multiCount / multiSetOption;
 * multiSort(h, view, cutout | null, views);  multiRender(h, views, frames, flags = 0): host-direct -- every GPU copies its
 * strip straight into the caller's page-locked frame(s) (allocFrame);  multiRenderDevice(h, views, flags) + multiRead(h, view,
 * frame): gathered on the first device;  multiSync(h);  multiDestroy(h). */
typedef struct gs_mhandle { gs_multi *m; gs_held held; } gs_mhandle;

static void multi_finalize(napi_env env, void *data, void *hint)
{
    (void)hint;
    gs_mhandle *h = (gs_mhandle *)data;
    if (!h) return;
    if (h->m) gs_multi_destroy(h->m);
    held_free(env, &h->held);
    free(h);
}

static napi_value throw_multi(napi_env env, gs_multi *m, int rc)
{
    char code[16];
    const char *msg = gs_multi_last_error(m);
    snprintf(code, sizeof code, "GS%d", rc);
    napi_throw_error(env, code, (msg && *msg) ? msg : "gs_splat call failed");
    return NULL;
}

static gs_mhandle *get_mhandle(napi_env env, napi_value v, int need_live)
{
    void *p = NULL;
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok || t != napi_external || napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "expected a handle returned by createMulti()");
        return NULL;
    }
    if (need_live && !((gs_mhandle *)p)->m) { napi_throw_error(env, "GS_DESTROYED", "the multi-GPU context has been destroyed"); return NULL; }
    return (gs_mhandle *)p;
}

static napi_value fn_create_multi(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv, NULL)) return NULL;
    uint32_t n = 0; bool is = false;
    if (napi_is_array(env, argv[0], &is) != napi_ok || !is || napi_get_array_length(env, argv[0], &n) != napi_ok || n < 1 || n > 64) {
        napi_throw_type_error(env, NULL, "createMulti: an array of 1..64 device ordinals"); return NULL;
    }
    int devs[64];
    for (uint32_t i = 0; i < n; i++) {
        napi_value e; int32_t d = 0;
        if (napi_get_element(env, argv[0], i, &e) != napi_ok || napi_get_value_int32(env, e, &d) != napi_ok) { napi_throw_type_error(env, NULL, "createMulti: device ordinals are integers"); return NULL; }
        devs[i] = d;
    }
    gs_multi *m = NULL;
    int rc = gs_create_multi(devs, (int)n, &m);
    if (rc != GS_OK) return throw_multi(env, NULL, rc);
    gs_mhandle *hd = (gs_mhandle *)calloc(1, sizeof *hd);
    if (!hd) { gs_multi_destroy(m); napi_throw_error(env, NULL, "out of memory"); return NULL; }
    hd->m = m;
    napi_value h;
    if (napi_create_external(env, hd, multi_finalize, NULL, &h) != napi_ok) { gs_multi_destroy(m); free(hd); napi_throw_error(env, NULL, "napi_create_external"); return NULL; }
    return h;
}

static napi_value fn_multi_destroy(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv, NULL)) return NULL;
    gs_mhandle *h = get_mhandle(env, argv[0], 0); if (!h) return NULL;
    if (h->m) { gs_multi_destroy(h->m); h->m = NULL; }
    held_release(env, &h->held);
    return NULL;
}

static napi_value multi_push_common(napi_env env, napi_callback_info info, int ply)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv, NULL)) return NULL;
    gs_mhandle *h = get_mhandle(env, argv[0], 1); if (!h) return NULL;
    void *data; size_t len;
    if (!get_bytes(env, argv[1], &data, &len)) { napi_throw_type_error(env, NULL, "expected an ArrayBuffer or TypedArray"); return NULL; }
    size_t n = len / 32;
    if (!ply && !is_nullish(env, argv[2])) {
        int64_t want = 0;
        NAPI_OK(napi_get_value_int64(env, argv[2], &want));
        if (want < 0) want = 0;
        if ((size_t)want < n) n = (size_t)want;
    }
    int rc = ply ? gs_multi_load_ply(h->m, data, len) : gs_multi_push_splat(h->m, data, n);
    if (rc != GS_OK) return throw_multi(env, h->m, rc);
    napi_value r;
    NAPI_OK(napi_create_double(env, (double)gs_multi_count(h->m), &r));
    return r;
}
static napi_value fn_multi_push_splat(napi_env env, napi_callback_info info) { return multi_push_common(env, info, 0); }
static napi_value fn_multi_load_ply(napi_env env, napi_callback_info info) { return multi_push_common(env, info, 1); }

static napi_value fn_multi_clear(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv, NULL)) return NULL;
    gs_mhandle *h = get_mhandle(env, argv[0], 1); if (!h) return NULL;
    int rc = gs_multi_clear(h->m);
    if (rc != GS_OK) return throw_multi(env, h->m, rc);
    return NULL;
}

static napi_value fn_multi_count(napi_env env, napi_callback_info info)
{
    napi_value argv[1], r;
    if (!get_args(env, info, 1, argv, NULL)) return NULL;
    gs_mhandle *h = get_mhandle(env, argv[0], 1); if (!h) return NULL;
    NAPI_OK(napi_create_double(env, (double)gs_multi_count(h->m), &r));
    return r;
}

static napi_value fn_multi_set_option(napi_env env, napi_callback_info info)
{
    napi_value argv[3]; int32_t opt = 0; int64_t val = 0;
    if (!get_args(env, info, 3, argv, NULL)) return NULL;
    gs_mhandle *h = get_mhandle(env, argv[0], 1); if (!h) return NULL;
    NAPI_OK(napi_get_value_int32(env, argv[1], &opt)); NAPI_OK(napi_get_value_int64(env, argv[2], &val));
    int rc = gs_multi_set_option(h->m, opt, val);
    if (rc != GS_OK) return throw_multi(env, h->m, rc);
    return NULL;
}

static napi_value fn_multi_sort(napi_env env, napi_callback_info info)
{
    napi_value argv[4];
    if (!get_args(env, info, 4, argv, NULL)) return NULL;
    gs_mhandle *h = get_mhandle(env, argv[0], 1); if (!h) return NULL;
    float view[4], cut[16]; const float *cutp = NULL;
    if (!get_sort_args(env, argv[1], argv[2], view, cut, &cutp)) return NULL;
    gs_render_params views[2]; uint32_t nv = 0;
    if (!fill_views(env, argv[3], views, &nv)) { napi_throw_type_error(env, NULL, "multiSort: bad view parameters"); return NULL; }
    int rc = gs_multi_sort(h->m, view, cutp, views, (int)nv);
    if (rc != GS_OK) return throw_multi(env, h->m, rc);
    return NULL;
}

/* multiRender(h, views, frames, flags = 0) -> frames.  frames: one Uint8Array (one view) or an array of them, fb_width x fb_height x 4 each */
static napi_value fn_multi_render(napi_env env, napi_callback_info info)
{
    napi_value argv[4]; uint32_t flags = 0;
    if (!get_args(env, info, 4, argv, NULL)) return NULL;
    gs_mhandle *h = get_mhandle(env, argv[0], 1); if (!h) return NULL;
    gs_render_params views[2]; uint32_t nv = 0;
    if (!fill_views(env, argv[1], views, &nv)) { napi_throw_type_error(env, NULL, "multiRender: bad view parameters"); return NULL; }
    if (!is_nullish(env, argv[3])) NAPI_OK(napi_get_value_uint32(env, argv[3], &flags));
    uint8_t *frames[2] = { NULL, NULL };
    bool is_arr = false;
    napi_is_array(env, argv[2], &is_arr);
    for (uint32_t v = 0; v < nv; v++) {
        napi_value fv = argv[2];
        if (is_arr && napi_get_element(env, argv[2], v, &fv) != napi_ok) { napi_throw_type_error(env, NULL, "multiRender: frames"); return NULL; }
        void *out; size_t len;
        if ((!is_arr && v > 0) || !get_bytes(env, fv, &out, &len) || len < (size_t)views[v].fb_width * (size_t)views[v].fb_height * 4) {
            napi_throw_range_error(env, NULL, "multiRender: a frame buffer is missing or too small"); return NULL;
        }
        frames[v] = (uint8_t *)out;
        if ((flags & GS_RENDER_ASYNC) && !held_add(env, &h->held, fv)) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
    }
    int rc = gs_multi_render(h->m, views, (int)nv, frames, 0, flags);
    if (rc != GS_OK) return throw_multi(env, h->m, rc);
    return argv[2];
}

static napi_value fn_multi_render_device(napi_env env, napi_callback_info info)
{
    napi_value argv[3]; uint32_t flags = 0;
    if (!get_args(env, info, 3, argv, NULL)) return NULL;
    gs_mhandle *h = get_mhandle(env, argv[0], 1); if (!h) return NULL;
    gs_render_params views[2]; uint32_t nv = 0;
    if (!fill_views(env, argv[1], views, &nv)) { napi_throw_type_error(env, NULL, "multiRenderDevice: bad view parameters"); return NULL; }
    if (!is_nullish(env, argv[2])) NAPI_OK(napi_get_value_uint32(env, argv[2], &flags));
    int rc = gs_multi_render_device(h->m, views, (int)nv, NULL, flags);
    if (rc != GS_OK) return throw_multi(env, h->m, rc);
    return NULL;
}

/* multiRead(h, view, frameUint8Array) -> frame */
static napi_value fn_multi_read(napi_env env, napi_callback_info info)
{
    napi_value argv[3]; int32_t view = 0;
    if (!get_args(env, info, 3, argv, NULL)) return NULL;
    gs_mhandle *h = get_mhandle(env, argv[0], 1); if (!h) return NULL;
    NAPI_OK(napi_get_value_int32(env, argv[1], &view));
    void *out; size_t len;
    if (!get_bytes(env, argv[2], &out, &len)) { napi_throw_type_error(env, NULL, "multiRead: frame buffer missing"); return NULL; }
    int rc = gs_multi_sync(h->m);
    if (rc != GS_OK) return throw_multi(env, h->m, rc);
    int w = 0, hh = 0;
    gs_ctx *root = gs_multi_ctx(h->m, 0);
    rc = gs_gathered_size(root, view, &w, &hh);
    if (rc != GS_OK) return throw_gs(env, root, rc);
    if (len < (size_t)w * (size_t)hh * 4) { napi_throw_range_error(env, NULL, "multiRead: frame buffer too small"); return NULL; }
    rc = gs_multi_read(h->m, view, (uint8_t *)out, 0);
    if (rc != GS_OK) return throw_multi(env, h->m, rc);
    return argv[2];
}

static napi_value fn_multi_sync(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv, NULL)) return NULL;
    gs_mhandle *h = get_mhandle(env, argv[0], 1); if (!h) return NULL;
    int rc = gs_multi_sync(h->m);
    if (rc == GS_OK || rc == GS_E_RETRY) held_release(env, &h->held);   /* (as fn_sync: kept while a copy may still be in flight) */
    if (rc != GS_OK) return throw_multi(env, h->m, rc);
    return NULL;
}

static napi_value fn_destroy(napi_env env, napi_callback_info info)      /* explicit teardown; GC finalizer is the fallback */
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv, NULL)) return NULL;
    gs_handle *h = get_handle(env, argv[0]);
    if (!h) return NULL;
    if (h->busy) { napi_throw_error(env, "GS_BUSY", "an asynchronous sort or render of this context is in flight"); return NULL; }
    if (h->ctx) { gs_destroy(h->ctx); h->ctx = NULL; }     /* HBM, streams and threads go now; the shell goes with the JS handle */
    held_release(env, &h->held);
    return NULL;
}

static napi_value init(napi_env env, napi_value exports)
{
    static const struct { const char *name; napi_callback fn; } fns[] = {
        { "create", fn_create }, { "destroy", fn_destroy }, { "clear", fn_clear }, { "pushSplat", fn_push_splat },
        { "pushMatrices", fn_push_matrices }, { "loadPly", fn_load_ply }, { "plyToSplat", fn_ply_to_splat }, { "plyToSplatGpu", fn_ply_to_splat_gpu },
        { "count", fn_count },
        { "sort", fn_sort }, { "sortAsync", fn_sort_async }, { "sortBegin", fn_sort_begin }, { "sortPoll", fn_sort_poll }, { "render", fn_render }, { "renderInto", fn_render_into },
        { "renderAsync", fn_render_async }, { "allocFrame", fn_alloc_frame }, { "sync", fn_sync },
        { "partition", fn_partition }, { "commUniqueId", fn_comm_unique_id }, { "commInit", fn_comm_init },
        { "sortGathered", fn_sort_gathered }, { "renderGathered", fn_render_gathered }, { "readGathered", fn_read_gathered }, { "setScene", fn_set_scene }, { "stats", fn_stats }, { "setOption", fn_set_option },
        { "modelViewMatrix", fn_model_view }, { "projectionMatrix", fn_projection }, { "tickUniforms", fn_tick },
        { "scaledSize", fn_scaled_size },
        { "createMulti", fn_create_multi }, { "multiDestroy", fn_multi_destroy }, { "multiPushSplat", fn_multi_push_splat },
        { "multiLoadPly", fn_multi_load_ply }, { "multiClear", fn_multi_clear }, { "multiCount", fn_multi_count },
        { "multiSetOption", fn_multi_set_option }, { "multiSort", fn_multi_sort }, { "multiRender", fn_multi_render },
        { "multiRenderDevice", fn_multi_render_device }, { "multiRead", fn_multi_read }, { "multiSync", fn_multi_sync },
    };
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok ||
            napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) {
            napi_throw_error(env, NULL, "gs_splat_napi: init failed");
            return NULL;
        }
    }
    napi_value v;
    if (napi_create_uint32(env, gs_version(), &v) == napi_ok) napi_set_named_property(env, exports, "version", v);
    return exports;
}

NAPI_MODULE(gs_splat_napi, init)

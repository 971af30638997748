// gs_multi.hip -- ONE host process driving several GPUs (SURVEY.md 8b plugin row, 8e): the consumer `north_star` names is a
// Node.js process, i.e. one JavaScript thread per page like the reference (index.js:1-23; several component instances per
// page: cutout-demo.html:24-25), and it cannot be one process per GPU.  A gs_multi owns one gs_ctx per device, replicates the
// splat buffer on them, and exposes sort / render with the single-context meaning: the viewport is split into the column
// strips gs_partition gives (XR: the eyes over the devices), every context sorts for its strip (gs_sort_gathered) and draws it.
// Where the pieces meet is the caller's choice (SURVEY.md 8e "measure both"):
//   gs_multi_render         host-direct: every GPU copies its strip straight into the caller's page-locked frame with a 2-D
//                           copy behind its kernels (gs_render with a row stride) -- no collective, no staging, no assembly;
//   gs_multi_render_device  the frame is gathered in HBM on devices[0] through the library's in-process transport (peer copies
//                           on the frames' own streams, gs_comm.hip) -- for consumers that keep the frame on a GPU.
// Each context has a feeder thread here (a gs_ctx is single-caller): a call posts one closure per context and -- for
// asynchronous frames -- returns at once, so the per-frame host cost does not grow with the number of GPUs.
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include "gs_internal.h"

namespace {

struct Feeder {
    gs_ctx *ctx = nullptr;
    int rank = 0;
    std::thread th;
    std::mutex m;
    std::condition_variable cv_work, cv_idle;
    std::deque<std::function<int(gs_ctx *)>> q;
    bool busy = false, stop = false;
    int rc = GS_OK;                                                // first failure since it was last collected ...
    char err[GS_ERRLEN] = "";                                      // ... and its message
};

void feeder_main(Feeder *f)
{
    (void)hipSetDevice(f->ctx->device);
    std::unique_lock<std::mutex> lk(f->m);
    for (;;) {
        f->cv_work.wait(lk, [&] { return f->stop || !f->q.empty(); });
        if (f->q.empty()) break;
        std::function<int(gs_ctx *)> fn = std::move(f->q.front());
        f->q.pop_front();
        f->busy = true;
        lk.unlock();
        const int rc = fn(f->ctx);
        lk.lock();
        if (rc != GS_OK && (f->rc == GS_OK || f->rc == GS_E_RETRY)) { f->rc = rc; snprintf(f->err, sizeof f->err, "device %d: %s", f->ctx->device, gs_last_error(f->ctx)); }
        f->busy = false;
        if (f->q.empty()) f->cv_idle.notify_all();
    }
}

}  // namespace

struct gs_multi {
    std::vector<Feeder *> f;
    char err[GS_ERRLEN] = "";
    // the last frame's arguments: a synchronous render that comes back with GS_E_RETRY is sorted and drawn again here
    float view[4] = { 0, 0, 0, 0 }, cutout[16]; bool has_cutout = false, have_sort = false;
    gs_render_params sviews[2]; int snviews = 0;
    bool dead = false;                                             // a call reached only some of the devices (post): nothing more is accepted
};

static thread_local char g_multi_err[GS_ERRLEN] = "";

namespace {

int post(gs_multi *m, const std::function<int(gs_ctx *, int)> &fn)
{
    // all or nothing: a collective (a gathered frame, a shared sort) queued on SOME feeders only would leave those ranks waiting
    // the transport's whole timeout for peers that never take part.  The closures -- what can really run out of memory: they
    // carry the call's arguments -- are all built before the first feeder sees anything; the hand-over then only moves them, and
    // if even that fails the context is marked unusable instead of staying half-posted.
    if (m->dead) { snprintf(m->err, sizeof m->err, "the multi-device context is unusable after a failed hand-over"); return GS_E_STATE; }
    std::vector<std::function<int(gs_ctx *)>> cl;
    try {                                                          // (no exception crosses the C ABI: a closure that cannot be queued is GS_E_OOM)
        cl.reserve(m->f.size());
        for (Feeder *f : m->f) { const int rank = f->rank; cl.emplace_back([fn, rank](gs_ctx *c) { return fn(c, rank); }); }
    } catch (...) { snprintf(m->err, sizeof m->err, "out of host memory"); return GS_E_OOM; }
    size_t k = 0;
    for (Feeder *f : m->f) {
        bool ok = true;
        { std::lock_guard<std::mutex> lk(f->m); try { f->q.push_back(std::move(cl[k])); } catch (...) { ok = false; } }
        if (!ok) {                                                  // (a queue node of a few hundred bytes could not be allocated)
            m->dead = true;
            snprintf(m->err, sizeof m->err, "out of host memory while a call was being handed to the devices: the multi-device context is unusable");
            return GS_E_OOM;
        }
        f->cv_work.notify_one();
        k++;
    }
    return GS_OK;
}

// wait until every feeder is idle; returns (and clears) the first failure, GS_E_RETRY only if nothing worse happened
int collect(gs_multi *m)
{
    int first = GS_OK;
    for (Feeder *f : m->f) {
        std::unique_lock<std::mutex> lk(f->m);
        f->cv_idle.wait(lk, [&] { return f->q.empty() && !f->busy; });
        if (f->rc != GS_OK && (first == GS_OK || (first == GS_E_RETRY && f->rc != GS_E_RETRY))) { first = f->rc; memcpy(m->err, f->err, sizeof m->err); }
        f->rc = GS_OK;
    }
    return first;
}

int run_all(gs_multi *m, const std::function<int(gs_ctx *, int)> &fn) { const int rp = post(m, fn), rc = collect(m); return rp != GS_OK ? rp : rc; }

int check_views(gs_multi *m, const gs_render_params *views, int nviews)
{
    if (!views || nviews < 1 || nviews > 2) { snprintf(m->err, sizeof m->err, "1 or 2 views"); return GS_E_BADARG; }
    return GS_OK;
}

// this rank's pieces of the frame `views` describe
int pieces_of(int rank, int world, const gs_render_params *views, int nviews, gs_piece *mine, int *nmine)
{
    int widths[2] = { views[0].fb_width, nviews > 1 ? views[1].fb_width : 0 };
    gs_piece pcs[128];
    const int np = gs_partition(nviews, widths, world, pcs, 128);
    if (np < 0) return GS_E_BADARG;
    *nmine = 0;
    for (int i = 0; i < np; i++) if (pcs[i].owner == rank) mine[(*nmine)++] = pcs[i];
    return GS_OK;
}

}  // namespace

extern "C" {

GS_API int gs_create_multi(const int *devices, int ndev, gs_multi **out)
{
    if (!out) return GS_E_BADARG;
    *out = nullptr;
    if (!devices || ndev < 1 || ndev > 64) { snprintf(g_multi_err, sizeof g_multi_err, "gs_create_multi: 1..64 devices"); return GS_E_BADARG; }
    gs_multi *m = new (std::nothrow) gs_multi();
    if (!m) { snprintf(g_multi_err, sizeof g_multi_err, "out of host memory"); return GS_E_OOM; }
    int rc = GS_OK;
    uint8_t id[GS_COMM_ID_BYTES];
    for (int i = 0; i < ndev && rc == GS_OK; i++) {
        gs_ctx *c = nullptr;
        rc = gs_create(devices[i], &c);
        if (rc != GS_OK) { snprintf(g_multi_err, sizeof g_multi_err, "%s", gs_last_error(nullptr)); break; }
        Feeder *f = new (std::nothrow) Feeder();
        if (!f) { gs_destroy(c); rc = GS_E_OOM; snprintf(g_multi_err, sizeof g_multi_err, "out of host memory"); break; }
        f->ctx = c; f->rank = i;
        m->f.push_back(f);
        // the ranks of one process: the in-process transport (no RCCL); joining never blocks, so this thread brings them all up
        if (i == 0) { rc = gs_set_option(c, GS_OPT_COMM_TRANSPORT, 1); if (rc == GS_OK) rc = gs_comm_unique_id(c, id); }
        if (rc == GS_OK) rc = gs_comm_init(c, id, i, ndev);
        if (rc != GS_OK) snprintf(g_multi_err, sizeof g_multi_err, "%s", gs_last_error(c));
    }
    if (rc == GS_OK) {
        try { for (Feeder *f : m->f) f->th = std::thread(feeder_main, f); }
        catch (...) { rc = GS_E_OOM; snprintf(g_multi_err, sizeof g_multi_err, "could not start a feeder thread"); }
    }
    if (rc != GS_OK) { gs_multi_destroy(m); return rc; }
    *out = m;
    return GS_OK;
}

GS_API int gs_multi_destroy(gs_multi *m)
{
    if (!m) return GS_OK;
    // a feeder that sits in a host-side receive of the in-process transport (its sender never posted: a device's frame failed, a
    // caller that tears down mid-frame) is released now, not after GS_COMM_TIMEOUT_S: nothing of this communicator is wanted any more.
    // Only when work is still queued: a quiet gs_multi keeps its hub intact until the endpoints go.
    for (Feeder *f : m->f) {
        bool busy;
        { std::lock_guard<std::mutex> lk(f->m); busy = f->busy || !f->q.empty(); }
        if (busy) { for (Feeder *g : m->f) gs_comm_cancel(g->ctx); break; }
    }
    for (Feeder *f : m->f) {
        { std::lock_guard<std::mutex> lk(f->m); f->stop = true; }
        f->cv_work.notify_one();
        if (f->th.joinable()) f->th.join();
    }
    // every context quiet before the first communicator endpoint goes (a peer may still be pulling a mailbox of another one)
    for (Feeder *f : m->f) (void)gs_sync(f->ctx);
    for (Feeder *f : m->f) { (void)gs_destroy(f->ctx); delete f; }
    delete m;
    return GS_OK;
}

GS_API const char *gs_multi_last_error(const gs_multi *m) { return m ? m->err : g_multi_err; }
GS_API int gs_multi_devices(const gs_multi *m) { return m ? (int)m->f.size() : 0; }
GS_API gs_ctx *gs_multi_ctx(gs_multi *m, int i) { return (m && i >= 0 && i < (int)m->f.size()) ? m->f[(size_t)i]->ctx : nullptr; }

GS_API int gs_multi_clear(gs_multi *m)
{
    if (!m) return GS_E_BADARG;
    m->have_sort = false;
    return run_all(m, [](gs_ctx *c, int) { return gs_clear(c); });
}

GS_API int gs_multi_push_splat(gs_multi *m, const void *rows, size_t nrows)
{
    if (!m) return GS_E_BADARG;
    m->have_sort = false;
    return run_all(m, [rows, nrows](gs_ctx *c, int) { return gs_push_splat(c, rows, nrows); });   // (the uploads of the devices run side by side)
}

GS_API int gs_multi_load_ply(gs_multi *m, const void *bytes, size_t nbytes)
{
    if (!m) return GS_E_BADARG;
    m->have_sort = false;
    return run_all(m, [bytes, nbytes](gs_ctx *c, int) { return gs_load_ply(c, bytes, nbytes); });
}

GS_API size_t gs_multi_count(const gs_multi *m) { return (m && !m->f.empty()) ? gs_count(m->f[0]->ctx) : 0; }

GS_API int gs_multi_set_option(gs_multi *m, int option, int64_t value)
{
    if (!m) return GS_E_BADARG;
    return run_all(m, [option, value](gs_ctx *c, int) { return gs_set_option(c, option, value); });
}

GS_API int gs_multi_sort(gs_multi *m, const float view[4], const float *cutout16, const gs_render_params *views, int nviews)
{
    if (!m || !view) return GS_E_BADARG;
    int rc = check_views(m, views, nviews);
    if (rc != GS_OK) return rc;
    // (a synchronous render that draws its frame again hands these very arrays back: memmove, self-assignment)
    memmove(m->view, view, sizeof m->view);
    m->has_cutout = cutout16 != nullptr;
    if (cutout16) memmove(m->cutout, cutout16, sizeof m->cutout);
    m->snviews = nviews;
    for (int v = 0; v < nviews; v++) if (&m->sviews[v] != &views[v]) m->sviews[v] = views[v];
    m->have_sort = true;
    struct A { float view[4], cutout[16]; bool has_cutout; gs_render_params views[2]; int nviews; } a;
    memcpy(a.view, m->view, sizeof a.view); memcpy(a.cutout, m->cutout, sizeof a.cutout); a.has_cutout = m->has_cutout;
    a.views[0] = views[0]; a.views[1] = views[nviews > 1 ? 1 : 0]; a.nviews = nviews;
    // (nothing is handed back: each context's own enqueue thread does the launching; failures surface at gs_multi_sync)
    return post(m, [a](gs_ctx *c, int) { return gs_sort_gathered(c, a.view, a.has_cutout ? a.cutout : nullptr, a.views, a.nviews); });
}

static int render_once(gs_multi *m, const gs_render_params *views, int nviews, uint8_t *const *host_frames, size_t stride,
                       void *const *device_frames, bool device, uint32_t flags)
{
    struct A { gs_render_params views[2]; int nviews; uint8_t *host[2]; void *dev[2]; bool has_dev; size_t stride; uint32_t flags; int world; } a;
    a.views[0] = views[0]; a.views[1] = views[nviews > 1 ? 1 : 0]; a.nviews = nviews; a.stride = stride; a.flags = flags | GS_RENDER_ASYNC;
    a.world = (int)m->f.size();
    for (int v = 0; v < 2; v++) { a.host[v] = (host_frames && v < nviews) ? host_frames[v] : nullptr; a.dev[v] = (device_frames && v < nviews) ? device_frames[v] : nullptr; }
    a.has_dev = device_frames != nullptr;
    if (device) {
        return post(m, [a](gs_ctx *c, int rank) { return gs_render_gathered(c, a.views, a.nviews, 0, (rank == 0 && a.has_dev) ? a.dev : nullptr, a.flags); });
    }
    return post(m, [a](gs_ctx *c, int rank) {
        gs_piece mine[128]; int n = 0;
        int rc = pieces_of(rank, a.world, a.views, a.nviews, mine, &n);
        for (int i = 0; rc == GS_OK && i < n; i++) {
            gs_render_params p = a.views[mine[i].view];
            p.x0 = mine[i].x0; p.x1 = mine[i].x1; p.flags = a.flags;
            const size_t st = a.stride ? a.stride : (size_t)p.fb_width * 4;
            rc = gs_render(c, &p, a.host[mine[i].view] + (size_t)p.x0 * 4, st);      // the strip lands in its columns of the caller's frame
        }
        return rc;
    });
}

static int render_multi(gs_multi *m, const gs_render_params *views, int nviews, uint8_t *const *host_frames, size_t stride,
                        void *const *device_frames, bool device, uint32_t flags)
{
    if (!m) return GS_E_BADARG;
    int rc = check_views(m, views, nviews);
    if (rc != GS_OK) return rc;
    if (!device) {
        if (!host_frames) { snprintf(m->err, sizeof m->err, "gs_multi_render: host_frames is NULL"); return GS_E_BADARG; }
        for (int v = 0; v < nviews; v++) {
            if (!host_frames[v]) { snprintf(m->err, sizeof m->err, "gs_multi_render: host_frames[%d] is NULL", v); return GS_E_BADARG; }
            if (stride && stride < (size_t)views[v].fb_width * 4) { snprintf(m->err, sizeof m->err, "stride %zu smaller than a row", stride); return GS_E_BADARG; }
        }
    }
    if (flags & GS_RENDER_COUNT_FRAGS) { snprintf(m->err, sizeof m->err, "counting renders are per context (gs_multi_ctx + gs_render_device)"); return GS_E_BADARG; }
    const bool async = (flags & GS_RENDER_ASYNC) != 0;
    rc = render_once(m, views, nviews, host_frames, stride, device_frames, device, flags);
    if (rc != GS_OK || async) return rc;
    // a synchronous frame: complete when this returns.  A context may report that its frame came back incomplete (GS_E_RETRY: it
    // outgrew a buffer, or needed the binning round it had skipped); every context then draws the frame again -- the gathered
    // form needs all of them anyway -- with the state the library has adapted meanwhile.
    for (int attempt = 0;; attempt++) {
        rc = run_all(m, [](gs_ctx *c, int) { return gs_sync(c); });
        if (rc != GS_E_RETRY) return rc;
        if (attempt >= 3 || !m->have_sort) return rc;
        rc = gs_multi_sort(m, m->view, m->has_cutout ? m->cutout : nullptr, m->sviews, m->snviews);
        if (rc != GS_OK) return rc;
        rc = render_once(m, views, nviews, host_frames, stride, device_frames, device, flags);
        if (rc != GS_OK) return rc;
    }
}

GS_API int gs_multi_render(gs_multi *m, const gs_render_params *views, int nviews, uint8_t *const *host_frames, size_t stride, uint32_t flags)
{
    return render_multi(m, views, nviews, host_frames, stride, nullptr, false, flags);
}

GS_API int gs_multi_render_device(gs_multi *m, const gs_render_params *views, int nviews, void *const *device_frames, uint32_t flags)
{
    return render_multi(m, views, nviews, nullptr, 0, device_frames, true, flags);
}

GS_API int gs_multi_read(gs_multi *m, int view, uint8_t *rgba_out, size_t stride)
{
    if (!m || !rgba_out) return GS_E_BADARG;
    int rc = collect(m);
    if (rc != GS_OK) return rc;
    rc = gs_read_gathered(m->f[0]->ctx, view, rgba_out, stride);
    if (rc != GS_OK) snprintf(m->err, sizeof m->err, "%s", gs_last_error(m->f[0]->ctx));
    return rc;
}

GS_API int gs_multi_sync(gs_multi *m)
{
    if (!m) return GS_E_BADARG;
    return run_all(m, [](gs_ctx *c, int) { return gs_sync(c); });
}

}  // extern "C"

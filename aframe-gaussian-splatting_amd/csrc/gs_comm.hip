// gs_comm.hip -- frames over several GPUs: partition of the viewport(s) into pieces, one RCCL communicator per
// context, the gather of a frame's pieces on its root queued on the frame's own pipeline-lane stream, and the kernel that
// assembles the row-major image(s) there.  The reference draws on one WebGL context (index.js:184-199); this is the part of
// the north star that has no counterpart in it.
//
// RCCL is reached through dlopen: the library must not drag a 500 MB dependency into single-GPU users, and a process that
// already holds a copy (torch ships its own librccl.so.1) gets that one.  send/recv pairs inside one group per frame: every
// peer uses its own xGMI link to the root (7 links in parallel, never ring-bound); messages are H x w x 4 bytes
// (1.04 MB at 1080p / 8, 4.15 MB at 4K / 8).
// Ordering: a communicator wants the same sequence of operations on every rank, one caller at a time, and frames in flight
// are enqueued by different threads (one worker per pipeline lane).  Every gathered frame takes a ticket on the caller's
// thread (same order on every rank, because every rank makes the same calls); a lane's worker issues its frame's group only
// when the tickets before it have been issued.
#include <dlfcn.h>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <vector>
#include "gs_internal.h"

extern "C" int gs_lane_call(gs_ctx *ctx, bool async, std::function<int(gs_ctx *)> call);   // gs_api.hip
extern "C" int gs_sort_two_views(gs_ctx *ctx, const float view[4], const float *cutout16);       // gs_api.hip

// the slice of rccl.h this file needs (types only; the functions come from dlsym)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                                          // 0 = ncclSuccess
enum { gsNcclUint8 = 1 };                                          // ncclUint8 / ncclChar share the element size

#define GS_ASSEMBLE_MAX 16                                          // pieces per k_assemble launch

struct GsComm {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    bool self_copy = false;
    std::mutex m;                                                  // tickets: gathers are issued in frame order, one at a time
    std::condition_variable cv;
    uint64_t next_ticket = 0, next_issue = 0;
};

#define FAILC(code, ...) do { snprintf(GS_ERRBUF(ctx), GS_ERRLEN, __VA_ARGS__); return (code); } while (0)

static int load_rccl(gs_ctx *ctx, GsComm *c)
{
    if (c->lib) return GS_OK;
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void *h = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);            // a copy this process already holds (e.g. torch's)
    for (int i = 0; !h && i < 3; i++) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) FAILC(GS_E_STATE, "RCCL not found (dlopen librccl.so.1): %s", dlerror());
#define SYM(field, name) do { *(void **)(&c->field) = dlsym(h, name); if (!c->field) { dlclose(h); FAILC(GS_E_STATE, "RCCL lacks %s", name); } } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    c->lib = h;
    return GS_OK;
}

#define NCCL_OK(ctx, c, call) do { const ncclResult_t _r = (call); if (_r != 0) {                                        \
        snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "%s failed: %s", #call, (c)->GetErrorString ? (c)->GetErrorString(_r) : "?"); \
        return GS_E_HIP; } } while (0)

namespace {

// staging (pieces, tight rows of w pixels each) -> row-major image(s) of W pixels per row; one thread per 16 bytes, all the
// pieces of a frame in ONE launch (blockIdx.y = piece): at 8 ranks the root would otherwise queue 8 launches per frame
struct AssemblePieces {
    int n;
    int x0[GS_ASSEMBLE_MAX], w[GS_ASSEMBLE_MAX], H[GS_ASSEMBLE_MAX], W[GS_ASSEMBLE_MAX];
    uint32_t off256[GS_ASSEMBLE_MAX];                               // staging offset / 256
    uint8_t *frame[GS_ASSEMBLE_MAX];
};

__global__ __launch_bounds__(256) void k_assemble(const uint8_t *__restrict__ stage, AssemblePieces a)
{
    const int pi = blockIdx.y;
    const int H = a.H[pi], w = a.w[pi], W = a.W[pi], x0 = a.x0[pi];
    const uint8_t *piece = stage + (size_t)a.off256[pi] * 256u;
    uint8_t *frame = a.frame[pi];
    const uint32_t per_row = (uint32_t)(w + 3) / 4u;                // 4 pixels = 16 bytes per thread
    const uint32_t total = per_row * (uint32_t)H;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t y = i / per_row, q = i % per_row;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(piece) + (size_t)y * w + q * 4u;
        uint32_t *dst = reinterpret_cast<uint32_t *>(frame) + (size_t)y * W + x0 + q * 4u;
        const int left = w - (int)(q * 4u);
        if (left >= 4 && ((w | W | x0) & 3) == 0) *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(src);
        else for (int k = 0; k < (left < 4 ? left : 4); k++) dst[k] = src[k];
    }
}

void split_strips(int view, int width, int r0, int nranks, std::vector<gs_piece> &out)
{
    const int tiles = (width + GS_TILE - 1) / GS_TILE;
    for (int k = 0; k < nranks; k++) {
        const int t0 = (int)((long long)tiles * k / nranks), t1 = (int)((long long)tiles * (k + 1) / nranks);
        const int x0 = t0 * GS_TILE, x1 = t1 * GS_TILE < width ? t1 * GS_TILE : width;
        if (x1 > x0) out.push_back(gs_piece{ view, x0, x1, r0 + k });
    }
}

struct GatherJob {
    std::vector<gs_piece> pieces;                                  // all pieces of the frame, gather order
    std::vector<size_t> off;                                       // staging offset of each
    int nviews, W[2], H[2], root;
    uint8_t *out[2];                                               // root: caller's device frames or nullptr
    uint64_t ticket;
};

int ensure_gather_buffers(gs_ctx *L, const GatherJob &j, size_t stage_bytes, bool is_root)
{
    gs_ctx *ctx = L;
    if (stage_bytes > L->gstage_cap) {
        GS_HIP(hipStreamSynchronize(L->stream));
        if (L->gstage) (void)hipFree(L->gstage);
        L->gstage = nullptr; L->gstage_cap = 0;
        GS_HIP(hipMalloc((void **)&L->gstage, stage_bytes));
        L->gstage_cap = stage_bytes;
    }
    for (int v = 0; is_root && v < j.nviews; v++) {
        const size_t fb = (size_t)j.W[v] * j.H[v] * 4;
        if (!j.out[v] && fb > L->gframe_cap[v]) {
            GS_HIP(hipStreamSynchronize(L->stream));
            if (L->gframe[v]) (void)hipFree(L->gframe[v]);
            L->gframe[v] = nullptr; L->gframe_cap[v] = 0;
            GS_HIP(hipMalloc((void **)&L->gframe[v], fb));
            L->gframe_cap[v] = fb;
        }
    }
    return GS_OK;
}

// runs on the frame's lane (worker thread for asynchronous frames) after the renders of the rank's own pieces were enqueued
int issue_gather(gs_ctx *L, const GatherJob &j)
{
    gs_ctx *ctx = L;
    gs_ctx *P = gs_root(L);
    GsComm *c = P->comm;
    const bool is_root = c->rank == j.root;
    int rc = GS_OK;
    {   // in ticket order, one group at a time
        std::unique_lock<std::mutex> lk(c->m);
        c->cv.wait(lk, [&] { return c->next_issue == j.ticket; });
        bool any = false;
        for (size_t i = 0; i < j.pieces.size(); i++) {
            const gs_piece &p = j.pieces[i];
            const bool mine = p.owner == c->rank;
            if ((mine && !is_root) || (is_root && (!mine || c->self_copy))) any = true;
        }
        if (any) {
            ncclResult_t r = c->GroupStart();
            for (size_t i = 0; r == 0 && i < j.pieces.size(); i++) {
                const gs_piece &p = j.pieces[i];
                const size_t bytes = (size_t)(p.x1 - p.x0) * j.H[p.view] * 4;
                const bool mine = p.owner == c->rank;
                // the root's own pieces are rendered where they are assembled from; with self_copy they travel through RCCL
                // like everybody's (rendered into the second half of the staging buffer, received into the first)
                if (mine && (!is_root || c->self_copy))
                    r = c->Send(L->gstage + (is_root ? L->gstage_cap / 2 : 0) + j.off[i], bytes, gsNcclUint8, j.root, c->comm, L->stream);
                if (r == 0 && is_root && (!mine || c->self_copy))
                    r = c->Recv(L->gstage + j.off[i], bytes, gsNcclUint8, p.owner, c->comm, L->stream);
            }
            const ncclResult_t r2 = c->GroupEnd();
            if (r != 0 || r2 != 0) {
                snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "RCCL gather failed: %s", c->GetErrorString(r != 0 ? r : r2));
                rc = GS_E_HIP;
            }
        }
        c->next_issue++;
        c->cv.notify_all();
    }
    if (rc != GS_OK || !is_root) return rc;
    for (size_t base = 0; base < j.pieces.size(); base += GS_ASSEMBLE_MAX) {
        AssemblePieces a;
        uint32_t most = 1;
        a.n = (int)(j.pieces.size() - base < GS_ASSEMBLE_MAX ? j.pieces.size() - base : GS_ASSEMBLE_MAX);
        for (int k = 0; k < a.n; k++) {
            const gs_piece &p = j.pieces[base + k];
            a.x0[k] = p.x0; a.w[k] = p.x1 - p.x0; a.H[k] = j.H[p.view]; a.W[k] = j.W[p.view];
            a.off256[k] = (uint32_t)(j.off[base + k] / 256);
            a.frame[k] = j.out[p.view] ? j.out[p.view] : L->gframe[p.view];
            const uint32_t work = (uint32_t)((a.w[k] + 3) / 4) * (uint32_t)a.H[k];
            if (work > most) most = work;
        }
        uint32_t g = gs_div_up(most, 256); if (g > 1024) g = 1024;
        hipLaunchKernelGGL(k_assemble, dim3(g, a.n), dim3(256), 0, L->stream, L->gstage, a);
    }
    GS_HIP(hipGetLastError());
    L->gviews = j.nviews;
    for (int v = 0; v < j.nviews; v++) { L->gw[v] = j.W[v]; L->gh[v] = j.H[v]; }
    return GS_OK;
}

}  // namespace

void gs_comm_free_lane(gs_ctx *lane)
{
    if (lane->gstage) { (void)hipFree(lane->gstage); lane->gstage = nullptr; lane->gstage_cap = 0; }
    for (int v = 0; v < 2; v++) if (lane->gframe[v]) { (void)hipFree(lane->gframe[v]); lane->gframe[v] = nullptr; lane->gframe_cap[v] = 0; }
}

extern "C" {

GS_API int gs_partition(int nviews, const int *widths, int world, gs_piece *out, int max_pieces)
{
    if (nviews < 1 || nviews > 2 || !widths || world < 1 || !out) return GS_E_BADARG;
    for (int v = 0; v < nviews; v++) if (widths[v] <= 0) return GS_E_BADARG;
    std::vector<gs_piece> p;
    if (nviews == 1) split_strips(0, widths[0], 0, world, p);
    else if (world == 1) { p.push_back(gs_piece{ 0, 0, widths[0], 0 }); p.push_back(gs_piece{ 1, 0, widths[1], 0 }); }
    else {
        const int n0 = (world + 1) / 2;                              // eye 0 -> ranks [0, n0), eye 1 -> the rest (eye k -> rank k at world 2)
        split_strips(0, widths[0], 0, n0, p);
        split_strips(1, widths[1], n0, world - n0, p);
    }
    if ((int)p.size() > max_pieces) return GS_E_BADARG;
    for (size_t i = 0; i < p.size(); i++) out[i] = p[i];
    return (int)p.size();
}

GS_API int gs_comm_unique_id(gs_ctx *ctx, void *id_out)
{
    if (!ctx || !id_out) return GS_E_BADARG;
    if (!ctx->comm) { ctx->comm = new (std::nothrow) GsComm(); if (!ctx->comm) FAILC(GS_E_OOM, "out of host memory"); }
    int rc = load_rccl(ctx, ctx->comm);
    if (rc != GS_OK) return rc;
    GS_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    NCCL_OK(ctx, ctx->comm, ctx->comm->GetUniqueId(&id));
    memcpy(id_out, &id, GS_COMM_ID_BYTES);
    return GS_OK;
}

GS_API int gs_comm_init(gs_ctx *ctx, const void *id, int rank, int world)
{
    if (!ctx || !id) return GS_E_BADARG;
    if (world < 1 || rank < 0 || rank >= world) FAILC(GS_E_BADARG, "gs_comm_init: rank %d of %d", rank, world);
    if (!ctx->comm) { ctx->comm = new (std::nothrow) GsComm(); if (!ctx->comm) FAILC(GS_E_OOM, "out of host memory"); }
    GsComm *c = ctx->comm;
    if (c->comm) FAILC(GS_E_STATE, "gs_comm_init: the context already joined a communicator");
    int rc = load_rccl(ctx, c);
    if (rc != GS_OK) return rc;
    GS_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(&uid, id, GS_COMM_ID_BYTES);
    NCCL_OK(ctx, c, c->CommInitRank(&c->comm, world, uid, rank));
    c->rank = rank; c->world = world;
    return GS_OK;
}

GS_API int gs_comm_destroy(gs_ctx *ctx)
{
    if (!ctx || !ctx->comm) return GS_OK;
    GsComm *c = ctx->comm;
    (void)gs_sync(ctx);                                              // nothing of ours is left in the communicator's streams
    if (c->comm) (void)c->CommDestroy(c->comm);
    // (the library handle stays open: RCCL does not survive being unloaded under a process that used it)
    delete c;
    ctx->comm = nullptr;
    return GS_OK;
}

GS_API int gs_render_gathered(gs_ctx *ctx, const gs_render_params *views, int nviews, int root, void *const *device_frames, uint32_t flags)
{
    if (!ctx || !views) return GS_E_BADARG;
    if (nviews < 1 || nviews > 2) FAILC(GS_E_BADARG, "gs_render_gathered: %d views (1 or 2)", nviews);
    GsComm *c = ctx->comm;
    const int world = c && c->comm ? c->world : 1, rank = c && c->comm ? c->rank : 0;
    if (root < 0 || root >= world) FAILC(GS_E_BADARG, "gs_render_gathered: root %d of %d ranks", root, world);
    if (flags & GS_RENDER_COUNT_FRAGS) FAILC(GS_E_BADARG, "gs_render_gathered: counting renders are per context (gs_render_device)");
    if (!c) { ctx->comm = c = new (std::nothrow) GsComm(); if (!c) FAILC(GS_E_OOM, "out of host memory"); }   // world 1 without RCCL: tickets only
    GatherJob j;
    j.nviews = nviews; j.root = root;
    int widths[2] = { 0, 0 };
    for (int v = 0; v < nviews; v++) { widths[v] = j.W[v] = views[v].fb_width; j.H[v] = views[v].fb_height; j.out[v] = device_frames ? (uint8_t *)device_frames[v] : nullptr; }
    for (int v = nviews; v < 2; v++) { j.W[v] = j.H[v] = 0; j.out[v] = nullptr; }
    gs_piece pcs[128];
    const int np = gs_partition(nviews, widths, world, pcs, 128);
    if (np < 0) FAILC(GS_E_BADARG, "gs_render_gathered: bad frame sizes or more than 128 pieces");
    size_t off = 0;
    for (int i = 0; i < np; i++) {
        j.pieces.push_back(pcs[i]); j.off.push_back(off);
        off += ((size_t)(pcs[i].x1 - pcs[i].x0) * j.H[pcs[i].view] * 4 + 255) & ~(size_t)255;
    }
    const bool is_root = rank == root;
    const bool self = is_root && c->self_copy && c->comm;
    const size_t stage_bytes = self ? 2 * off : off;               // (self copy: rendered into the upper half, received into the lower)
    const bool async = (flags & GS_RENDER_ASYNC) != 0;
    GS_HIP(hipSetDevice(ctx->device));
    gs_ctx *L = ctx->lanes[ctx->cur];                              // the frame's lane: where its gs_sort ran
    // buffers: on the caller's thread, before anything of this frame is enqueued (the lane's previous frame is drained by
    // gs_lane_call below only if they must grow)
    if (stage_bytes > L->gstage_cap || (is_root && ((!j.out[0] && (size_t)j.W[0] * j.H[0] * 4 > L->gframe_cap[0]) ||
                                                    (nviews > 1 && !j.out[1] && (size_t)j.W[1] * j.H[1] * 4 > L->gframe_cap[1])))) {
        int rc = gs_lane_call(ctx, false, [&](gs_ctx *lane) { return ensure_gather_buffers(lane, j, stage_bytes, is_root); });
        if (rc != GS_OK) return rc;
    }
    // this rank's own pieces: ordinary strip renders into their staging slots
    for (int i = 0; i < np; i++) {
        if (pcs[i].owner != rank) continue;
        gs_render_params p = views[pcs[i].view];
        p.x0 = pcs[i].x0; p.x1 = pcs[i].x1;
        p.flags = (flags & ~(uint32_t)GS_RENDER_ASYNC) | (async ? GS_RENDER_ASYNC : 0u);
        GsFrameUniforms u;
        int rc = gs_fill_uniforms(ctx, &p, u);
        if (rc != GS_OK) return rc;
        rc = gs_render_uniforms(ctx, u, L->gstage + (self ? L->gstage_cap / 2 : 0) + j.off[i], nullptr, 0);
        if (rc != GS_OK) return rc;
    }
    { std::lock_guard<std::mutex> lk(c->m); j.ticket = c->next_ticket++; }
    int rc = gs_lane_call(ctx, async, [j](gs_ctx *lane) { return issue_gather(lane, j); });
    if (rc == GS_OK && !async) GS_HIP(hipStreamSynchronize(L->stream));   // a synchronous call returns with the frame(s) complete on the root
    return rc;
}

GS_API int gs_sort_gathered(gs_ctx *ctx, const float view[4], const float *cutout16, const gs_render_params *views, int nviews)
{
    if (!ctx || !views) return GS_E_BADARG;
    if (nviews < 1 || nviews > 2) FAILC(GS_E_BADARG, "gs_sort_gathered: %d views (1 or 2)", nviews);
    GsComm *c = ctx->comm;
    const int world = c && c->comm ? c->world : 1, rank = c && c->comm ? c->rank : 0;
    int widths[2] = { views[0].fb_width, nviews > 1 ? views[1].fb_width : 0 };
    gs_piece pcs[128];
    const int np = gs_partition(nviews, widths, world, pcs, 128);
    if (np < 0) FAILC(GS_E_BADARG, "gs_sort_gathered: bad frame sizes or more than 128 pieces");
    int mine = -1, count = 0;
    for (int i = 0; i < np; i++) if (pcs[i].owner == rank) { mine = i; count++; }
    if (count == 2) return gs_sort_two_views(ctx, view, cutout16);                // both eyes here: the whole order, once
    if (count != 1) return gs_sort(ctx, view, cutout16, nullptr, nullptr);       // nothing to draw (or an unusual partition): the whole order
    gs_render_params p = views[pcs[mine].view];
    p.x0 = pcs[mine].x0; p.x1 = pcs[mine].x1;
    return gs_sort_for(ctx, view, cutout16, &p, nullptr, nullptr);
}

GS_API int gs_gathered_size(gs_ctx *ctx, int view, int *width, int *height)
{
    if (!ctx || !width || !height) return GS_E_BADARG;
    int rc = gs_lane_call(ctx, false, [](gs_ctx *) { return GS_OK; });   // the lane's worker has enqueued (and recorded) the frame
    if (rc != GS_OK) return rc;
    gs_ctx *L = ctx->lanes[ctx->cur];
    if (view < 0 || view >= L->gviews) FAILC(GS_E_STATE, "gs_gathered_size: no gathered frame for view %d on this context", view);
    *width = L->gw[view]; *height = L->gh[view];
    return GS_OK;
}

GS_API int gs_read_gathered(gs_ctx *ctx, int view, uint8_t *rgba_out, size_t stride)
{
    if (!ctx || !rgba_out) return GS_E_BADARG;
    GS_HIP(hipSetDevice(ctx->device));
    gs_ctx *L = ctx->lanes[ctx->cur];
    int rc = gs_lane_call(ctx, false, [](gs_ctx *) { return GS_OK; });   // the lane's worker has enqueued everything
    if (rc != GS_OK) return rc;
    GS_HIP(hipStreamSynchronize(L->stream));
    if (view < 0 || view >= L->gviews || !L->gframe[view]) FAILC(GS_E_STATE, "gs_read_gathered: no gathered frame for view %d on this context (root only, context-owned frames only)", view);
    const size_t row = (size_t)L->gw[view] * 4;
    if (!stride) stride = row;
    if (stride < row) FAILC(GS_E_BADARG, "stride %zu smaller than a row (%zu bytes)", stride, row);
    GS_HIP(hipMemcpy2D(rgba_out, stride, L->gframe[view], row, row, (size_t)L->gh[view], hipMemcpyDeviceToHost));
    return GS_OK;
}

}  // extern "C"

int gs_comm_set_self_copy(gs_ctx *ctx, bool on)
{
    if (!ctx->comm) { ctx->comm = new (std::nothrow) GsComm(); if (!ctx->comm) return GS_E_OOM; }
    ctx->comm->self_copy = on;
    return GS_OK;
}

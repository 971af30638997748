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
#include <atomic>
#include <chrono>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <unistd.h>
#include "gs_internal.h"

extern "C" int gs_lane_call(gs_ctx *ctx, bool async, std::function<int(gs_ctx *)> call, bool always = false);   // gs_api.hip
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
    ncclComm_t comm = nullptr;                                     // RCCL's communicator, or -- in-process transport -- a GsLoopEndpoint
    int rank = 0, world = 1;
    int transport = 0;                                             // GS_OPT_COMM_TRANSPORT: what gs_comm_unique_id makes an id for (0 RCCL, 1 in-process)
    bool loop = false;                                             // the communicator joined is an in-process one
    bool self_copy = false;
    std::mutex m;                                                  // tickets: gathers are issued in frame order, one at a time
    std::condition_variable cv;
    uint64_t next_ticket = 0, next_issue = 0;
    uint64_t share_seq = 0;                                        // frames sorted through GS_OPT_SORT_SHARE so far: frame f belongs to rank f mod world
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

// ---------------------------------------------------------------- in-process transport (GS_OPT_COMM_TRANSPORT = 1)
// Several contexts of ONE process -- on different GPUs (gs_create_multi: one host process drives the node) or on the same one
// (the GPU test tier runs the whole multi-rank path, world 2 / 3 / 8, on a single device) -- exchange their pieces without
// RCCL, behind the same five calls.  The semantics are ncclSend / ncclRecv's: messages between a pair of ranks match in
// order; a send returns at once and its source buffer may be reused in stream order; a receive is complete in stream order.
//   send   the piece is copied into a mailbox buffer on the sender's device on the sender's stream, an event recorded
//          behind the copy, and the buffer posted to the (source, destination) queue of the hub;
//   recv   (performed at GroupEnd, after the group's sends: a rank may send to itself) waits ON THE HOST until the matching
//          send has been posted -- bounded: a peer that never sends fails the frame after GS_LOOP_TIMEOUT_S instead of hanging
//          it --, makes its stream wait for the sender's event, pulls the buffer (a peer copy when the devices differ), records
//          the buffer's release event and returns it to the pool; the next sender that takes it waits for that event.
// A stream never waits for work that has not been submitted yet, so the hardware queues cannot deadlock whatever the number
// of contexts sharing them.
#define GS_LOOP_MAGIC "GSLOOPBK"
#ifndef GS_LOOP_TIMEOUT_S
#define GS_LOOP_TIMEOUT_S 60                                       // environment GS_COMM_TIMEOUT_S overrides (read when a communicator is opened)
#endif

namespace {

struct GsLoopBuf {
    void *p = nullptr; size_t cap = 0, bytes = 0; int dev = -1;
    hipEvent_t ready = nullptr;                                    // sender's device: the copy into the buffer is done
    hipEvent_t done = nullptr; int done_dev = -1; bool done_valid = false;   // receiver's device: the buffer has been read
};

struct GsLoopHub {
    std::mutex m;
    std::condition_variable cv;
    int world = 0, refs = 0, timeout_s = GS_LOOP_TIMEOUT_S;
    int jitter_us = 0;                                             // test hook (GS_COMM_TEST_JITTER_US): every post is held back by a pseudo-random time below this
    std::atomic<uint32_t> jitter_seq{0};
    bool failed = false;                                           // a rank gave up: everybody waiting fails too
    std::map<std::pair<int, int>, std::deque<GsLoopBuf *>> box;    // (source, destination) -> posted messages, in order
    std::vector<GsLoopBuf *> pool;                                 // free buffers
    std::string key;
};

struct GsLoopEndpoint { GsLoopHub *hub; int rank; int dev; };

std::mutex g_loop_m;
std::map<std::string, GsLoopHub *> g_loop_hubs;
std::atomic<uint64_t> g_loop_serial{1};

struct LoopRecv { void *buf; size_t bytes; int peer; GsLoopEndpoint *ep; hipStream_t st; };
thread_local std::vector<LoopRecv> tl_loop_recvs;
thread_local int tl_loop_group = 0;
thread_local char tl_loop_err[160] = "";
enum { LOOP_OK = 0, LOOP_ERR = 1 };

int loop_fail(const char *what, hipError_t e = hipSuccess)
{
    if (e != hipSuccess) snprintf(tl_loop_err, sizeof tl_loop_err, "%s: %s", what, hipGetErrorString(e));
    else snprintf(tl_loop_err, sizeof tl_loop_err, "%s", what);
    return LOOP_ERR;
}
#define LOOP_HIP(call) do { const hipError_t _e = (call); if (_e != hipSuccess) return loop_fail(#call, _e); } while (0)

void loop_free_buf(GsLoopBuf *b)
{
    int cur = 0; (void)hipGetDevice(&cur);
    (void)hipSetDevice(b->dev);
    if (b->p) (void)hipFree(b->p);
    if (b->ready) (void)hipEventDestroy(b->ready);
    if (b->done) { (void)hipSetDevice(b->done_dev); (void)hipEventDestroy(b->done); }
    (void)hipSetDevice(cur);
    delete b;
}

ncclResult_t loop_send(const void *src, size_t bytes, int, int peer, ncclComm_t comm, hipStream_t st)
{
    GsLoopEndpoint *ep = reinterpret_cast<GsLoopEndpoint *>(comm);
    GsLoopHub *h = ep->hub;
    if (peer < 0 || peer >= h->world) return loop_fail("send: peer out of range");
    GsLoopBuf *b = nullptr;
    {   // smallest free buffer of this device that is large enough
        std::lock_guard<std::mutex> lk(h->m);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < h->pool.size(); i++)
            if (h->pool[i]->dev == ep->dev && h->pool[i]->cap >= bytes && (best == (size_t)-1 || h->pool[i]->cap < h->pool[best]->cap)) best = i;
        if (best != (size_t)-1) { b = h->pool[best]; h->pool.erase(h->pool.begin() + (long)best); }
    }
    if (!b) {
        b = new (std::nothrow) GsLoopBuf();
        if (!b) return loop_fail("send: out of host memory");
        b->dev = ep->dev; b->cap = (bytes + 0xFFFFF) & ~(size_t)0xFFFFF;
        hipError_t e = hipMalloc(&b->p, b->cap);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b->ready, hipEventDisableTiming);
        if (e != hipSuccess) { loop_free_buf(b); return loop_fail("send: mailbox allocation", e); }
    }
    if (b->done_valid) { LOOP_HIP(hipStreamWaitEvent(st, b->done, 0)); b->done_valid = false; }   // its previous reader has finished
    b->bytes = bytes;
    if (bytes) LOOP_HIP(hipMemcpyAsync(b->p, src, bytes, hipMemcpyDeviceToDevice, st));
    LOOP_HIP(hipEventRecord(b->ready, st));
    if (h->jitter_us > 0) {
        // RCCL orders the messages of ONE pair of ranks and nothing else: the pieces of a frame may reach the root in any order across
        // its peers, and a peer's piece of the next frame before another peer's piece of this one.  The test hook makes that happen here.
        uint32_t x = (h->jitter_seq.fetch_add(1) + 1u) * 2654435761u ^ ((uint32_t)ep->rank * 40503u);
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        usleep(x % (uint32_t)h->jitter_us);
    }
    { std::lock_guard<std::mutex> lk(h->m); h->box[std::make_pair(ep->rank, peer)].push_back(b); }
    h->cv.notify_all();
    return LOOP_OK;
}

int loop_recv_now(const LoopRecv &r)
{
    GsLoopHub *h = r.ep->hub;
    GsLoopBuf *b = nullptr;
    {
        std::unique_lock<std::mutex> lk(h->m);
        std::deque<GsLoopBuf *> &q = h->box[std::make_pair(r.peer, r.ep->rank)];
        if (!h->cv.wait_for(lk, std::chrono::seconds(h->timeout_s), [&] { return h->failed || !q.empty(); })) {
            h->failed = true; h->cv.notify_all();
            snprintf(tl_loop_err, sizeof tl_loop_err, "recv: rank %d posted nothing for rank %d within %d s", r.peer, r.ep->rank, h->timeout_s);
            return LOOP_ERR;
        }
        if (q.empty()) return loop_fail("recv: another rank of the in-process communicator failed");
        b = q.front(); q.pop_front();
    }
    int rc = LOOP_OK;
    if (b->bytes != r.bytes) { snprintf(tl_loop_err, sizeof tl_loop_err, "recv: rank %d sent %zu bytes, %zu expected", r.peer, b->bytes, r.bytes); rc = LOOP_ERR; }
    hipError_t e = hipSuccess;
    if (rc == LOOP_OK) e = hipStreamWaitEvent(r.st, b->ready, 0);
    if (rc == LOOP_OK && e == hipSuccess && r.bytes) e = hipMemcpyAsync(r.buf, b->p, r.bytes, hipMemcpyDefault, r.st);
    if (rc == LOOP_OK && e == hipSuccess) {
        if (b->done && b->done_dev != r.ep->dev) { (void)hipEventDestroy(b->done); b->done = nullptr; }
        if (!b->done) { e = hipEventCreateWithFlags(&b->done, hipEventDisableTiming); b->done_dev = r.ep->dev; }
        if (e == hipSuccess) e = hipEventRecord(b->done, r.st);
        if (e == hipSuccess) b->done_valid = true;
    }
    if (e != hipSuccess) rc = loop_fail("recv", e);
    { std::lock_guard<std::mutex> lk(h->m); h->pool.push_back(b); if (rc != LOOP_OK) h->failed = true; }
    if (rc != LOOP_OK) h->cv.notify_all();
    return rc;
}

ncclResult_t loop_recv(void *dst, size_t bytes, int, int peer, ncclComm_t comm, hipStream_t st)
{
    GsLoopEndpoint *ep = reinterpret_cast<GsLoopEndpoint *>(comm);
    if (peer < 0 || peer >= ep->hub->world) return loop_fail("recv: peer out of range");
    const LoopRecv r = { dst, bytes, peer, ep, st };
    if (tl_loop_group > 0) { tl_loop_recvs.push_back(r); return LOOP_OK; }
    return loop_recv_now(r);
}

ncclResult_t loop_group_start() { tl_loop_group++; return LOOP_OK; }
ncclResult_t loop_group_end()
{
    if (tl_loop_group > 0 && --tl_loop_group > 0) return LOOP_OK;
    int rc = LOOP_OK;
    for (size_t i = 0; i < tl_loop_recvs.size(); i++) if (rc == LOOP_OK) rc = loop_recv_now(tl_loop_recvs[i]);
    tl_loop_recvs.clear();
    return rc;
}
const char *loop_error_string(ncclResult_t) { return tl_loop_err; }

ncclResult_t loop_comm_destroy(ncclComm_t comm)
{
    GsLoopEndpoint *ep = reinterpret_cast<GsLoopEndpoint *>(comm);
    GsLoopHub *h = ep->hub;
    bool last;
    { std::lock_guard<std::mutex> g(g_loop_m); last = --h->refs == 0; if (last) g_loop_hubs.erase(h->key); }
    if (last) {                                                    // (every rank has drained its streams: gs_comm_destroy syncs first)
        for (auto &kv : h->box) for (GsLoopBuf *b : kv.second) loop_free_buf(b);
        for (GsLoopBuf *b : h->pool) loop_free_buf(b);
        delete h;
    }
    delete ep;
    return LOOP_OK;
}

// join (or open) the hub an id names; never blocks: ranks may join in any order, from one thread or many
int loop_join(gs_ctx *ctx, GsComm *c, const void *id, int rank, int world)
{
    const std::string key((const char *)id, GS_COMM_ID_BYTES);
    GsLoopHub *h = nullptr;
    {
        std::lock_guard<std::mutex> g(g_loop_m);
        auto it = g_loop_hubs.find(key);
        if (it == g_loop_hubs.end()) {
            h = new (std::nothrow) GsLoopHub();
            if (!h) FAILC(GS_E_OOM, "out of host memory");
            h->world = world; h->key = key;
            const char *t = getenv("GS_COMM_TIMEOUT_S");
            if (t && atoi(t) > 0) h->timeout_s = atoi(t);
            const char *jt = getenv("GS_COMM_TEST_JITTER_US");
            if (jt && atoi(jt) > 0) h->jitter_us = atoi(jt);
            g_loop_hubs[key] = h;
        } else h = it->second;
        if (h->world != world) FAILC(GS_E_BADARG, "gs_comm_init: this in-process communicator has %d ranks, not %d", h->world, world);
        h->refs++;
    }
    GsLoopEndpoint *ep = new (std::nothrow) GsLoopEndpoint{ h, rank, ctx->device };
    if (!ep) {
        { std::lock_guard<std::mutex> g(g_loop_m); if (--h->refs == 0) { g_loop_hubs.erase(h->key); delete h; } }
        FAILC(GS_E_OOM, "out of host memory");
    }
    // contexts on different GPUs pull each other's mailboxes directly where the platform allows (else the runtime stages the copy)
    int ndev = 0; (void)hipGetDeviceCount(&ndev);
    for (int d = 0; d < ndev; d++) if (d != ctx->device) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, ctx->device, d) == hipSuccess && can) { const hipError_t e = hipDeviceEnablePeerAccess(d, 0); (void)e; (void)hipGetLastError(); }
    }
    c->Send = loop_send; c->Recv = loop_recv; c->GroupStart = loop_group_start; c->GroupEnd = loop_group_end;
    c->GetErrorString = loop_error_string; c->CommDestroy = loop_comm_destroy;
    c->comm = reinterpret_cast<ncclComm_t>(ep);
    c->loop = true; c->rank = rank; c->world = world;
    return GS_OK;
}

}  // namespace

namespace {

// staging (pieces, tight rows of w pixels each) -> row-major image(s) of W pixels per row; one thread per 16 bytes, all the
// pieces of a frame in ONE launch (blockIdx.y = piece): at 8 ranks the root would otherwise queue 8 launches per frame
struct AssemblePieces {
    int n;
    int x0[GS_ASSEMBLE_MAX], w[GS_ASSEMBLE_MAX], H[GS_ASSEMBLE_MAX], W[GS_ASSEMBLE_MAX];
    uint32_t off256[GS_ASSEMBLE_MAX];                               // staging offset / 256
    uint8_t *frame[GS_ASSEMBLE_MAX];
    GsControl *ctl; int first;                                      // the root lane's control block; first launch of the frame's assembly
};

// Every piece travels with a trailer behind its pixels: the completion word of the render that drew it (GsFrameUniforms::status: 0 =
// complete; an asynchronous piece that skipped its second binning round and had an unsaturated tile, or outgrew its pair buffers, is
// not).  The root ORs the trailers into its lane's frame_status -- what gs_frame_status_device() names for a gathered frame -- and
// raises its own sticky flag, so that ITS gs_sync() reports the frame (GS_E_RETRY) even when only a peer's piece was incomplete.
#define GS_PIECE_TRAILER 16u
__global__ __launch_bounds__(256) void k_assemble(const uint8_t *__restrict__ stage, AssemblePieces a)
{
    const int pi = blockIdx.y;
    const int H = a.H[pi], w = a.w[pi], W = a.W[pi], x0 = a.x0[pi];
    const uint8_t *piece = stage + (size_t)a.off256[pi] * 256u;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        uint32_t v = 0;
        for (int k = 0; k < a.n; k++) v |= *reinterpret_cast<const uint32_t *>(stage + (size_t)a.off256[k] * 256u + (size_t)a.w[k] * a.H[k] * 4u);
        a.ctl->frame_status = a.first ? v : (a.ctl->frame_status | v);
        if (v) a.ctl->round1_missed = 1;
    }
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
    {   // in ticket order, one group at a time (the ticket's holder works outside the mutex: the caller's thread takes the next
        // frames' tickets meanwhile, and an in-process receive may wait for its sender here)
        { std::unique_lock<std::mutex> lk(c->m); c->cv.wait(lk, [&] { return c->next_issue == j.ticket; }); }
        bool any = false;
        for (size_t i = 0; i < j.pieces.size(); i++) {
            const gs_piece &p = j.pieces[i];
            const bool mine = p.owner == c->rank;
            if ((mine && !is_root) || (is_root && (!mine || c->self_copy))) any = true;
        }
        if (any && c->comm) {
            ncclResult_t r = c->GroupStart();
            for (size_t i = 0; r == 0 && i < j.pieces.size(); i++) {
                const gs_piece &p = j.pieces[i];
                const size_t bytes = (size_t)(p.x1 - p.x0) * j.H[p.view] * 4 + GS_PIECE_TRAILER;   // (pixels + the piece's completion word)
                const bool mine = p.owner == c->rank;
                // the root's own pieces are rendered where they are assembled from; with self_copy they travel through the
                // transport like everybody's (rendered into the second half of the staging buffer, received into the first)
                if (mine && (!is_root || c->self_copy))
                    r = c->Send(L->gstage + (is_root ? L->gstage_cap / 2 : 0) + j.off[i], bytes, gsNcclUint8, j.root, c->comm, L->stream);
                if (r == 0 && is_root && (!mine || c->self_copy))
                    r = c->Recv(L->gstage + j.off[i], bytes, gsNcclUint8, p.owner, c->comm, L->stream);
            }
            const ncclResult_t r2 = c->GroupEnd();
            if (r != 0 || r2 != 0) {
                snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "%s gather failed: %s", c->loop ? "in-process" : "RCCL", c->GetErrorString(r != 0 ? r : r2));
                rc = GS_E_HIP;
            }
        }
        { std::lock_guard<std::mutex> lk(c->m); c->next_issue++; }
        c->cv.notify_all();
    }
    if (rc != GS_OK || !is_root) return rc;
    for (size_t base = 0; base < j.pieces.size(); base += GS_ASSEMBLE_MAX) {
        AssemblePieces a;
        a.ctl = L->ctl; a.first = base == 0 ? 1 : 0;
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

// ---- GS_OPT_SORT_SHARE: the ranks take turns sorting.  The owner of a frame runs the ordinary near-only sort of the whole view
// (gs_run_sort: the same kernels, hence the same order on whoever runs it) and sends two messages to every peer -- the head of its
// control block (min / max, V, the record count P', V': what the projection needs to address the order) and the first `cap`
// records of the order; a peer receives them where its own sort would have written them.  One send per peer, each over its own
// xGMI link; the messages have a fixed size (the host does not know P'), and an order that does not fit is flagged, not truncated
// silently.
#define GS_SHARE_HEAD offsetof(GsControl, n_visible)

__global__ void k_share_check(GsControl *ctl, uint32_t cap)
{
    if (ctl->n_sorted > cap) { ctl->n_sorted = cap; ctl->order_incomplete = 1; ctl->round1_missed = 1; }   // the frame is drawn from a truncated order: reported (GS_E_RETRY)
}

struct ShareJob {
    float view[4], cutout[16]; bool has_cutout;
    uint32_t near_req, cap; int owner; uint64_t ticket;
};

int issue_shared_sort(gs_ctx *L, const ShareJob &j)
{
    gs_ctx *ctx = L;
    GsComm *c = gs_root(L)->comm;
    const bool mine = c->rank == j.owner;
    int rc = GS_OK;
    // (the peers receive j.cap records: a tail sort keeps a whole segment of the order more than asked for -- any number of records)
    if (mine) { L->no_tail_sort = true; rc = gs_run_sort(L, j.view, j.has_cutout ? j.cutout : nullptr, nullptr, j.near_req); L->no_tail_sort = false; }
    else gs_remember_sort(L, j.view, j.has_cutout ? j.cutout : nullptr, nullptr, j.near_req);   // (what a fall-back to a whole local sort starts from)
    { std::unique_lock<std::mutex> lk(c->m); c->cv.wait(lk, [&] { return c->next_issue == j.ticket; }); }
    {   // whatever happened to the owner's sort, the exchange is issued: the peers wait for it
        ncclResult_t r = c->GroupStart();
        if (mine) {
            for (int p = 0; r == 0 && p < c->world; p++) {
                if (p == c->rank) continue;
                r = c->Send(L->ctl, GS_SHARE_HEAD, gsNcclUint8, p, c->comm, L->stream);
                if (r == 0) r = c->Send(L->val_a, (size_t)j.cap * 4, gsNcclUint8, p, c->comm, L->stream);
            }
        } else {
            r = c->Recv(L->ctl, GS_SHARE_HEAD, gsNcclUint8, j.owner, c->comm, L->stream);
            if (r == 0) r = c->Recv(L->val_a, (size_t)j.cap * 4, gsNcclUint8, j.owner, c->comm, L->stream);
        }
        const ncclResult_t r2 = c->GroupEnd();
        if ((r != 0 || r2 != 0) && rc == GS_OK) {
            snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "%s exchange of the shared sort failed: %s", c->loop ? "in-process" : "RCCL", c->GetErrorString(r != 0 ? r : r2));
            rc = GS_E_HIP;
        }
    }
    { std::lock_guard<std::mutex> lk(c->m); c->next_issue++; }
    c->cv.notify_all();
    if (!mine) {
        L->sorted = L->val_a; L->have_sort = true;
        hipLaunchKernelGGL(k_share_check, dim3(1), dim3(1), 0, L->stream, L->ctl, j.cap);   // (the owner holds its whole order)
    }
    return rc;
}

}  // namespace

// The in-process transport's receives wait ON THE HOST for their sender to post (GS_COMM_TIMEOUT_S): a rank that is being torn down
// while a feeder or enqueue thread of it -- or of a peer -- sits in such a wait must not make the teardown last a minute.  Cancelling
// fails the hub (as a rank that timed out does): every waiting receive returns at once, the communicator is dead for all its ranks.
void gs_comm_cancel(gs_ctx *ctx)
{
    if (!ctx || !ctx->comm || !ctx->comm->loop || !ctx->comm->comm) return;
    GsLoopHub *h = reinterpret_cast<GsLoopEndpoint *>(ctx->comm->comm)->hub;
    { std::lock_guard<std::mutex> lk(h->m); h->failed = true; }
    h->cv.notify_all();
}

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
    if (ctx->comm->transport == 1) {                             // in-process: the id only has to be unique in this process
        memset(id_out, 0, GS_COMM_ID_BYTES);
        snprintf((char *)id_out, GS_COMM_ID_BYTES, GS_LOOP_MAGIC "%ld-%llu", (long)getpid(), (unsigned long long)g_loop_serial.fetch_add(1));
        return GS_OK;
    }
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
    GS_HIP(hipSetDevice(ctx->device));
    if (memcmp(id, GS_LOOP_MAGIC, sizeof(GS_LOOP_MAGIC) - 1) == 0) return loop_join(ctx, c, id, rank, world);
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
        off += ((size_t)(pcs[i].x1 - pcs[i].x0) * j.H[pcs[i].view] * 4 + GS_PIECE_TRAILER + 255) & ~(size_t)255;
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
    // The frame's ticket is taken before anything of it is enqueued, and from here on the gather IS issued whatever happens to
    // this rank's own renders: the peers' receives (and the tickets behind this one) wait for it.
    { std::lock_guard<std::mutex> lk(c->m); j.ticket = c->next_ticket++; }
    int first = GS_OK;
    char first_err[GS_ERRLEN] = "";
    // this rank's own pieces: ordinary strip renders into their staging slots
    for (int i = 0; i < np && first == GS_OK; i++) {
        if (pcs[i].owner != rank) continue;
        gs_render_params p = views[pcs[i].view];
        p.x0 = pcs[i].x0; p.x1 = pcs[i].x1;
        p.flags = (flags & ~(uint32_t)GS_RENDER_ASYNC) | (async ? GS_RENDER_ASYNC : 0u);
        GsFrameUniforms u;
        first = gs_fill_uniforms(ctx, &p, u);
        uint8_t *slot = L->gstage + (self ? L->gstage_cap / 2 : 0) + j.off[i];
        u.status = reinterpret_cast<uint32_t *>(slot + (size_t)(pcs[i].x1 - pcs[i].x0) * j.H[pcs[i].view] * 4);   // the piece's trailer
        if (first == GS_OK) first = gs_render_uniforms(ctx, u, slot, nullptr, 0);
        if (first != GS_OK) memcpy(first_err, ctx->err, sizeof first_err);
    }
    int rc = gs_lane_call(ctx, async && first == GS_OK, [j](gs_ctx *lane) { return issue_gather(lane, j); }, /*always=*/true);
    if (first != GS_OK) { memcpy(ctx->err, first_err, sizeof ctx->err); return first; }
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
    // GS_OPT_SORT_SHARE: the ranks take turns sorting (frame f: rank f mod world) and exchange the nearest part of the order
    if (world > 1 && ctx->sort_share_permille > 0 && ctx->n > 0 && ctx->n <= ((size_t)1 << 25) && !ctx->wide_pairs && ctx->renderable) {
        ShareJob j;
        memcpy(j.view, view, sizeof j.view);
        j.has_cutout = cutout16 != nullptr;
        if (cutout16) memcpy(j.cutout, cutout16, sizeof j.cutout);
        const double nr = ceil((double)ctx->sort_share_permille / 1000.0 * (double)ctx->n);
        j.near_req = nr < 1 ? 1u : (uint32_t)nr;
        // room for what the near-only rule lets through beyond near_req (the threshold is a whole depth-histogram bin and ties)
        uint64_t cap = (uint64_t)j.near_req * 2u + 65536u;
        if (cap > ctx->n) cap = ctx->n;
        j.cap = (uint32_t)cap;
        // the frame's lane first (it can fail: scratch for a new lane), THEN the turn in the rota and the ticket: a rank that
        // left here with them taken and nothing queued would be out of step with its peers for good
        const int rb = gs_sort_call_begin(ctx, view, cutout16);
        if (rb != GS_OK) return rb;
        j.owner = (int)(c->share_seq++ % (uint64_t)world);
        { std::lock_guard<std::mutex> lk(c->m); j.ticket = c->next_ticket++; }
        std::function<int(gs_ctx *)> call;
        try { call = [j](gs_ctx *lane) { return issue_shared_sort(lane, j); }; }
        catch (...) {                                              // (the closure itself: run the exchange without one)
            const int rd = gs_lane_call(ctx, false, [](gs_ctx *) { return GS_OK; }, false);
            const int ri = issue_shared_sort(ctx->lanes[gs_frame_lane(ctx)], j);
            return rd != GS_OK ? rd : ri;
        }
        return gs_sort_call_issue(ctx, &call);
    }
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

int gs_comm_set_transport(gs_ctx *ctx, int transport)
{
    if (!ctx->comm) { ctx->comm = new (std::nothrow) GsComm(); if (!ctx->comm) return GS_E_OOM; }
    ctx->comm->transport = transport;
    return GS_OK;
}

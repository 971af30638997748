// gs_api.hip -- the C ABI (include/gs_splat.h): context lifetime, HBM residency, ingest, sort / render
// entry points, stats.  No compute happens on the host here; every hot-path stage is a HIP kernel and the
// library refuses to exist without a device (no CPU fallback).
#include <stddef.h>
#include <stdlib.h>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <set>
#include <vector>
#include "gs_internal.h"
#include "gs_ply.h"
#include "gs_host_tables.h"

#ifndef GS_NEAR_FLOOR_MULT
#define GS_NEAR_FLOOR_MULT 1.3f     // the adaptive share never again shrinks below this x the share that failed (1.2: 5 % faster on one orbit, redraws on another)
#endif
static thread_local char g_create_err[512] = "";
thread_local char *gs_tl_err = nullptr;

#define CHECK_CTX(ctx) do { if (!(ctx)) return GS_E_BADARG; } while (0)
#define FAIL(code, ...) do { snprintf(GS_ERRBUF(ctx), GS_ERRLEN, __VA_ARGS__); return (code); } while (0)

template <typename T> static int dev_alloc(gs_ctx *ctx, T **p, size_t count)
{
    *p = nullptr;
    if (!count) count = 1;
    hipError_t e = hipMalloc((void **)p, count * sizeof(T));
    if (e != hipSuccess) {
        snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? GS_E_OOM : GS_E_HIP;
    }
    return GS_OK;
}
template <typename T> static void dev_free(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }
#define TRY(x) do { int _rc = (x); if (_rc != GS_OK) return _rc; } while (0)

// ---- transparent re-rendering (GS_OPT_AUTO_RETRY).  An asynchronous frame can come back incomplete: it skipped its second
// binning round and a tile had not saturated, or it outgrew the pair buffers.  The lane's control block says so at gs_sync() --
// per lane, not per frame -- so every lane keeps a log of what it was asked since the last gs_sync(): the sort's arguments and
// the renders that followed (uniforms, output pointers).  gs_sync() then draws the logged frames of a flagged lane (and of the
// lane / twin that shares its stream) again, synchronously, both rounds, into the same buffers.  Written and read on the caller's
// thread only.  Not decided here: gathered frames (the other ranks have to take part), frames sharing an output buffer with
// another logged frame (drawing the older one again would overwrite the newer), data or scene changed meanwhile.
struct GsFrameRec {
    float view[4], cutout[16]; bool has_cutout, has_strip; GsSortStrip strip;
    int nrender; bool undecidable;
    struct { GsFrameUniforms u; void *dev; uint8_t *host; size_t stride; } r[2];
};
struct GsFrameLog { std::vector<GsFrameRec> recs; bool open = false; };
#define GS_LOG_MAX 4096

static GsFrameRec *log_open_rec(gs_ctx *L) { return (L->log && L->log->open && !L->log->recs.empty()) ? &L->log->recs.back() : nullptr; }
static void log_sort(gs_ctx *L, const float view[4], const float *cutout16, const GsSortStrip *strip)
{
    if (!gs_root(L)->auto_retry) return;
    if (!L->log) { L->log = new (std::nothrow) GsFrameLog(); if (!L->log) return; }
    GsFrameLog *g = L->log;
    if (g->open && !g->recs.empty() && g->recs.back().nrender == 0) g->recs.pop_back();   // a sort nothing was drawn from
    if (g->recs.size() >= GS_LOG_MAX) { g->recs.back().undecidable = true; g->open = true; return; }   // (a caller that never syncs)
    GsFrameRec r;
    memcpy(r.view, view, sizeof r.view);
    r.has_cutout = cutout16 != nullptr;
    if (cutout16) memcpy(r.cutout, cutout16, sizeof r.cutout);
    r.has_strip = strip != nullptr;
    if (strip) r.strip = *strip;
    r.nrender = 0; r.undecidable = false;
    try { g->recs.push_back(r); g->open = true; } catch (...) { g->open = false; }
}
static void log_render(gs_ctx *L, const GsFrameUniforms &u, void *dev, uint8_t *host, size_t stride)
{
    GsFrameRec *r = log_open_rec(L);
    if (!r) return;
    if (r->nrender >= 2) { r->undecidable = true; return; }
    r->r[r->nrender].u = u; r->r[r->nrender].dev = dev; r->r[r->nrender].host = host; r->r[r->nrender].stride = stride;
    r->nrender++;
}
static void log_undecidable(gs_ctx *L) { GsFrameRec *r = log_open_rec(L); if (r) r->undecidable = true; }
// after gs_sync: only the open frame stays (renders may still follow its sort), with nothing drawn
static void log_reset(gs_ctx *L)
{
    if (!L->log) return;
    GsFrameLog *g = L->log;
    if (g->open && !g->recs.empty()) { GsFrameRec last = g->recs.back(); last.nrender = 0; last.undecidable = false; g->recs.clear(); g->recs.push_back(last); }
    else g->recs.clear();
}

static int lane_drain(gs_ctx *L, bool flush = false);
static void lane_stop_worker(gs_ctx *L);

static int ensure_radix_tables(gs_ctx *ctx, size_t items)
{
    // a histogram row per radix chunk (sized for the short geometry's chunks) + the digit totals
    const size_t need_hist = (size_t)GS_RADIX_MAX_BINS * (gs_div_up(items, GS_CHUNK_S) + 16);
    if (need_hist > ctx->hist_cap) { dev_free(ctx->hist); ctx->hist_cap = 0; TRY(dev_alloc(ctx, &ctx->hist, need_hist)); ctx->hist_cap = need_hist; }
    // the MSD depth sort's group rows: 256 words per GS_MSD_GROUP chunks (of the short geometry), for at most GS_MSD_MAX_N splats
    const size_t msd_items = items < GS_MSD_MAX_N ? items : (size_t)GS_MSD_MAX_N;
    const size_t need_grp = (size_t)256 * (gs_div_up(gs_div_up(msd_items, GS_CHUNK_S), GS_MSD_GROUP) + 2);
    // (+ k_seg_sort's work items behind them: 4 words of header and 4 per item, at most 256 segments + one block per 4096 records)
    const size_t need_tab = 4 + 4 * ((size_t)512 + 256 + msd_items / 4096 + 16);   // (room for GS_SEG_GRID items at least: every workgroup reads ITS first item unconditionally)
    if (need_grp > ctx->msd_grp_cap) {
        dev_free(ctx->msd_grp); ctx->msd_grp_cap = 0; ctx->msd_tab = nullptr;
        TRY(dev_alloc(ctx, &ctx->msd_grp, need_grp + need_tab));
        ctx->msd_grp_cap = need_grp; ctx->msd_tab = ctx->msd_grp + need_grp;
    }
    const size_t need_aux = (size_t)GS_RADIX_MAX_BINS * 2;
    if (need_aux > ctx->aux_cap) { dev_free(ctx->radix_aux); TRY(dev_alloc(ctx, &ctx->radix_aux, need_aux)); ctx->aux_cap = need_aux; }
    return GS_OK;
}

static int ensure_scan_scratch(gs_ctx *ctx)
{
    TRY(ensure_radix_tables(ctx, ctx->scratch_cap > ctx->pair_cap ? ctx->scratch_cap : ctx->pair_cap));   // the larger of the two sorts
    const size_t need_spine = gs_div_up(ctx->scratch_cap, GS_BLOCK) + 16;   // project/emit chunks of 256 splats
    if (need_spine > ctx->spine_cap) {
        dev_free(ctx->spine); ctx->spine_cap = 0;
        TRY(dev_alloc(ctx, &ctx->spine, need_spine)); ctx->spine_cap = need_spine;
    }
    return GS_OK;
}

int gs_ensure_radix_scratch(gs_ctx *ctx, size_t items) { return ensure_radix_tables(ctx, items); }

int gs_ensure_pair_capacity(gs_ctx *ctx, size_t pairs)
{
    if (pairs <= ctx->pair_cap) return GS_OK;
    size_t cap = ctx->pair_cap ? ctx->pair_cap : (size_t)1 << 22;
    while (cap < pairs) cap += cap / 2 + 1;
    cap = (cap + GS_CHUNK_L - 1) / GS_CHUNK_L * GS_CHUNK_L;
    if (cap > 0xFFFF0000ull) FAIL(GS_E_OOM, "pair list of %zu entries exceeds the 32-bit index space", pairs);
    dev_free(ctx->pair_a); dev_free(ctx->pair_b); dev_free(ctx->emit_extra);
    ctx->pair_cap = 0;
    TRY(dev_alloc(ctx, &ctx->pair_a, cap)); TRY(dev_alloc(ctx, &ctx->pair_b, cap));
    TRY(dev_alloc(ctx, &ctx->emit_extra, cap / GS_EMIT_PAIRS + 2));
    ctx->pair_cap = cap;
    return ensure_scan_scratch(ctx);
}

// every lane idle (the resident arrays are about to change, or the caller wants the device quiet)
static int drain_all(gs_ctx *ctx)
{
    gs_ctx *P = gs_root(ctx);
    int first = GS_OK;
    // what follows may change what the frames in the logs were drawn from (more splats, another scene image, other options):
    // gs_sync() will not draw such frames again by itself
    for (int i = 0; i < GS_MAX_LANES; i++)
        if (P->lanes[i] && P->lanes[i]->log) for (const GsFrameRec &r : P->lanes[i]->log->recs) if (r.nrender) P->log_stale = true;
    for (int i = 0; i < GS_MAX_LANES; i++) {
        gs_ctx *L = P->lanes[i];
        if (!L) continue;
        const int rc = lane_drain(L, true);                        // a worker's failure surfaces at the next gs_sync()
        if (rc != GS_OK && first == GS_OK) { first = rc; if (L != ctx) memcpy(ctx->err, L->err, sizeof ctx->err); }
        if (L->stream) GS_HIP(hipStreamSynchronize(L->stream));
    }
    return first;
}

// per-frame scratch of one lane (sort keys, projected records, pair lists, scan tables) for `cap` splats
static int ensure_lane_scratch(gs_ctx *ctx, size_t cap)
{
    if (cap <= ctx->scratch_cap) return GS_OK;
    GS_HIP(hipStreamSynchronize(ctx->stream));
    dev_free(ctx->depth); dev_free(ctx->key_a); dev_free(ctx->kv_b); dev_free(ctx->val_a);
    dev_free(ctx->proj); dev_free(ctx->rect); dev_free(ctx->tile_count); dev_free(ctx->zwin);
    ctx->scratch_cap = 0; ctx->have_sort = false; ctx->sorted = nullptr;
    TRY(dev_alloc(ctx, &ctx->depth, cap));
    TRY(dev_alloc(ctx, &ctx->key_a, cap)); TRY(dev_alloc(ctx, &ctx->kv_b, cap)); TRY(dev_alloc(ctx, &ctx->val_a, cap));
    TRY(dev_alloc(ctx, &ctx->proj, cap)); TRY(dev_alloc(ctx, &ctx->rect, cap));
    TRY(dev_alloc(ctx, &ctx->tile_count, cap)); TRY(dev_alloc(ctx, &ctx->zwin, cap));
    ctx->scratch_cap = cap;
    TRY(gs_ensure_pair_capacity(ctx, cap * 8 > ((size_t)1 << 22) ? cap * 8 : (size_t)1 << 22));
    return ensure_scan_scratch(ctx);
}

// grow the resident per-splat arrays (owner only) to hold at least `want` splats, preserving the data
static int ensure_capacity(gs_ctx *ctx, size_t want)
{
    if (want <= ctx->cap) return GS_OK;
    if (want > 0x7FFFFFF0ull) FAIL(GS_E_BADARG, "more than 2^31 splats");
    size_t cap = ctx->cap ? ctx->cap : (size_t)1 << 16;
    while (cap < want) cap *= 2;
    TRY(drain_all(ctx));
    float4 *sr = nullptr; uint4 *sp = nullptr; float *br = nullptr;
    TRY(dev_alloc(ctx, &sp, cap * 2)); TRY(dev_alloc(ctx, &sr, cap)); TRY(dev_alloc(ctx, &br, cap));
    if (ctx->n) {
        GS_HIP(hipMemcpyAsync(sp, ctx->splat, ctx->n * 2 * sizeof(uint4), hipMemcpyDeviceToDevice, ctx->stream));
        GS_HIP(hipMemcpyAsync(sr, ctx->sort_rows, ctx->n * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
        GS_HIP(hipMemcpyAsync(br, ctx->bound_r, ctx->n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        GS_HIP(hipStreamSynchronize(ctx->stream));
    }
    dev_free(ctx->splat); dev_free(ctx->sort_rows); dev_free(ctx->bound_r);
    ctx->splat = sp; ctx->sort_rows = sr; ctx->bound_r = br;
    ctx->cap = cap;
    for (int i = 0; i < GS_MAX_LANES; i++) if (ctx->lanes[i]) { ctx->lanes[i]->have_sort = false; ctx->lanes[i]->sorted = nullptr; }
    return ensure_lane_scratch(ctx, cap);                       // lane 0 now; the other lanes when they are next used
}


// ---------------------------------------------------------------- profiling ring (HIP events on the context's stream)

hipEvent_t gs_prof_event(gs_ctx *ctx, int k)
{
    if (!ctx->profile || !ctx->ring) return nullptr;
    if (ctx->profile_blend_only && k != 4 && k != 5) return nullptr;   // GS_OPT_PROFILE = 2 / 3: only the events around the blend
    if (ctx->profile_every > 1 && (ctx->profile_tick % ctx->profile_every) != 0) return nullptr;   // ... of every 4th frame
    const uint32_t slot = ctx->ring_head % GS_PROF_RING;
    ctx->ring_flags[slot] |= (uint8_t)(1u << k);
    return ctx->ring[slot * GS_PROF_EVENTS + k];
}

// fold the timings of every pending slot into the stats; the stream must be idle
static int prof_drain(gs_ctx *ctx)
{
    for (; ctx->ring_pending; ctx->ring_pending--) {
        const uint32_t slot = (ctx->ring_head - ctx->ring_pending) % GS_PROF_RING;
        hipEvent_t *e = ctx->ring + slot * GS_PROF_EVENTS;
        const uint8_t f = ctx->ring_flags[slot];
        ctx->ring_flags[slot] = 0;
        float ms;
        if ((f & 3) == 3) { GS_HIP(hipEventElapsedTime(&ms, e[0], e[1])); ctx->stats.ms_sort = ms; ctx->stats.sum_ms_sort += ms; }
        if ((f & 0x7C) == 0x30) {                            // blend-only profiling
            float d;
            GS_HIP(hipEventElapsedTime(&d, e[4], e[5]));
            ctx->stats.ms_blend = d; ctx->stats.sum_ms_blend += d;
            ctx->stats.prof_frames++;
        }
        if ((f & 0x7C) == 0x7C) {
            float a, b, d, r1;
            GS_HIP(hipEventElapsedTime(&a, e[2], e[3])); GS_HIP(hipEventElapsedTime(&b, e[3], e[4])); GS_HIP(hipEventElapsedTime(&d, e[4], e[5]));
            GS_HIP(hipEventElapsedTime(&r1, e[5], e[6]));
            b += r1;                                       // round 1 (binning against the unsaturated tiles + their blend)
            ctx->stats.ms_project = a; ctx->stats.ms_bin = b; ctx->stats.ms_blend = d; ctx->stats.ms_render = a + b + d;
            ctx->stats.sum_ms_project += a; ctx->stats.sum_ms_bin += b; ctx->stats.sum_ms_blend += d;
            ctx->stats.prof_frames++;
        }
    }
    return GS_OK;
}

// a frame has been enqueued: move to the next slot (draining first if the ring is full)
static int prof_advance(gs_ctx *ctx)
{
    if (!ctx->profile || !ctx->ring) return GS_OK;
    ctx->profile_tick++;
    ctx->ring_head++; ctx->ring_pending++;
    if (ctx->ring_pending >= GS_PROF_RING - 1) {
        GS_HIP(hipStreamSynchronize(ctx->stream));
        return prof_drain(ctx);
    }
    return GS_OK;
}

// after a stream sync: publish the counters of the last completed frame and react to pair-buffer overflow
// the share of splats binned first was too small for a frame drawn with `frac_used`: never again below 1.3 x that, now 1.5 x that
static void share_raise(gs_ctx *ctx /* owner */, float frac_used)
{
    const float fl = frac_used * GS_NEAR_FLOOR_MULT > 1.0f ? 1.0f : frac_used * GS_NEAR_FLOOR_MULT;
    if (fl > ctx->near_floor) ctx->near_floor = fl;
    float nf = frac_used * 1.5f; if (nf < ctx->near_floor) nf = ctx->near_floor; if (nf > 1.0f) nf = 1.0f;
    if (nf > ctx->near_frac) ctx->near_frac = nf;
    ctx->clean_frames = 0; ctx->skip_hold = 32;
}

// the speculative stash of near-only sorts failed: at once out of use for a while if the view does not suit it (kind 2), at the second
// miss within a few collections if the threshold bin keeps outrunning the hint -- one miss is a camera jump, two in a row are a view
// whose threshold is not smooth (the 20 M cloud seen from outside: the bin moves by up to three from one 3-degree pose to the next), and
// every miss costs the redraw of the lanes' logged frames.  (A collected sort that took the path pays one unit of the 24 back.)
static void gs_spec_back_off(gs_ctx *ctx /* owner */, bool unsuited)
{
    ctx->near_spec_miss_credit += 24u;
    if (!unsuited && ctx->near_spec_miss_credit <= 32u) return;
    ctx->near_spec_miss_credit = 0;
    ctx->near_spec_backoff = ctx->near_spec_backoff ? (ctx->near_spec_backoff >= 32768u ? 65536u : ctx->near_spec_backoff * 2u) : 512u;
    __atomic_store_n(&ctx->near_spec_hold, ctx->near_spec_backoff, __ATOMIC_RELAXED);
}

#ifndef GS_NEED_MARGIN_MIN
#define GS_NEED_MARGIN_MIN 1.04f    // what the margin on the measured share shrinks to while no frame misses (it starts at GS_NEED_MARGIN)
#endif
#ifndef GS_NEED_MARGIN
#define GS_NEED_MARGIN 1.15         // the share binned first = what the collected frames' tiles needed x this (the walked share ended 1.17-1.3 x above the share that had failed)
#endif
// The share of splats binned first, from what the blends of the collected frames MEASURED (GsControl::need_near: per tile, how many of
// the nearest splats it read before its pixels were saturated).  Rounds 1-4 walked the share: x 0.9 per collection from 25 % until a
// share failed, x 1.5 then -- a fresh context needed dozens of collections (and one failure) to arrive, and a 20 M-splat scene drew
// its first hundred frames with 5 M positions in the first round (`cold_orbit`: 63 frames/s).  One collection is enough now.
static void share_from_need(gs_ctx *ctx /* owner */, uint32_t need, uint32_t frames)
{
    if (need) ctx->stats.need_splats = need;
    if (!need || ctx->near_fixed_permille > 0 || !ctx->n) return;
    // what the share follows: the largest need of the recent past -- sixteen buckets of at least sixteen frames each (a collection of
    // queued frames is a bucket; synchronous frames, a collection each, share one) -- so that a pose which needs less does not take
    // away what another pose of the same orbit needs: bench.py's orbit is 120 poses whose needs lie between 80 and 125 permille, and a
    // caller may queue hundreds of frames, all drawn with ONE share, before it collects.  Up at once, down when the window has moved on.
    const int HN = (int)(sizeof ctx->need_hist / sizeof ctx->need_hist[0]);
    if (ctx->need_hist_frames[ctx->need_hist_pos] >= 16u) {
        ctx->need_hist_pos = (ctx->need_hist_pos + 1) % HN;
        ctx->need_hist[ctx->need_hist_pos] = 0; ctx->need_hist_frames[ctx->need_hist_pos] = 0;
    }
    if (need > ctx->need_hist[ctx->need_hist_pos]) ctx->need_hist[ctx->need_hist_pos] = need;
    // (frames gs_sync is drawing AGAIN -- adapt_frozen -- measure, but are no new evidence of a camera at rest: they neither age the window
    // nor shrink the margin that their own miss has just raised)
    if (!ctx->adapt_frozen) ctx->need_hist_frames[ctx->need_hist_pos] += frames ? frames : 1u;
    uint32_t m = 0;
    for (int k = 0; k < HN; k++) if (ctx->need_hist[k] > m) m = ctx->need_hist[k];
    float target = 1.0f;                                         // 0xFFFFFFFF: a tile that no share saturates (sky): one round over everything
    if (m != 0xFFFFFFFFu) {
        // the margin on top follows what the camera does: 1.15 to begin with, a hundredth less with every collection that nothing missed
        // -- with every eight frames of a collection of queued frames: a caller that collects every 24 frames was at 1.10 after a whole
        // lap of 120 new poses without a miss, i.e. had binned 6 % more than it came to need for all of it --
        // (down to 1.04: a still or periodic camera needs none, and 10 % of share are 10 % of the binning and the blend's staging), a tenth
        // more after a miss
        if (ctx->need_margin < GS_NEED_MARGIN_MIN) ctx->need_margin = (float)GS_NEED_MARGIN;
        const float less = ctx->adapt_frozen ? 0.0f : 0.01f * (float)(frames > 8u ? frames / 8u : 1u);
        ctx->need_margin = ctx->need_margin - less < GS_NEED_MARGIN_MIN ? GS_NEED_MARGIN_MIN : ctx->need_margin - less;
        target = (float)((double)m * (double)ctx->need_margin / (double)ctx->n);
        if (target > 0.85f) target = 1.0f;                       // (two rounds over nearly everything cost more than one)
    }
    if (target >= 1.0f && ctx->last_kept) {
        // One round over everything -- but "everything" is the V splats the sort KEEPS, and a cut-out or a strip may keep a few per cent
        // of the N resident ones (the cut-out demo: 226 K of 6.3 M).  A share of 100 % sizes the round for N positions: grids, and row
        // tables of N / 256 chunks per tile row that the row scan reads in full -- project 34 us instead of 24, binning 73 instead of 61.
        // A share that covers V with a quarter to spare draws the same single round (nothing lies beyond it: no tile is flagged) on
        // tables of that size; should V grow past it between two collections, the tiles that want more flag their frames like any
        // other share that was too small.
        const double cover = ((double)ctx->last_kept * 1.25 + 4096.0) / (double)ctx->n;
        if (cover < 0.85) target = (float)cover;
    }
    if (target < ctx->near_floor) target = ctx->near_floor;      // (a walked share's floor, where there is one)
    // (never fewer than 4096 positions: until round 6 never less than one permille -- 21 K positions of a 20 M scene, eight times what a
    // frame sorted for its frustum needs there)
    { const float fl = 4096.0f / (float)ctx->n; if (target < fl) target = fl > 1.0f ? 1.0f : fl; }
    ctx->near_frac = target;
    ctx->share_measured = true;
}

// a MISS under a measured share (round 1 skipped, a tile unsaturated; the frames are drawn again): the frames that missed needed more
// than `frac_used`, how much more the frames drawn again will measure.  No floor -- the walk's "never again below 1.3 x the share that
// failed" ratchets (1.3 x 92, then 1.3 x 120 = 156 permille over poses that need 80-115) where the measurement simply follows.
static void share_missed(gs_ctx *ctx /* owner */, float frac_used)
{
    float nf = frac_used * 1.3f; if (nf > 1.0f) nf = 1.0f;
    if (nf > ctx->near_frac) ctx->near_frac = nf;
    if (ctx->need_margin < GS_NEED_MARGIN_MIN) ctx->need_margin = (float)GS_NEED_MARGIN;
    ctx->need_margin = ctx->need_margin + 0.1f > 1.3f ? 1.3f : ctx->need_margin + 0.1f;
    const uint32_t as_need = (uint32_t)((double)nf * (double)ctx->n / (double)ctx->need_margin);   // (the window remembers it like a measurement)
    if (as_need > ctx->need_hist[ctx->need_hist_pos]) ctx->need_hist[ctx->need_hist_pos] = as_need;
    ctx->clean_frames = 0; ctx->skip_hold = 32;
}

// Every lane's words are brought near what the collected frames measured: a lane whose words are far below it (a lane that has drawn
// nothing yet -- its block is zero --, or nothing of this view) would have EVERY tile of its next frame issue its atomic: 8160 on one
// line, ~90 us -- the first frame of five lanes after a run of synchronous frames, i.e. the fill of bench.py's twenty-frame region.
static int seed_need_words(gs_ctx *ctx /* owner */, uint32_t need, bool idle_only = false)
{
    if (!need) return GS_OK;
    const uint32_t seed = need == 0xFFFFFFFFu ? need : (uint32_t)((uint64_t)need * 9u / 10u);
    for (int i = 0; i < GS_MAX_LANES; i++) {
        gs_ctx *L = ctx->lanes[i];
        if (!L || !L->ctl) continue;
        // (from a synchronous frame: a lane with queued, uncollected frames keeps its words -- the seed its next projection would store is the
        // host's ESTIMATE, and the larger maxima the frames in flight recorded would be lost before gs_sync reads them: ADVICE r5)
        if (idle_only && L->async_pending) continue;
        const uint32_t have = L->need_word_est;
        const bool stale = need == 0xFFFFFFFFu ? have != 0xFFFFFFFFu : (have != 0xFFFFFFFFu && (uint64_t)have * 10u < (uint64_t)need * 8u);
        if (!stale) continue;
        L->need_seed_pending = seed ? seed : 1u;
        L->need_word_est = seed;
    }
    return GS_OK;
}

static int collect_status(gs_ctx *lane, bool *overflowed, bool *share_failed = nullptr, uint32_t *spec_failed = nullptr, uint32_t *need_out = nullptr,
                          uint32_t *frames_out = nullptr)
{
    gs_ctx *ctx = gs_root(lane);                                // the adaptive share is one state for all lanes ...
    const GsControl *c = lane->ctl_host;                        // ... fed by each lane's own counters
    lane->stats.n_sorted = c->n_kept; lane->stats.n_visible = c->n_visible; lane->stats.n_pairs = c->n_pairs_frame;
    lane->stats.acc_frames = c->acc_frames; lane->stats.acc_sorted = c->acc_sorted; lane->stats.acc_visible = c->acc_visible;
    lane->stats.acc_pairs = c->acc_pairs; lane->stats.sort_records = c->n_sorted;
    lane->stats.sort_mode = c->near_sorted;
    ctx->last_kept = c->n_kept;
    if (c->n_pairs_frame) { ctx->last_pairs = c->n_pairs_frame; ctx->last_visible = c->n_visible; }
    if (c->n_pairs_frame) __atomic_store_n(&ctx->run_hint, c->n_runs, __ATOMIC_RELAXED);
    if (c->n_pairs_frame) {                                     // sizing hint for the next frames' pair sort (any lane's worker may read it)
        const uint64_t h = (uint64_t)c->n_pairs_frame + c->n_pairs_frame / 4 + GS_CHUNK_L;
        __atomic_store_n(&ctx->pair_hint, h > 0xFFFFFFFFull ? 0u : (uint32_t)h, __ATOMIC_RELAXED);
    }
    // Adapt the share of splats binned in round 0.  An "event" = a frame whose round 0 left tiles unsaturated (round 1
    // re-binned for them): the share grows x1.5 and will never again shrink below 1.3 x the share that failed; without
    // events it shrinks 10 % per collected frame until the first event, 2 % afterwards.  After 16 clean frames round 1 is not even launched (11 empty kernels
    // cost ~50 us): blend<0> raises round1_missed if that was wrong, and the frame is completed / re-rendered.
    uint32_t need = 0;
    for (uint32_t k = 0; k < GS_NEED_WORDS; k++) if (c->need_near[k] > need) need = c->need_near[k];
    if (need) {
        // The words are not cleared but SEEDED with 0.9 x the maximum: a tile issues its atomic only when it needs more than the word it
        // read when it started, so after a clear every tile of the first frames would -- 8160 atomics on one cache line serialise at
        // ~11 ns each: the first six frames after every gs_sync took twice as long.  Seeded, only the tiles within 10 % of the maximum
        // speak up; a need that falls is followed 10 % per collection (the word is then an upper bound), one that rises at once.
        // 0xFFFFFFFF (a tile nothing saturates) is kept, and dropped once in sixteen seedings: the scene may have changed.
        // (... every fourth collection of the lane: the host follows the maximum of the last eight collections anyway, and a queue entry
        // per lane and gs_sync is 60 us of a twenty-frame region)
        lane->need_word_est = need;                                // (what the lane's words hold now: their maximum)
        if ((++lane->need_probe & 3u) == 0u) {
            const uint32_t seed = need == 0xFFFFFFFFu ? ((lane->need_probe & 63u) ? 0xFFFFFFFFu : 0u) : (uint32_t)((uint64_t)need * 9u / 10u);
            lane->need_seed_pending = seed ? seed : 1u;            // (written by the lane's next frame itself: GsFrameUniforms::need_seed)
            lane->need_word_est = seed;
        }
    }
    // (frames drawn from a truncated order measure nothing.  Frames gs_sync draws AGAIN do -- both rounds, every tile's need recorded, the
    // sky's 0xFFFFFFFF included -- and what they measure is what the frames that missed had needed: the share is right after ONE miss.
    // Ignored like the rest of their counters, a camera that left the cloud took eight collections x 24 redrawn frames to get to 100 %)
    if (c->order_incomplete) need = 0;
    if (need_out && need > *need_out) *need_out = need;
    if (ctx->adapt_frozen || c->order_incomplete) {
        // gs_sync is drawing flagged frames again (redraw_flagged_frames) with the uniforms they were queued with: their share has
        // been dealt with once already -- every one of them would count as a new failure and multiply the share by 1.5.
        // order_incomplete: the frames were drawn from an order that lacked splats (a stash overflowed, the speculative stash could
        // not be vouched for): tiles that did not saturate say nothing about the SHARE -- the frames are drawn again from a whole
        // sort either way.  (Counting them raised the share 23 -> 40 permille outside the cloud at 20 M, past the 1/32 the chunk
        // stashes need: 3 718 -> 1 898 frames/s for good, after one overflow.)
        lane->seen_unsat_events = c->unsat_events; lane->seen_acc_frames = c->acc_frames;
    } else if (ctx->near_fixed_permille <= 0 && lane->stats.n_tiles) {
        const uint32_t events = c->unsat_events - lane->seen_unsat_events;
        const uint64_t frames = c->acc_frames >= lane->seen_acc_frames ? c->acc_frames - lane->seen_acc_frames : 1;
        lane->seen_unsat_events = c->unsat_events; lane->seen_acc_frames = c->acc_frames;
        if (frames_out) *frames_out += (uint32_t)(frames ? frames : 1);
        if (lane->last_two_rounds) {
            // An "event" -- round 0 left tiles unsaturated and round 1 finished them -- is a failure of a WALKED share; with the blend's
            // measurement it is not: round 1 recorded what those tiles needed, and the share is set from that (share_from_need).  Counted as a
            // failure it left a floor of 1.3 x a share that the measurement already covers (bench.py's region: need 111 K splats, the walk of
            // the pre-roll's two-round frames ended at 1.3 x 120 = 156 permille where 1.15 x 106 = 122 do).  A MISS -- round 1 skipped and a
            // tile unsaturated: the frame is drawn again -- is a failure either way.
            // (need == 0xFFFFFFFF -- a tile no share saturates: sky -- is a measurement too: share_from_need goes to one round, or to the share that
            // covers V, at once; counted as a failure it walked there x 1.3 per collection and came back with the margin at 1.3 and a hold: ADVICE r5)
            if ((events && !need) || c->round1_missed) {
                // the share proved too small: raised ONCE per collection by the caller (share_failed) -- the frames of all the lanes a
                // gs_sync collects were drawn with the same share, and six lanes reporting the same failure used to multiply it by
                // 1.5^6 and to leave the floor at 1.3 x 1.5^5 of the share that had really failed (a cold context ended up at 70 %)
                if (share_failed) *share_failed = true;
            } else {
                ctx->clean_frames += (uint32_t)(frames ? frames : 1);
                if (!need) {
                    // no measurement (counting renders, the pixel-split blend: no need word recorded): the walk of rounds 1-4 -- fast descent (x0.9) until a share
                    // has proved too small once, then a slow drift (x0.98) above the floor
                    float nf = ctx->near_frac * (ctx->near_floor > 0.0f ? 0.98f : 0.9f);
                    if (nf < ctx->near_floor) nf = ctx->near_floor; if (nf < 0.001f) nf = 0.001f;
                    if (nf < ctx->near_frac) ctx->near_frac = nf;
                }
                const uint32_t fr = (uint32_t)(frames ? frames : 1);     // the hold is counted in frames, not in collections
                ctx->skip_hold = ctx->skip_hold > fr ? ctx->skip_hold - fr : 0;
            }
            ctx->single_round_frames = 0;
        } else if (!need && ctx->near_frac >= 1.0f && ++ctx->single_round_frames >= 64) {
            ctx->near_frac = 0.5f; ctx->near_floor = 0.0f; ctx->single_round_frames = 0; ctx->clean_frames = 0;   // (unmeasured: re-probe occlusion now and then)
        }
    }
    if (c->near_overflow) {
        // a chunk of a near-only sort had more survivors than its stash holds (the frame was flagged and is drawn again from a
        // whole sort): this context keeps to the whole-length passes from now on
        __atomic_store_n(&ctx->near_stash_off, true, __ATOMIC_RELAXED);   // (the lanes' enqueue threads read these flags while they launch sorts)
        GS_HIP(hipMemsetAsync(&lane->ctl->near_overflow, 0, sizeof(uint32_t), lane->stream));
    }
    if (c->near_sorted == 1u || c->near_sorted == 2u) {          // (3: a tail sort -- no threshold, no hint)
        __atomic_store_n(&ctx->near_spec, true, __ATOMIC_RELAXED);    // a near-only sort has run: the threshold-bin hint exists (gs_near_spec_ok)
        const uint32_t h = __atomic_load_n(&ctx->near_spec_hold, __ATOMIC_RELAXED);
        if (h) __atomic_store_n(&ctx->near_spec_hold, h - 1u, __ATOMIC_RELAXED);
    }
    if (c->near_sorted == 2u) { ctx->stats.spec_sorts++; if (ctx->near_spec_miss_credit) ctx->near_spec_miss_credit--; }
    if (c->spec_fail) {
        // a near-only sort could not vouch for the candidates its depth pass had stashed (the frame was flagged and is drawn again
        // from a whole sort).  1: the hint was behind -- it is exact now; 2: a stash overflowed / the depth range does not suit the path
        ctx->stats.spec_misses++;
        // (the frames a gs_sync collects were sorted by ONE hint: the lanes that missed it together -- a camera jump -- are one event)
        if (spec_failed) { if (c->spec_fail > *spec_failed) *spec_failed = c->spec_fail; }
        else gs_spec_back_off(ctx, c->spec_fail == 2u);
        GS_HIP(hipMemsetAsync(&lane->ctl->spec_fail, 0, sizeof(uint32_t), lane->stream));
    }
    lane->stats.unsat_tiles = lane->last_two_rounds ? c->unsat_round0 : 0;
    lane->stats.near_permille = (uint32_t)(ctx->near_frac * 1000.0f + 0.5f);
    *overflowed = c->overflow_sticky != 0;
    if (*overflowed) {
        const size_t need = (size_t)c->max_total + c->max_total / 4 + 1;
        GS_HIP(hipMemsetAsync(&lane->ctl->overflow_sticky, 0, 2 * sizeof(uint32_t), lane->stream));
        GS_HIP(hipStreamSynchronize(lane->stream));
        if (gs_ensure_pair_capacity(lane, need) != GS_OK) { if (lane != ctx) memcpy(ctx->err, lane->err, sizeof ctx->err); return GS_E_OOM; }
    }
    return GS_OK;
}

// ---------------------------------------------------------------- enqueue threads
// Enqueuing one frame costs the host ~100 us (18 kernel launches), about what the GPU needs for a frame once three of
// them overlap -- so each lane gets a worker thread that does the launching: gs_sort() / gs_render_device(ASYNC) only
// hand it the frame's uniforms, and the launches of the frames in flight proceed in parallel on the lanes' own streams.
// Everything a worker touches is lane-local; the caller's thread touches a lane only after lane_drain().

struct GsLaneCmd {
    gs_ctx *target;                                            // the lane (or twin) the command is for
    int type;                                                  // 0 = sort, 1 = asynchronous render, 2 = call (gs_comm.hip: the frame's gather)
    float view[4], cutout[16]; bool has_cutout;
    bool has_strip; GsSortStrip strip; uint32_t near_req;      // (sort: how much of the order the frame is expected to need; 0 = all)
    GsFrameUniforms u; void *device_rgba; uint8_t *host_rgba; size_t stride;
    std::function<int(gs_ctx *)> call;
};

struct GsLaneWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv_work, cv_idle;
    std::deque<GsLaneCmd> q;
    bool busy = false, stop = false;
    bool flush = false;                                        // a drain is waiting: do not hold a frame back for a partner
    uint64_t n_pairs = 0, n_single = 0;                        // frames sent out in pairs / alone (GS_DEBUG_PAIRS=1 prints them at shutdown)
    int rc = GS_OK;                                            // first failure since the last drain ...
    char err[GS_ERRLEN] = "";                                  // ... and its message: the worker never writes the lane's err itself
};

static int render_async_on_lane(gs_ctx *ctx, const GsFrameUniforms &u, void *device_rgba, uint8_t *host_rgba, size_t stride);

static int ensure_frame_buffers(gs_ctx *ctx, const GsFrameUniforms &u, bool need_fb);
static int prof_advance(gs_ctx *ctx);

// Near-only sorts (GS_OPT_SORT_NEAR).  A frame whose single binning round covers the nearest near_count splats (round 1 skipped
// optimistically, gs_fill_uniforms) reads nothing else of the order, so its sort lets only the splats go on that can be among
// them (gs_sort.hip: an exact rule on the sort key; the positions it fills equal the whole order's).  The request is made when
// the sort is issued, from the state the render's uniforms will be made from; a render that turns out to need more of the
// order (another share, a counting render, round 1 after all, gs_download of the order) sorts again in full first.
// May a frame leave its second binning round out (and its sort be near-only)?  After 16 clean frames -- but not while the share is
// still on its fast way DOWN from the default (no floor yet): that descent ends with a share that is too small, and with round 1
// still launched the frame that finds it is completed by round 1 (an "event") instead of being flagged and drawn again with
// everything queued behind it.  (A scene that never fails ends at the minimum share: skippable from there.)
static inline bool round1_skippable(const gs_ctx *ctx /* owner */)
{
    // (a MEASURED share -- share_from_need -- needs neither the failure that gives the walked share its floor nor sixteen frames of proof)
    return ctx->near_fixed_permille <= 0 && ctx->clean_frames >= (ctx->share_measured ? 4u : 16u) && !ctx->skip_hold &&
           (ctx->near_floor > 0.0f || ctx->near_frac <= 0.0011f || ctx->share_measured);
}

static uint32_t sort_near_request(const gs_ctx *ctx /* owner */)
{
    if (!ctx->sort_near_opt || !ctx->renderable) return 0;
    // (short sorts are bound by their launches and dependent round trips, not by their records: measured again in round 5 with the
    // four-launch MSD sort -- at 1 M splats the depth histogram and the threshold search cost the depth and bucket kernels 10 us, the
    // two kernels behind them save 2 with a fifth of the records: 7 170 against 7 590 frames/s one frame at a time, the same pipelined)
    // ... so up to GS_MSD_MAX_N splats a near-only sort is a TAIL sort (gs_sort.hip): the four launches of the whole sort, of which the
    // last two leave out the segments before the one the frame's first position lies in -- nothing is added in front
    const bool tail = gs_msd_enabled() && ctx->n <= (size_t)GS_MSD_MAX_N && !ctx->wide_pairs;
    if (ctx->sort_near_opt == 1 && ctx->n < ((size_t)1 << 22) && !tail) return 0;
    if (ctx->wide_pairs || ctx->n > ((size_t)1 << 25)) return 0;
    if (!round1_skippable(ctx) || ctx->near_frac >= 1.0f) return 0;
    const double nc = ceil((double)ctx->near_frac * (double)ctx->n);
    // (a cutout or a strip that keeps little more than the frame reads anyway: the histogram and the threshold would buy nothing)
    if (ctx->sort_near_opt == 1 && ctx->last_kept && nc * 2.0 > (double)ctx->last_kept) return 0;
    return nc < 1 ? 1u : (uint32_t)nc;
}
static inline bool sort_covers(uint32_t near_req, const GsFrameUniforms &u)
{
    return near_req == 0 || (u.near_count != 0xFFFFFFFFu && u.skip_round1 && u.near_count <= near_req);
}
static int ensure_full_sort(gs_ctx *L)
{
    if (!L->have_sort || !L->sort_near_req) return GS_OK;
    return gs_run_sort(L, L->sv_view, L->sv_has_cutout ? L->sv_cutout : nullptr, L->sv_has_strip ? &L->sv_strip : nullptr, 0);
}
static int ensure_sort_covers(gs_ctx *L, const GsFrameUniforms &u) { return sort_covers(L->sort_near_req, u) ? GS_OK : ensure_full_sort(L); }

// GS_OPT_FRAME_BATCH: the frames of a lane and of its twin, when both are waiting, go out as ONE chain of launches
static int run_frame_pair(gs_ctx *A, gs_ctx *B, const GsLaneCmd &s0, const GsLaneCmd &r0, const GsLaneCmd &s1, const GsLaneCmd &r1)
{
    gs_ctx *ctx = A;
    gs_ctx *S[2] = { A, B };
    const float *view[2] = { s0.view, s1.view };
    const float *cut[2] = { s0.has_cutout ? s0.cutout : nullptr, s1.has_cutout ? s1.cutout : nullptr };
    const GsSortStrip *strip[2] = { s0.has_strip ? &s0.strip : nullptr, s1.has_strip ? &s1.strip : nullptr };
    const GsFrameUniforms U[2] = { r0.u, r1.u };
    const uint32_t near[2] = { sort_covers(s0.near_req, U[0]) ? s0.near_req : 0u, sort_covers(s1.near_req, U[1]) ? s1.near_req : 0u };
    TRY(gs_run_sort2(S, view, cut, strip, near));
    uint8_t *dev[2] = { (uint8_t *)r0.device_rgba, (uint8_t *)r1.device_rgba };
    TRY(ensure_frame_buffers(A, U[0], dev[0] == nullptr));
    TRY(ensure_frame_buffers(B, U[1], dev[1] == nullptr));
    TRY(gs_run_render2(S, U, dev));
    const GsLaneCmd *r[2] = { &r0, &r1 };
    for (int k = 0; k < 2; k++) {
        if (!r[k]->host_rgba) continue;
        const size_t sw = (size_t)(U[k].x1 - U[k].x0);
        const uint8_t *src = dev[k] ? dev[k] : S[k]->fb;
        GS_HIP(hipMemcpy2DAsync(r[k]->host_rgba, r[k]->stride ? r[k]->stride : sw * 4, src, sw * 4, sw * 4, (size_t)U[k].H, hipMemcpyDeviceToHost, A->stream));
    }
    return prof_advance(A);                                     // (the pair's HIP events sit in the primary lane's ring)
}

// What follows the sort `s0` that was just popped (a frame on the primary lane L) in the queue?
//   GS_Q_PAIR    its render [+ its gather call] and the twin's sort, render [+ gather call], of a kind that may share launches
//                (n_take commands to pop; calls = 1: gathered frames of one piece each, the gathers issued after the shared
//                kernels in frame order);
//   GS_Q_STEREO  the renders of TWO views of this frame (both XR eyes drawn by this context), which may share launches;
//   GS_Q_SINGLE  enough to know that neither will form;  GS_Q_WAIT  not enough yet.
enum { GS_Q_WAIT, GS_Q_PAIR, GS_Q_STEREO, GS_Q_SINGLE };
static int classify_queue(const gs_ctx *L, const GsLaneCmd &s0, const std::deque<GsLaneCmd> &q, int *n_take, int *calls)
{
    const gs_ctx *T = L->twin;
    if (q.empty()) return GS_Q_WAIT;
    if (q[0].type != 1 || q[0].target != L) return GS_Q_SINGLE;
    if (q.size() < 2) return GS_Q_WAIT;
    if (q[1].type == 1) {                                          // a second view of the same frame
        const bool ok = q[1].target == L && !s0.has_strip && !q[0].host_rgba && !q[1].host_rgba && gs_frames_batchable(q[0].u, q[1].u) && L->n == T->n;
        return ok ? GS_Q_STEREO : GS_Q_SINGLE;
    }
    const int c = (q[1].type == 2 && q[1].target == L) ? 1 : 0;
    const size_t need = 3 + 2 * (size_t)c;
    if (q.size() < (size_t)(2 + c)) return GS_Q_WAIT;
    const GsLaneCmd &s1 = q[1 + c];
    if (s1.type != 0 || s1.target != T || s1.has_strip != s0.has_strip) return GS_Q_SINGLE;
    if (q.size() < (size_t)(3 + c)) return GS_Q_WAIT;
    const GsLaneCmd &r1 = q[2 + c];
    if (r1.type != 1 || r1.target != T || !gs_frames_batchable(q[0].u, r1.u) || L->n != T->n) return GS_Q_SINGLE;
    if (c) {
        if (q.size() < need) return GS_Q_WAIT;
        if (q[3 + c].type != 2 || q[3 + c].target != T) return GS_Q_SINGLE;
    }
    *n_take = (int)need; *calls = c;
    return GS_Q_PAIR;
}

// one sort, then the two views in ONE chain of launches: the second view on the twin's scratch, from the lane's order
static int run_two_views(gs_ctx *A, gs_ctx *B, const GsLaneCmd &s0, const GsLaneCmd &r0, const GsLaneCmd &r1)
{
    gs_ctx *ctx = A;
    const uint32_t near = (sort_covers(s0.near_req, r0.u) && sort_covers(s0.near_req, r1.u)) ? s0.near_req : 0u;
    TRY(gs_run_sort(A, s0.view, s0.has_cutout ? s0.cutout : nullptr, nullptr, near));
    // the twin's kernels read the order and its length through THEIR lane: point it at the lane's (counts: the head of the control block)
    GS_HIP(hipMemcpyAsync(B->ctl, A->ctl, offsetof(GsControl, n_visible), hipMemcpyDeviceToDevice, A->stream));
    B->sorted = A->sorted; B->have_sort = true; B->sort_near_req = 0;   // (nothing of its own to sort again: the pair was checked above)
    gs_ctx *S[2] = { A, B };
    const GsFrameUniforms U[2] = { r0.u, r1.u };
    uint8_t *dev[2] = { (uint8_t *)r0.device_rgba, (uint8_t *)r1.device_rgba };
    TRY(ensure_frame_buffers(A, U[0], dev[0] == nullptr));
    TRY(ensure_frame_buffers(B, U[1], dev[1] == nullptr));
    TRY(gs_run_render2(S, U, dev));
    return prof_advance(A);
}

static void lane_worker_main(gs_ctx *L)
{
    GsLaneWorker *w = L->worker;
    (void)hipSetDevice(L->device);
    char scratch[GS_ERRLEN] = "";
    gs_tl_err = scratch;                                       // GS_HIP / FAIL on this thread write here
    std::unique_lock<std::mutex> lk(w->m);
    for (;;) {
        w->cv_work.wait(lk, [&] { return w->stop || !w->q.empty(); });
        if (w->q.empty()) break;                               // stop requested and nothing left to do
        GsLaneCmd c = w->q.front(); w->q.pop_front();
        w->busy = true;
        int rc = GS_OK;
        gs_ctx *T = c.target;
        bool paired = false, stereo = false;
        int n_take = 0, calls = 0;
        GsLaneCmd pc[5];                                           // the rest of a pair: render 0 [call 0] sort 1 render 1 [call 1]
        if (c.type == 0 && T == L && L->twin && gs_root(L)->frame_batch == 2 && w->rc == GS_OK) {
            // the sort of a frame on the primary lane: if its render and the twin's frame are queued behind it (the caller is
            // normally several frames ahead of this thread; give it a moment if not), the two frames share their launches
            int kind = GS_Q_WAIT;
            w->cv_work.wait_for(lk, std::chrono::microseconds(200), [&] { kind = classify_queue(L, c, w->q, &n_take, &calls); return w->stop || w->flush || kind != GS_Q_WAIT; });
            if (kind == GS_Q_PAIR) {
                for (int k = 0; k < n_take; k++) { pc[k] = w->q.front(); w->q.pop_front(); }
                paired = true;
            } else if (kind == GS_Q_STEREO) {
                pc[0] = w->q.front(); w->q.pop_front(); pc[1] = w->q.front(); w->q.pop_front();
                stereo = true;
                L->twin->inflight++;                               // the second view runs on the twin's scratch: a drain of the twin waits for it
            }
        }
        // (w->rc is the caller's to reset under the mutex -- lane_drain of the lane OR of its twin, which shares this thread -- so the
        // launches below decide on a copy taken while the mutex is still held: ThreadSanitizer, round 4)
        const int prior_rc = w->rc;
        lk.unlock();
        static const bool dbg_slow = getenv("GS_DEBUG_WORKER") != nullptr;   // (diagnostic: commands that kept this thread longer than 60 us)
        const auto dbg_t0 = std::chrono::steady_clock::now();
        // (a call is run even after a failure: the gather of a frame must be issued on every rank, or the others wait for it)
        if (stereo) rc = run_two_views(L, L->twin, c, pc[0], pc[1]);
        else if (paired) {
            rc = run_frame_pair(L, L->twin, c, pc[0], pc[1 + calls], pc[2 + calls]);
            if (calls) {                                           // the two gathers, in frame order, behind the shared kernels
                const int g0 = pc[1].call(L), g1 = pc[4].call(L->twin);
                if (rc == GS_OK) rc = g0 != GS_OK ? g0 : g1;
            }
        }
        else if (c.type == 2) { const int r2 = c.call(T); if (prior_rc == GS_OK) rc = r2; }
        else if (prior_rc == GS_OK) rc = c.type == 0 ? gs_run_sort(T, c.view, c.has_cutout ? c.cutout : nullptr, c.has_strip ? &c.strip : nullptr, c.near_req)
                                                  : render_async_on_lane(T, c.u, c.device_rgba, c.host_rgba, c.stride);
        if (dbg_slow) {
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - dbg_t0).count();
            if (us > (getenv("GS_DEBUG_WORKER_US") ? atof(getenv("GS_DEBUG_WORKER_US")) : 60.0)) fprintf(stderr, "[gs] worker %p: %s for lane %p took %.0f us\n", (void *)L, stereo ? "two views" : paired ? "a pair" : c.type == 0 ? "a sort" : c.type == 1 ? "a render" : "a call", (void *)T, us);
        }
        lk.lock();
        if (rc != GS_OK && w->rc == GS_OK) { w->rc = rc; memcpy(w->err, scratch, sizeof w->err); }
        w->busy = false;
        if (w->q.empty()) w->flush = false;
        if (stereo) { L->inflight -= 3; L->twin->inflight--; w->n_pairs += 2; L->twin->async_pending = true; }   // (the second view's counters sit in the twin's control block)
        else if (paired) { L->inflight -= 2 + calls; L->twin->inflight -= 2 + calls; w->n_pairs += 2; } else { T->inflight -= 1; if (c.type == 1) w->n_single++; }
        w->cv_idle.notify_all();
    }
}

// wait until the lane's worker has enqueued everything it was given; returns (and clears) its first failure
static int lane_drain(gs_ctx *L, bool flush)
{
    GsLaneWorker *w = L->exec ? L->exec->worker : L->worker;
    if (!w) return GS_OK;
    std::unique_lock<std::mutex> lk(w->m);
    if (flush && L->inflight) { w->flush = true; w->cv_work.notify_all(); }   // (gs_sync / drain_all: a frame waiting for its partner goes out now)
    // everything handed over FOR THIS LANE has been executed (its twin's commands may still be queued: the two only share the thread)
    w->cv_idle.wait(lk, [&] { return L->inflight == 0; });
    const int rc = w->rc; w->rc = GS_OK;
    if (rc != GS_OK) memcpy(L->err, w->err, sizeof L->err);    // on the caller's thread, under the worker's mutex
    return rc;
}

static int ensure_worker(gs_ctx *L)
{
    gs_ctx *E = L->exec ? L->exec : L;
    if (!E->worker) {
        E->worker = new (std::nothrow) GsLaneWorker();
        if (!E->worker) return GS_E_OOM;
        try { E->worker->th = std::thread(lane_worker_main, E); }
        catch (...) {                                              // no exception crosses the C ABI
            delete E->worker; E->worker = nullptr;
            snprintf(L->err, sizeof L->err, "could not start the lane's enqueue thread");
            return GS_E_OOM;
        }
    }
    return GS_OK;
}

static int lane_push(gs_ctx *L, GsLaneCmd c)
{
    gs_ctx *E = L->exec ? L->exec : L;
    if (ensure_worker(L) != GS_OK) return GS_E_OOM;
    GsLaneWorker *w = E->worker;
    c.target = L;
    { std::lock_guard<std::mutex> lk(w->m); w->q.push_back(c); L->inflight++; }
    w->cv_work.notify_all();
    return GS_OK;
}

static void lane_stop_worker(gs_ctx *L)
{
    GsLaneWorker *w = L->worker;
    if (!w) return;
    { std::lock_guard<std::mutex> lk(w->m); w->stop = true; }
    w->cv_work.notify_one();
    if (w->th.joinable()) w->th.join();
    if (getenv("GS_DEBUG_PAIRS")) fprintf(stderr, "[gs] lane %p: %llu frames in pairs, %llu alone\n", (void *)L, (unsigned long long)w->n_pairs, (unsigned long long)w->n_single);
    delete w;
    L->worker = nullptr;
}

// ---------------------------------------------------------------- lanes (frame pipelining)

#define LANE_HIP(L, call) do { hipError_t _e = (call); if (_e != hipSuccess) {                                             \
        snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__);    \
        return (_e == hipErrorOutOfMemory) ? GS_E_OOM : GS_E_HIP; } } while (0)

// The first dispatch of a hardware queue that asks for PRIVATE (scratch) memory -- or for more of it per lane than the queue has had
// so far -- stops at the command processor until the runtime has allocated it: ~140 us, once per queue.  A few kernels here ask
// for some (k_project<0, true>: 36 bytes of spill slots the compiler keeps although every SGPR spill went to VGPR lanes; the paired
// k_seg_sort: 12), and which of the lanes' queues has met one of them before a caller starts timing is a matter of which frames
// went out alone and which in pairs: round 5's twenty-frame region lost 80 us to one such stop on a queue the pre-roll had only
// fed pairs (tools/prof_api.py: the launch call returned, the kernel started 140 us later).  So every lane's stream asks for more
// than any kernel will, once, when the lane is made.
__global__ void k_touch_private(uint32_t *out, uint32_t n)
{
    volatile uint32_t a[GS_TOUCH_PRIVATE_WORDS];                  // (volatile + a run-time index: stays in private memory)
    for (uint32_t i = 0; i < GS_TOUCH_PRIVATE_WORDS; i++) a[i] = i * n;
    out[0] = a[n % GS_TOUCH_PRIVATE_WORDS];
}

// stream, control block, per-workgroup partial slots, pinned mirror: what every lane owns besides its scratch
static hipError_t init_frame_resources(gs_ctx *c, gs_ctx *primary = nullptr)
{
    hipError_t e;
#define IFR(call) do { e = (call); if (e != hipSuccess) return e; } while (0)
    if (primary) { c->stream = primary->stream; c->own_stream = false; c->exec = primary; }   // a twin: frames on its primary's stream and thread
    else { IFR(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; c->exec = c; }
    IFR(hipMalloc((void **)&c->ctl, sizeof(GsControl)));
    IFR(hipMemset(c->ctl, 0, sizeof(GsControl)));
    IFR(hipMalloc((void **)&c->part_min, GS_MAX_PART * sizeof(unsigned long long)));
    IFR(hipMalloc((void **)&c->part_max, GS_MAX_PART * sizeof(unsigned long long)));
    IFR(hipMalloc((void **)&c->part_cnt, GS_MAX_PART * sizeof(uint32_t)));
    IFR(hipMalloc((void **)&c->part_valid, GS_MAX_PART * sizeof(uint32_t)));
    IFR(hipMalloc((void **)&c->part_vis, GS_MAX_PART * sizeof(uint32_t)));
    for (int k = 0; k < 2; k++) { IFR(hipMalloc((void **)&c->dhist[k], GS_DH_WORDS * sizeof(uint32_t))); IFR(hipMemset(c->dhist[k], 0, GS_DH_WORDS * sizeof(uint32_t))); }
    c->dh_next = 0; c->dh_dirty = nullptr;
    c->sort_near_req = 0;
    IFR(hipHostMalloc((void **)&c->ctl_host, sizeof(GsControl), hipHostMallocDefault));
    memset(c->ctl_host, 0, sizeof(GsControl));
    IFR(hipEventCreateWithFlags(&c->ev_frame, hipEventDisableTiming | hipEventReleaseToDevice));
    IFR(hipEventCreateWithFlags(&c->ev_gate, hipEventDisableTiming | hipEventReleaseToDevice));
    // (the queue's private memory: above.  One launch: 32 .. 512 of them -- a deeper warm-up of the new queue -- changed nothing)
    if (!primary) { k_touch_private<<<1, 64, 0, c->stream>>>(c->part_cnt, 1u); IFR(hipGetLastError()); }
#undef IFR
    return hipSuccess;
}

static void free_frame_resources(gs_ctx *c)
{
    if (c->exec == c || !c->exec) lane_stop_worker(c);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    delete c->log; c->log = nullptr;
    dev_free(c->depth); dev_free(c->key_a); dev_free(c->kv_b); dev_free(c->val_a);
    dev_free(c->hist); dev_free(c->radix_aux); dev_free(c->msd_grp); dev_free(c->spine);
    dev_free(c->proj); dev_free(c->rect); dev_free(c->tile_count); dev_free(c->zwin);
    dev_free(c->pair_a); dev_free(c->pair_b); dev_free(c->emit_extra); dev_free(c->row_cnt); dev_free(c->row_tot); dev_free(c->seg_diff);
    gs_comm_free_lane(c);
    dev_free(c->tile_range); dev_free(c->fb); dev_free(c->ctl); dev_free(c->state); dev_free(c->unsat_mask);
    dev_free(c->part_min); dev_free(c->part_max); dev_free(c->part_cnt); dev_free(c->part_valid); dev_free(c->part_vis); dev_free(c->dhist[0]); dev_free(c->dhist[1]);
    if (c->ctl_host) { (void)hipHostFree(c->ctl_host); c->ctl_host = nullptr; }
    if (c->ring) { for (int i = 0; i < GS_PROF_RING * GS_PROF_EVENTS; i++) if (c->ring[i]) (void)hipEventDestroy(c->ring[i]); free(c->ring); c->ring = nullptr; }
    free(c->ring_flags); c->ring_flags = nullptr;
    if (c->ev_frame) (void)hipEventDestroy(c->ev_frame);
    if (c->ev_gate) (void)hipEventDestroy(c->ev_gate);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = nullptr;
}

// switch the HIP-event profiling of one lane on/off (allocates its ring on first use, restarts its accumulators)
static int set_profile(gs_ctx *ctx, bool on, bool blend_only, uint32_t every = 1)
{
    TRY(lane_drain(ctx));
    GS_HIP(hipStreamSynchronize(ctx->stream));
    TRY(prof_drain(ctx));
    if (on && !ctx->ring) {
        ctx->ring = (hipEvent_t *)calloc(GS_PROF_RING * GS_PROF_EVENTS, sizeof(hipEvent_t));
        ctx->ring_flags = (uint8_t *)calloc(GS_PROF_RING, 1);
        if (!ctx->ring || !ctx->ring_flags) FAIL(GS_E_OOM, "out of host memory");
        for (int i = 0; i < GS_PROF_RING * GS_PROF_EVENTS; i++) GS_HIP(hipEventCreate(&ctx->ring[i]));
    }
    if (on && !ctx->profile) {                                   // (re)start accumulation
        ctx->stats.prof_frames = 0;
        ctx->stats.sum_ms_sort = ctx->stats.sum_ms_project = ctx->stats.sum_ms_bin = ctx->stats.sum_ms_blend = 0;
        GS_HIP(hipMemsetAsync(&ctx->ctl->acc_frames, 0, offsetof(GsControl, need_near) - offsetof(GsControl, acc_frames), ctx->stream));
        ctx->seen_acc_frames = 0;
    }
    ctx->profile = on; ctx->profile_blend_only = on && blend_only; ctx->profile_every = on ? every : 1; ctx->profile_tick = 0;
    return GS_OK;
}

// lane i of the owner `ctx`, created on first use, with the owner's current resident arrays / options and scratch for them
static int get_lane(gs_ctx *ctx, int i, gs_ctx **out)
{
    gs_ctx *L = ctx->lanes[i];
    if (!L) {
        gs_ctx *primary = nullptr;
        if (i >= GS_MAX_PRIMARY) TRY(get_lane(ctx, i - GS_MAX_PRIMARY, &primary));   // a twin: its primary lane first
        L = new (std::nothrow) gs_ctx();
        if (!L) FAIL(GS_E_OOM, "out of host memory");
        memset(L, 0, sizeof *L);
        L->parent = ctx; L->device = ctx->device;
        const hipError_t e = init_frame_resources(L, primary);
        if (e != hipSuccess) {
            snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "creating pipeline lane %d failed: %s", i, hipGetErrorString(e));
            free_frame_resources(L); delete L;
            return e == hipErrorOutOfMemory ? GS_E_OOM : GS_E_HIP;
        }
        ctx->lanes[i] = L;
        if (primary) {                                           // (the primary's enqueue thread reads `twin` under its mutex)
            if (primary->worker) { std::lock_guard<std::mutex> lk(primary->worker->m); primary->twin = L; }
            else primary->twin = L;
        }
        if (ctx->profile && set_profile(L, true, ctx->profile_blend_only, ctx->profile_every) != GS_OK) { memcpy(ctx->err, L->err, sizeof ctx->err); return GS_E_HIP; }
    }
    if (lane_drain(L) != GS_OK) { if (L != ctx) memcpy(ctx->err, L->err, sizeof ctx->err); return GS_E_HIP; }   // its worker is idle from here on
    if (L != ctx) {
        L->splat = ctx->splat; L->sort_rows = ctx->sort_rows; L->bound_r = ctx->bound_r; L->pow10tab = ctx->pow10tab;
        L->n = ctx->n; L->cap = ctx->cap; L->renderable = ctx->renderable;
        L->scene_depth = ctx->scene_depth; L->scene_rgba = ctx->scene_rgba; L->scene_w = ctx->scene_w; L->scene_h = ctx->scene_h;
        L->record_staged = ctx->record_staged; L->t_eps = ctx->t_eps; L->wide_pairs = ctx->wide_pairs;
        if (ensure_lane_scratch(L, ctx->cap) != GS_OK) { memcpy(ctx->err, L->err, sizeof ctx->err); return GS_E_OOM; }
    }
    *out = L;
    return GS_OK;
}

// Every lane the options imply -- GS_OPT_PIPELINE_DEPTH lanes, their twins with GS_OPT_FRAME_BATCH = 2 -- created now, with scratch for the
// resident splats and their enqueue threads: a caller that asks for pipelining on a loaded context pays for the lanes THERE (half a
// millisecond each: a stream, a pinned control block, two dozen allocations, a thread) and not in its first six frames
// (`cold_orbit`: 2.6 of the first lap's 11.5 ms).  Lanes of a context nobody asked to pipeline are still created on first use.
static int prepare_lanes(gs_ctx *ctx)
{
    if (!ctx->n || ctx->user_stream || !ctx->renderable) return GS_OK;
    for (int i = 0; i < ctx->pipe_depth; i++) {
        gs_ctx *L = nullptr;
        TRY(get_lane(ctx, i, &L));
        if (ctx->enqueue_threads && ensure_worker(L) != GS_OK) FAIL(GS_E_OOM, "out of host memory");
        if (ctx->frame_batch == 2 && ctx->enqueue_threads) TRY(get_lane(ctx, i + GS_MAX_PRIMARY, &L));
    }
    return GS_OK;
}

// after drain_all(): give every lane the owner's current resident arrays, scene inputs and options
static void refresh_lanes(gs_ctx *ctx)
{
    for (int i = 1; i < GS_MAX_LANES; i++) {
        gs_ctx *L = ctx->lanes[i];
        if (!L) continue;
        L->splat = ctx->splat; L->sort_rows = ctx->sort_rows; L->bound_r = ctx->bound_r; L->pow10tab = ctx->pow10tab;
        L->n = ctx->n; L->cap = ctx->cap; L->renderable = ctx->renderable;
        L->scene_depth = ctx->scene_depth; L->scene_rgba = ctx->scene_rgba; L->scene_w = ctx->scene_w; L->scene_h = ctx->scene_h;
        L->record_staged = ctx->record_staged; L->t_eps = ctx->t_eps; L->wide_pairs = ctx->wide_pairs;
    }
}

// the lane a NEW frame goes to: the next one if the current frame was handed off asynchronously
// do asynchronous frames move on to another lane / twin?  (more than one lane, or one lane whose frames pair with its twin's)
static inline bool gs_rotates(const gs_ctx *ctx) { return ctx->pipe_depth > 1 || (ctx->frame_batch == 2 && ctx->enqueue_threads); }

static int next_frame_lane(const gs_ctx *ctx, int *rot = nullptr, bool solo = false)
{
    if (!(ctx->cur_async && !ctx->user_stream && gs_rotates(ctx))) { if (rot) *rot = ctx->rot; return ctx->cur; }
    if (ctx->frame_batch == 2 && ctx->enqueue_threads && solo) {
        // a frame of two views (XR eyes on one GPU): primary lanes only -- the twin's scratch is the second view's
        const int lane = (ctx->cur % GS_MAX_PRIMARY + 1) % ctx->pipe_depth;
        if (rot) *rot = 2 * lane;
        return lane;
    }
    if (ctx->frame_batch == 2 && ctx->enqueue_threads) {
        // lane 0, its twin, lane 1, its twin, ...: the two frames of a pair sit behind each other in one enqueue thread's queue
        const int r = (ctx->rot + 1) % (2 * ctx->pipe_depth);
        if (rot) *rot = r;
        return r / 2 + (r % 2) * GS_MAX_PRIMARY;
    }
    if (rot) *rot = 0;
    return ctx->cur >= GS_MAX_PRIMARY ? 0 : (ctx->cur + 1) % ctx->pipe_depth;
}

static int lane_rc(gs_ctx *ctx, gs_ctx *L, int rc)
{
    if (rc != GS_OK && L != ctx) memcpy(ctx->err, L->err, sizeof ctx->err);
    return rc;
}

extern "C" {

GS_API uint32_t gs_version(void) { return 0x000500; }

GS_API int gs_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

GS_API const char *gs_last_error(const gs_ctx *ctx) { return ctx ? ctx->err : g_create_err; }

GS_API int gs_create(int device, gs_ctx **out)
{
    if (!out) return GS_E_BADARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        snprintf(g_create_err, sizeof g_create_err, "no HIP device available (%s); this library has no CPU fallback",
                 e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return GS_E_NODEVICE;
    }
    if (device < 0 || device >= ndev) {
        snprintf(g_create_err, sizeof g_create_err, "device %d out of range (have %d)", device, ndev);
        return GS_E_BADARG;
    }
    gs_ctx *ctx = new (std::nothrow) gs_ctx();
    if (!ctx) { snprintf(g_create_err, sizeof g_create_err, "out of host memory"); return GS_E_OOM; }
    memset(ctx, 0, sizeof *ctx);
    ctx->device = device; ctx->renderable = true; ctx->t_eps = 1.0f / 1024.0f; ctx->near_frac = 0.25f;
    ctx->lanes[0] = ctx; ctx->pipe_depth = 3; ctx->enqueue_threads = true; ctx->frame_batch = 1; ctx->exec = ctx; ctx->sort_near_opt = 1; ctx->auto_retry = true; ctx->subtile_opt = 1;
    { const char *e = getenv("GS_SPEC_STASH"); ctx->near_spec_opt = !(e && e[0] == '0'); }   // (A/B: near-only sorts without the speculative stash)
#define CREATE_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) {                                              \
        snprintf(g_create_err, sizeof g_create_err, "%s failed: %s", #call, hipGetErrorString(_e)); gs_destroy(ctx);      \
        return GS_E_HIP; } } while (0)
    CREATE_HIP(hipSetDevice(device));
    CREATE_HIP(init_frame_resources(ctx));
    {
        double tab[GS_POW10_ENTRIES];                           // a few KB: on the stack
        gs_build_pow10_table(tab);
        CREATE_HIP(hipMalloc((void **)&ctx->pow10tab, sizeof tab));
        CREATE_HIP(hipMemcpy(ctx->pow10tab, tab, sizeof tab, hipMemcpyHostToDevice));
    }
#undef CREATE_HIP
    *out = ctx;
    return GS_OK;
}

GS_API int gs_destroy(gs_ctx *ctx)
{
    if (!ctx) return GS_OK;
    (void)hipSetDevice(ctx->device);
    (void)gs_comm_destroy(ctx);
    for (int i = GS_MAX_LANES - 1; i >= 1; i--)                  // twins before the lanes whose streams they borrow
        if (ctx->lanes[i]) { free_frame_resources(ctx->lanes[i]); delete ctx->lanes[i]; ctx->lanes[i] = nullptr; }
    free_frame_resources(ctx);
    dev_free(ctx->splat); dev_free(ctx->sort_rows); dev_free(ctx->bound_r); dev_free(ctx->pow10tab);
    dev_free(ctx->scene_depth); dev_free(ctx->scene_rgba);
    if (ctx->ev_sort) (void)hipEventDestroy(ctx->ev_sort);
    delete ctx;
    return GS_OK;
}

GS_API int gs_clear(gs_ctx *ctx)
{
    CHECK_CTX(ctx);
    GS_HIP(hipSetDevice(ctx->device));
    TRY(drain_all(ctx));
    ctx->n = 0; ctx->renderable = true; ctx->pair_hint = 0; ctx->run_hint = 0; ctx->last_pairs = 0; ctx->last_visible = 0;
    ctx->near_frac = 0.25f; ctx->near_floor = 0.0f; ctx->clean_frames = 0; ctx->skip_hold = 0; ctx->single_round_frames = 0; ctx->last_kept = 0; ctx->share_measured = false; ctx->need_margin = 0.0f; ctx->cold_sorts = 0; ctx->cold_frames = 0; ctx->share_kind = 0; memset(ctx->need_hist, 0, sizeof ctx->need_hist);
    ctx->near_stash_off = false; ctx->near_spec = false; ctx->near_spec_hold = 0; ctx->near_spec_backoff = 0; ctx->near_spec_miss_credit = 0;
    for (int i = 0; i < GS_MAX_LANES; i++) {
        gs_ctx *L = ctx->lanes[i];
        if (!L) continue;
        L->have_sort = false; L->sorted = nullptr; L->n = 0;
        memset(&L->stats, 0, sizeof L->stats);
    }
    ctx->cur = 0; ctx->rot = 0; ctx->cur_async = false; ctx->pend_lane = 0;   // (a sort begun before the clear is dropped, like the worker's data: index.js:573-575)
    return GS_OK;
}

GS_API size_t gs_count(const gs_ctx *ctx) { return ctx ? ctx->n : 0; }

// pack `nrows` .splat rows that already sit in device memory behind the resident splats (capacity ensured by the caller)
static int append_device_rows(gs_ctx *ctx, const uint4 *rows_dev, size_t nrows)
{
    int rc = gs_launch_pack(ctx, rows_dev, ctx->n, nrows);
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (rc == GS_OK && e != hipSuccess) { snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "pack failed: %s", hipGetErrorString(e)); rc = GS_E_HIP; }
    if (rc != GS_OK) return rc;
    ctx->n += nrows; ctx->renderable = true;
    for (int i = 0; i < GS_MAX_LANES; i++) if (ctx->lanes[i]) ctx->lanes[i]->have_sort = false;
    ctx->stats.n_splats = ctx->n;
    return GS_OK;
}

GS_API int gs_push_splat(gs_ctx *ctx, const void *rows, size_t nrows)
{
    CHECK_CTX(ctx);
    if (nrows == 0) return GS_OK;                               // pushDataBuffer: vertexCount <= 0 -> return (index.js:333-335)
    if (!rows) FAIL(GS_E_BADARG, "gs_push_splat: rows is NULL");
    if (!ctx->renderable && ctx->n) FAIL(GS_E_STATE, "gs_push_splat after gs_push_matrices: mixed ingest is not supported");
    GS_HIP(hipSetDevice(ctx->device));
    TRY(drain_all(ctx));
    TRY(ensure_capacity(ctx, ctx->n + nrows));
    uint4 *stage = nullptr;
    TRY(dev_alloc(ctx, &stage, nrows * 2));
    hipError_t e = hipMemcpyAsync(stage, rows, nrows * 32, hipMemcpyHostToDevice, ctx->stream);
    int rc = GS_OK;
    if (e != hipSuccess) { snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "upload failed: %s", hipGetErrorString(e)); rc = GS_E_HIP; }
    if (rc == GS_OK) rc = append_device_rows(ctx, stage, nrows);
    else (void)hipStreamSynchronize(ctx->stream);
    dev_free(stage);
    return rc;
}

GS_API int gs_push_matrices(gs_ctx *ctx, const float *matrices, size_t nrows)
{
    CHECK_CTX(ctx);
    if (nrows == 0) return GS_OK;
    if (!matrices) FAIL(GS_E_BADARG, "gs_push_matrices: matrices is NULL");
    if (ctx->renderable && ctx->n) FAIL(GS_E_STATE, "gs_push_matrices after gs_push_splat: mixed ingest is not supported");
    GS_HIP(hipSetDevice(ctx->device));
    TRY(drain_all(ctx));
    TRY(ensure_capacity(ctx, ctx->n + nrows));
    // strided H2D copy: only elements 12..15 of every 16-float row are ever read (index.js:520-548)
    GS_HIP(hipMemcpy2DAsync(ctx->sort_rows + ctx->n, 16, matrices + 12, 64, 16, nrows, hipMemcpyHostToDevice, ctx->stream));
    GS_HIP(hipStreamSynchronize(ctx->stream));
    ctx->n += nrows; ctx->renderable = false;
    for (int i = 0; i < GS_MAX_LANES; i++) if (ctx->lanes[i]) ctx->lanes[i]->have_sort = false;
    ctx->stats.n_splats = ctx->n;
    return GS_OK;
}

// .ply -> .splat rows on the GPU (gs_ply.hip); rows_dev receives a device buffer of *nrows x 32 B that the caller frees.
// A NaN importance (engine-defined order in the reference) is handed to the host converter so that there is ONE
// definition of that case.
static int ply_rows_to_device(gs_ctx *ctx, const void *bytes, size_t nbytes, uint4 **rows_dev, size_t *nrows)
{
    *rows_dev = nullptr; *nrows = 0;
    gsm::PlyLayout L;
    size_t n = 0, data_start = 0;
    int rc = gs_ply_plan(bytes, nbytes, &L, &n, &data_start, ctx->err, sizeof ctx->err);
    *nrows = n;
    if (rc != GS_OK || !n) return rc;
    GS_HIP(hipSetDevice(ctx->device));
    TRY(drain_all(ctx));                                        // the converter borrows lane 0's stream and radix scratch
    uint4 *rows = nullptr;
    TRY(dev_alloc(ctx, &rows, n * 2));
    bool had_nan = false;
    rc = gs_ply_rows_device(ctx, (const uint8_t *)bytes + data_start, L, n, rows, &had_nan);
    if (rc == GS_OK && had_nan) {
        std::vector<uint8_t> host;
        try { host.resize(n * 32); } catch (...) { dev_free(rows); FAIL(GS_E_OOM, "out of host memory for %zu rows", n); }
        rc = gs_ply_to_splat(bytes, nbytes, host.data(), &n, ctx->err, sizeof ctx->err);
        if (rc == GS_OK && hipMemcpy(rows, host.data(), n * 32, hipMemcpyHostToDevice) != hipSuccess) {
            snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "upload failed"); rc = GS_E_HIP;
        }
    }
    if (rc != GS_OK) { dev_free(rows); return rc; }
    *rows_dev = rows;
    return GS_OK;
}

GS_API int gs_load_ply(gs_ctx *ctx, const void *bytes, size_t nbytes)
{
    CHECK_CTX(ctx);
    if (!bytes) FAIL(GS_E_BADARG, "gs_load_ply: bytes is NULL");
    if (!ctx->renderable && ctx->n) FAIL(GS_E_STATE, "gs_load_ply after gs_push_matrices: mixed ingest is not supported");
    uint4 *rows = nullptr; size_t n = 0;
    TRY(ply_rows_to_device(ctx, bytes, nbytes, &rows, &n));
    if (!n) return GS_OK;
    int rc = ensure_capacity(ctx, ctx->n + n);
    if (rc == GS_OK) rc = append_device_rows(ctx, rows, n);         // the rows never leave HBM
    dev_free(rows);
    return rc;
}

GS_API int gs_ply_to_splat_gpu(gs_ctx *ctx, const void *bytes, size_t nbytes, void *out_rows, size_t *out_nrows)
{
    CHECK_CTX(ctx);
    if (!bytes || !out_nrows) FAIL(GS_E_BADARG, "gs_ply_to_splat_gpu: NULL argument");
    if (!out_rows) {                                             // size query: header + property checks only
        gsm::PlyLayout L; size_t ds = 0;
        return gs_ply_plan(bytes, nbytes, &L, out_nrows, &ds, ctx->err, sizeof ctx->err);
    }
    uint4 *rows = nullptr;
    TRY(ply_rows_to_device(ctx, bytes, nbytes, &rows, out_nrows));
    if (!*out_nrows) return GS_OK;
    const hipError_t e = hipMemcpy(out_rows, rows, *out_nrows * 32, hipMemcpyDeviceToHost);
    dev_free(rows);
    if (e != hipSuccess) FAIL(GS_E_HIP, "download failed: %s", hipGetErrorString(e));
    return GS_OK;
}

static int sort_common(gs_ctx *ctx, const float view[4], const float *cutout16, const GsSortStrip *strip, uint32_t *out_idx, uint32_t *out_n, bool solo = false);

GS_API int gs_sort(gs_ctx *ctx, const float view[4], const float *cutout16, uint32_t *out_idx, uint32_t *out_n)
{
    return sort_common(ctx, view, cutout16, nullptr, out_idx, out_n);
}

// the sort of a frame of two views drawn by this context (gs_sort_gathered): with GS_OPT_FRAME_BATCH the frame stays on a primary lane
int gs_sort_two_views(gs_ctx *ctx, const float view[4], const float *cutout16) { return sort_common(ctx, view, cutout16, nullptr, nullptr, nullptr, true); }

GS_API int gs_sort_for(gs_ctx *ctx, const float view[4], const float *cutout16, const gs_render_params *strip, uint32_t *out_idx, uint32_t *out_n)
{
    CHECK_CTX(ctx);
    if (!strip) return sort_common(ctx, view, cutout16, nullptr, out_idx, out_n);
    if (strip->fb_width <= 0 || strip->x0 < 0 || strip->x1 > strip->fb_width || strip->x0 >= strip->x1)
        FAIL(GS_E_BADARG, "gs_sort_for: bad strip [%d,%d) for width %d", strip->x0, strip->x1, strip->fb_width);
    GsSortStrip st;
    memcpy(st.mv, strip->model_view, sizeof st.mv); memcpy(st.proj, strip->projection, sizeof st.proj);
    st.focal = strip->focal > 0 ? strip->focal : (float)(((double)strip->fb_height / 2.0) * fabs((double)strip->projection[5]));
    st.vw = (float)strip->fb_width; st.vh = (float)strip->fb_height; st.x0 = strip->x0; st.x1 = strip->x1;
    // a perspective projection whose w does not depend on x or y (three.js PerspectiveCamera, WebXR eye frusta); anything else
    // is sorted whole
    const float *P = strip->projection;
    const bool persp = P[3] == 0.0f && P[7] == 0.0f && P[15] == 0.0f && P[11] != 0.0f;
    return sort_common(ctx, view, cutout16, persp ? &st : nullptr, out_idx, out_n);
}

static int sort_common(gs_ctx *ctx, const float view[4], const float *cutout16, const GsSortStrip *strip, uint32_t *out_idx, uint32_t *out_n, bool solo)
{
    CHECK_CTX(ctx);
    if (!view) FAIL(GS_E_BADARG, "gs_sort: view is NULL");
    if (ctx->n == 0) {                                          // sort before any push (index.js:588-590)
        if (out_idx) out_idx[0] = 0;
        if (out_n) *out_n = 1;
        ctx->lanes[ctx->cur]->have_sort = false; ctx->lanes[ctx->cur]->stats.n_sorted = 0;
        if (ctx->lanes[ctx->cur]->log) ctx->lanes[ctx->cur]->log->open = false;
        return GS_OK;
    }
    GS_HIP(hipSetDevice(ctx->device));
    // a sort starts a frame: it goes to the next lane if the previous frame was handed off with GS_RENDER_ASYNC
    gs_ctx *L = nullptr;
    int rot = 0;
    const int lane = next_frame_lane(ctx, &rot, solo);
    TRY(get_lane(ctx, lane, &L));
    if (solo && ctx->frame_batch == 2 && lane < GS_MAX_PRIMARY) { gs_ctx *T = nullptr; TRY(get_lane(ctx, lane + GS_MAX_PRIMARY, &T)); }   // (the second view's scratch)
    if (ctx->pend_lane > 0 && lane == ctx->pend_lane - 1) FAIL(GS_E_STATE, "gs_sort: this frame's lane holds the sort begun with gs_sort_begin: collect it with gs_sort_poll first");
    ctx->cur = lane; ctx->rot = rot; ctx->cur_async = false;
    log_sort(L, view, cutout16, strip);
    {   // The share of splats binned first is measured in POSITIONS of the order the frames are drawn from.  A strip's order (gs_sort_for: the
        // splats that can reach the strip -- the whole frame: the frustum) is a sub-sequence of the whole one: "the nearest 30 000" of it reach
        // five times as deep.  When the kind of order changes, what was measured on the other kind is dropped (round 6: bench.py's frames sorted
        // for their frustum, then its one-frame-at-a-time extras with gs_sort, drew those with a share a fifth of what they needed and were
        // drawn again by gs_sync until the share had crept up).
        const uint32_t kind = strip ? 2u : 1u;                      // (strips of one frame differ less from each other than any of them from the whole order:
                                                                    // a context that draws several strips in turn keeps one measurement, as before)
        if (ctx->share_kind && ctx->share_kind != kind && ctx->near_fixed_permille <= 0) {
            ctx->near_frac = 0.25f; ctx->near_floor = 0.0f; ctx->share_measured = false; ctx->need_margin = 0.0f; ctx->cold_sorts = 0; ctx->cold_frames = 0;
            ctx->clean_frames = 0; ctx->skip_hold = 0;
            memset(ctx->need_hist, 0, sizeof ctx->need_hist); memset(ctx->need_hist_frames, 0, sizeof ctx->need_hist_frames);
            for (int i = 0; i < GS_MAX_LANES; i++) if (ctx->lanes[i]) { ctx->lanes[i]->need_seed_pending = 1u; ctx->lanes[i]->need_word_est = 0; }
        }
        ctx->share_kind = kind;
    }
    const uint32_t near_req = (out_idx || out_n) ? 0u : sort_near_request(ctx);     // (the caller wants the order itself: all of it)
    // (a context that has not measured its share yet draws its next frame synchronously -- gs_render_uniforms --: its sort on the caller's
    // thread then, not on an enqueue thread that was created a moment ago and has to be woken first: 0.14 ms of that call's 0.7)
    // (... at most two sorts in a row: a context whose frames never measure -- counting renders -- keeps its threads)
    const bool cold = !ctx->share_measured && ctx->near_fixed_permille <= 0 && ctx->cold_sorts < 2u;   // (get_lane has drained the lane: nothing of it is waiting on its thread)
    if (cold) ctx->cold_sorts++; else if (ctx->share_measured) { ctx->cold_sorts = 0; ctx->cold_frames = 0; }
    if (ctx->enqueue_threads && gs_rotates(ctx) && !ctx->user_stream && !out_idx && !out_n && !cold) {
        // nothing to hand back: the lane's worker thread does the launching (a failure surfaces at gs_sync())
        GsLaneCmd c;
        c.type = 0; memcpy(c.view, view, sizeof c.view);
        c.has_cutout = cutout16 != nullptr;
        if (cutout16) memcpy(c.cutout, cutout16, sizeof c.cutout);
        c.has_strip = strip != nullptr;
        if (strip) c.strip = *strip;
        c.near_req = near_req;
        c.device_rgba = nullptr; c.host_rgba = nullptr; c.stride = 0;
        L->have_sort = true;                                    // (set again by the worker; the render command follows it)
        if (lane_push(L, c) != GS_OK) FAIL(GS_E_OOM, "out of host memory");
        return GS_OK;
    }
    TRY(lane_rc(ctx, L, gs_run_sort(L, view, cutout16, strip, near_req)));
    if (out_idx || out_n) {
        LANE_HIP(L, hipMemcpyAsync(L->ctl_host, L->ctl, sizeof(GsControl), hipMemcpyDeviceToHost, L->stream));
        LANE_HIP(L, hipStreamSynchronize(L->stream));
        const uint32_t V = L->ctl_host->n_kept;
        L->stats.n_sorted = V; L->sorted_n_host = V;
        if (out_n) *out_n = V;
        if (out_idx && V) GS_HIP(hipMemcpy(out_idx, L->sorted, (size_t)V * 4, hipMemcpyDeviceToHost));
    }
    return GS_OK;
}

// ---- the reference's single-flight rhythm (include/gs_splat.h: gs_sort_begin / gs_sort_poll; index.js:201-207, 438-455)
GS_API int gs_sort_begin(gs_ctx *ctx, const float view[4], const float *cutout16)
{
    CHECK_CTX(ctx);
    if (!view) FAIL(GS_E_BADARG, "gs_sort_begin: view is NULL");
    if (ctx->pend_lane) FAIL(GS_E_STATE, "gs_sort_begin: a sort is in flight (sortReady is false, index.js:439-440): collect it with gs_sort_poll");
    if (ctx->n == 0) { ctx->pend_lane = -1; return GS_OK; }     // sort before any push: the reply will be [0] (index.js:588-590)
    GS_HIP(hipSetDevice(ctx->device));
    // the lane the sort runs on: a primary lane other than the one whose order the draws use (created on first use, whatever
    // GS_OPT_PIPELINE_DEPTH says: the order in flight needs scratch of its own)
    const int front = ctx->cur % GS_MAX_PRIMARY;
    const int back = (front + 1) % (ctx->pipe_depth > 2 ? ctx->pipe_depth : 2);
    gs_ctx *B = nullptr;
    TRY(get_lane(ctx, back, &B));                                // (drained: nothing of it waits on its enqueue thread)
    if (!ctx->ev_sort) GS_HIP(hipEventCreateWithFlags(&ctx->ev_sort, hipEventDisableTiming));
    const int rc = lane_rc(ctx, B, gs_run_sort(B, view, cutout16, nullptr, 0));
    B->have_sort = false;                                        // (not an order to draw from until it has been collected)
    if (rc != GS_OK) return rc;
    LANE_HIP(B, hipMemcpyAsync(B->ctl_host, B->ctl, sizeof(GsControl), hipMemcpyDeviceToHost, B->stream));
    LANE_HIP(B, hipEventRecord(ctx->ev_sort, B->stream));
    ctx->pend_lane = back + 1; ctx->pend_n = ctx->n; ctx->pend_gen = B->sort_gen;
    memcpy(ctx->pend_view, view, sizeof ctx->pend_view); ctx->pend_has_cutout = cutout16 != nullptr;
    if (cutout16) memcpy(ctx->pend_cutout, cutout16, sizeof ctx->pend_cutout);
    return GS_OK;
}

GS_API int gs_sort_poll(gs_ctx *ctx, int wait, uint32_t *out_idx, uint32_t *out_n, int *done)
{
    CHECK_CTX(ctx);
    if (!done) FAIL(GS_E_BADARG, "gs_sort_poll: done is NULL");
    *done = 1;
    if (out_n) *out_n = 0;
    if (!ctx->pend_lane) return GS_OK;
    if (ctx->pend_lane < 0) {                                    // begun before any push
        ctx->pend_lane = 0;
        if (ctx->n == 0) { if (out_idx) out_idx[0] = 0; if (out_n) *out_n = 1; return GS_OK; }
        FAIL(GS_E_STATE, "gs_sort_poll: the sort was begun on an empty context and splats were pushed since: begin it again");
    }
    GS_HIP(hipSetDevice(ctx->device));
    const int back = ctx->pend_lane - 1;
    gs_ctx *B = ctx->lanes[back];
    if (!wait) {
        const hipError_t e = hipEventQuery(ctx->ev_sort);
        if (e == hipErrorNotReady) { *done = 0; return GS_OK; }
        if (e != hipSuccess) { ctx->pend_lane = 0; FAIL(GS_E_HIP, "hipEventQuery failed: %s", hipGetErrorString(e)); }
    } else GS_HIP(hipEventSynchronize(ctx->ev_sort));
    ctx->pend_lane = 0;
    if (ctx->pend_n != ctx->n || B->n != ctx->n || !B->sorted || B->sort_gen != ctx->pend_gen) {
        // the resident data changed under the sort (pushes drain every lane and drop the lanes' orders), or another sort has used the lane's
        // scratch since (gs_sync drawing a flagged frame of this lane again): over what is resident now
        gs_ctx *L = nullptr;
        TRY(get_lane(ctx, back, &L));
        TRY(lane_rc(ctx, L, gs_run_sort(L, ctx->pend_view, ctx->pend_has_cutout ? ctx->pend_cutout : nullptr, nullptr, 0)));
        LANE_HIP(L, hipMemcpyAsync(L->ctl_host, L->ctl, sizeof(GsControl), hipMemcpyDeviceToHost, L->stream));
        LANE_HIP(L, hipStreamSynchronize(L->stream));
        B = L;
    }
    const uint32_t V = B->ctl_host->n_kept;
    B->have_sort = true; B->stats.n_sorted = V; B->sorted_n_host = V;
    if (out_n) *out_n = V;
    if (out_idx && V) GS_HIP(hipMemcpy(out_idx, B->sorted, (size_t)V * 4, hipMemcpyDeviceToHost));
    // the new order is the one the draws use from here on (the reply handler, index.js:201-207)
    ctx->cur = back; ctx->rot = 2 * back; ctx->cur_async = false;
    log_sort(B, B->sv_view, B->sv_has_cutout ? B->sv_cutout : nullptr, nullptr);
    return GS_OK;
}

// Begin a frame whose sort is a call (gs_comm.hip: the order comes from -- or goes to -- the other ranks).  Two steps, so that the
// caller can take what must stay in step across the ranks (its turn in the sort rota, the communicator ticket) only after
// everything that can fail without the call having been queued -- here: lane selection and the lane's scratch -- has succeeded.
int gs_sort_call_begin(gs_ctx *ctx, const float view[4], const float *cutout16)
{
    CHECK_CTX(ctx);
    GS_HIP(hipSetDevice(ctx->device));
    gs_ctx *L = nullptr;
    int rot = 0;
    const int lane = next_frame_lane(ctx, &rot, false);
    TRY(get_lane(ctx, lane, &L));
    ctx->cur = lane; ctx->rot = rot; ctx->cur_async = false;
    log_sort(L, view, cutout16, nullptr);
    log_undecidable(L);                                            // (the other ranks are part of this frame's sort)
    L->have_sort = true;
    return GS_OK;
}

// ... and hand the call to the frame's lane: queued behind the lane's work, or run here.  The call is ALWAYS run.
int gs_sort_call_issue(gs_ctx *ctx, void *call_p)
{
    std::function<int(gs_ctx *)> &call = *static_cast<std::function<int(gs_ctx *)> *>(call_p);
    gs_ctx *L = ctx->lanes[ctx->cur];
    if (ctx->enqueue_threads && gs_rotates(ctx) && !ctx->user_stream) {
        GsLaneCmd c;
        c.type = 2; c.has_cutout = false; c.has_strip = false; c.device_rgba = nullptr; c.host_rgba = nullptr; c.stride = 0;
        bool queued = false;
        try { c.call = call; queued = lane_push(L, c) == GS_OK; } catch (...) { queued = false; }
        if (queued) return GS_OK;                                  // (not a frame's end: the rotation is decided by its render)
    }
    const int rc = lane_rc(ctx, L, lane_drain(L));                 // run here, whatever came before: the other ranks wait for this exchange
    const int rc2 = lane_rc(ctx, L, call(L));
    return rc != GS_OK ? rc : rc2;
}

int gs_sort_by_call(gs_ctx *ctx, const float view[4], const float *cutout16, void *call_p)
{
    TRY(gs_sort_call_begin(ctx, view, cutout16));
    return gs_sort_call_issue(ctx, call_p);
}

int gs_fill_uniforms(gs_ctx *ctx /* owner: options, adaptive share, scene */, const gs_render_params *p, GsFrameUniforms &u)
{
    if (!p) FAIL(GS_E_BADARG, "render params NULL");
    if (p->fb_width <= 0 || p->fb_height <= 0 || p->fb_width > 65535 * GS_TILE || p->fb_height > 65535 * GS_TILE)
        FAIL(GS_E_BADARG, "bad framebuffer size %dx%d", p->fb_width, p->fb_height);
    if (p->x0 < 0 || p->x1 > p->fb_width || p->x0 >= p->x1) FAIL(GS_E_BADARG, "bad strip [%d,%d) for width %d", p->x0, p->x1, p->fb_width);
    memcpy(u.mv, p->model_view, sizeof u.mv); memcpy(u.proj, p->projection, sizeof u.proj);
    u.W = p->fb_width; u.H = p->fb_height; u.x0 = p->x0; u.x1 = p->x1;
    u.out_pitch = p->x1 - p->x0;
    u.rc_stride = 0;                                               // (the binning is chosen per round: gs_render.hip)
    u.status = nullptr;                                            // (the frame's lane supplies its own word: gs_render_uniforms)
    u.need_seed = 0;
    u.x1b = p->x0 + ((p->x1 - p->x0 + 3) & ~3);
    if (u.x1b > p->fb_width) u.x1b = p->fb_width;
    u.vw = (float)p->fb_width; u.vh = (float)p->fb_height;
    // focal = (viewport.w / 2) * |P[5]| in JS f64, uploaded as a float uniform (index.js:191-194)
    u.focal = p->focal > 0 ? p->focal : (float)(((double)p->fb_height / 2.0) * fabs((double)p->projection[5]));
    u.tiles_x = (p->x1 - p->x0 + GS_TILE - 1) / GS_TILE; u.tiles_y = (p->fb_height + GS_TILE - 1) / GS_TILE;
    // the pair sort keys on the tile id in two digits of at most 9 bits (GS_RADIX_MAX_BINS): 2^18 tiles = 67 Mpixel per strip
    if ((uint64_t)u.tiles_x * (uint64_t)u.tiles_y > (1ull << 18))
        FAIL(GS_E_BADARG, "strip of %dx%d pixels has %llu tiles; at most 262144 (16x16-pixel tiles) are supported: render it in column strips",
             p->x1 - p->x0, p->fb_height, (unsigned long long)u.tiles_x * (unsigned long long)u.tiles_y);
    memcpy(u.bg, p->background, sizeof u.bg);
    u.t_eps = ctx->t_eps; u.flags = p->flags; u.record_staged = ctx->record_staged; u.split_min = ctx->blend_split_min;
    u.mask_words = (uint32_t)(u.tiles_x + 31) / 32;
    // round 0 covers the nearest near_frac * N splats; counting / no-early-out renders need every fragment -> one round
    const float frac = ctx->near_fixed_permille > 0 ? ctx->near_fixed_permille / 1000.0f : ctx->near_frac;
    if ((p->flags & (GS_RENDER_COUNT_FRAGS | GS_RENDER_NO_EARLY_OUT)) || frac >= 1.0f) u.near_count = 0xFFFFFFFFu;
    else { const double nc = ceil((double)frac * (double)ctx->n); u.near_count = nc < 1 ? 1u : (uint32_t)nc; }
    u.has_depth = ctx->scene_depth != nullptr; u.has_scene_rgba = ctx->scene_rgba != nullptr;
    if ((u.has_depth || u.has_scene_rgba) && (ctx->scene_w != p->fb_width || ctx->scene_h != p->fb_height))
        FAIL(GS_E_BADARG, "scene inputs are %dx%d but the frame is %dx%d", ctx->scene_w, ctx->scene_h, p->fb_width, p->fb_height);
    u.skip_round1 = (u.near_count != 0xFFFFFFFFu && round1_skippable(ctx)) ? 1u : 0u;
    // sub-tile lists in the blend (GS_OPT_SUBTILE): where the last collected frame's visible splats touched few tiles each
    static const double subtile_ratio = getenv("GS_SUBTILE_RATIO") ? atof(getenv("GS_SUBTILE_RATIO")) : 16.0;
    u.subtile = ctx->subtile_opt == 2 ? 1u : (ctx->subtile_opt == 1 && ctx->last_visible && (double)ctx->last_pairs < subtile_ratio * (double)ctx->last_visible ? 1u : 0u);
    return GS_OK;
}

// per-frame buffers of lane `ctx` for the viewport of `u`
static int ensure_frame_buffers(gs_ctx *ctx, const GsFrameUniforms &u, bool need_fb)
{
    const size_t sw = (size_t)(u.x1 - u.x0), fb_bytes = sw * (size_t)u.H * 4;
    const size_t ntiles = (size_t)u.tiles_x * u.tiles_y;
    if (ntiles > ctx->tile_cap) { dev_free(ctx->tile_range); TRY(dev_alloc(ctx, &ctx->tile_range, ntiles)); ctx->tile_cap = ntiles; }
    if (need_fb && fb_bytes > ctx->fb_cap) { dev_free(ctx->fb); TRY(dev_alloc(ctx, &ctx->fb, fb_bytes)); ctx->fb_cap = fb_bytes; }
    if (!ctx->pair_cap) TRY(gs_ensure_pair_capacity(ctx, (size_t)1 << 22));
    const size_t mask_total = (size_t)u.tiles_y * u.mask_words;
    if (mask_total > ctx->mask_cap) { dev_free(ctx->unsat_mask); TRY(dev_alloc(ctx, &ctx->unsat_mask, mask_total)); ctx->mask_cap = mask_total; }
    if (ntiles * 256 > ctx->state_cap) { dev_free(ctx->state); TRY(dev_alloc(ctx, &ctx->state, ntiles * 256)); ctx->state_cap = ntiles * 256; }
    ctx->last_two_rounds = u.near_count != 0xFFFFFFFFu && ctx->n && ctx->have_sort;
    ctx->stats.n_tiles = ntiles; ctx->stats.blend_launches = 1;
    return GS_OK;
}

// pipelined frame on lane `ctx`: enqueue and return; gs_sync() reads the control block back (its flags and accumulators
// are cumulative, so one read-back per gs_sync() covers every frame since the last one -- a per-frame copy would be one
// more queue entry per frame).  An overflowing frame shows the background only and is reported (GS_E_RETRY) there.
// Runs on the lane's worker thread when GS_OPT_ENQUEUE_THREADS is on.
static int render_async_on_lane(gs_ctx *ctx, const GsFrameUniforms &u, void *device_rgba, uint8_t *host_rgba, size_t stride)
{
    TRY(ensure_frame_buffers(ctx, u, device_rgba == nullptr));
    TRY(ensure_sort_covers(ctx, u));
    TRY(gs_run_render(ctx, u, (uint8_t *)device_rgba));
    if (host_rgba) {
        // the frame follows its kernels on the lane's stream (page-locked destination: a real asynchronous copy that
        // overlaps the next frames' kernels; pageable memory works too, through the runtime's staging)
        const size_t sw = (size_t)(u.x1 - u.x0);
        const uint8_t *src = device_rgba ? (const uint8_t *)device_rgba : ctx->fb;
        GS_HIP(hipMemcpy2DAsync(host_rgba, stride ? stride : sw * 4, src, sw * 4, sw * 4, (size_t)u.H, hipMemcpyDeviceToHost, ctx->stream));
    }
    return prof_advance(ctx);
}

static bool gs_debug_cold() { static const bool on = getenv("GS_DEBUG_COLD") != nullptr; return on; }   // (diagnostic: what a synchronous frame's call spends where)

// synchronous frame on lane `ctx` (the owner supplies options and the adaptive share through fill_uniforms)
static int render_sync_on_lane(gs_ctx *ctx, const GsFrameUniforms &u, void *device_rgba, uint8_t *host_rgba, size_t stride)
{
    const size_t sw = (size_t)(u.x1 - u.x0);
    TRY(ensure_frame_buffers(ctx, u, device_rgba == nullptr));
    TRY(ensure_sort_covers(ctx, u));
    if (host_rgba) {
        if (!stride) stride = sw * 4;
        if (stride < sw * 4) FAIL(GS_E_BADARG, "stride %zu smaller than a row (%zu bytes)", stride, sw * 4);
    }
    // the frame's way to the host is queued right behind its kernels, before the control block has told whether the frame is
    // complete (one stream synchronisation per frame instead of two); the rare frame that is not is drawn and copied again
    auto queue_copy = [&]() -> int {
        if (!host_rgba) return GS_OK;
        const uint8_t *src = device_rgba ? (const uint8_t *)device_rgba : ctx->fb;
        // (a tight frame is ONE linear copy: the 2-D form goes through a slower path of the runtime even when pitch == width)
        if (stride == sw * 4) GS_HIP(hipMemcpyAsync(host_rgba, src, sw * 4 * (size_t)u.H, hipMemcpyDeviceToHost, ctx->stream));
        else GS_HIP(hipMemcpy2DAsync(host_rgba, stride, src, sw * 4, sw * 4, (size_t)u.H, hipMemcpyDeviceToHost, ctx->stream));
        return GS_OK;
    };
    for (int attempt = 0;; attempt++) {
        TRY(gs_run_render(ctx, u, (uint8_t *)device_rgba));
        GS_HIP(hipMemcpyAsync(ctx->ctl_host, ctx->ctl, sizeof(GsControl), hipMemcpyDeviceToHost, ctx->stream));
        TRY(queue_copy());
        GS_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->profile && ctx->ring) { ctx->ring_head++; ctx->ring_pending++; TRY(prof_drain(ctx)); }
        if (ctx->ctl_host->order_incomplete) {
            // the order this frame was drawn from was incomplete (a near-only sort whose survivors overflowed a chunk's stash, an
            // exchanged order cut short): round 0 itself was blended from the wrong positions.  Sort in full, draw the whole frame again.
            gs_ctx *P = gs_root(ctx);
            GS_HIP(hipMemsetAsync(&ctx->ctl->order_incomplete, 0, sizeof(uint32_t), ctx->stream));
            GS_HIP(hipMemsetAsync(&ctx->ctl->round1_missed, 0, sizeof(uint32_t), ctx->stream));
            if (ctx->ctl_host->near_overflow) { __atomic_store_n(&P->near_stash_off, true, __ATOMIC_RELAXED); GS_HIP(hipMemsetAsync(&ctx->ctl->near_overflow, 0, sizeof(uint32_t), ctx->stream)); }
            if (ctx->ctl_host->spec_fail) { P->stats.spec_misses++; gs_spec_back_off(P, ctx->ctl_host->spec_fail == 2u); GS_HIP(hipMemsetAsync(&ctx->ctl->spec_fail, 0, sizeof(uint32_t), ctx->stream)); }
            if (attempt >= 2) FAIL(GS_E_HIP, "the sorted order keeps coming back incomplete");
            TRY(gs_run_sort(ctx, ctx->sv_view, ctx->sv_has_cutout ? ctx->sv_cutout : nullptr, ctx->sv_has_strip ? &ctx->sv_strip : nullptr, 0));
            P->stats.retried_frames++;
            continue;
        }
        if (ctx->ctl_host->round1_missed) {
            // round 1 was skipped but a tile did not saturate: its mask bit and per-pixel state are intact -- finish it now
            GS_HIP(hipMemsetAsync(&ctx->ctl->round1_missed, 0, sizeof(uint32_t), ctx->stream));
            TRY(ensure_full_sort(ctx));                             // (round 1 reads the far part of the order)
            TRY(gs_run_round1(ctx, u, (uint8_t *)device_rgba));
            GS_HIP(hipMemcpyAsync(ctx->ctl_host, ctx->ctl, sizeof(GsControl), hipMemcpyDeviceToHost, ctx->stream));
            TRY(queue_copy());
            GS_HIP(hipStreamSynchronize(ctx->stream));
            ctx->ctl_host->round1_missed = 1;                      // seen by the adaptation below
        }
        bool over = false, failed = false;
        uint32_t need = 0, nfr = 0;
        const float frac_used = gs_root(ctx)->near_frac;
        TRY(collect_status(ctx, &over, &failed, nullptr, &need, &nfr));
        if (failed) { if (gs_root(ctx)->share_measured) share_missed(gs_root(ctx), frac_used); else share_raise(gs_root(ctx), frac_used); }
        else { share_from_need(gs_root(ctx), need, nfr); TRY(seed_need_words(gs_root(ctx), need, true)); }
        if (!over) break;
        if (attempt >= 2) FAIL(GS_E_OOM, "pair list keeps overflowing (%u pairs)", ctx->ctl_host->scan_total);
    }
    ctx->async_pending = false;
    if (u.flags & GS_RENDER_COUNT_FRAGS) ctx->stats.n_frags = ctx->ctl_host->n_frags;
    return GS_OK;
}

// the frame's lane runs `call` after everything handed to it so far: on its worker thread for asynchronous frames (so the
// caller keeps enqueuing), on the caller's thread otherwise.  The call's failure surfaces like a render's.
// always: the call runs even if something handed to the lane before it has failed, or the hand-over itself fails (the gather of
// a frame, which the other ranks wait for); the earlier failure is what is returned.
int gs_lane_call(gs_ctx *ctx, bool async, std::function<int(gs_ctx *)> call, bool always)
{
    gs_ctx *L = ctx->lanes[ctx->cur];
    if (always) log_undecidable(L);                                // (a frame's gather: every rank would have to draw it again)
    if (async && ctx->enqueue_threads && gs_rotates(ctx) && !ctx->user_stream) {
        GsLaneCmd c;
        c.type = 2; c.has_cutout = false; c.has_strip = false; c.device_rgba = nullptr; c.host_rgba = nullptr; c.stride = 0; c.call = call;
        L->async_pending = true; ctx->cur_async = true;
        if (lane_push(L, c) == GS_OK) return GS_OK;
        if (!always) FAIL(GS_E_OOM, "out of host memory");
    }
    const int rc = lane_rc(ctx, L, lane_drain(L));
    if (rc != GS_OK && !always) return rc;
    const int rc2 = lane_rc(ctx, L, call(L));
    return rc != GS_OK ? rc : rc2;
}

static int render_common(gs_ctx *ctx, const gs_render_params *p, void *device_rgba, uint8_t *host_rgba, size_t stride)
{
    GsFrameUniforms u;
    TRY(gs_fill_uniforms(ctx, p, u));
    return gs_render_uniforms(ctx, u, device_rgba, host_rgba, stride);
}

int gs_render_uniforms(gs_ctx *ctx, const GsFrameUniforms &u_in, void *device_rgba, uint8_t *host_rgba, size_t stride)
{
    if (!ctx->renderable && ctx->n) FAIL(GS_E_STATE, "context was fed worker matrices only (gs_push_matrices): it can sort but not render");
    GS_HIP(hipSetDevice(ctx->device));
    GsFrameUniforms u = u_in;
    if (host_rgba && !device_rgba) {
        const size_t row = (size_t)(u.x1 - u.x0) * 4, st = stride ? stride : row;
        if (st < row) FAIL(GS_E_BADARG, "stride %zu smaller than a row (%zu bytes)", st, row);
    }
    gs_ctx *L = ctx->lanes[ctx->cur];                           // the frame's lane: where its gs_sort ran
    // the completion word of this frame: the next word of the lane's ring (a gathered piece brings its own: the piece's trailer)
    if (!L->async_pending) L->status_base = L->status_seq;       // (nothing of the lane waits for a collection: the frames to come count from here)
    if (!u.status) { u.status = &L->ctl->status_ring[L->status_seq % GS_STATUS_RING]; L->status_seq++; }
    L->status_cur = u.status;
    L->stats.subtile = u.subtile;
    u.need_seed = L->need_seed_pending; L->need_seed_pending = 0;  // (a seed for the lane's need words travels with its next frame)
    bool async = (u.flags & GS_RENDER_ASYNC) && !(u.flags & GS_RENDER_COUNT_FRAGS);
    // A context that has not MEASURED its share yet (fresh, cleared, the share un-pinned) draws its first two-round frame synchronously
    // even when asked to queue it: the frames queued behind it then use the share it measured instead of the 25 % every context starts
    // with (at 20 M splats: 5 M positions in the first round of every frame until the first gs_sync).  The call returns with the frame
    // complete; its status needs no gs_sync.
    // (... at most two such frames in a row, like the cold sorts: a context whose blends never record a need -- counting renders, every tile
    // saturated by round 0 of a walked share -- would otherwise draw EVERY queued frame synchronously,
    // with a drain of every other lane each time: ADVICE r5)
    if (async && !ctx->share_measured && ctx->near_fixed_permille <= 0 && u.near_count != 0xFFFFFFFFu && ctx->n && ctx->cold_frames < 2u) {
        async = false; u.flags &= ~(uint32_t)GS_RENDER_ASYNC;
        ctx->cold_frames++;
        // ... and the lanes that exist get their per-frame buffers for this frame's size now (tile ranges, per-pixel state, row tables:
        // a dozen allocations each), while nothing is in flight, instead of one lane per frame over the next five.  (BEFORE the frame's
        // kernels go out, not under them: allocations made while the device works take longer than the wait they would hide -- the
        // call 0.93 ms instead of 0.66, tools/cold_probe.py)
        const auto dbg_t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < GS_MAX_LANES; i++) {
            gs_ctx *Li = ctx->lanes[i];
            if (!Li || Li == L || Li->scratch_cap < ctx->cap) continue;
            if (lane_drain(Li) != GS_OK) continue;
            (void)ensure_frame_buffers(Li, u, device_rgba == nullptr);
            if (const size_t e = (u.tiles_x <= 256 && u.tiles_y <= 256 && !ctx->bin_mode) ? gs_row_tables_entries(ctx->n, (uint32_t)u.tiles_y) : 0) (void)gs_row_tables_ensure(Li, e);
        }
        if (gs_debug_cold()) fprintf(stderr, "[gs] cold frame: the other lanes' buffers took %.0f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - dbg_t0).count());
    }
    if (async) {
        if (host_rgba && stride && stride < (size_t)(u.x1 - u.x0) * 4) FAIL(GS_E_BADARG, "stride %zu smaller than a row (%zu bytes)", stride, (size_t)(u.x1 - u.x0) * 4);
        L->async_pending = true; ctx->cur_async = true;
        log_render(L, u, device_rgba, host_rgba, stride);
        if (ctx->enqueue_threads && gs_rotates(ctx) && !ctx->user_stream) {
            GsLaneCmd c;
            c.type = 1; c.has_cutout = false; c.has_strip = false; c.u = u; c.device_rgba = device_rgba; c.host_rgba = host_rgba; c.stride = stride;
            if (lane_push(L, c) != GS_OK) FAIL(GS_E_OOM, "out of host memory");
            return GS_OK;
        }
        TRY(lane_rc(ctx, L, lane_drain(L)));
        return lane_rc(ctx, L, render_async_on_lane(L, u, device_rgba, host_rgba, stride));
    }
    const auto dbg_t1 = std::chrono::steady_clock::now();
    TRY(lane_rc(ctx, L, lane_drain(L)));                         // whatever its worker still had to enqueue comes first
    const auto dbg_t2 = std::chrono::steady_clock::now();
    const int rcs = lane_rc(ctx, L, render_sync_on_lane(L, u, device_rgba, host_rgba, stride));
    if (gs_debug_cold()) fprintf(stderr, "[gs] synchronous frame: drain %.0f us, draw %.0f us\n", std::chrono::duration<double, std::micro>(dbg_t2 - dbg_t1).count(),
                                         std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - dbg_t2).count());
    return rcs;
}

GS_API int gs_render(gs_ctx *ctx, const gs_render_params *p, uint8_t *rgba_out, size_t stride)
{
    CHECK_CTX(ctx);
    if (!rgba_out) FAIL(GS_E_BADARG, "gs_render: rgba_out is NULL");
    return render_common(ctx, p, nullptr, rgba_out, stride);
}

GS_API int gs_render_device(gs_ctx *ctx, const gs_render_params *p, void *device_rgba)
{
    CHECK_CTX(ctx);
    return render_common(ctx, p, device_rgba, nullptr, 0);
}

GS_API int gs_render_stereo(gs_ctx *ctx, const gs_render_params eyes[2], uint8_t *rgba_out[2], size_t stride)
{
    CHECK_CTX(ctx);
    if (!eyes || !rgba_out || !rgba_out[0] || !rgba_out[1]) FAIL(GS_E_BADARG, "gs_render_stereo: NULL argument");
    TRY(render_common(ctx, &eyes[0], nullptr, rgba_out[0], stride));
    return render_common(ctx, &eyes[1], nullptr, rgba_out[1], stride);
}

GS_API int gs_set_scene(gs_ctx *ctx, const float *depth, const uint8_t *rgba, int fb_width, int fb_height)
{
    CHECK_CTX(ctx);
    GS_HIP(hipSetDevice(ctx->device));
    TRY(drain_all(ctx));
    dev_free(ctx->scene_depth); dev_free(ctx->scene_rgba);
    ctx->scene_w = ctx->scene_h = 0;
    refresh_lanes(ctx);
    if (!depth && !rgba) return GS_OK;
    if (fb_width <= 0 || fb_height <= 0) FAIL(GS_E_BADARG, "gs_set_scene: bad size %dx%d", fb_width, fb_height);
    const size_t px = (size_t)fb_width * fb_height;
    if (depth) { TRY(dev_alloc(ctx, &ctx->scene_depth, px)); GS_HIP(hipMemcpy(ctx->scene_depth, depth, px * 4, hipMemcpyHostToDevice)); }
    if (rgba) { TRY(dev_alloc(ctx, &ctx->scene_rgba, px)); GS_HIP(hipMemcpy(ctx->scene_rgba, rgba, px * 4, hipMemcpyHostToDevice)); }
    ctx->scene_w = fb_width; ctx->scene_h = fb_height;
    refresh_lanes(ctx);
    return GS_OK;
}

GS_API void *gs_host_alloc(size_t nbytes)
{
    void *p = nullptr;
    if (!nbytes || hipHostMalloc(&p, nbytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

GS_API void gs_host_free(void *p) { if (p) (void)hipHostFree(p); }

// gs_sync found incomplete asynchronous frames on the lanes of bad_unit[]: draw the frames logged for those lanes again, in the
// order they were asked, synchronously (both binning rounds; a pair overflow grows the buffers and repeats inside
// render_sync_on_lane).  Every lane is drained and idle.  GS_E_RETRY when the library cannot decide alone (see GsFrameRec).
static int redraw_flagged_frames_impl(gs_ctx *ctx, const bool bad_unit[GS_MAX_PRIMARY]);
static int redraw_flagged_frames(gs_ctx *ctx, const bool bad_unit[GS_MAX_PRIMARY])
{
    // (the walk copies the log and keeps a set of outputs: out of host memory there, the caller draws the frames again itself)
    try { return redraw_flagged_frames_impl(ctx, bad_unit); } catch (...) { ctx->adapt_frozen = false; return GS_E_RETRY; }
}
static int redraw_flagged_frames_impl(gs_ctx *ctx, const bool bad_unit[GS_MAX_PRIMARY])
{
    // an output written by more than one logged frame (any lane): drawing one of them again could overwrite a newer image
    std::set<const void *> outs;
    for (int i = 0; i < GS_MAX_LANES; i++) {
        gs_ctx *L = ctx->lanes[i];
        if (!L || !L->log) continue;
        for (const GsFrameRec &r : L->log->recs)
            for (int k = 0; k < r.nrender; k++) {
                const void *o = r.r[k].host ? (const void *)r.r[k].host : (r.r[k].dev ? (const void *)r.r[k].dev : (const void *)L);   // (the lane's own framebuffer)
                if (o != (const void *)L && !outs.insert(o).second) return GS_E_RETRY;
            }
    }
    for (int i = 0; i < GS_MAX_LANES; i++) {
        gs_ctx *L = ctx->lanes[i];
        if (!L || !L->log || !bad_unit[i % GS_MAX_PRIMARY]) continue;
        for (const GsFrameRec &r : L->log->recs) if (r.nrender && r.undecidable) return GS_E_RETRY;
    }
    uint32_t redrawn = 0;
    struct Thaw { gs_ctx *c; ~Thaw() { c->adapt_frozen = false; } } thaw{ ctx };
    ctx->adapt_frozen = true;
    for (int i = 0; i < GS_MAX_LANES; i++) {
        gs_ctx *L = ctx->lanes[i];
        if (!L || !L->log || !bad_unit[i % GS_MAX_PRIMARY]) continue;
        const std::vector<GsFrameRec> recs = L->log->recs;          // (render_sync_on_lane does not log, but keep the walk independent of it)
        // which of them: the frames whose completion word (read back with the control block) is not 0 -- as long as the lane's ring still
        // holds the word of every render since the last collection; else, or for a render whose word is not in the ring, all of them
        const bool words_valid = L->status_seq - L->status_base <= GS_STATUS_RING;
        for (const GsFrameRec &r : recs) {
            if (!r.nrender) continue;
            bool complete = words_valid;
            for (int k = 0; k < r.nrender && complete; k++) {
                const uint32_t *w = r.r[k].u.status;
                if (!w || w < L->ctl->status_ring || w >= L->ctl->status_ring + GS_STATUS_RING) complete = false;
                else if (L->ctl_host->status_ring[w - L->ctl->status_ring] != 0u) complete = false;
            }
            if (complete) continue;
            TRY(lane_rc(ctx, L, gs_run_sort(L, r.view, r.has_cutout ? r.cutout : nullptr, r.has_strip ? &r.strip : nullptr, 0)));
            for (int k = 0; k < r.nrender; k++) {
                GsFrameUniforms u = r.r[k].u;
                u.flags &= ~(uint32_t)GS_RENDER_ASYNC;
                u.skip_round1 = 0;                                   // both rounds: complete by construction
                u.need_seed = 0;                                     // (the seed it carried has been written)
                TRY(lane_rc(ctx, L, render_sync_on_lane(L, u, r.r[k].dev, r.r[k].host, r.r[k].stride)));
                redrawn++;
            }
        }
    }
    ctx->stats.retried_frames += redrawn;
    return GS_OK;
}

GS_API int gs_sync(gs_ctx *ctx)
{
    CHECK_CTX(ctx);
    GS_HIP(hipSetDevice(ctx->device));
    bool any_missed = false, any_over = false, share_failed = false;
    uint32_t spec_failed = 0, need = 0, nfr = 0;
    const float frac_used = ctx->near_frac;                       // what every frame queued since the last collection was drawn with
    // whatever way this call ends, the frames logged so far are not drawn again by a LATER gs_sync (their output buffers may be
    // gone by then): a failure below leaves no records behind
    struct LogGuard { gs_ctx *c; ~LogGuard() { for (int i = 0; i < GS_MAX_LANES; i++) if (c->lanes[i]) log_reset(c->lanes[i]); } } log_guard{ ctx };
    bool bad_unit[GS_MAX_PRIMARY] = { false, false, false, false };   // a lane or its twin (one stream, one worker) reported an incomplete frame
    uint32_t want = 0;
    // everything is about to be drained: the first frame after this goes to a twin's slot, i.e. out at once and alone (an idle GPU
    // should not wait for a partner frame), the pairs start with the frame after it
    if (ctx->frame_batch == 2) ctx->rot &= ~1;
    // The control blocks of the lanes with asynchronous frames follow the frames' kernels on the lanes' own streams, ALL of them before
    // the first stream is waited for: six blocking copies one after the other (three lanes and their twins, 20-30 us each) were the
    // last 150 us of every gs_sync() -- a tenth of a region of twenty frames (tools/prof_lanes.py).
    for (int i = 0; i < GS_MAX_LANES; i++) {
        gs_ctx *L = ctx->lanes[i];
        if (L) TRY(lane_rc(ctx, L, lane_drain(L, true)));        // (every launch of every lane is in its stream)
    }
    for (int i = 0; i < GS_MAX_LANES; i++) {
        gs_ctx *L = ctx->lanes[i];
        if (L && L->async_pending) LANE_HIP(L, hipMemcpyAsync(L->ctl_host, L->ctl, sizeof(GsControl), hipMemcpyDeviceToHost, L->stream));
    }
    for (int i = 0; i < GS_MAX_LANES; i++) {
        gs_ctx *L = ctx->lanes[i];
        if (!L) continue;
        LANE_HIP(L, hipStreamSynchronize(L->stream));
        TRY(lane_rc(ctx, L, prof_drain(L)));
        if (!L->async_pending) continue;
        L->async_pending = false;
        const bool missed = L->ctl_host->round1_missed != 0;
        static const bool dbg_near = getenv("GS_DEBUG_NEAR") != nullptr;       // (what raised the share: printed per collected lane)
        if (dbg_near && (L->ctl_host->round1_missed || L->ctl_host->unsat_events != L->seen_unsat_events)) {
            const GsControl *c = L->ctl_host;
            fprintf(stderr, "[gs] sync lane %d: order_incomplete %u spec_fail %u (T %u chunk limit %u) near_overflow %u missed %u unsat_events %u (seen %u) V %u P %u V' %u near_sorted %u req %u frames %u two_rounds %d near_frac %.4f clean %u hold %u\n", i, c->order_incomplete, c->spec_fail, c->spec_dbg >> 16, c->spec_dbg & 0xFFFFu, c->near_overflow, c->round1_missed,
                    c->unsat_events, L->seen_unsat_events, c->n_kept, c->n_sorted, c->n_valid, c->near_sorted, L->sort_near_req, c->acc_frames, (int)L->last_two_rounds,
                    ctx->near_frac, ctx->clean_frames, ctx->skip_hold);
        }
        if (missed) LANE_HIP(L, hipMemsetAsync(&L->ctl->round1_missed, 0, sizeof(uint32_t), L->stream));
        if (L->ctl_host->order_incomplete) LANE_HIP(L, hipMemsetAsync(&L->ctl->order_incomplete, 0, sizeof(uint32_t), L->stream));
        bool over = false;
        TRY(collect_status(L, &over, &share_failed, &spec_failed, &need, &nfr));
        any_missed |= missed; any_over |= over;
        if (missed || over) bad_unit[i % GS_MAX_PRIMARY] = true;
        if (over && L->ctl_host->max_total > want) want = L->ctl_host->max_total;
    }
    if (share_failed) { if (ctx->share_measured) share_missed(ctx, frac_used); else share_raise(ctx, frac_used); }   // (once, whatever the number of lanes that saw it)
    else { share_from_need(ctx, need, nfr); TRY(seed_need_words(ctx, need)); }   // ... else what the collected frames' tiles needed (the maximum over the lanes)
    if (spec_failed) gs_spec_back_off(ctx, spec_failed == 2u);
    if (any_over)                                                // one retry for all lanes: each gets room for the largest demand seen
        for (int i = 0; i < GS_MAX_LANES; i++)
            if (ctx->lanes[i] && ctx->lanes[i]->pair_cap)
                TRY(lane_rc(ctx, ctx->lanes[i], gs_ensure_pair_capacity(ctx->lanes[i], (size_t)want + want / 4 + 1)));
    const bool stale = ctx->log_stale;
    ctx->log_stale = false;
    int rc = GS_OK;
    if (any_missed || any_over) {
        rc = (ctx->auto_retry && !stale) ? redraw_flagged_frames(ctx, bad_unit) : GS_E_RETRY;
        if (rc == GS_E_RETRY) {
            if (any_missed) snprintf(ctx->err, sizeof ctx->err, "an asynchronous frame skipped its second binning round but a tile had not saturated; "
                                     "the share of splats binned first was raised - render the frames since the previous gs_sync() again");
            else snprintf(ctx->err, sizeof ctx->err, "an asynchronous frame needed %u pairs and overflowed the pair buffers; they were enlarged - "
                          "render the frames since the previous gs_sync() again", want);
        }
    }
    return rc;                                                       // (log_guard resets the logs)
}

GS_API int gs_set_stream(gs_ctx *ctx, void *hip_stream)
{
    CHECK_CTX(ctx);
    GS_HIP(hipSetDevice(ctx->device));
    TRY(drain_all(ctx));
    if (ctx->own_stream) { (void)hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
    if (hip_stream) ctx->stream = (hipStream_t)hip_stream;
    else { GS_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->own_stream = true; }
    // on a caller-owned stream every frame is ordered with the caller's own work on it: no lane rotation
    ctx->user_stream = hip_stream != nullptr;
    ctx->cur = 0; ctx->rot = 0; ctx->cur_async = false;
    return GS_OK;
}

GS_API int gs_wait_stream(gs_ctx *ctx, void *hip_stream)
{
    CHECK_CTX(ctx);
    GS_HIP(hipSetDevice(ctx->device));
    gs_ctx *L = nullptr;
    TRY(get_lane(ctx, next_frame_lane(ctx), &L));               // the lane the next gs_sort() will use
    GS_HIP(hipEventRecord(L->ev_gate, (hipStream_t)hip_stream));
    GS_HIP(hipStreamWaitEvent(L->stream, L->ev_gate, 0));
    return GS_OK;
}

GS_API int gs_frame_lane(gs_ctx *ctx) { return ctx ? ctx->cur : -1; }

GS_API void *gs_lane_stream(gs_ctx *ctx, int lane)
{
    if (!ctx || lane < 0 || lane >= GS_MAX_LANES || !ctx->lanes[lane]) return nullptr;
    (void)lane_drain(ctx->lanes[lane]);                          // everything handed to the lane so far is in its stream
    return (void *)ctx->lanes[lane]->stream;
}

GS_API int gs_stream_wait_frame(gs_ctx *ctx, void *hip_stream)
{
    CHECK_CTX(ctx);
    GS_HIP(hipSetDevice(ctx->device));
    gs_ctx *L = ctx->lanes[ctx->cur];
    TRY(lane_rc(ctx, L, lane_drain(L)));
    GS_HIP(hipEventRecord(L->ev_frame, L->stream));
    GS_HIP(hipStreamWaitEvent((hipStream_t)hip_stream, L->ev_frame, 0));
    return GS_OK;
}

GS_API int gs_frame_status_device(gs_ctx *ctx, void **device_word)
{
    CHECK_CTX(ctx);
    if (!device_word) FAIL(GS_E_BADARG, "gs_frame_status_device: device_word is NULL");
    gs_ctx *L = ctx->lanes[ctx->cur];
    TRY(lane_rc(ctx, L, lane_drain(L)));                         // (the frame's kernels are in the stream: the word is theirs from here on)
    // (the word the frame's last render was given; on the root of a gathered frame: where k_assemble ORs the pieces' words)
    *device_word = (L->status_cur && L->status_cur >= L->ctl->status_ring && L->status_cur < L->ctl->status_ring + GS_STATUS_RING) ? (void *)L->status_cur
                                                                                                                          : (void *)&L->ctl->frame_status;
    return GS_OK;
}

GS_API void *gs_frame_stream(gs_ctx *ctx)
{
    if (!ctx) return nullptr;
    (void)lane_drain(ctx->lanes[ctx->cur]);                      // the frame's kernels are in the stream when this returns
    return (void *)ctx->lanes[ctx->cur]->stream;
}

GS_API int gs_set_option(gs_ctx *ctx, int option, int64_t value)
{
    CHECK_CTX(ctx);
    switch (option) {
    case GS_OPT_PROFILE:
        GS_HIP(hipSetDevice(ctx->device));
        for (int i = 0; i < GS_MAX_LANES; i++)
            if (ctx->lanes[i]) TRY(lane_rc(ctx, ctx->lanes[i], set_profile(ctx->lanes[i], value != 0, value == 2 || value == 3, value == 3 ? 4u : 1u)));
        return GS_OK;
    case GS_OPT_NEAR_PERMILLE:
        if (value < 0 || value > 1000) FAIL(GS_E_BADARG, "near permille must be 0 (adaptive) .. 1000 (single round)");
        ctx->near_fixed_permille = (int)value;
        if (value == 0) {
            // adapt from scratch: the default share, nothing measured -- on the device too (each lane's next frame clears its need words:
            // a word that still says "a tile no share saturates" from frames long gone would keep the share at 100 % for 64 collections)
            ctx->near_frac = 0.25f; ctx->share_measured = false; ctx->need_margin = 0.0f; ctx->cold_sorts = 0; ctx->cold_frames = 0; memset(ctx->need_hist, 0, sizeof ctx->need_hist);
            for (int i = 0; i < GS_MAX_LANES; i++) if (ctx->lanes[i]) { ctx->lanes[i]->need_seed_pending = 1u; ctx->lanes[i]->need_word_est = 0; }
        }
        return GS_OK;
    case GS_OPT_RECORD_STAGED: ctx->record_staged = value == 2 ? 2u : (value != 0 ? 1u : 0u); return GS_OK;
    case GS_OPT_TERMINATION:
        if (value < 2) FAIL(GS_E_BADARG, "termination 1/eps must be >= 2");
        ctx->t_eps = 1.0f / (float)value; return GS_OK;
    case GS_OPT_WIDE_PAIRS:
        GS_HIP(hipSetDevice(ctx->device));
        TRY(drain_all(ctx));
        if (value < 0 || value > 1) FAIL(GS_E_BADARG, "wide records: 0 (automatic) or 1 (the depth sort's general 8-byte records whatever the number of splats)");
        ctx->wide_pairs = value == 1; refresh_lanes(ctx);
        return GS_OK;
    case GS_OPT_BINNING:
        GS_HIP(hipSetDevice(ctx->device));
        TRY(drain_all(ctx));
        if (value < 0 || value > 1) FAIL(GS_E_BADARG, "binning: 0 (span lists wherever the strip fits) or 1 (pair records and two stable radix passes)");
        ctx->bin_mode = (int)value;
        return GS_OK;
    case GS_OPT_ENQUEUE_THREADS:
        GS_HIP(hipSetDevice(ctx->device));
        TRY(drain_all(ctx));
        ctx->enqueue_threads = value != 0;
        return GS_OK;
    case GS_OPT_BLEND_SPLIT:
        if (value < 0 || value > 0x7FFFFFFF) FAIL(GS_E_BADARG, "blend split: 0 (off) or the list length from which a tile is split");
        ctx->blend_split_min = (uint32_t)value;
        return GS_OK;
    case GS_OPT_COMM_SELF_COPY:
        GS_HIP(hipSetDevice(ctx->device));
        TRY(drain_all(ctx));
        return gs_comm_set_self_copy(ctx, value != 0);
    case GS_OPT_SORT_SHARE:
        if (value < 0 || value > 1000) FAIL(GS_E_BADARG, "sort share: 0 (off) or the permille of the splats whose order the ranks exchange");
        GS_HIP(hipSetDevice(ctx->device));
        TRY(drain_all(ctx));
        ctx->sort_share_permille = (int)value;
        return GS_OK;
    case GS_OPT_AUTO_RETRY:
        GS_HIP(hipSetDevice(ctx->device));
        TRY(drain_all(ctx));
        ctx->auto_retry = value != 0;
        return GS_OK;
    case GS_OPT_COMM_TRANSPORT:
        if (value != 0 && value != 1) FAIL(GS_E_BADARG, "comm transport: 0 (RCCL) or 1 (in-process)");
        return gs_comm_set_transport(ctx, (int)value);
    case GS_OPT_PIPELINE_DEPTH:
        if (value < 1 || value > GS_MAX_PRIMARY) FAIL(GS_E_BADARG, "pipeline depth must be 1..%d", GS_MAX_PRIMARY);
        GS_HIP(hipSetDevice(ctx->device));
        TRY(drain_all(ctx));
        ctx->pipe_depth = (int)value;
        ctx->cur = 0; ctx->rot = 0; ctx->cur_async = false;
        return prepare_lanes(ctx);
    case GS_OPT_FRAME_BATCH:
        if (value != 1 && value != 2) FAIL(GS_E_BADARG, "frame batch must be 1 (off) or 2");
        GS_HIP(hipSetDevice(ctx->device));
        TRY(drain_all(ctx));
        ctx->frame_batch = (int)value;
        ctx->cur = 0; ctx->rot = 0; ctx->cur_async = false;
        return prepare_lanes(ctx);
    case GS_OPT_SUBTILE:
        if (value < 0 || value > 2) FAIL(GS_E_BADARG, "sub-tile lists: 0 (off), 1 (where splats are small) or 2 (always)");
        ctx->subtile_opt = (int)value;
        return GS_OK;
    case GS_OPT_SORT_NEAR:
        if (value < 0 || value > 2) FAIL(GS_E_BADARG, "near-only sorts: 0 (off), 1 (scenes of 4 M splats and more) or 2 (always)");
        GS_HIP(hipSetDevice(ctx->device));
        TRY(drain_all(ctx));
        ctx->sort_near_opt = (int)value;
        return GS_OK;
    default: FAIL(GS_E_BADARG, "unknown option %d", option);
    }
}

GS_API int gs_get_stats(gs_ctx *ctx, gs_stats *out)
{
    CHECK_CTX(ctx);
    if (!out) FAIL(GS_E_BADARG, "gs_get_stats: out is NULL");
    // per-frame figures: the current frame's lane; accumulators: summed over the lanes
    for (int i = 0; i < GS_MAX_LANES; i++) if (ctx->lanes[i]) (void)lane_drain(ctx->lanes[i]);
    gs_stats s = ctx->lanes[ctx->cur]->stats;
    s.prof_frames = 0; s.sum_ms_sort = s.sum_ms_project = s.sum_ms_bin = s.sum_ms_blend = 0;
    s.acc_frames = 0; s.acc_sorted = s.acc_visible = s.acc_pairs = 0;
    for (int i = 0; i < GS_MAX_LANES; i++) {
        const gs_ctx *L = ctx->lanes[i];
        if (!L) continue;
        s.prof_frames += L->stats.prof_frames;
        s.sum_ms_sort += L->stats.sum_ms_sort; s.sum_ms_project += L->stats.sum_ms_project;
        s.sum_ms_bin += L->stats.sum_ms_bin; s.sum_ms_blend += L->stats.sum_ms_blend;
        s.acc_frames += L->stats.acc_frames; s.acc_sorted += L->stats.acc_sorted;
        s.acc_visible += L->stats.acc_visible; s.acc_pairs += L->stats.acc_pairs;
    }
    s.n_splats = ctx->n;
    s.retried_frames = ctx->stats.retried_frames; s.spec_sorts = ctx->stats.spec_sorts; s.spec_misses = ctx->stats.spec_misses; s.need_splats = ctx->stats.need_splats;
    s.near_permille = (uint32_t)(ctx->near_frac * 1000.0f + 0.5f);
    *out = s;
    return GS_OK;
}

GS_API int gs_download(gs_ctx *ctx, int which, void *out, size_t nbytes)
{
    CHECK_CTX(ctx);
    if (!out) FAIL(GS_E_BADARG, "gs_download: out is NULL");
    GS_HIP(hipSetDevice(ctx->device));
    gs_ctx *L = ctx->lanes[ctx->cur];                           // per-frame buffers: the current frame's lane
    TRY(lane_rc(ctx, L, lane_drain(L)));
    GS_HIP(hipStreamSynchronize(L->stream));
    const void *src = nullptr; size_t have = 0;
    const size_t V = L->stats.n_sorted;
    if (which == GS_BUF_CENTER_SCALE || which == GS_BUF_COV_COLOR) {     // de-interleave the 32-byte splat records
        if (!ctx->renderable && ctx->n) FAIL(GS_E_STATE, "context holds worker rows only");
        if (nbytes > ctx->n * 16 || nbytes % 16) FAIL(GS_E_BADARG, "buffer %d holds %zu bytes, %zu requested", which, ctx->n * 16, nbytes);
        if (nbytes) GS_HIP(hipMemcpy2D(out, 16, (const char *)ctx->splat + (which == GS_BUF_COV_COLOR ? 16 : 0), 32, 16, nbytes / 16,
                                        hipMemcpyDeviceToHost));
        return GS_OK;
    }
    switch (which) {
    case GS_BUF_SORT_ROWS: src = ctx->sort_rows; have = ctx->n * 16; break;
    case GS_BUF_SORTED:
        if (L->have_sort && L->sort_near_req) { TRY(lane_rc(ctx, L, ensure_full_sort(L))); GS_HIP(hipStreamSynchronize(L->stream)); }
        src = L->sorted; have = L->have_sort ? V * 4 : 0; break;
    case GS_BUF_PROJECTED: src = L->proj; have = L->have_sort ? V * 32 : 0; break;
    case GS_BUF_TILE_COUNT: src = L->tile_count; have = L->have_sort ? V * 4 : 0; break;
    case GS_BUF_TILE_STATS: src = L->tile_range; have = L->tile_cap * 8; break;
    case GS_BUF_UNSAT_MASK: src = L->unsat_mask; have = L->mask_cap * 4; break;
    default: FAIL(GS_E_BADARG, "unknown buffer %d", which);
    }
    if (nbytes > have) FAIL(GS_E_BADARG, "buffer %d holds %zu bytes, %zu requested", which, have, nbytes);
    if (nbytes && (!ctx->renderable && (which == GS_BUF_CENTER_SCALE || which == GS_BUF_COV_COLOR)))
        FAIL(GS_E_STATE, "context holds worker rows only");
    if (nbytes) GS_HIP(hipMemcpy(out, src, nbytes, hipMemcpyDeviceToHost));
    return GS_OK;
}

}  // extern "C"

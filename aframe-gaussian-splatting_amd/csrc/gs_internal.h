// gs_internal.h -- context layout and kernel-launcher declarations shared by the .hip files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/gs_splat.h"
#include "gs_device_math.h"

// The kernels are written for gfx950 (CDNA4) and nothing else: wave64 ballots, packed-fp32 with op_sel, v_min_f64 / v_max_f64 and
// s_waitcnt forms are written out as gfx9 inline assembly.  Another --offload-arch fails HERE, with a sentence, not in the assembler.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libgs_splat_hip is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

// ---------------------------------------------------------------- geometry of the work decomposition
#define GS_TILE 16                 // screen tile edge (pixels): 16x16 = one 256-thread workgroup
#define GS_BLOCK 256               // threads per workgroup everywhere (4 wavefronts of 64)
#define GS_CHUNK_S 2048            // items per radix chunk = histogram row: short inputs (256-thread workgroups) ...
#define GS_CHUNK_L 4096            // ... and long ones (512-thread workgroups); 8 items per thread in both
#ifndef GS_RADIX_LARGE_N
#define GS_RADIX_LARGE_N (3u << 20) // inputs expected to be longer than this take the long geometry
#endif
#define GS_RADIX_MAX_BINS 512      // up to 9-bit digits (depth key = 17 bits = 8 + 9)
#define GS_TOUCH_PRIVATE_WORDS 16u   /* private dwords per lane a new lane's stream is made to allocate (gs_api.hip: k_touch_private) */
#define GS_STATUS_RING 64u             /* completion words per lane (GsControl::status_ring) */
#define GS_NEED_WORDS 32u          // GsControl::need_near
#define GS_MSD_GROUP 32u           // the MSD depth sort (gs_sort.hip): radix chunks per group row
#define GS_MSD_MAX_N (1u << 21)    // ... takes sorts of at most this many splats.  Its records (low bucket byte << 24 | index) would hold 2^24, but a chunk's
                                   // offsets are summed from ~sqrt(chunks) x 2 rows by the scatter itself: 46 at 1 M splats (two round trips under the
                                   // ranking), 79 at 6 M (five, in the open: C3's sort went from 129 to 175 us per frame); longer sorts keep the two LSD
                                   // passes with their scan launches -- they are not launch-bound
#define GS_PROF_RING 256           // frames of HIP-event timings kept in flight
#define GS_PROF_EVENTS 7
#define GS_MAX_PART 8192           // upper bound on the grid of any kernel that writes per-workgroup partials
#define GS_CULLED_KEY 65536u       // depth-sort key of culled / dropped splats (sorts behind every bucket)
#define GS_MAX_PRIMARY 4           // lanes with a stream and an enqueue thread of their own (GS_OPT_PIPELINE_DEPTH)
#define GS_MAX_LANES 8             // ... + their twins (GS_OPT_FRAME_BATCH): lanes[GS_MAX_PRIMARY + i] shares stream and worker of lanes[i]
#ifndef GS_EMIT_PAIRS
#define GS_EMIT_PAIRS 2048u        // pair slots written per k_emit work item (a slice of one chunk's pairs).  Every slice re-reads its chunk's 256 x 44 B:
                                   // 2048 instead of round 2's 1024 halves those re-reads (+2 % frames/s pipelined, -0.7 % for a frame alone; 4096: -4 % alone)
#endif
#ifndef GS_EMIT_RUNS
#define GS_EMIT_RUNS 512u          // tile-row runs expanded per pass inside an item
#endif

// strip: sort only the splats that can reach columns [x0, x1) of the frame these uniforms draw (gs_sort_for); nullptr = all
struct GsSortStrip { float mv[16], proj[16]; float focal, vw, vh; int32_t x0, x1; };   // (vh > 0: the rows [0, vh) are tested too)
#define GS_DEPTH_BINS 2048u        // depth histogram of a near-only sort: sign-less f32 bits >> 20 (exponent + 3 mantissa bits)
#define GS_DH_COPIES 8u            // ... kept in this many copies (workgroup % copies)
#define GS_DEPTH_COARSE (GS_DEPTH_BINS / 32u)   // ... with sums over 32 consecutive bins behind the copies
#define GS_DH_WORDS (GS_DH_COPIES * (GS_DEPTH_BINS + GS_DEPTH_COARSE))
#ifndef GS_NEAR_STASH
#define GS_NEAR_STASH 512u         // survivors a 4096-item chunk of a near-only sort may stash (one eighth; the share asked for is <= 1/32)
#endif

// Device-resident control block: every data-dependent count lives here so that no stage needs a
// host round trip; kernels read their problem size from it (grid-stride over chunks).
struct GsControl {
    unsigned long long min_enc;    // ordered-u64 encoding of min kept depth (f64)
    unsigned long long max_enc;    // ... max
    unsigned long long n_frags;    // fragment counter (GS_RENDER_COUNT_FRAGS)
    uint32_t n_total;              // N at the time of the sort
    uint32_t n_kept;               // V : survivors of the sort culls  (= reference validCount)
    uint32_t n_sorted;             // records that went through the depth sort: V' = those with a bucket inside the table (compact records: the rest is
                                   // the zero tail), or only the nearest P of them (near-only sort)
    uint32_t near_sorted;          // 1 / 2 / 3 (2: through the depth pass' own candidate stash; 3: a tail sort, cut at a segment boundary by k_msd_scatter): `sorted` holds the order's last P valid positions [V' - P, V') only (gs_run_sort with near_req)
    uint32_t n_valid;              // V' of the whole order, counted by a near-only sort (position of sorted[0] = n_valid - n_sorted)
    uint32_t order_incomplete;     // sticky: `sorted` does not hold everything it claims -- a near-only sort's chunk stash overflowed, an exchanged
                                   // order was cut short (host clears).  The frames drawn from it are flagged round1_missed too (asynchronous frames:
                                   // drawn again from a whole sort by gs_sync); a synchronous frame sorts in full and draws BOTH rounds again
    uint32_t n_visible;            // Vp: splats that pass the vertex-shader culls
    uint32_t n_pairs;              // I : (tile, splat) pairs
    uint32_t pair_overflow;        // set when I exceeded the pair capacity (pairs clamped to 0)
    uint32_t scan_total;           // scratch: total of the last scan
    uint32_t overflow_sticky;      // like pair_overflow but only ever cleared by the host (asynchronous frames)
    uint32_t max_total;            // largest per-FRAME pair demand (sum over rounds) seen since the host last cleared it
    uint32_t j_lo, j_hi;           // sorted-position range of the current binning round
    uint32_t unsat_count;          // tiles left unsaturated by round 0 (counted by its blend)
    uint32_t unsat_round0;         // copy taken when round 1 begins (what the host adapts near_count on)
    uint32_t n_pairs_frame;        // I summed over the rounds of the frame
    uint32_t want_frame;           // pair demand of the frame so far (counts rounds that overflowed, too)
    uint32_t unsat_events;         // frames whose round 0 left tiles unsaturated (monotonic)
    uint32_t n_emit_extra;         // k_emit work items beyond one per chunk (k_pairs_check -> k_emit of the same round)
    uint32_t n_runs;               // span-list binning: tile-row runs of the frame (k_lists; the host's hint for k_seg_count)
    uint32_t round1_missed;        // sticky: round 1 was skipped optimistically but a tile needed it (host clears)
    uint32_t vis_total;            // visible splats of the current binning round (k_pairs_check)
    uint32_t near_overflow;        // sticky: a near-only sort's survivors did not fit a chunk's stash (host clears; the frame is also
                                   // flagged order_incomplete + round1_missed: it is drawn again from a whole sort)
    uint32_t spec_fail;            // sticky: a near-only sort whose depth pass stashed the candidates itself (k_sort_depth<.., SPEC>) could not vouch for
                                   // them -- 1: the threshold hint was behind (transient), 2: a stash overflowed / the depth range does not suit
                                   // the path (the context stops using it); the frame is flagged order_incomplete + round1_missed (host clears)
    uint32_t spec_dbg;             // (GS_DEBUG_NEAR) exact threshold bin << 16 | the limit of a chunk that failed the check
    uint32_t near_bin_hint;        // OWNER's block only: the threshold depth bin the context's last near-only sort found (any lane's kernels write it)
    uint32_t frame_status;         // the completion word of the lane's last frame (GsFrameUniforms::status points here unless the frame is a gathered piece;
                                   // on the root of a gathered frame: the OR of all its pieces' words, written by k_assemble)
    // What the share of splats binned first has to be (round 5: measured, not walked).  A tile's blend knows how far into its list it
    // read before its 256 pixels were saturated; the sorted position of that entry says how many of the NEAREST splats had to be binned
    // for the tile: V - position.  need_near[tile % GS_NEED_WORDS] = max over the tiles of the frames drawn since the host last cleared
    // it (one fire-and-forget atomicMax per tile, spread over 32 words: a single word serialises at ~11 ns per tile); 0xFFFFFFFF: a tile
    // was not saturated by the whole order (sky: no share helps it).  The host sets near_count from the maximum (gs_api.hip).
    uint32_t acc_frames;           // frames rendered since profiling was switched on
    unsigned long long acc_sorted, acc_visible, acc_pairs;   // sums of V, Vp, I over those frames
    // (a cache line of its own: atomics drop the line from the XCDs' L2s, and the counts above are read by every kernel.  A tile's wave
    // reads its word when it starts and issues the atomic only if its need is larger: a dozen atomics per frame instead of 8160 --
    // 8160 atomics on ONE LINE serialise at ~11 ns each whatever the word: the blend went from 43 to 64 us)
    alignas(128) uint32_t need_near[32];
    // The completion words of the lane's last GS_STATUS_RING renders (GsFrameUniforms::status points at one of them unless the frame is a
    // gathered piece): render k of the lane since its last collection owns word k % GS_STATUS_RING.  gs_sync() reads them with the rest of
    // the block and draws again exactly the logged frames whose word is not 0 -- while a lane has queued no more renders than the ring
    // holds; beyond that every logged frame of a flagged lane is drawn again, as before round 5.
    alignas(128) uint32_t status_ring[GS_STATUS_RING];
};

struct GsFrameUniforms {           // per-render constants, passed by value to kernels
    float mv[16];
    float proj[16];
    float focal, vw, vh;
    int32_t W, H;                  // full viewport
    int32_t x0, x1;                // strip
    int32_t out_pitch;             // pixels per row of the output image (x1 - x0: a tight strip)
    int32_t x1b;                   // x1 rounded up to a multiple of 4 pixels from x0, clipped to W: what is binned and blended (x1: written)
    int32_t tiles_x, tiles_y;      // tile grid of the strip (origin at pixel x0, row 0 = top)
    float bg[4];
    float t_eps;                   // early-out threshold on transmittance
    uint32_t flags;                // GS_RENDER_* of the call | GS_FRAME_* (internal)
    uint32_t record_staged;        // GS_OPT_RECORD_STAGED: blend overwrites the tile-range table with (staged, length)
    uint32_t near_count;           // round 0 bins the nearest near_count splats; 0xFFFFFFFF = single round (everything)
    uint32_t mask_words;           // 32-bit words per tile row of the unsaturated-tile mask
    uint32_t skip_round1;          // round 1 is not launched for this frame (optimistic; blend<0> raises round1_missed)
    uint32_t has_depth, has_scene_rgba;   // scene compositing inputs present (gs_set_scene)
    uint32_t split_min;            // GS_OPT_BLEND_SPLIT: tiles whose list has at least this many entries are blended by GS_SPLIT_WAVES
                                   // wavefronts (k_blend<.., GS_SPLIT_WAVES>), the others by one; 0 = all by one
    uint32_t need_seed;            // != 0: the frame's projection first sets the lane's need_near words to this (1: to zero) -- how the host seeds them
                                   // after a collection: a hipMemsetD32Async per lane cost gs_sync() 20 us of host time each
    uint32_t *status;              // the frame's completion word (gs_frame_status_device): 0 = complete; bit 0: the second binning round was skipped and a
                                   // tile was not saturated, bit 1: the pair buffers overflowed, bit 2: drawn from an incomplete order -- the frame is drawn
                                   // again at gs_sync().  Written by the frame's own kernels (k_project<0> resets it, the blend raises the bits); the
                                   // lane's control block by default, the trailer of the piece for a gathered frame (it travels with the piece)
    uint32_t rc_stride;            // span-list binning (GS_OPT_BINNING): chunks per tile row of the row-count table; 0 = pair records + radix passes.
                                   // A tile's list entries are then the sorted positions themselves, else (tile, position) records
    uint32_t subtile;              // GS_OPT_SUBTILE: k_blend splits a staged batch into the lists of the tile's sixteen 4x4-pixel blocks where that
                                   // shortens the walk (gs_render.hip: same pixels either way)
};

struct GsLaneWorker;
struct GsFrameLog;                 // gs_api.hip: what was asked of a lane since the last gs_sync (transparent re-rendering)
struct gs_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    char err[512];

    // Frame pipelining.  A context owns up to GS_MAX_LANES "lanes": lane 0 is the context itself, the others are sibling
    // gs_ctx records with their OWN stream, per-frame scratch, control block and profiling ring, whose resident arrays
    // (splat, sort_rows, pow10tab, scene_*) alias the owner's.  Asynchronous frames rotate over the lanes, so the kernel
    // chain of frame k+1 runs under the tail of frame k.
    gs_ctx *parent;                // non-null for lanes 1..: the owning context
    gs_ctx *lanes[GS_MAX_LANES];   // owner only; [0] = this
    gs_ctx *exec;                  // the lane whose stream and enqueue thread run this lane's frames: itself, or -- a twin -- its primary
    gs_ctx *twin;                  // primary lanes: the twin, once created
    int frame_batch;               // owner: GS_OPT_FRAME_BATCH (1 = off, 2 = consecutive asynchronous frames share their launches)
    int rot;                       // owner: position in the rotation over (lane, twin) slots while batching
    int inflight;                  // commands handed to the enqueue thread for THIS lane and not yet executed (under the worker's mutex)
    int pipe_depth;                // GS_OPT_PIPELINE_DEPTH (1 = no rotation)
    int cur;                       // lane of the current frame (chosen by the last gs_sort)
    bool cur_async;                // the current frame was rendered with GS_RENDER_ASYNC: the next gs_sort moves on
    bool user_stream;              // gs_set_stream gave lane 0 a caller-owned stream: no rotation
    size_t scratch_cap;            // splats the per-frame scratch of THIS lane is sized for
    hipEvent_t ev_frame, ev_gate;  // gs_stream_wait_frame / gs_wait_stream
    struct GsLaneWorker *worker;   // enqueue thread of this lane (GS_OPT_ENQUEUE_THREADS), created on first use
    bool enqueue_threads;          // owner: asynchronous frames are enqueued by the lanes' worker threads

    // resident splat data (append-only; capacity doubles)
    size_t n, cap;
    bool renderable;               // false once matrices-only rows were pushed
    // the reference's two data textures, interleaved into ONE 32-byte record per splat so that the projection's
    // gather by sorted index touches one cache line per splat instead of two:
    //   [0] float4 (cx, cy, -z, max|Sigma|/32767)   centerAndScaleData  index.js:378-382
    //   [1] uint4  (6 x int16 Sigma, RGBA8)         covAndColorData     index.js:384-394
    uint4 *splat;                  // N x 2 x 16 B
    float4 *sort_rows;             // N x worker-row elements 12..15                 index.js:396-401
    float *bound_r;                // N x upper bound of the splat's largest standard deviation in object space: sqrt(3 max|Sigma_ij|)
                                   // (Gershgorin), +inf where unknown -- what gs_sort_for's strip test needs, 4 B per splat
    double *pow10tab;              // parseInt table (gs_host_tables.h)

    // sort scratch (sized by cap)
    float *depth;                  // stored f32 depth or +inf for culled
    uint32_t *key_a, *val_a;       // bucket keys in, sorted indices out
    uint2 *kv_b;                   // (key,val) records between the two passes
    uint32_t *sorted;              // alias of the buffer holding the final order
    uint32_t sorted_n_host;        // V as last read back (only when the caller asked for it)
    bool have_sort;
    uint32_t sort_gen;             // lane: sorts run on this lane's scratch so far (gs_sort_poll: was the posted sort's result overwritten -- a redraw by gs_sync?)
    // near-only sorts (GS_OPT_SORT_NEAR): histograms of the kept depths (GS_DH_COPIES x GS_DEPTH_BINS words; two buffers: a sort fills
    // one and clears the one the previous sort filled), the request the last sort of this lane ran with (0 = whole order) and its arguments
    uint32_t *dhist[2]; int dh_next; uint32_t *dh_dirty;
    uint32_t sort_near_req;
    bool no_tail_sort;             // lane: this sort's near-only form must hold AT MOST ~2 x near_req records (the shared sort's exchange buffer): the histogram form, not a tail sort
    uint32_t status_seq, status_base; // lane: renders handed to the lane so far (which word of the ring the next one gets) / its value when the first frame of the collection under way was queued
    uint32_t *status_cur;          // lane: the word of the render handed over last (gs_frame_status_device)
    uint32_t cold_sorts;           // owner: sorts run on the caller's thread because the share had not been measured yet (at most two in a row)
    uint32_t share_kind;           // owner: what kind of order the share was measured on -- 1: whole orders (gs_sort), 2: strips' orders (gs_sort_for);
                                   // 0: nothing yet.  A sort of another kind starts the measurement afresh (positions of a strip's order are not positions of the whole)
    uint32_t cold_frames;          // owner: queued frames drawn synchronously for the same reason (at most two in a row: a context whose frames
                                   // never measure -- counting renders -- keeps its pipelining)
    float sv_view[4], sv_cutout[16]; bool sv_has_cutout, sv_has_strip; GsSortStrip sv_strip;
    int sort_near_opt;             // owner: GS_OPT_SORT_NEAR
    bool near_stash_off;           // owner: a chunk's stash overflowed once: near-only sorts keep to the two whole-length passes
    bool near_spec;                // owner: a near-only sort has been collected (the threshold-bin hint exists)
    uint32_t near_spec_hold;       // owner: collections of near-only frames (one per lane and gs_sync) still to come before the speculative stash is tried again (0 = in use).  A sort
                                   // that overflowed its candidate stash or a run of misses sets it to near_spec_backoff, which doubles each time
                                   // (512 ... 65536): a scene the path does not suit pays one redraw in ever more frames, a camera that only
                                   // passed through such a view gets the path back
    uint32_t near_spec_backoff, near_spec_miss_credit;
    int near_spec_opt;             // owner: 0 = never stash speculatively (GS_SPEC_STASH=0 in the environment: A/B)
    uint32_t last_kept;            // owner: V of the last collected frame (a near-only sort pays only where V is well above the share read)

    // radix / scan scratch
    uint32_t *hist;  size_t hist_cap;       // digit-histogram rows H[radix chunks][bins], scanned in place
    uint32_t *radix_aux; size_t aux_cap;    // digit totals [GS_RADIX_MAX_BINS]
    uint32_t *spine; size_t spine_cap;      // per-256-splat totals of tiles touched (project -> emit)
    uint32_t *msd_grp; size_t msd_grp_cap;  // MSD depth sort: H[group of GS_MSD_GROUP radix chunks][bucket >> 8], words
    uint32_t *msd_tab;                      // ... and k_seg_sort's work items (written by k_msd_scatter), behind the group rows in the same allocation

    // render scratch
    gsm::Projected *proj;          // V records, sorted order
    uint2 *rect;                   // V x (tx0 | ty0<<16, tx1 | ty1<<16), strip-local tile coords
    uint32_t *tile_count;          // V
    uint2 *emit_extra;             // pair_cap / GS_EMIT_PAIRS + 2 (chunk, slice) items of the chunks with more than GS_EMIT_PAIRS pairs
    float *zwin;                   // V window depth of each sorted splat (written only while a scene depth buffer is set)
    float *scene_depth; uint32_t *scene_rgba; int scene_w, scene_h;   // gs_set_scene
    uint2 *pair_a, *pair_b; size_t pair_cap;   // (tile id, sorted position) records, ping-pong
    uint32_t blend_split_min;               // owner: GS_OPT_BLEND_SPLIT
    uint32_t pair_hint;                     // owner: pairs a frame is expected to bin (1.25 x the last collected frame's; 0 = unknown)
    uint32_t run_hint;                      // owner: tile-row runs of the last collected frame (span-list binning; 0 = unknown)
    uint32_t *row_cnt; size_t row_cnt_cap;  // span-list binning: [tile row][256-splat chunk] runs | tiles << 9 (k_project) -> runs before the chunk (k_row_scan)
    uint2 *row_tot; size_t row_tot_cap;     // ... and per tile row (runs, tiles) of the round
    int *seg_diff;                          // ... and per k_lists item (tile row, segment of its runs) the segment's difference array over the tile columns
    int bin_mode;                           // owner: GS_OPT_BINNING
    int subtile_opt;                        // owner: GS_OPT_SUBTILE (0 off, 1 where the last collected frames' splats were small, 2 always)
    uint32_t last_pairs, last_visible;      // owner: I and Vp of the last collected frame (what GS_OPT_SUBTILE = 1 decides on)
    uint2 *tile_range; size_t tile_cap;     // per tile [start,end) into the sorted pair list
    uint8_t *fb; size_t fb_cap;             // RGBA8 strip
    // multi-GPU frames (gs_comm.hip)
    struct GsComm *comm;                    // owner: the communicator this context joined (gs_comm_init)
    uint8_t *gstage; size_t gstage_cap;     // lane: the pieces of a gathered frame, tight rows each (its own; on the root everybody's)
    uint8_t *gframe[2]; size_t gframe_cap[2];  // lane, root only: the assembled row-major image(s) of the last gathered frame
    int gviews, gw[2], gh[2];               // ... and what they hold
    float4 *state; size_t state_cap;        // per tile 64 lanes x 4 float4: (T, r, g, b) of each lane's 4 pixels, round 0 -> 1
    uint32_t *unsat_mask; size_t mask_cap;  // one bit per tile (rows of mask_words words): left unsaturated by round 0
    float near_frac;                        // round 0 covers the nearest near_frac * N splats (adapted from unsat_round0)
    int near_fixed_permille;                // > 0: fixed by GS_OPT_NEAR_PERMILLE instead of adapted
    bool last_two_rounds;                   // the last enqueued frame ran the two-round path (its unsat count is meaningful)
    float near_floor;                       // never shrink the share below this (1.3 x the share that last proved too small)
    bool share_measured;                    // owner: near_frac comes from a measurement (GsControl::need_near) -- share_from_need, gs_api.hip
    uint32_t need_probe;                    // lane: its collections with a measurement (every fourth re-seeds its words)
    uint32_t need_word_est;                 // lane: what its need_near words hold at least (host-side estimate: seed_need_words)
    uint32_t need_seed_pending;             // lane: the seed its next frame's projection writes into the words (GsFrameUniforms::need_seed; 0 = none)
    float need_margin;                      // owner: the factor on top of the measured need (1.15 ... 1.04 while nothing misses, + 0.1 per miss)
    uint32_t need_hist[16], need_hist_frames[16]; int need_hist_pos;   // owner: the needs of the last collections and the frames each covered (share_from_need)
    uint32_t clean_frames, skip_hold;       // collected frames since the last unsaturated one / frames to keep round 1 on
    uint32_t seen_unsat_events; uint64_t seen_acc_frames;
    uint32_t single_round_frames;           // consecutive collected frames at near_frac == 1 (re-probe occlusion now and then)
    // per-workgroup partial reductions (instead of same-address global atomics, which serialise at ~11 ns each)
    unsigned long long *part_min, *part_max;   // [GS_MAX_PART]
    uint32_t *part_cnt, *part_valid, *part_vis; // [GS_MAX_PART]
    GsControl *ctl;                // device
    GsControl *ctl_host;           // pinned host mirror

    // options / stats
    bool profile;
    bool profile_blend_only;       // GS_OPT_PROFILE = 2 / 3: HIP events around the blend kernel only (2 instead of 7 per frame)
    uint32_t profile_every;        // GS_OPT_PROFILE = 3: ... and only on every 4th frame of the lane (0 / 1 = every frame)
    uint32_t profile_tick;
    uint32_t record_staged;        // GS_OPT_RECORD_STAGED (1 = entries staged, 2 = entries evaluated)
    bool wide_pairs;               // GS_OPT_WIDE_PAIRS = 1: the depth sort uses its general 8-byte (key, index) records whatever N (those of N > 2^25)
    float t_eps;
    // profiling ring: GS_PROF_RING slots x GS_PROF_EVENTS events (sort begin/end, render begin, after project, after
    // binning, after blend of round 0, end of round 1)
    hipEvent_t *ring; uint8_t *ring_flags; uint32_t ring_head, ring_pending;
    bool async_pending;            // frames were enqueued with GS_RENDER_ASYNC since the last gs_sync
    GsFrameLog *log;               // lane: the frames handed to it since the last gs_sync (caller's thread only)
    bool auto_retry;               // owner: GS_OPT_AUTO_RETRY
    int sort_share_permille;       // owner: GS_OPT_SORT_SHARE (0 = every rank sorts every frame)
    bool adapt_frozen;             // owner: gs_sync is drawing flagged frames again: their counters do not feed the adaptive share
    bool log_stale;                // owner: the resident data / scene changed under frames that are still in the logs
    int pend_lane;                 // owner: lane + 1 of the sort begun with gs_sort_begin and not yet collected by gs_sort_poll (0 = none; -1: begun before
                                   // any push: the reference's [0] reply is owed)
    hipEvent_t ev_sort;            // owner: behind that sort's kernels and the copy of its control block
    size_t pend_n;                 // owner: splats resident when it was begun
    uint32_t pend_gen;             // owner: the lane's sort_gen right after it was enqueued
    float pend_view[4], pend_cutout[16]; bool pend_has_cutout;   // owner: its arguments (a re-run does not trust the lane's saved ones)
    gs_stats stats;
};

// Where a failure message goes.  A lane's enqueue thread (gs_api.hip) points gs_tl_err at ITS OWN buffer: lane 0 is the
// owner context itself, and the caller's thread may be writing or reading ctx->err at the same moment; lane_drain()
// copies the worker's message into the lane's err on the caller's thread.
extern thread_local char *gs_tl_err;
#define GS_ERRLEN 512
#define GS_ERRBUF(ctx) (gs_tl_err ? gs_tl_err : (ctx)->err)

#define GS_HIP(call)                                                                                     \
    do {                                                                                                 \
        hipError_t _e = (call);                                                                          \
        if (_e != hipSuccess) {                                                                          \
            snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e),   \
                     __FILE__, __LINE__);                                                                \
            return (_e == hipErrorOutOfMemory) ? GS_E_OOM : GS_E_HIP;                                    \
        }                                                                                                \
    } while (0)

static inline uint32_t gs_div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// XCD-aware chunk order for the streaming kernels.  Workgroup i runs on XCD i % 8 and every XCD has its own L2, so with
// chunk = workgroup index neighbouring chunks -- which write neighbouring segments of the same digit run, i.e. the two
// halves of one 128-byte line -- sit in different L2s and each goes to HBM as a partial line.  Giving XCD k the k-th
// contiguous eighth of the chunks lets those segments merge in one L2.  v = virtual workgroup index (grid a multiple of
// 8, grid-strided); returns false for the padding slots of the last eighths.
__device__ __forceinline__ bool gs_xcd_chunk(uint32_t v, uint32_t nchunks, uint32_t &chunk) { return gsm::xcd_chunk(v, nchunks, chunk); }

// ---- two frames per launch (GS_OPT_FRAME_BATCH)
// A frame of 1 M splats is a chain of 18 short dependent kernels, most of them at the launch floor.  Two frames that take the
// same path can share every launch: grid (x, 2), blockIdx.y = the frame, each with its OWN argument list -- its own scratch,
// control block, uniforms, output.  The kernels' bodies are __device__ functions `k_xxx_body(args...)` (the plain kernels are
// thin wrappers around them); k_twin calls a body with the argument pack blockIdx.y selects.
template <class... A> struct GsPack;
template <> struct GsPack<> {};
template <class H, class... T> struct GsPack<H, T...> { H h; GsPack<T...> t; };
static inline GsPack<> gs_pack_make() { return GsPack<>(); }
template <class H, class... T> static inline GsPack<H, T...> gs_pack_make(H h, T... t) { GsPack<H, T...> p; p.h = h; p.t = gs_pack_make(t...); return p; }
// F: a type with `template <class... A> static __device__ void call(const A &...)` that forwards to the body (host code may
// not name a __device__ function, but it may name such a type): GS_BODY(F_name, k_xxx_body<...>)
#define GS_BODY(Name, ...) struct Name { template <class... A> static __device__ __forceinline__ void call(const A &... a) { __VA_ARGS__(a...); } }
template <class F, class... B> __device__ __forceinline__ void gs_pack_call(const GsPack<> &, const B &... b) { F::call(b...); }
template <class F, class H, class... T, class... B> __device__ __forceinline__ void gs_pack_call(const GsPack<H, T...> &p, const B &... b) { gs_pack_call<F>(p.t, b..., p.h); }
// (the two packs as ONE array argument indexed by blockIdx.y: with `if (blockIdx.y) call(p1); else call(p0);` the compiler loads BOTH packs'
// words into scalar registers ahead of the branch, and a body with large uniforms -- the depth pass' strip test, the projection -- then
// parks scalars in lanes of a vector register: v_readlane was 23 % of the vector instructions of the paired depth pass)
template <class P> struct GsTwinArgs { P p[2]; };
template <class F, int NT, class P> __global__ __launch_bounds__(NT) void k_twin(GsTwinArgs<P> a) { gs_pack_call<F>(a.p[blockIdx.y]); }
// launch body F for two frames: gs_twin<F, threads>(grid_x, stream, pack0, pack1)
template <class F, int NT, class P> static inline void gs_twin(uint32_t grid_x, hipStream_t st, const P &p0, const P &p1)
{
    GsTwinArgs<P> a; a.p[0] = p0; a.p[1] = p1;
    hipLaunchKernelGGL((k_twin<F, NT, P>), dim3(grid_x, 2), dim3(NT), 0, st, a);
}
// ... with a register budget: MINW = the waves per SIMD the kernel must leave room for (512 / MINW vector registers), for bodies whose
// unrolled loops the compiler would otherwise give 180-250 registers -- workgroups that then wait for half a SIMD's register file to
// drain while the other frames' blends hold it
template <class F, int NT, int MINW, class P> __global__ __launch_bounds__(NT, MINW) void k_twin_w(GsTwinArgs<P> a) { gs_pack_call<F>(a.p[blockIdx.y]); }
template <class F, int NT, int MINW, class P> static inline void gs_twin_w(uint32_t grid_x, hipStream_t st, const P &p0, const P &p1)
{
    GsTwinArgs<P> a; a.p[0] = p0; a.p[1] = p1;
    hipLaunchKernelGGL((k_twin_w<F, NT, MINW, P>), dim3(grid_x, 2), dim3(NT), 0, st, a);
}

// ---- gs_prims.hip
// One stable LSD radix pass over n = *n_ptr items on digit (key >> shift) & (2^bits-1).
// record formats: GS_RADIX_KEYS    in: a plain key array whose value is the element index; out: the values alone (final pass)
//                 GS_RADIX_PACKED  (key,val) uint2 records
//                 GS_RADIX_KEYONLY 4-byte records that are their own payload (in and out)
//                 GS_RADIX_KEYIDX  4-byte records `remaining key bits << idx_bits | element index`: out of a GS_RADIX_KEYS pass
//                                  (key >> (shift + bits) goes on top of the index), in of the final pass (digit = record >> shift
//                                  with shift = idx_bits, value = the low idx_bits bits) -- half the traffic of (key,val) records
// max_n:     upper bound of *n_ptr (sizes the scratch); hint_n: the count to expect (0 = max_n) -- it picks the grid and
//            between one- and two-level offsets, nothing that affects the result.
// have_hist: the caller's producer kernel already filled ctx->hist[chunk][digit] for this digit (skips the histogram launch).
// zero_key:  value-only output stores 0 for items with this key (0xFFFFFFFF = never).
// idx_bits:  GS_RADIX_KEYIDX output: bits reserved for the element index.
// count_out: GS_RADIX_KEYS input: where the pass leaves the number of records that took a slot (skipped ones do not).
// fill_to:   GS_RADIX_KEYIDX -> GS_RADIX_KEYS: the output is zero-filled from *n_ptr up to *fill_to.
#define GS_RADIX_KEYS 0
#define GS_RADIX_PACKED 1
#define GS_RADIX_KEYONLY 2
#define GS_RADIX_KEYIDX 3
#define GS_RADIX_SKIP 0xFFFFFFFFu   // GS_RADIX_KEYS input only: a record with this key is neither counted nor scattered (compaction)
int gs_launch_radix_pass(gs_ctx *ctx, const void *in, int in_fmt, void *out, int out_fmt, const uint32_t *n_ptr,
                         uint32_t max_n, uint32_t hint_n, int shift, int bits, bool have_hist = false, uint32_t zero_key = 0xFFFFFFFFu,
                         int idx_bits = 0, uint32_t *count_out = nullptr, const uint32_t *fill_to = nullptr);
// the same pass over two frames' records, one launch per kernel (S[k]: histogram rows / totals of frame k)
int gs_launch_radix_pass2(gs_ctx *const S[2], const void *const in[2], int in_fmt, void *const out[2], int out_fmt, const uint32_t *const n_ptr[2],
                          uint32_t max_n, uint32_t hint_n, int shift, int bits, bool have_hist, uint32_t zero_key, int idx_bits,
                          uint32_t *const count_out[2], const uint32_t *const fill_to[2]);
// The last two launches of the MSD depth sort (gs_sort.hip: depth -> bucket + rows -> THIS): keys = ctx->key_a (16-bit bucket or
// GS_RADIX_SKIP per splat), rows = ctx->hist H[chunk][bucket >> 8], group rows = ctx->msd_grp -> ctx->val_a (the index list, zero tail
// [V', V) included), ctl->n_sorted = V'.  rec = ctx->kv_b used as 4-byte records between the two.
// near: a near-only sort (no zero tail: k_project supplies the positions behind the records).
int gs_launch_msd_sort(gs_ctx *ctx, uint32_t n, uint32_t tail_req);
int gs_launch_msd_sort2(gs_ctx *const S[2], uint32_t n, const uint32_t tail_req[2]);
bool gs_msd_enabled();             // (GS_SORT_MSD=0 in the environment: the two LSD passes everywhere)
// grid used by the radix kernels for hint_n items (a producer that pre-fills the histogram rows uses the same chunking)
uint32_t gs_radix_grid(uint32_t hint_n);
// chunk length (GS_CHUNK_S / GS_CHUNK_L) a pass expecting hint_n items works with
uint32_t gs_radix_chunk(uint32_t hint_n);
// words per histogram row of a pass with nbins digits (rows are read 16 bytes at a time)
static __host__ __device__ __forceinline__ uint32_t gs_radix_row_stride(uint32_t nbins) { return nbins < 4u ? 4u : nbins; }
// ---- gs_pack.hip
int gs_launch_pack(gs_ctx *ctx, const uint4 *rows_dev, size_t first, size_t nrows);
// ---- gs_sort.hip
int gs_run_sort(gs_ctx *ctx, const float view[4], const float *cutout16, const GsSortStrip *strip = nullptr, uint32_t near_req = 0);
// what the lane's order was made from and how much of it exists (a render that needs more sorts again in full by itself)
void gs_remember_sort(gs_ctx *L, const float view[4], const float *cutout16, const GsSortStrip *strip, uint32_t near_req);
int gs_run_sort2(gs_ctx *const S[2], const float *const view[2], const float *const cutout16[2], const GsSortStrip *const strip[2], const uint32_t near_req[2]);   // two frames per launch
// ---- gs_render.hip
int gs_run_render(gs_ctx *ctx, const GsFrameUniforms &u, uint8_t *device_out);
int gs_run_round1(gs_ctx *ctx, const GsFrameUniforms &u, uint8_t *device_out);
// span-list binning: the row tables of lane `ctx` for `entries` (tile rows x 256-splat chunks) words; and an upper bound of what a frame of
// `tiles_y` tile rows over `n` sorted positions asks for (so that a context can size them before its first queued frames)
int gs_row_tables_ensure(gs_ctx *ctx, size_t entries);
size_t gs_row_tables_entries(size_t n, uint32_t tiles_y);
// two frames per launch (GS_OPT_FRAME_BATCH): whether two frames qualify, and the batched form of gs_run_render for those that do
bool gs_frames_batchable(const GsFrameUniforms &a, const GsFrameUniforms &b);
int gs_run_render2(gs_ctx *const S[2], const GsFrameUniforms U[2], uint8_t *const device_out[2]);
// ---- gs_ply.hip / gs_host.cpp
namespace gsm { struct PlyLayout; }
int gs_ply_rows_device(gs_ctx *ctx, const uint8_t *host_data, const gsm::PlyLayout &layout, size_t n, uint4 *rows_out, bool *had_nan);
extern "C" int gs_ply_plan(const void *bytes, size_t nbytes, gsm::PlyLayout *layout, size_t *nrows, size_t *data_start, char *err, size_t errlen);
// ---- gs_api.hip
static inline gs_ctx *gs_root(gs_ctx *c) { return c->parent ? c->parent : c; }
// (defined inside gs_api.hip's extern "C" block: C linkage, hidden visibility)
extern "C" int gs_fill_uniforms(gs_ctx *ctx, const gs_render_params *p, GsFrameUniforms &u);
// render the current frame's strip `u` on its lane (asynchronously if u.flags says so) into device_rgba / host_rgba
extern "C" int gs_render_uniforms(gs_ctx *ctx, const GsFrameUniforms &u, void *device_rgba, uint8_t *host_rgba, size_t stride);
// ---- gs_comm.hip
void gs_comm_free_lane(gs_ctx *lane);
void gs_comm_cancel(gs_ctx *ctx);                                  // in-process transport: fail the hub so that every waiting receive returns at once (teardown)
int gs_comm_set_self_copy(gs_ctx *ctx, bool on);
int gs_comm_set_transport(gs_ctx *ctx, int transport);
// begin a frame on its lane like gs_sort() does (lane rotation, frame log) and run `call` where the sort's kernels would be
// enqueued: the frame's order comes from -- or goes to -- the other ranks (gs_comm.hip, GS_OPT_SORT_SHARE)
extern "C" int gs_sort_by_call(gs_ctx *ctx, const float view[4], const float *cutout16, void *call /* std::function<int(gs_ctx *)> * */);
// the same in two steps: everything that can fail before the call is queued, then the hand-over (the call is always run)
extern "C" int gs_sort_call_begin(gs_ctx *ctx, const float view[4], const float *cutout16);
extern "C" int gs_sort_call_issue(gs_ctx *ctx, void *call /* std::function<int(gs_ctx *)> * */);
int gs_ensure_pair_capacity(gs_ctx *ctx, size_t pairs);
int gs_ensure_radix_scratch(gs_ctx *ctx, size_t items);         // histogram / totals scratch for a radix sort of `items` records
// event k (0..GS_PROF_EVENTS-1) of the current profiling slot, or nullptr when profiling is off
hipEvent_t gs_prof_event(gs_ctx *ctx, int k);
#define GS_PROF_RECORD(ctx, k) do { hipEvent_t _pev = gs_prof_event(ctx, k); if (_pev) GS_HIP(hipEventRecord(_pev, (ctx)->stream)); } while (0)

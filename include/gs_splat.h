/*
 * gs_splat.h -- C ABI of the MI355X-native Gaussian-splat hot path.
 *
 * Drop-in boundary for the hot path of the A-Frame `gaussian_splatting`
 * component (reference: quadjr/aframe-gaussian-splatting, file index.js).
 * Every entry point names the reference interface it replaces (file:line).
 * Plain pointers and sizes only; no C++ / torch types.  All functions return
 * GS_OK (0) or a negative gs_status; the message for the last failure on a
 * context is available from gs_last_error().
 *
 * Threading: a gs_ctx is single-caller (the reference is single-flight too:
 * `sortReady`, index.js:206/220/439-440).  Several contexts may coexist (the
 * reference allows several component instances per page, cutout-demo.html:24-25).
 *
 * There is NO CPU fallback: if no HIP device is usable gs_create() fails with
 * GS_E_NODEVICE / GS_E_HIP and nothing else can be called.
 */
#ifndef GS_SPLAT_H
#define GS_SPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_API __attribute__((visibility("default")))

typedef struct gs_ctx gs_ctx;

typedef enum gs_status {
    GS_OK = 0,
    GS_E_BADARG = -1,      /* null / out-of-range argument                                              */
    GS_E_PLY_HEADER = -2,  /* "Unable to read .ply file header"           (index.js:606-607)            */
    GS_E_PLY_PROP = -3,    /* "<prop> not found"                          (index.js:643)                */
    GS_E_HIP = -4,         /* a HIP runtime call failed                                                 */
    GS_E_OOM = -5,         /* host or device allocation failed                                          */
    GS_E_NODEVICE = -6,    /* no usable gfx950 device                                                   */
    GS_E_STATE = -7,       /* call not valid in this state (e.g. render after matrices-only push)       */
    GS_E_PLY_DATA = -8,    /* vertex data shorter than the header promises (DataView RangeError in JS)  */
    GS_E_RETRY = -9        /* gs_sync(): an asynchronous frame came back incomplete (it outgrew the pair buffers, or needed the
                              binning round it had skipped) AND the library could not draw it again by itself: gathered frames
                              (every rank has to take part), frames that share an output buffer, data pushed meanwhile, or
                              GS_OPT_AUTO_RETRY = 0.  The frames since the previous gs_sync() must be rendered again.  Otherwise
                              gs_sync() re-renders such frames into the same buffers before it returns.                  */
} gs_status;

/* ---- lifetime ----------------------------------------------------------------------------------- */

/* One context = one component instance on one GPU (replaces the Worker spawn + GL resource creation,
 * index.js:229-236 and initGL index.js:25-66).  `device` is the HIP device ordinal. */
GS_API int gs_create(int device, gs_ctx **out);
GS_API int gs_destroy(gs_ctx *ctx);
/* Last error text for ctx; ctx == NULL returns the last gs_create() failure of the calling thread. */
GS_API const char *gs_last_error(const gs_ctx *ctx);
/* Library/ABI version, e.g. 0x000100 = 0.1.0 */
GS_API uint32_t gs_version(void);
/* HIP devices this process can use (0 = none: gs_create will fail) -- what a host would pass to gs_create_multi */
GS_API int gs_device_count(void);

/* ---- data ingest (reference: worker `clear`/`push`, pushDataBuffer, processPlyBuffer) --------------- */

/* worker {method:"clear"} (index.js:573-575) + loadedVertexCount = 0 (index.js:226). */
GS_API int gs_clear(gs_ctx *ctx);

/* pushDataBuffer(buffer, vertexCount) (index.js:328-437): append `nrows` 32-byte .splat rows
 * (f32 pos[3], f32 scale[3], u8 rgba[4], u8 quat_wxyz[4]).  Packs on the GPU, bit-exactly as the
 * reference does, into the 16 B centre/scale record, the 16 B covariance/colour record and the
 * worker's sort row.  Rows are borrowed for the duration of the call. */
GS_API int gs_push_splat(gs_ctx *ctx, const void *rows, size_t nrows);

/* worker {method:"push", matrices} (index.js:576-586): append `nrows` 16-float worker rows; only
 * elements 12..15 are read, as in sortSplats (index.js:520-548).  A context fed this way can sort
 * but not render. */
GS_API int gs_push_matrices(gs_ctx *ctx, const float *matrices, size_t nrows);

/* processPlyBuffer(inputBuffer) (index.js:600-745) followed by pushDataBuffer: parse a binary
 * little-endian PLY, order rows by importance, convert to .splat rows, append.  The header is parsed on
 * the host; importance, the stable descending sort and the row conversion run on the GPU and the rows go
 * from HBM straight into the pack kernel. */
GS_API int gs_load_ply(gs_ctx *ctx, const void *bytes, size_t nbytes);

/* processPlyBuffer alone, converted on the context's GPU: same bytes as gs_ply_to_splat (both evaluate the
 * same f64 arithmetic, incl. the engine's Math.exp).  out_rows == NULL: size query (header and property
 * checks only).  Errors are reported through gs_last_error(). */
GS_API int gs_ply_to_splat_gpu(gs_ctx *ctx, const void *bytes, size_t nbytes, void *out_rows, size_t *out_nrows);

/* processPlyBuffer alone (host side): PLY bytes -> .splat rows.  Call with out_rows == NULL to get
 * *out_nrows, then again with a buffer of 32 * *out_nrows bytes.  err (optional) receives the message
 * the reference would throw. */
GS_API int gs_ply_to_splat(const void *bytes, size_t nbytes, void *out_rows, size_t *out_nrows, char *err,
                           size_t errlen);

/* Number of splats resident (reference: loadedVertexCount / matrices.length/16). */
GS_API size_t gs_count(const gs_ctx *ctx);

/* ---- sort (reference: worker {method:"sort"} -> sortSplats, index.js:507-570, 587-596) --------------- */

/* view = row 2 of gsModelViewMatrix (index.js:441-442); cutout16 = column-major object->unit-box
 * matrix or NULL (index.js:443-452).  The order is kept on the device for gs_render().
 * out_idx (optional, capacity >= max(gs_count(),1)) receives the reference's Uint32Array bit-exactly,
 * *out_n its length.  Before any push the reference answers Uint32Array(1) = [0] (index.js:588-590):
 * so does this (out_n = 1, out_idx[0] = 0). */
GS_API int gs_sort(gs_ctx *ctx, const float view[4], const float *cutout16, uint32_t *out_idx, uint32_t *out_n);

/* The reference's single-flight rhythm (index.js:201-207, 438-455): tick() POSTS the sort to the worker and returns; every frame
 * three.js draws until the worker's reply arrives uses the last COMPLETED order (the reply handler installs the new index list and
 * re-arms sortReady).  One caller thread, no blocking:
 *   gs_sort_begin  enqueues the whole sort for (view, cutout16) on a pipeline lane other than the one the current order lives on and
 *                  returns at once.  GS_E_STATE while a begun sort has not been collected (the reference's `sortReady` guard,
 *                  index.js:439-440).  gs_render* / gs_render_stereo keep drawing from the order of the last completed gs_sort /
 *                  gs_sort_poll meanwhile (before the first one: nothing, as the reference's instanceCount starts at 0).
 *   gs_sort_poll   *done = 0 if the sort is still running (wait == 0), else installs its order as the one gs_render* draws from,
 *                  hands it back like gs_sort (out_idx optional, capacity >= max(gs_count(), 1); *out_n its length; begun before any
 *                  push: [0], index.js:588-590) and sets *done = 1.  wait != 0 blocks until then.  Nothing begun: *done = 1, *out_n = 0.
 * Splats pushed between the two calls: the sort is run again over what is resident when it is collected. */
GS_API int gs_sort_begin(gs_ctx *ctx, const float view[4], const float *cutout16);
GS_API int gs_sort_poll(gs_ctx *ctx, int wait, uint32_t *out_idx, uint32_t *out_n, int *done);

/* ---- render (reference: vertex shader + rasteriser + fragment shader + blend, index.js:77-195) ------- */

#define GS_RENDER_FLIP_Y 1u      /* rows bottom-up (WebGL readPixels order) instead of top-down              */
#define GS_RENDER_COUNT_FRAGS 2u /* no early termination; count reference-equivalent splat-fragments         */
#define GS_RENDER_NO_EARLY_OUT 4u/* blend every fragment (parity debugging)                                   */
#define GS_RENDER_COUNT_EVALUATED 16u /* with GS_RENDER_COUNT_FRAGS: leave early termination ON and count the fragments
                                    the blend really evaluates (one binning round; what the reference-equivalent count shrinks to) */
#define GS_RENDER_ASYNC 8u       /* enqueue the frame and return; completion, status and statistics are collected by
                                    gs_sync().  The reference renders every frame without waiting for the GPU either
                                    (index.js:184-207).  gs_render_device: the frame stays in HBM.  gs_render: the frame
                                    is copied to rgba_out behind its kernels, on the frame's own stream -- rgba_out must
                                    stay valid and unread until gs_sync(), one buffer per frame in flight, and should be
                                    page-locked (gs_host_alloc) for the copy to overlap the following frames.           */

typedef struct gs_render_params {
    float model_view[16]; /* gsModelViewMatrix, column-major (getModelViewMatrix, index.js:467-487)      */
    float projection[16]; /* gsProjectionMatrix, column-major (getProjectionMatrix, index.js:456-466)    */
    int32_t fb_width;     /* `viewport` uniform, device pixels (index.js:189-193)                        */
    int32_t fb_height;
    int32_t x0, x1;       /* column strip [x0,x1) to produce; x0 = 0, x1 = fb_width for the whole frame  */
    float focal;          /* `focal` uniform; <= 0: computed as fb_height/2*|projection[5]| (index.js:191) */
    float background[4];  /* destination before blending; demos: opaque black sky (index.html:14)        */
    uint32_t flags;       /* GS_RENDER_*                                                                  */
} gs_render_params;

/* Draw the last gs_sort() order into a tightly described RGBA8 image in caller-owned host memory.
 * rgba_out: (x1-x0) x fb_height pixels, `stride` bytes per row (0 = tight), row 0 = top. */
GS_API int gs_render(gs_ctx *ctx, const gs_render_params *p, uint8_t *rgba_out, size_t stride);
/* Strips [x0,x1) whose x0 is a multiple of 4 (the multi-GPU partition uses multiples of 16), whatever x1, reproduce the
 * corresponding columns of the full frame bit for bit; other strips within 1 LSB (early termination works on groups of 4
 * pixels counted from x0). */
/* Same, but the strip stays on the GPU: device_rgba is a device pointer (e.g. a torch tensor's
 * data_ptr, tight rows) or NULL to render into the context's own framebuffer only. */
GS_API int gs_render_device(gs_ctx *ctx, const gs_render_params *p, void *device_rgba);
/* gs_sort for ONE column strip of a frame that is then drawn with exactly these parameters (strip->x0, x1; NULL = gs_sort):
 * splats whose quad cannot reach the strip are left out by a conservative bound (4 min(sqrt(2 lambda1), 1024) + 2 pixels
 * around the projected centre, lambda1 <= (|J| |A| sigma_max)^2 + 0.3, index.js:127-149), the others come in the reference's
 * order -- the bucket scale still uses the depth range of EVERY splat the reference keeps (index.js:552-558).  The strip's
 * pixels are bit-identical to those drawn from the full order; what shrinks is the sort (and everything downstream) on a GPU
 * that owns one strip of eight.  out_idx / out_n: the strip's sub-sequence of the reference's Uint32Array.  Not for a sort
 * that outlives its camera (the reference draws with the latest completed order, index.js:201-207): render with the pose
 * it was sorted for. */
GS_API int gs_sort_for(gs_ctx *ctx, const float view[4], const float *cutout16, const gs_render_params *strip,
                       uint32_t *out_idx, uint32_t *out_n);

/* XR: two eyes share one sort order from the head camera (index.js:441) -- two params, two images. */
GS_API int gs_render_stereo(gs_ctx *ctx, const gs_render_params eyes[2], uint8_t *rgba_out[2], size_t stride);

/* Scene compositing inputs (reference: the splat mesh is drawn in three.js' transparent pass with depthTest: true,
 * depthWrite: false over the opaque scene, index.js:177-181): an optional window-space depth buffer of that scene
 * (GL convention: 0 = near plane, 1 = far plane; a fragment survives iff its depth zndc*0.5+0.5 <= the buffer, LEQUAL)
 * and an optional RGBA8 colour image that replaces the constant background.  Both fb_width x fb_height, tightly packed,
 * row 0 = top, host memory (copied); NULL for either = not used; gs_set_scene(ctx, NULL, NULL, 0, 0) clears.
 * Renders whose fb_width/fb_height differ from the scene's fail with GS_E_BADARG. */
GS_API int gs_set_scene(gs_ctx *ctx, const float *depth, const uint8_t *rgba, int fb_width, int fb_height);

/* Page-locked host memory for framebuffers (gs_render copies into it at PCIe speed, no staging copy; the N-API addon hands
 * it to JavaScript as an external ArrayBuffer).  Free with gs_host_free. */
GS_API void *gs_host_alloc(size_t nbytes);
GS_API void gs_host_free(void *p);

/* Block until all work queued on the context's stream is done; collects the status and statistics of frames
 * rendered with GS_RENDER_ASYNC on any lane (GS_E_RETRY if one of them overflowed the pair buffers). */
GS_API int gs_sync(gs_ctx *ctx);
/* Run the context's kernels on a caller-owned hipStream_t (e.g. torch's current stream). NULL = own stream.
 * On a caller-owned stream every frame is ordered with the caller's other work on it, so frames are not pipelined
 * over lanes (GS_OPT_PIPELINE_DEPTH is ignored); use the two calls below to couple pipelined frames to other streams. */
GS_API int gs_set_stream(gs_ctx *ctx, void *hip_stream);
/* GPU-side ordering between pipelined frames and the caller's own streams (no host blocking; what such a consumer reads may be an
 * INCOMPLETE frame: test gs_frame_status_device()'s word, below):
 * gs_wait_stream: the NEXT frame (next gs_sort + its renders) starts only after everything queued on hip_stream so
 *                 far, e.g. a collective that still reads the buffer that frame will render into;
 * gs_stream_wait_frame: hip_stream waits for the frame enqueued LAST (its gs_sort + renders so far), e.g. before a
 *                 collective or copy on that stream consumes the frame's device_rgba. */
/* The hipStream_t the current frame (last gs_sort + its renders) was enqueued on: work the caller queues on it -- a
 * collective over the frame's device_rgba, a copy -- is ordered after the frame and before the frame that will reuse
 * this lane, without any cross-stream event (those cost ~50 us of pipeline stall each on this platform). */
GS_API void *gs_frame_stream(gs_ctx *ctx);
/* ASYNCHRONOUS FRAMES ARE SPECULATIVE until gs_sync().  A frame queued with GS_RENDER_ASYNC may skip its second binning round (the
 * share of splats binned first has been measured over the frames before) or bin into buffers sized from the frames before; if a tile
 * then turns out not to be saturated, or the buffers to be too small, the frame is INCOMPLETE and gs_sync() draws it again (or asks
 * for it: GS_E_RETRY).  The reference never draws from an incomplete order either (index.js:201-207: the draw uses the latest
 * COMPLETED sort).  A consumer that reads the frame on the GPU before gs_sync() -- work queued on gs_frame_stream(), a stream coupled
 * through gs_stream_wait_frame(), the library's own gather -- must therefore test the frame's completion word:
 *   gs_frame_status_device(): *device_word = the device address of ONE uint32 of the current frame's lane, written by the frame's own
 *   kernels: 0 = complete; non-zero = the frame will be drawn again at gs_sync() (bit 0: a tile was not saturated, bit 1: the pair
 *   buffers overflowed, bit 2: the sorted order was incomplete).  Valid once the frame's kernels have run (in stream order behind
 *   the frame) until the lane has been handed 64 more renders (every lane keeps a ring of 64 words: gs_sync() itself reads them and
 *   draws again exactly the frames whose word is not 0); for a gathered frame,
 *   on the root: the OR of the words of all its pieces (each piece travels with its own word, the root's gs_sync() reports the
 *   frame even if only a peer's piece was incomplete).  Several renders of one frame on one context (gs_render_stereo): the word
 *   describes the last of them. */
GS_API int gs_frame_status_device(gs_ctx *ctx, void **device_word);
/* The same in two steps, for callers that want to keep enqueuing: gs_frame_lane() names the lane of the current frame
 * (no waiting); gs_lane_stream() returns that lane's stream once its worker thread has enqueued everything handed to
 * the lane so far.  Queuing the follow-up work of frame k only after frame k+1 (or k+2) has been handed over keeps the
 * caller from waiting for the worker; it must still be queued before the lane is given its next frame. */
GS_API int gs_frame_lane(gs_ctx *ctx);
GS_API void *gs_lane_stream(gs_ctx *ctx, int lane);
GS_API int gs_wait_stream(gs_ctx *ctx, void *hip_stream);
GS_API int gs_stream_wait_frame(gs_ctx *ctx, void *hip_stream);

/* ---- several GPUs (no reference counterpart: the reference draws on one WebGL context) ------------------------------
 * The viewport is split into tile-aligned column strips (or, for XR, the two eyes go to different GPUs), the splat buffer is
 * replicated, every context sorts for the same view and renders its own pieces, and the pieces are gathered on one root
 * over RCCL (xGMI).  One gs_ctx per GPU: one process per GPU (the launcher distributes the id: torch.distributed, MPI, a
 * file) or several contexts in one process.  RCCL is loaded (dlopen "librccl.so.1") by gs_comm_init, not before: a
 * single-GPU user never needs it.  Every rank must issue the same sequence of gs_sort / gs_render_*_gathered calls. */
#define GS_COMM_ID_BYTES 128
/* rank 0: create the id all ranks pass to gs_comm_init (ncclGetUniqueId) */
GS_API int gs_comm_unique_id(gs_ctx *ctx, void *id_out);
/* join the communicator as `rank` of `world` (ncclCommInitRank; collective: returns when every rank has called).
 * In-process transport (GS_OPT_COMM_TRANSPORT = 1): never blocks.  A receive of that transport waits ON THE HOST for its sender to
 * post (GS_COMM_TIMEOUT_S, default 60 s), so ONE thread that drives several such ranks with SYNCHRONOUS gathered frames has to call
 * the non-root ranks first and the root last (queued frames -- GS_RENDER_ASYNC -- and gs_create_multi, one thread per rank, have no
 * such order).  A failed exchange is final, as with an aborted RCCL communicator: the rank that gave up has skipped an operation
 * its peers performed; destroy the communicator on every rank and join a new one. */
GS_API int gs_comm_init(gs_ctx *ctx, const void *id, int rank, int world);
GS_API int gs_comm_destroy(gs_ctx *ctx);

/* Who renders what: nviews images of widths[v] pixels over `world` ranks -> pieces (view, [x0,x1), owner rank), in the
 * order they are gathered.  One view: tile-aligned column strips, as even as possible, the last strip takes the ragged
 * edge.  Two views (XR eyes, index.js:13-15): world 1 renders both; otherwise the ranks are divided between the eyes (eye k ->
 * rank k at world 2) and each eye is split in strips over its ranks.  Returns the number of pieces (<= max_pieces) or < 0. */
typedef struct gs_piece { int32_t view, x0, x1, owner; } gs_piece;
GS_API int gs_partition(int nviews, const int *widths, int world, gs_piece *out, int max_pieces);

/* One frame over all ranks: each renders its pieces of the view(s) with the order of its last gs_sort() and sends them to
 * `root`, where they are assembled into row-major RGBA8 images: device_frames[v] (device memory, fb_width x fb_height x 4
 * tight) or, if NULL, buffers of the context readable with gs_read_gathered().  views[v].x0/x1 are ignored (the partition
 * sets them).  flags: GS_RENDER_ASYNC enqueues and returns (completion and status at gs_sync()); GS_RENDER_FLIP_Y applies
 * per view.  nviews = 2 is the XR frame: two eyes, ONE shared sort from the head camera (index.js:441). */
GS_API int gs_render_gathered(gs_ctx *ctx, const gs_render_params *views, int nviews, int root, void *const *device_frames,
                              uint32_t flags);
/* The sort of a gathered frame: gs_sort_for the piece gs_partition gives this rank when it owns exactly one column strip of
 * one view (the sort, projection and binning of an 8-GPU frame then shrink with the strip instead of being replicated), a plain
 * gs_sort otherwise.  Call it with the views the following gs_render_gathered draws. */
GS_API int gs_sort_gathered(gs_ctx *ctx, const float view[4], const float *cutout16, const gs_render_params *views, int nviews);
/* root: copy view `view` of the last gathered frame (after gs_sync() for asynchronous frames) to host memory */
GS_API int gs_read_gathered(gs_ctx *ctx, int view, uint8_t *rgba_out, size_t stride);
/* ... and its size in pixels (so that a binding can check the caller's buffer before the copy) */
GS_API int gs_gathered_size(gs_ctx *ctx, int view, int *width, int *height);
/* ---- one host process, several GPUs (the reference is one JavaScript thread per page, index.js:1-23, with
 * several component instances allowed, cutout-demo.html:24-25 -- a Node.js consumer cannot be one process per GPU) ----------
 * A gs_multi owns one context per entry of `devices` (an ordinal may repeat: several "devices" on one GPU, which is how the
 * single-GPU test tier runs it), keeps the splat buffer replicated on them and splits every frame as gs_partition says: column
 * strips of one view, or the two XR eyes over the devices.  The calls have the single-context meaning and signatures; each
 * context is fed by its own thread, so an asynchronous frame costs the caller the same whatever the number of GPUs.
 * The communicator between the contexts is the in-process transport (GS_OPT_COMM_TRANSPORT = 1), set up by gs_create_multi. */
typedef struct gs_multi gs_multi;
GS_API int gs_create_multi(const int *devices, int ndev, gs_multi **out);
GS_API int gs_multi_destroy(gs_multi *m);
GS_API const char *gs_multi_last_error(const gs_multi *m);   /* m == NULL: the last gs_create_multi failure of the calling thread */
GS_API int gs_multi_devices(const gs_multi *m);
/* the context on devices[i] (statistics, downloads, per-context options); only between gs_multi_sync() and the next frame */
GS_API gs_ctx *gs_multi_ctx(gs_multi *m, int i);
GS_API int gs_multi_clear(gs_multi *m);                                              /* gs_clear on every device          */
GS_API int gs_multi_push_splat(gs_multi *m, const void *rows, size_t nrows);         /* gs_push_splat, uploads in parallel */
GS_API int gs_multi_load_ply(gs_multi *m, const void *bytes, size_t nbytes);
GS_API size_t gs_multi_count(const gs_multi *m);
GS_API int gs_multi_set_option(gs_multi *m, int option, int64_t value);              /* gs_set_option on every device     */
/* tick + worker sort for the frame `views` describe (one view, or the two XR eyes with the head camera's view row, index.js:441):
 * every device sorts what its own piece of the frame needs (gs_sort_gathered).  Asynchronous: failures surface at the next
 * synchronous render or gs_multi_sync(). */
GS_API int gs_multi_sort(gs_multi *m, const float view[4], const float *cutout16, const gs_render_params *views, int nviews);
/* HOST-DIRECT frame: host_frames[v] is the caller's fb_width x fb_height RGBA8 image (`stride` bytes per row, 0 = tight; page-
 * locked memory from gs_host_alloc lets the copies overlap the next frames); every device copies its strip straight into its
 * columns behind its kernels -- no collective, no staging, no assembly.  Without GS_RENDER_ASYNC the call returns with the frame
 * complete (and has drawn it again by itself if a device reported GS_E_RETRY); with it, completion and status come from
 * gs_multi_sync(), one frame buffer per frame in flight. */
GS_API int gs_multi_render(gs_multi *m, const gs_render_params *views, int nviews, uint8_t *const *host_frames, size_t stride, uint32_t flags);
/* DEVICE frame: gathered on devices[0] through the in-process transport (peer copies on the frames' own streams) into
 * device_frames[v] (memory of devices[0]) or, if NULL, into buffers read with gs_multi_read(). */
GS_API int gs_multi_render_device(gs_multi *m, const gs_render_params *views, int nviews, void *const *device_frames, uint32_t flags);
GS_API int gs_multi_read(gs_multi *m, int view, uint8_t *rgba_out, size_t stride);
/* gs_sync on every device: GS_E_RETRY if any asynchronous frame since the last sync has to be drawn again */
GS_API int gs_multi_sync(gs_multi *m);

#define GS_OPT_BLEND_SPLIT 9    /* 0 (default): one wavefront blends each tile, 4 pixels per lane.  L > 0: tiles whose list has at least
                                   L entries (try 512) are blended by FOUR wavefronts, one pixel per lane -- for frames in which few
                                   tiles carry long lists (a cut-out scene: the kernel otherwise lasts as long as ONE wavefront's
                                   serial walk of the busiest tile's list).  Same fragments and per-pixel operations; a pixel of such
                                   a tile stops exactly when it falls below the termination threshold instead of with its lane's
                                   other three (within the 1 LSB tolerance), so images are bit-identical only between renders with
                                   the same setting AND the same binning share: off by default.  The rule is per tile: strips
                                   still equal the full frame bit for bit. */
#define GS_OPT_FRAME_BATCH 10   /* 1 (default): every frame is its own chain of launches.  2: consecutive asynchronous frames
                                   (gs_sort without an output array + gs_render / gs_render_device with GS_RENDER_ASYNC) are
                                   paired: the two frames of a pair share every kernel launch (grid (x, 2): each frame keeps
                                   its own sort, projection, binning and blend on its own scratch; pixels are identical to
                                   unpaired rendering), and 2 x GS_OPT_PIPELINE_DEPTH frames are in flight.  A frame of
                                   ~1 M splats is a chain of 18 short kernels at the launch floor: sharing them is worth
                                   ~20 % in frames/s.  Pairing happens in the lanes' enqueue threads when both frames are
                                   queued (needs GS_OPT_ENQUEUE_THREADS); gathered frames of one piece per rank pair as well
                                   (their gathers follow the shared kernels in frame order); frames that differ in size,
                                   flags or path go out alone.                                                              */
#define GS_OPT_SORT_NEAR 11     /* near-only depth sorts.  A frame whose second binning round is skipped (the share has been clean
                                   for a few frames) reads only the nearest share of the order; its gs_sort then leaves out the
                                   splats that cannot be among those, so that the positions the frame reads hold exactly what the
                                   whole order holds there.  Up to 2 M splats (the four-launch sort): a TAIL sort -- the order is cut
                                   at the boundary of the 256 depth segments the sort works in anyway, the segments before the cut
                                   are neither scattered nor sorted.  Longer inputs: an exact threshold on the sort key from a
                                   histogram of the depths, applied before the radix passes.  A render that needs more of the
                                   order after all (another share, a counting render, round 1, gs_download of the order) sorts
                                   again in full by itself.  Sorts that return the order (out_idx / out_n) are always complete.
                                   0: off; 1 (default): tail sorts up to 2 M splats, the histogram form from 4 M (between the two
                                   its extra work costs what it saves); 2: always.                                               */
#define GS_OPT_SORT_SHARE 15    /* several ranks (gs_sort_gathered), value = permille P of the splats, 0 = off (default).  The depth sort is the
                                   part of a frame that does not shrink with a rank's strip: every rank keys and sorts all N splats of every
                                   frame (at 20 M splats 230 of a strip frame's 290 us).  With P > 0 the ranks take turns: frame f is
                                   sorted by rank f mod world alone -- a near-only sort of the nearest P/1000 * N splats of the WHOLE view,
                                   the same kernels, hence the same order -- which sends it (a few MB, one send per peer, each over its own
                                   xGMI link) to the others; those receive it where their own sort would have run.  Every rank then draws
                                   its pieces from that order exactly as after gs_sort(): same pixels.  A frame that needs more of the
                                   order than was exchanged (its first binning round reads more than P/1000 * N splats, or did not skip
                                   the second one) sorts again in full locally, as after any near-only sort.  Set the same value on every
                                   rank, before the frames; needs N <= 2^25.                                                        */
/* (option 14, GS_OPT_HOST_WRITE -- the blend kernel storing its tiles straight into a page-locked host frame -- was measured in rounds 3
   and 4: no faster for a frame alone, 25 % slower with frames in flight; removed in round 5.  The copy engine delivers host frames.) */
#define GS_OPT_AUTO_RETRY 13    /* 1 (default): gs_sync() draws an asynchronous frame that came back incomplete again by itself -- same sort
                                   arguments, same uniforms, same output buffers, both binning rounds -- before it returns; GS_E_RETRY is
                                   left for the cases it cannot decide alone (see gs_status).  0: every such frame is reported.       */
#define GS_OPT_COMM_TRANSPORT 12 /* what gs_comm_unique_id makes an id for.  0 (default): RCCL over xGMI, one process per GPU or several contexts
                                   of one process.  1: the in-process transport -- contexts of ONE process (on different GPUs, or all on
                                   the same one) exchange their pieces through mailbox buffers and peer copies on their own streams,
                                   ncclSend / ncclRecv semantics, no RCCL; gs_comm_init with such an id never blocks, so one thread can
                                   bring all ranks up (gs_create_multi does).  Set it on the context that creates the id; the other
                                   ranks recognise the id.                                                                      */
#define GS_OPT_COMM_SELF_COPY 8 /* value != 0: the root sends its own pieces to itself through RCCL too instead of rendering them
                                   in place (exercises send/recv on a single-GPU box; slower) */

/* ---- uniforms / camera helpers (host side; reference: tick + camera matrices, index.js:438-487) ------ */

/* getModelViewMatrix: camera.matrixWorld, object.matrixWorld (column-major f64) -> gsModelViewMatrix */
GS_API void gs_model_view_matrix(const double cam_world[16], const double obj_world[16], double out[16]);
/* getProjectionMatrix: camera.projectionMatrix -> gsProjectionMatrix */
GS_API void gs_projection_matrix(const double proj[16], double out[16]);
/* tick: the two messages posted to the worker: view[4] and (if cutout_world != NULL) cutout[16] */
GS_API void gs_tick_uniforms(const double cam_world[16], const double obj_world[16], const double *cutout_world,
                             float view[4], float cutout[16]);
/* onBeforeRender focal (index.js:191) */
GS_API double gs_focal(const double gs_proj[16], double viewport_h);
/* init: pixelRatio / xrPixelRatio (index.js:10-15): drawing-buffer size = floor(css * ratio) when ratio > 0 */
GS_API void gs_scaled_size(int css_w, int css_h, double ratio, int *out_w, int *out_h);

/* ---- stats / introspection ---------------------------------------------------------------------- */

typedef struct gs_stats {
    uint64_t n_splats;    /* N resident                                                               */
    uint64_t n_sorted;    /* V  = length of the last sort result                                      */
    uint64_t n_visible;   /* Vp = splats surviving the vertex-shader culls in the last render          */
    uint64_t n_pairs;     /* I  = (tile,splat) pairs binned in the last render                         */
    uint64_t n_frags;     /* reference-equivalent fragments (last GS_RENDER_COUNT_FRAGS render)        */
    uint64_t n_tiles;     /* tiles in the last rendered strip                                         */
    float ms_sort;        /* GPU time of the last gs_sort (HIP events; 0 unless profiling is on)       */
    float ms_project;
    float ms_bin;
    float ms_blend;
    float ms_render;      /* project + bin + blend                                                    */
    uint32_t blend_launches;
    /* accumulated since GS_OPT_PROFILE was last switched on (collected at gs_sync / synchronous renders)     */
    uint32_t prof_frames; /* frames whose HIP-event timings are summed below                                  */
    float sum_ms_sort, sum_ms_project, sum_ms_bin, sum_ms_blend;
    uint64_t acc_frames;  /* frames rendered (device-side counter)                                            */
    uint64_t acc_sorted;  /* sum of V over those frames                                                       */
    uint64_t acc_visible; /* sum of Vp                                                                        */
    uint64_t acc_pairs;   /* sum of I                                                                         */
    uint32_t unsat_tiles; /* tiles the nearest-splats round left unsaturated in the last collected frame             */
    uint32_t near_permille;/* share of the splats binned in that first round (adapted, or GS_OPT_NEAR_PERMILLE)        */
    uint32_t sort_records;/* records the last collected frame's depth sort carried through its second pass: the kept
                             splats with a valid bucket, or the nearest few of them (GS_OPT_SORT_NEAR)               */
    uint32_t retried_frames;/* asynchronous frames gs_sync() drew again by itself since the context was created (GS_OPT_AUTO_RETRY) */
    uint32_t spec_sorts;  /* collected frames whose near-only sort took its candidates from the depth pass' own stash (no depth
                             array written: a threshold hint from the previous frames decides what is stashed) ...              */
    uint32_t spec_misses; /* ... and those of them whose candidates could not be vouched for (drawn again from a whole sort)    */
    uint32_t need_splats; /* how many of the NEAREST splats the last collected frames' tiles read before they were saturated (max over
                             the tiles; 0xFFFFFFFF: a tile no share saturates): what near_permille is set from, with a margin of
                             15 % that shrinks to 4 % while no frame misses                                                        */
    uint32_t sort_mode;   /* how the last collected frame's depth sort ran: 0 = the whole order (the reference's, index.js:507-570); near-only
                             forms (GS_OPT_SORT_NEAR): 1 = threshold from a depth histogram, 2 = candidates from the depth pass' own stash,
                             3 = tail sort (the segments the frame does not read neither scattered nor sorted)                       */
    uint32_t subtile;     /* 1 if the last frame's blend walked sub-tile lists (GS_OPT_SUBTILE)                                        */
} gs_stats;

#define GS_OPT_PROFILE 1        /* 1: bracket every stage with HIP events on the frame's stream (7 per frame); 2: only
                                   the blend kernel (2 per frame; sum_ms_blend / prof_frames); 3: like 2 but only on every
                                   4th frame of a lane (an event pair costs ~5 % of the pipelined frame rate); 0: off */
#define GS_OPT_TERMINATION 2    /* value = 1/eps: a pixel stops blending once its transmittance T < eps (default 1024, SURVEY.md 8a row 9:
                                   what lies behind then weighs < eps = 0.25 LSB of RGBA8 in total; the image stays within 1 LSB of the
                                   back-to-front result for any eps < 1/256.  4096 was the round-1 default: 9 % fewer frames/s) */
#define GS_OPT_NEAR_PERMILLE 3  /* occlusion-aware binning: 0 = adapt (default), 1..999 = bin that share of the nearest
                                   splats first and the rest only against unsaturated tiles, 1000 = single round      */
#define GS_OPT_RECORD_STAGED 4  /* value != 0: each render overwrites the tile-range table with (list entries staged,
                                   list length) per tile, readable with gs_download(GS_BUF_TILE_STATS) (measurement aid);
                                   value 2 records the entries the tile evaluated before it saturated instead of staged */
#define GS_OPT_PIPELINE_DEPTH 5 /* 1..4 (default 3): frames enqueued with GS_RENDER_ASYNC rotate over this many internal lanes
                                   (own HIP stream + own per-frame scratch, resident splat data shared), so the kernel
                                   chain of frame k+1 runs under the tail of frame k -- as the reference overlaps its
                                   worker sort with drawing.  A gs_sort() begins a frame; it moves to the next lane when
                                   the previous frame was handed off asynchronously.  1 = strictly one frame at a time   */
#define GS_OPT_WIDE_PAIRS 6     /* 1: the depth sort carries its general 8-byte (key, index) records whatever the number of splats -- the format of
                                   N > 2^25, where 4-byte records no longer hold bucket and index -- (same order; a test hook for that format on
                                   small inputs).  0 (default): 4-byte records up to 2^25 splats.  (Rounds 2-5: also the binning's record
                                   formats; since round 6 its (tile, position) records have ONE form, 8 bytes.)                     */
#define GS_OPT_BINNING 16       /* how a binning round turns visible splats into per-tile lists (same lists, same images).  0 (default): span
                                   lists -- every splat becomes one run of tiles per tile row it touches, and the runs of a tile row, in
                                   sorted order, are expanded into the row's tile lists by one thread per tile column: four launches per
                                   round (project, row scan, runs, lists; five with the segment counts of frames that hold many runs per
                                   tile row) -- wherever a strip has at most 256 tile columns and rows (4096 x 4096 pixels), i.e. for every
                                   BASELINE configuration; beyond that, and with 1: (tile, splat) pair records sorted by two stable radix
                                   passes (rounds 1-3; eight launches per round).                                                */
#define GS_OPT_SUBTILE 17       /* how the blend walks a tile's list (same pixels, bit for bit, whatever the value).  The reference's rasteriser shades
                                   only the fragments a quad covers (index.js:52-66, 166-176); a 16x16 tile's wavefront evaluates a list entry for
                                   all 256 pixels.  With sub-tile lists the wavefront splits every batch of 64 entries into the lists of the
                                   tile's sixteen 4x4-pixel blocks (which blocks an entry's ellipse can reach is worked out from its record when
                                   it is staged) and takes as many steps as the longest of them -- worth it where splats are small against a
                                   tile (the cloud seen from outside: a list entry covers a fifth of its tile).  0: never; 1 (default): in
                                   frames that follow collected frames whose visible splats touched fewer than 16 tiles each on average
                                   (GS_SUBTILE_RATIO in the environment overrides the 16); 2: always (a batch of large splats is still walked
                                   whole: the decision is per batch).                                                              */
#define GS_OPT_ENQUEUE_THREADS 7 /* default 1: gs_sort() (without an output array) and gs_render_device(GS_RENDER_ASYNC) hand the
                                   frame to a worker thread of its pipeline lane, which does the ~18 kernel launches, so the
                                   launches of the frames in flight run in parallel; failures surface at gs_sync().  0: the
                                   calling thread launches everything itself                                            */
GS_API int gs_set_option(gs_ctx *ctx, int option, int64_t value);
GS_API int gs_get_stats(gs_ctx *ctx, gs_stats *out);

/* Copy a device-resident array back to the host (parity tests / debugging). */
#define GS_BUF_CENTER_SCALE 0 /* N x 4 f32   centerAndScaleData                                         */
#define GS_BUF_COV_COLOR 1    /* N x 4 u32   covAndColorData                                            */
#define GS_BUF_SORT_ROWS 2    /* N x 4 f32   worker row elements 12..15                                 */
#define GS_BUF_SORTED 3       /* V u32       last sort result                                           */
#define GS_BUF_PROJECTED 4    /* V x 8 f32   projected records of the last render (sorted order)         */
#define GS_BUF_TILE_COUNT 5   /* V u32       tiles touched per sorted splat in the last render            */
#define GS_BUF_TILE_STATS 6   /* tiles x 2 u32  (entries staged, list length) after a GS_OPT_RECORD_STAGED render  */
#define GS_BUF_UNSAT_MASK 7   /* tiles_y x ceil(tiles_x / 32) u32: one bit per tile the first binning round of the last render left
                                 unsaturated (measurement aid)                                                         */
GS_API int gs_download(gs_ctx *ctx, int which, void *out, size_t nbytes);

#ifdef __cplusplus
}
#endif
#endif /* GS_SPLAT_H */

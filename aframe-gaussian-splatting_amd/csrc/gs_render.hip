// gs_render.hip -- the GPU half of the reference (vertex shader index.js:77-165, rasteriser, fragment shader
// index.js:166-176, blend state index.js:177-181) as a tile-binned HIP pipeline:
//
//   k_project      one thread per SORTED splat (not 6 like the instanced quad): gather one 32 B record,
//                  Sigma' = (J W) Sigma (J W)^T, eigen axes, EXACT per-tile-row coverage count   [HBM gather]
//   k_pairs_check  spine scan of tiles-touched -> chunk offsets, total I, the extra work items of heavy chunks
//   k_emit         (tile id, sorted position) records in splat order, work handed out by pair slots
//   radix x2       stable sort of the records by tile id only: the input is already in depth order, so each
//                  tile's list inherits the reference's draw order with no depth key            [gs_prims]
//   k_tile_ranges  [start,end) of every tile in the sorted list
//   k_blend        one wavefront per 16x16 tile, 4 pixels per lane, LDS-staged batches of projected records,
//                  FRONT-to-back traversal (reverse of the back-to-front list) with per-pixel transmittance
//                  accumulators and wave-wide early termination; one rounding to RGBA8 at the end.
//
// Occlusion-aware binning in two rounds.  Front-to-back blending stops a pixel at T < eps, so whatever lies behind a
// saturated tile is never read -- in the benchmark scene 81 % of the (tile, splat) records.  Round 0 therefore bins
// and blends only the NEAREST `near_count` splats of the sorted order; tiles in which every pixel terminated are
// final.  Tiles that did not saturate keep their exact fp32 per-pixel state (T, premultiplied RGB) and set a bit in
// a tile mask; round 1 bins the remaining (farther) splats against the masked tiles only and continues those tiles
// from the saved state.  The arithmetic per pixel is the same sequence of operations as a single pass, so the
// result is bit-identical; only work that could not contribute is skipped.  When no tile is left unsaturated the
// kernels of round 1 find an empty range and return at once.
//
// The fixed-function rasteriser + ROP of the reference have no structural counterpart; parity is defined at
// the pixel level against oracle/gs_oracle.c (docs/LAB_NOTES.md "Pixel parity argument").
#include "gs_internal.h"

namespace {

// Sums by data-parallel-primitive moves instead of __shfl_xor steps (each of those a ds_bpermute round trip plus its address arithmetic:
// ~5 vector instructions where these take one or two).  row16_sum: every lane gets the sum over its row of 16 lanes (quads, then the two
// mirrors); wave_sum: the wave's sum (rows 0 -> 1 and 2 -> 3, then rows 0-1 -> 2-3: lane 63 holds it).
#define GS_DPP_ADD(V, CTRL, ROWS) (V) += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(V), (CTRL), (ROWS), 0xF, false)
__device__ __forceinline__ uint32_t row16_sum(uint32_t v)
{
#ifdef GS_DEBUG_DPP_MAX
    const uint32_t v0_ = v;
#endif
    GS_DPP_ADD(v, 0xB1, 0xF);      // quad_perm:[1,0,3,2]
    GS_DPP_ADD(v, 0x4E, 0xF);      // quad_perm:[2,3,0,1]
    GS_DPP_ADD(v, 0x141, 0xF);     // row_half_mirror
    GS_DPP_ADD(v, 0x140, 0xF);     // row_mirror
#ifdef GS_DEBUG_DPP_MAX
    { uint32_t c = v0_; for (int m = 8; m >= 1; m >>= 1) c += __shfl_xor(c, m, 16); if (c != v) __builtin_trap(); }
#endif
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#ifdef GS_DEBUG_DPP_MAX
    uint32_t c_ = v; for (int m = 32; m >= 1; m >>= 1) c_ += __shfl_xor(c_, m, 64);
#endif
    v = row16_sum(v);
    GS_DPP_ADD(v, 0x142, 0xA);     // row_bcast:15 -> rows 1, 3
    GS_DPP_ADD(v, 0x143, 0xC);     // row_bcast:31 -> rows 2, 3
    v = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
#ifdef GS_DEBUG_DPP_MAX
    if (c_ != v) __builtin_trap();
#endif
    return v;
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(v, d, 64); if (lane >= d) v += t; }
    return v;
}

// bits [t0, t0+n) of one row of the unsaturated-tile mask
__device__ __forceinline__ uint32_t mask_bits(const uint32_t *__restrict__ row, uint32_t word, uint32_t t0, uint32_t n)
{
    const uint32_t lo = word * 32u, hi = lo + 32u;
    const uint32_t a = max(t0, lo), b = min(t0 + n, hi);
    if (a >= b) return 0u;
    const uint32_t span = b - a;
    const uint32_t m = (span == 32u ? 0xFFFFFFFFu : ((1u << span) - 1u)) << (a - lo);
    return row[word] & m;
}
__device__ __forceinline__ uint32_t mask_count(const uint32_t *__restrict__ row, uint32_t t0, uint32_t n)
{
    uint32_t c = 0;
    for (uint32_t w = t0 >> 5; w <= ((t0 + n - 1) >> 5); w++) c += __popc(mask_bits(row, w, t0, n));
    return c;
}

// Sorted-position range [lo, hi) of a round.  Round 0: the nearest `near_count` splats (the sorted order is
// far -> near, so they are its tail).  Round 1: everything farther, or nothing if every tile saturated in round 0
// (ctl->unsat_count is final once round 0's blend has completed).
template <int ROUND>
__device__ __forceinline__ void round_range(const GsControl *ctl, uint32_t near_count, uint32_t &lo, uint32_t &hi)
{
    const uint32_t V = ctl->n_kept;
    const uint32_t nn = min(V, near_count);
    if (ROUND == 0) { lo = V - nn; hi = V; }
    else if (ctl->unsat_count == 0 || nn == V) { lo = 0; hi = 0; }
    else { lo = 0; hi = V - nn; }
}

// Vertex-shader stage.  One thread per sorted splat, 256 splats per workgroup pass; the pass also leaves the chunk's
// total tiles-touched in spine[chunk] (first level of the pair-offset scan).
// Splats whose bounding box spans more than 16 tile rows (few, but up to 68 rows each) are queued in LDS and their
// exact per-row tile counts are summed by a whole wavefront (one lane per tile row).  ROUND 1 counts only tiles whose
// bit is set in the unsaturated-tile mask.
// RUNS (span-list binning, below): the chunk also leaves, per tile row, how many of its splats touch the row (runs) and how many
// tiles they touch there, packed `runs | tiles << 9` (at most 256 runs of at most 256 tiles), in row_cnt[row][chunk]; `spine` is
// not written.
template <int ROUND, bool RUNS>
__device__ __forceinline__ void k_project_body(const uint32_t *__restrict__ sorted, const uint4 *__restrict__ splat,
                                               const GsFrameUniforms &u, gsm::Projected *__restrict__ proj, uint2 *__restrict__ rect,
                                               uint32_t *__restrict__ tile_count, uint32_t *__restrict__ spine,
                                               uint32_t *__restrict__ part_vis, const uint32_t *__restrict__ mask,
                                               float *__restrict__ zwin, GsControl *ctl)
{
    __shared__ float s_rec[GS_BLOCK][6];
    __shared__ uint32_t s_rows[GS_BLOCK], s_j[GS_BLOCK];
    __shared__ uint32_t s_nbig, s_nmid, s_vis, s_sum;
    __shared__ uint32_t s_rc[RUNS ? GS_BLOCK : 1];                  // RUNS: the chunk's (runs | tiles << 9) per tile row
    uint32_t *__restrict__ row_cnt = spine;                         // RUNS: the table takes the spine's argument slot
    uint32_t j_lo, j_hi;
    round_range<ROUND>(ctl, u.near_count, j_lo, j_hi);
    // a near-only sort holds positions [V' - P, V') of the order; the positions behind V' are the reference's zero tail
    const bool near_sorted = ctl->near_sorted != 0u;
    const uint32_t n_rec = ctl->n_sorted, j_base = near_sorted ? ctl->n_valid - n_rec : 0u;
    const uint32_t nchunks = (j_hi - j_lo + GS_BLOCK - 1) / GS_BLOCK;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_vis = 0;
    // the frame's completion word starts here (the first kernel of the frame's render; its sort may have left the order incomplete)
    if (ROUND == 0 && blockIdx.x == 0 && threadIdx.x == 0 && u.status) *u.status = ctl->order_incomplete ? 4u : 0u;
    if (ROUND == 0 && blockIdx.x == 0 && u.need_seed && threadIdx.x < GS_NEED_WORDS) ctl->need_near[threadIdx.x] = u.need_seed == 1u ? 0u : u.need_seed;   // (the host's seed: gs_api.hip)
    for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        if (threadIdx.x == 0) { s_sum = 0; s_nbig = 0; s_nmid = 0; }
        if (RUNS) s_rc[threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t j = j_lo + c * GS_BLOCK + threadIdx.x;
        uint32_t count = 0;
        bool queued = false;
        if (j < j_hi) {
            const uint32_t idx = !near_sorted ? sorted[j] : (j - j_base < n_rec ? sorted[j - j_base] : 0u);
            const uint4 cs4 = splat[2 * (size_t)idx], cc4 = splat[2 * (size_t)idx + 1];   // one 32-byte record, one line
            const float cs[4] = { __uint_as_float(cs4.x), __uint_as_float(cs4.y), __uint_as_float(cs4.z), __uint_as_float(cs4.w) };
            const uint32_t cc[4] = { cc4.x, cc4.y, cc4.z, cc4.w };
            gsm::Projected p; gsm::ProjExtra x;
            if (gsm::project_splat(cs, cc, u.mv, u.proj, u.focal, u.vw, u.vh, p, x)) {
                float xmin, xmax, ymin, ymax;
                gsm::splat_pixel_bounds(p, x, xmin, xmax, ymin, ymax);
                // clamp in float (bounds can be far outside the int range), then to the strip / screen
                const float fx0 = fmaxf(xmin, (float)u.x0), fx1 = fminf(xmax, (float)(u.x1b - 1));
                const float fy0 = fmaxf(ymin, 0.0f), fy1 = fminf(ymax, (float)(u.H - 1));
                if (fx0 <= fx1 && fy0 <= fy1) {
                    const int ix0 = (int)fx0, ix1 = (int)fx1, jy0 = (int)fy0, jy1 = (int)fy1;
                    const int r0 = u.H - 1 - jy1, r1 = u.H - 1 - jy0;       // GL rows (y up) -> image rows (top-down)
                    const uint32_t tx0 = (uint32_t)(ix0 - u.x0) / GS_TILE, tx1 = (uint32_t)(ix1 - u.x0) / GS_TILE;
                    const uint32_t ty0 = (uint32_t)r0 / GS_TILE, ty1 = (uint32_t)r1 / GS_TILE;
                    if (!RUNS || ty1 - ty0 >= 2) rect[j] = make_uint2(tx0 | (ty0 << 16), tx1 | (ty1 << 16));   // (RUNS, one or two tile rows: the runs themselves, below)
                    float4 *dst = reinterpret_cast<float4 *>(proj + j);
                    dst[0] = make_float4(p.cx, p.cy, p.ax, p.ay);
                    dst[1] = make_float4(p.bx, p.by, __uint_as_float(p.rgba), p.alpha);
                    if (u.has_depth) zwin[j] = x.zndc * 0.5f + 0.5f;           // gl_FragCoord.z of every fragment of the quad
                    if (ty1 - ty0 >= 2) {
                        // three or more tile rows: counted cooperatively, one lane per row -- by 16-lane groups up to 16 rows
                        // (queued from the front), by a whole wavefront beyond (queued from the back of the same arrays)
                        const uint32_t q = (ty1 - ty0 >= 16) ? (GS_BLOCK - 1u - atomicAdd(&s_nbig, 1u)) : atomicAdd(&s_nmid, 1u);
                        s_rec[q][0] = p.cx; s_rec[q][1] = p.cy; s_rec[q][2] = p.ax; s_rec[q][3] = p.ay; s_rec[q][4] = p.bx; s_rec[q][5] = p.by;
                        s_rows[q] = ty0 | (ty1 << 16); s_j[q] = j;
                        queued = true;
                    } else {
                        // exact coverage: per tile row, the contiguous run of tiles the ellipse touches
                        gsm::EllipseRows e;
                        gsm::ellipse_rows_setup(p, e);
                        uint32_t runs = 0, present = 0;                // RUNS: the (at most two) runs as k_emit_runs wants them
                        for (uint32_t ty = ty0; ty <= ty1; ty++) {
                            uint32_t a, n;
                            gsm::splat_tile_row(p, e, (int)ty, u.H, u.x0, u.x1b, a, n);
                            if (RUNS && n) { runs |= (a | ((n - 1u) << 8)) << (16u * (ty - ty0)); present |= 1u << (ty - ty0); }
                            if (ROUND == 1 && n) n = mask_count(mask + ty * u.mask_words, a, n);
                            if (RUNS && n) atomicAdd(&s_rc[ty], 1u | (n << 9));
                            count += n;
                        }
                        // (span lists: a splat of one or two tile rows hands its runs on -- first tile | tiles - 1 << 8 per row, strips of at most
                        // 256 tile columns -- so that k_emit_runs neither reads its projected record nor repeats the band arithmetic twice: in a
                        // frame of small splats that was 60 % of that kernel's instructions.  Bit 14 says so; tx0 of a queued splat is below 256)
                        if (RUNS) rect[j] = make_uint2((ty0 << 16) | ((ty1 - ty0) << 15) | (1u << 14) | (present << 12), runs);
                    }
                }
            }
            if (!queued) tile_count[j] = count;
        }
        uint32_t vis = count ? 1u : 0u, sum = count;
        __syncthreads();
        const uint32_t nmid = s_nmid;
        for (uint32_t mi = (uint32_t)w * 4u + ((uint32_t)lane >> 4); mi < ((nmid + 15u) & ~15u); mi += 16u) {   // 16 lanes per splat
            uint32_t n = 0;
            if (mi < nmid) {
                gsm::Projected p;
                p.cx = s_rec[mi][0]; p.cy = s_rec[mi][1]; p.ax = s_rec[mi][2]; p.ay = s_rec[mi][3]; p.bx = s_rec[mi][4]; p.by = s_rec[mi][5];
                gsm::EllipseRows e;
                gsm::ellipse_rows_setup(p, e);
                const uint32_t ty = (s_rows[mi] & 0xFFFF) + ((uint32_t)lane & 15u);
                if (ty <= (s_rows[mi] >> 16)) {
                    uint32_t a;
                    gsm::splat_tile_row(p, e, (int)ty, u.H, u.x0, u.x1b, a, n);
                    if (ROUND == 1 && n) n = mask_count(mask + ty * u.mask_words, a, n);
                    if (RUNS && n) atomicAdd(&s_rc[ty], 1u | (n << 9));
                }
            }
            n = row16_sum(n);
            if ((lane & 15) == 0 && mi < nmid) {
                tile_count[s_j[mi]] = n; sum += n; if (n) vis++;
            }
        }
        const uint32_t nbig = s_nbig;
        for (uint32_t bq = w; bq < nbig; bq += 4) {                   // one wavefront per queued splat of more than 16 tile rows
            const uint32_t bi = GS_BLOCK - 1u - bq;
            gsm::Projected p;
            p.cx = s_rec[bi][0]; p.cy = s_rec[bi][1]; p.ax = s_rec[bi][2]; p.ay = s_rec[bi][3]; p.bx = s_rec[bi][4]; p.by = s_rec[bi][5];
            gsm::EllipseRows e;
            gsm::ellipse_rows_setup(p, e);
            const uint32_t ty0 = s_rows[bi] & 0xFFFF, ty1 = s_rows[bi] >> 16;
            uint32_t rsum = 0;
            for (uint32_t ty = ty0 + lane; ty <= ty1; ty += 64) {
                uint32_t a, n;
                gsm::splat_tile_row(p, e, (int)ty, u.H, u.x0, u.x1b, a, n);
                if (ROUND == 1 && n) n = mask_count(mask + ty * u.mask_words, a, n);
                if (RUNS && n) atomicAdd(&s_rc[ty], 1u | (n << 9));
                rsum += n;
            }
            rsum = wave_sum(rsum);
            if (lane == 0) { tile_count[s_j[bi]] = rsum; sum += rsum; if (rsum) vis++; }
        }
        vis = wave_sum(vis); sum = wave_sum(sum);
        if (lane == 0) { if (vis) atomicAdd(&s_vis, vis); if (sum) atomicAdd(&s_sum, sum); }
        __syncthreads();
        if (RUNS) { if (threadIdx.x < (uint32_t)u.tiles_y) row_cnt[(size_t)threadIdx.x * u.rc_stride + c] = s_rc[threadIdx.x]; }
        else if (threadIdx.x == 0) spine[c] = s_sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) part_vis[blockIdx.x] = s_vis;
}

template <int ROUND, bool RUNS>
__global__ __launch_bounds__(GS_BLOCK) void k_project(const uint32_t *__restrict__ sorted, const uint4 *__restrict__ splat,
                                                      GsFrameUniforms u, gsm::Projected *__restrict__ proj, uint2 *__restrict__ rect,
                                                      uint32_t *__restrict__ tile_count, uint32_t *__restrict__ spine,
                                                      uint32_t *__restrict__ part_vis, const uint32_t *__restrict__ mask,
                                                      float *__restrict__ zwin, GsControl *ctl)
{
    k_project_body<ROUND, RUNS>(sorted, splat, u, proj, rect, tile_count, spine, part_vis, mask, zwin, ctl);
}

// One workgroup: exclusive scan of the per-chunk totals (spine) -> chunk base offsets, I = grand total (refused and
// flagged if it does not fit the pair buffers), Vp from the partials, frame accumulators -- and k_emit's extra work items:
// a chunk whose splats touch more than GS_EMIT_PAIRS tiles in total (the nearest, largest splats sit next to each other
// in the sorted order) is written by several workgroups, GS_EMIT_PAIRS pair slots each; (chunk, slice) of every slice
// after a chunk's first goes to `extra`.
#define GS_SPINE_CACHED 8u
template <int ROUND>
__device__ __forceinline__ void k_pairs_check_body(GsControl *ctl, uint32_t pair_cap, uint32_t *__restrict__ spine,
                                                   const uint32_t *__restrict__ part_vis, uint32_t nparts, uint32_t near_count,
                                                   int last_round, uint32_t *__restrict__ mask, uint32_t mask_total_words,
                                                   uint2 *__restrict__ extra)
{
    __shared__ uint32_t s_vis, s_wave[4], s_wave_e[4], s_tv[GS_SPINE_CACHED][GS_BLOCK], s_eb[GS_SPINE_CACHED][GS_BLOCK];
    __shared__ unsigned long long s_total64[4];
    if (threadIdx.x == 0) s_vis = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int round = ROUND;
    uint32_t j_lo, j_hi;
    round_range<ROUND>(ctl, near_count, j_lo, j_hi);
    if (ROUND == 0) for (uint32_t i = threadIdx.x; i < mask_total_words; i += GS_BLOCK) mask[i] = 0u;   // read by blend<0> onwards
    const uint32_t nsp = (j_hi - j_lo + GS_BLOCK - 1) / GS_BLOCK;
    uint32_t v = 0;
    if (nsp) for (uint32_t i = threadIdx.x; i < nparts; i += GS_BLOCK) v += part_vis[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if (lane == 0 && v) atomicAdd(&s_vis, v);
    // spine scan: each thread owns a contiguous slice
    const uint32_t per = (nsp + GS_BLOCK - 1) / GS_BLOCK;
    const uint32_t lo = min(threadIdx.x * per, nsp), hi = min(lo + per, nsp);
    uint32_t s = 0;
    unsigned long long s64 = 0;                                    // the demand itself in 64 bits: a sum past 2^32 must not wrap under pair_cap
    // A short slice is read once -- all its loads in flight together -- and parked in LDS: both sweeps then run as small rolled
    // loops (this is ONE workgroup executing cold code once per frame: every instruction-cache line it touches is a memory
    // round trip, and a sweep that re-reads global memory pays one more per element, serialised by its stores).
    const bool cached = per <= GS_SPINE_CACHED;
    if (cached) {
        uint32_t tv[GS_SPINE_CACHED];
#pragma unroll
        for (uint32_t k = 0; k < GS_SPINE_CACHED; k++) tv[k] = lo + k < hi ? spine[lo + k] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < GS_SPINE_CACHED; k++) s_tv[k][threadIdx.x] = tv[k];
    }
    uint32_t ne = 0;                                                // extra slices of this thread's chunks
#define GS_EXTRA_OF(t) ((t) > GS_EMIT_PAIRS ? ((t) - 1u) / GS_EMIT_PAIRS : 0u)
    for (uint32_t i = lo, k = 0; i < hi; i++, k++) { const uint32_t t = cached ? s_tv[k][threadIdx.x] : spine[i]; s += t; s64 += t; ne += GS_EXTRA_OF(t); }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s64 += ((unsigned long long)__shfl_xor((uint32_t)(s64 >> 32), m, 64) << 32) + __shfl_xor((uint32_t)s64, m, 64);
    if (lane == 0) s_total64[w] = s64;
    const uint32_t inc = wave_incl_scan_u32(s, lane), inc_e = wave_incl_scan_u32(ne, lane);
    if (lane == 63) { s_wave[w] = inc; s_wave_e[w] = inc_e; }
    __syncthreads();
    uint32_t base = 0, total = 0, base_e = 0, total_e = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t t = s_wave[k], te = s_wave_e[k]; if (k < w) { base += t; base_e += te; } total += t; total_e += te; }
    const unsigned long long total64 = s_total64[0] + s_total64[1] + s_total64[2] + s_total64[3];
    if (total64 > 0xFFFFFFFFull) total = 0xFFFFFFFFu;              // saturate: larger than any pair_cap (<= 0xFFFF0000)
    uint32_t run = base + inc - s, q = base_e + inc_e - ne;
    const bool fits = total <= pair_cap;                            // (then at most pair_cap / GS_EMIT_PAIRS extra slices exist)
    for (uint32_t i = lo, k = 0; i < hi; i++, k++) {
        const uint32_t t = cached ? s_tv[k][threadIdx.x] : spine[i];
        spine[i] = run; run += t;
        const uint32_t n2 = fits ? GS_EXTRA_OF(t) : 0u;
        if (cached) s_eb[k][threadIdx.x] = q;
        else for (uint32_t k2 = 0; k2 < n2; k2++) extra[q + k2] = make_uint2(i, k2 + 1u);
        q += n2;
    }
    if (cached && fits) {
        // the extra slices are written with the chunks dealt out round-robin: the heavy chunks are neighbours (the nearest
        // splats), i.e. all in the slices of a few threads, which would write hundreds of entries each
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nsp; i += GS_BLOCK) {
            const uint32_t owner = i / per, k = i % per;
            const uint32_t n2 = GS_EXTRA_OF(s_tv[k][owner]), qb = s_eb[k][owner];
            for (uint32_t k2 = 0; k2 < n2; k2++) extra[qb + k2] = make_uint2(i, k2 + 1u);
        }
    }
#undef GS_EXTRA_OF
    if (threadIdx.x == 0) {
        ctl->vis_total = 0;
        ctl->n_emit_extra = fits ? total_e : 0u;
        if (ROUND == 0) { ctl->n_visible = 0; ctl->n_pairs_frame = 0; ctl->want_frame = 0; }
        else { ctl->unsat_round0 = ctl->unsat_count; if (ctl->unsat_count) ctl->unsat_events += 1; }
        ctl->j_lo = j_lo; ctl->j_hi = j_hi;                          // for k_tile_ranges / k_blend of this round
        ctl->scan_total = total;
        ctl->want_frame = (ctl->want_frame + total < total) ? 0xFFFFFFFFu : ctl->want_frame + total;   // (saturating)
        if (ctl->want_frame > ctl->max_total) ctl->max_total = ctl->want_frame;
        // a round that does not fit, or that follows one of this frame that did not (k_emit wrote nothing then), bins nothing:
        // the frame is re-rendered with larger buffers, and no kernel downstream may walk records that were never written
        if (total > pair_cap) { ctl->pair_overflow = 1; ctl->overflow_sticky = 1; ctl->n_pairs = 0; }
        else if (round == 0) { ctl->pair_overflow = 0; ctl->n_pairs = total; }
        else ctl->n_pairs = ctl->pair_overflow ? 0u : total;
        ctl->n_visible += s_vis; ctl->n_pairs_frame += ctl->n_pairs;
        if (ROUND == 0 && near_count != 0xFFFFFFFFu) ctl->unsat_count = 0;   // counted by blend<0>, read by round 1
        if (last_round) {
            ctl->acc_frames += 1; ctl->acc_sorted += ctl->n_kept; ctl->acc_visible += ctl->n_visible; ctl->acc_pairs += ctl->n_pairs_frame;
        }
    }
}

template <int ROUND>
__global__ __launch_bounds__(GS_BLOCK) void k_pairs_check(GsControl *ctl, uint32_t pair_cap, uint32_t *__restrict__ spine,
                                                          const uint32_t *__restrict__ part_vis, uint32_t nparts, uint32_t near_count,
                                                          int last_round, uint32_t *__restrict__ mask, uint32_t mask_total_words,
                                                          uint2 *__restrict__ extra)
{
    k_pairs_check_body<ROUND>(ctl, pair_cap, spine, part_vis, nparts, near_count, last_round, mask, mask_total_words, extra);
}

// (tile id, sorted position) records in splat order: pair slot = spine[chunk] + the in-chunk exclusive scan of tile_count
// + the pair's index inside its splat (tile rows top to bottom, tiles left to right), so no offset array goes to memory.
//
// Work is handed out by PAIRS, not by splats: the tiles-per-splat distribution is extremely skewed (headline scene: 31 000
// visible splats of the first round touch 1.7 M tiles, 10 % of them 60 % of that; a chunk of 256 consecutive splats holds
// 580 pairs on average and 16 000 at the near end), and a splat-per-lane or splat-per-wavefront expansion leaves the
// kernel waiting for the few workgroups that own the near chunks.  An item = GS_EMIT_PAIRS consecutive pair slots of one
// chunk (the first slice of every chunk, plus k_pairs_check's `extra` slices of the heavy ones, taken first).  Per item:
//   1. scan tile_count over the chunk; the splats whose slot range meets the slice are compacted into LDS;
//   2. their tile-row runs (splat, row) -> (first tile, length), GS_EMIT_RUNS at a time, one run per thread slot, by bisection
//      of the row-count scan; exclusive scan of the lengths;
//   3. one thread per pair slot of the slice: the run by bisection of that scan, consecutive threads write consecutive
//      slots.
// ROUND 1 writes only the tiles whose bit is set in the unsaturated-tile mask (a run's length is then its popcount).
// one pair record: 8 bytes (tile, sorted position).  (Rounds 2-5 also had two 4-byte forms -- tile | position, tile | index among the round's
// visible splats -- for frames whose tile and position bits fit 32: since round 6 every frame a 4-byte record could serve is binned with span
// lists, and the records are what is left for strips beyond 4096 pixels: docs/LAB_NOTES.md.)
// exclusive scan of one value per thread over the workgroup (256 threads), total returned in `total`
__device__ __forceinline__ uint32_t block_exscan(uint32_t v, uint32_t *s_w, int lane, int w, uint32_t &total)
{
    const uint32_t inc = wave_incl_scan_u32(v, lane);
    __syncthreads();                                                // (readers of the previous use are done)
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    uint32_t base = 0;
    total = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t t = s_w[k]; if (k < w) base += t; total += t; }
    return base + inc - v;
}

// tile column of the kth (0-based) unsaturated tile among [t0, t0 + n) of one tile row
__device__ __forceinline__ uint32_t nth_masked_tile(const uint32_t *__restrict__ mask_row, uint32_t t0, uint32_t n, uint32_t kth)
{
    for (uint32_t w = t0 >> 5; w <= ((t0 + n - 1u) >> 5); w++) {
        uint32_t bits = mask_bits(mask_row, w, t0, n);
        const uint32_t c = (uint32_t)__popc(bits);
        if (kth < c) { while (kth--) bits &= bits - 1u; return w * 32u + (uint32_t)__ffs(bits) - 1u; }
        kth -= c;
    }
    return t0;                                                      // (not reached: kth < the run's popcount)
}

template <int ROUND>
__device__ __forceinline__ void k_emit_body(const gsm::Projected *__restrict__ proj, const uint2 *__restrict__ rect,
                                            const uint32_t *__restrict__ tile_count, const uint32_t *__restrict__ spine,
                                            const uint2 *__restrict__ extra, const GsFrameUniforms &u, uint2 *__restrict__ pairs,
                                            const uint32_t *__restrict__ mask, const GsControl *ctl)
{
    __shared__ float s_rec[GS_BLOCK][6];                            // the slice's splats: projected record,
    __shared__ uint32_t s_sp[GS_BLOCK], s_ty[GS_BLOCK];             // index in the chunk (or among the round's visible splats), first | last tile row,
    __shared__ uint32_t s_rb[GS_BLOCK + 1];                         // exclusive scan of their tile-row counts
    __shared__ uint32_t s_rt[GS_EMIT_RUNS], s_rx[GS_EMIT_RUNS];     // a batch of runs: first tile | splat << 24, exclusive scan of lengths
    __shared__ uint32_t s_rn[ROUND == 1 ? GS_EMIT_RUNS : 1];        // ROUND 1: the run's unmasked length
    __shared__ uint32_t s_w[4], s_first;
    if (ctl->pair_overflow) return;
    const uint32_t j_lo = ctl->j_lo, j_hi = ctl->j_hi;               // set by k_pairs_check of this round
    const uint32_t nchunks = (j_hi - j_lo + GS_BLOCK - 1) / GS_BLOCK, n_extra = ctl->n_emit_extra;
    const uint32_t tiles_x = (uint32_t)u.tiles_x;
    const uint32_t tid = threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t item = blockIdx.x; item < n_extra + nchunks; item += gridDim.x) {
        uint32_t c = item - n_extra, sub = 0;
        if (item < n_extra) { const uint2 e = extra[item]; c = e.x; sub = e.y; }
        const uint32_t base = spine[c];
        const uint32_t j = j_lo + c * GS_BLOCK + tid;
        // one memory round trip for everything the item may need: count, rectangle and record of every position of the
        // chunk (the two latter are only meaningful -- and only used -- where the count is not zero)
        uint32_t cnt = 0;
        uint2 rc = make_uint2(0u, 0u);
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb4 = ra;
        if (j < j_hi) {
            cnt = tile_count[j]; rc = rect[j];
            const float4 *src = reinterpret_cast<const float4 *>(proj + j);
            ra = src[0]; rb4 = src[1];
        }
        uint32_t n_c;
        const uint32_t ex = block_exscan(cnt, s_w, lane, w, n_c);
        const uint32_t q0 = sub * GS_EMIT_PAIRS;
        if (q0 >= n_c) continue;                                     // (uniform) a chunk without pairs
        const uint32_t vid = c * GS_BLOCK + tid;                        // position in the round
        const uint32_t q1 = n_c - q0 > GS_EMIT_PAIRS ? q0 + GS_EMIT_PAIRS : n_c;
        const bool needed = cnt != 0u && ex < q1 && ex + cnt > q0;
        const uint32_t nr = needed ? (rc.y >> 16) - (rc.x >> 16) + 1u : 0u;
        uint32_t K, R;
        const uint32_t k = block_exscan(needed ? 1u : 0u, s_w, lane, w, K);
        const uint32_t rb = block_exscan(nr, s_w, lane, w, R);
        if (needed) {
            s_rec[k][0] = ra.x; s_rec[k][1] = ra.y; s_rec[k][2] = ra.z; s_rec[k][3] = ra.w; s_rec[k][4] = rb4.x; s_rec[k][5] = rb4.y;
            s_sp[k] = vid; s_ty[k] = (rc.x >> 16) | (rc.y & 0xFFFF0000u); s_rb[k] = rb;   // (vid: position in the round, or visible index)
            if (k == 0) s_first = ex;
        }
        __syncthreads();
        const uint32_t o_first = s_first;                             // chunk-relative slot of the first compacted splat's first pair
        const uint32_t ktop = K > 1u ? 1u << (31 - __clz((int)(K - 1u))) : 0u;   // first bisection step over K compacted splats
        uint32_t xc = 0;                                              // pairs of the compacted splats before this batch of runs
        for (uint32_t rb0 = 0; rb0 < R; rb0 += GS_EMIT_RUNS) {
            // (the bisections of a thread's runs, and of its pairs below, are interleaved step by step: each step is one LDS
            // round trip, and four independent ones in flight cost about as much as one)
            constexpr uint32_t RPT = GS_EMIT_RUNS / GS_BLOCK;         // runs per thread and pass
            uint32_t nm[RPT], kk[RPT], tsum = 0;
#pragma unroll
            for (uint32_t i = 0; i < RPT; i++) kk[i] = 0;             // the largest kk with s_rb[kk] <= r
            for (uint32_t step = ktop; step; step >>= 1) {
#pragma unroll
                for (uint32_t i = 0; i < RPT; i++) {
                    const uint32_t r = rb0 + tid * RPT + i, t = kk[i] + step;
                    if (t < K && s_rb[t] <= r) kk[i] = t;
                }
            }
#pragma unroll
            for (uint32_t i = 0; i < RPT; i++) {
                const uint32_t idx = tid * RPT + i, r = rb0 + idx, k1 = kk[i];
                nm[i] = 0;
                if (r < R) {
                    const uint32_t row = (s_ty[k1] & 0xFFFFu) + (r - s_rb[k1]);
                    gsm::Projected p;
                    p.cx = s_rec[k1][0]; p.cy = s_rec[k1][1]; p.ax = s_rec[k1][2]; p.ay = s_rec[k1][3]; p.bx = s_rec[k1][4]; p.by = s_rec[k1][5];
                    gsm::EllipseRows e;
                    gsm::ellipse_rows_setup(p, e);
                    uint32_t t0, n;
                    gsm::splat_tile_row(p, e, (int)row, u.H, u.x0, u.x1b, t0, n);
                    nm[i] = (ROUND == 1 && n) ? mask_count(mask + row * u.mask_words, t0, n) : n;
                    s_rt[idx] = (row * tiles_x + t0) | (k1 << 24);
                    if (ROUND == 1) s_rn[idx] = n;
                }
                tsum += nm[i];
            }
            uint32_t pb;
            uint32_t x = block_exscan(tsum, s_w, lane, w, pb);
#pragma unroll
            for (uint32_t i = 0; i < RPT; i++) { s_rx[tid * RPT + i] = x; x += nm[i]; }
            __syncthreads();
            // the batch's pairs 0 .. pb-1 sit at the chunk-relative slots s0 .. s0+pb-1; this item writes those inside [q0, q1)
            const uint32_t s0 = o_first + xc;
            const uint32_t p_lo = q0 > s0 ? q0 - s0 : 0u, p_hi = q1 > s0 ? (q1 - s0 < pb ? q1 - s0 : pb) : 0u;
            const uint32_t nrun = R - rb0 < GS_EMIT_RUNS ? R - rb0 : GS_EMIT_RUNS;
            const uint32_t rtop = 1u << (31 - __clz((int)nrun));      // (the slots past the batch's runs hold pb: never taken)
            constexpr uint32_t PPT = 4;                               // pairs per thread and pass
            for (uint32_t pp = p_lo + tid; pp < p_hi; pp += PPT * GS_BLOCK) {
                uint32_t run[PPT];                                     // the largest run with s_rx[run] <= p (it has a positive length)
#pragma unroll
                for (uint32_t i = 0; i < PPT; i++) run[i] = 0;
                for (uint32_t step = rtop; step; step >>= 1) {
#pragma unroll
                    for (uint32_t i = 0; i < PPT; i++) { const uint32_t t = run[i] + step; if (t < GS_EMIT_RUNS && s_rx[t] <= pp + i * GS_BLOCK) run[i] = t; }
                }
#pragma unroll
                for (uint32_t i = 0; i < PPT; i++) {
                    const uint32_t p = pp + i * GS_BLOCK;
                    if (p >= p_hi) break;
                    const uint32_t rt = s_rt[run[i]], k1 = rt >> 24, first = rt & 0xFFFFFFu, within = p - s_rx[run[i]];
                    uint32_t tile = first + within;
                    if (ROUND == 1) {
                        const uint32_t row = first / tiles_x, t0 = first % tiles_x;
                        tile = row * tiles_x + nth_masked_tile(mask + row * u.mask_words, t0, s_rn[run[i]], within);
                    }
                    const uint32_t jrel = s_sp[k1];
                    pairs[base + s0 + p] = make_uint2(tile, j_lo + jrel);
                }
            }
            xc += pb;
            if (s0 + pb >= q1) break;                                  // (uniform) the rest lies behind the slice
            __syncthreads();                                          // the run tables are rewritten by the next batch
        }
        __syncthreads();                                              // the splat tables are rewritten by the next item
    }
}

template <int ROUND>
__global__ __launch_bounds__(GS_BLOCK) void k_emit(const gsm::Projected *__restrict__ proj, const uint2 *__restrict__ rect,
                                                   const uint32_t *__restrict__ tile_count, const uint32_t *__restrict__ spine,
                                                   const uint2 *__restrict__ extra, GsFrameUniforms u, uint2 *__restrict__ pairs,
                                                   const uint32_t *__restrict__ mask, const GsControl *ctl)
{
    k_emit_body<ROUND>(proj, rect, tile_count, spine, extra, u, pairs, mask, ctl);
}

// [start,end) of every tile in the sorted pair list, written for ALL tiles (empty ones get an empty range at the
// position where they would be), so no clearing pass is needed.
__device__ __forceinline__ void k_tile_ranges_body(const uint2 *__restrict__ p64, uint2 *__restrict__ range,
                                                   uint32_t ntiles, int round, const GsControl *ctl)
{
#define GS_PAIR_TILE(i) (p64[i].x)
    if (round == 1 && ctl->j_hi == 0) return;                      // nothing left for round 1: its blend returns too
    const uint32_t I = ctl->n_pairs;
    if (I == 0) {
        for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < ntiles; t += gridDim.x * blockDim.x) range[t] = make_uint2(0u, 0u);
        return;
    }
    // the empty tiles before the first and after the last listed tile (a cutout scene leaves thousands): by the whole grid
    const uint32_t first = GS_PAIR_TILE(0), last = GS_PAIR_TILE(I - 1);
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < ntiles; t += gridDim.x * blockDim.x) {
        if (t < first) range[t] = make_uint2(0u, 0u);
        else if (t > last) range[t] = make_uint2(I, I);
    }
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < I; p += gridDim.x * blockDim.x) {
        const uint32_t k = GS_PAIR_TILE(p);
        if (p == 0) {
            range[k].x = 0;
        } else {
            const uint32_t kp = GS_PAIR_TILE(p - 1);
            if (kp != k) {
                range[kp].y = p;
                for (uint32_t t = kp + 1; t < k; t++) range[t] = make_uint2(p, p);
                range[k].x = p;
            }
        }
        if (p == I - 1) range[k].y = I;
    }
}
#undef GS_PAIR_TILE

__global__ __launch_bounds__(GS_BLOCK) void k_tile_ranges(const uint2 *__restrict__ pairs, uint2 *__restrict__ range,
                                                          uint32_t ntiles, int round, const GsControl *ctl)
{
    k_tile_ranges_body(pairs, range, ntiles, round, ctl);
}

// ================================================================= span-list binning (GS_OPT_BINNING; round 4)
// What the fixed-function rasteriser did for the reference (index.js:52-66, 158-163), as a scan-line rasteriser does it: a splat
// meets a tile row in ONE run of tiles (exact coverage, above), so a splat is first turned into its runs -- (first tile, length)
// per tile row it touches -- and the runs of a tile row, kept in sorted-splat order, are then expanded into the row's per-tile
// lists by one thread per tile COLUMN: thread x walks the row's runs in order and appends the splat to its list whenever the run
// covers column x.  A column's appends happen in run order, i.e. in draw order: the lists come out stable without a single key
// being sorted, and there are no (tile, splat) records before the lists themselves.  Against the pair records + two stable radix
// passes of rounds 1-3 (hist / scan / scatter twice + emit + ranges = 8 launches, ~20 bytes of traffic per pair) a binning round
// is 4 launches, and the work is per RUN (a tenth of the pairs in the headline scene), not per pair.
//   k_project<.., RUNS>  also counts, per 256-splat chunk and tile row, the chunk's runs and tiles:        row_cnt[row][chunk]
//   k_row_scan           one workgroup per tile row: row_cnt[row][.] <- runs before the chunk; (runs, tiles) of the row: row_tot
//   k_emit_runs          per chunk: the runs again, each written to ITS ROW's segment of the run arrays at
//                        start(row) + runs before the chunk + rank inside the chunk.  The rank -- how many earlier positions of
//                        the chunk touch the same row -- comes from a 256-bit occupancy word per tile row that the chunk's
//                        threads OR their bit into: order-free to build, a popcount to read
//   k_seg_count          item = (tile row, segment of its runs): the segment's difference array over the tile columns (+1 at a run's
//                        first column, -1 behind its last), one row of ints per item
//   k_lists              same items: per-column counts of the runs BEFORE the segment and of ALL the row's runs from the rows of
//                        k_seg_count (added up, one prefix sum) -> the tiles' ranges and each column's write cursor; then the
//                        walk.  Also the round's bookkeeping in the control block (what k_pairs_check does for the pair
//                        records): block 0.
// ROUND 1 counts, ranks and appends only tiles whose bit is set in the unsaturated-tile mask (a run stays one record; its
// columns are filtered by the walk).  Tile lists hold the sorted positions themselves (GsFrameUniforms::rc_stride != 0 says so).
#ifndef GS_LIST_SEG
#define GS_LIST_SEG 256u            // runs a k_lists item walks at least ...
#endif
#ifndef GS_LIST_LOADS
#define GS_LIST_LOADS 16u           // run records a thread of k_seg_count has in flight
#endif
#ifndef GS_SEGC_RUNS_PER_ROW
#define GS_SEGC_RUNS_PER_ROW 4096u  // frames expected to hold more runs than this per tile row count their segments in a launch of their own (k_seg_count)
#endif
#define GS_LIST_SEGS 16u            // ... and a tile row is cut into at most this many items (long rows: longer walks, not more recounts)
__device__ __forceinline__ uint32_t list_seg_len(uint32_t nr)
{
    const uint32_t s = (nr + GS_LIST_SEGS - 1u) / GS_LIST_SEGS;
    return s <= GS_LIST_SEG ? GS_LIST_SEG : ((s + 63u) & ~63u);
}

// sum of one 64-bit value per thread over the workgroup
__device__ __forceinline__ unsigned long long block_sum64(unsigned long long v, unsigned long long *s_p, int lane, int w)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += ((unsigned long long)__shfl_xor((uint32_t)(v >> 32), m, 64) << 32) + __shfl_xor((uint32_t)v, m, 64);
    __syncthreads();
    if (lane == 0) s_p[w] = v;
    __syncthreads();
    return s_p[0] + s_p[1] + s_p[2] + s_p[3];
}

// Prefix sums over the (at most 256) tile rows of a round's (runs, tiles) totals, by the workgroup's FIRST wavefront alone (four
// rows per lane, one shuffle scan per column): first run / first tile of every row, and (ITEMS) the first k_lists item of every
// row.  Every workgroup of k_emit_runs and k_lists needs them; as four workgroup-wide scans they were a quarter of k_lists'
// instructions.  s_*[r] = sum over the rows before r, s_*[tiles_y .. 256] = the totals.  Returns the tiles of all rows (64 bits).
template <bool ITEMS>
__device__ __forceinline__ unsigned long long row_prefixes(const uint2 *__restrict__ row_tot, uint32_t tiles_y, uint32_t *s_rrun, uint32_t *s_rpair,
                                                           uint32_t *s_item, unsigned long long *s_total)
{
    if (threadIdx.x < 64) {
        const uint32_t l = threadIdx.x;
        uint2 rt[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) rt[k] = 4u * l + k < tiles_y ? row_tot[4u * l + k] : make_uint2(0u, 0u);
        uint32_t sr = 0, sp = 0, si = 0, ni[4];
        unsigned long long sp64 = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            ni[k] = (ITEMS && 4u * l + k < tiles_y) ? max(1u, (rt[k].x + list_seg_len(rt[k].x) - 1u) / list_seg_len(rt[k].x)) : 0u;   // (every row has an item: its ranges are written)
            sr += rt[k].x; sp += rt[k].y; sp64 += rt[k].y; si += ni[k];
        }
        uint32_t er = wave_incl_scan_u32(sr, (int)l) - sr, ep = wave_incl_scan_u32(sp, (int)l) - sp, ei = ITEMS ? wave_incl_scan_u32(si, (int)l) - si : 0u;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sp64 += ((unsigned long long)__shfl_xor((uint32_t)(sp64 >> 32), m, 64) << 32) + __shfl_xor((uint32_t)sp64, m, 64);
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            s_rrun[4u * l + k] = er; s_rpair[4u * l + k] = ep; if (ITEMS) s_item[4u * l + k] = ei;
            er += rt[k].x; ep += rt[k].y; ei += ni[k];
        }
        if (l == 63) { s_rrun[GS_BLOCK] = er; s_rpair[GS_BLOCK] = ep; if (ITEMS) s_item[GS_BLOCK] = ei; *s_total = sp64; }
    }
    __syncthreads();
    return *s_total;
}

template <int ROUND>
__device__ __forceinline__ void k_row_scan_body(uint32_t *__restrict__ row_cnt, uint2 *__restrict__ row_tot, const GsControl *ctl,
                                                uint32_t near_count, uint32_t rc_stride, uint32_t tiles_y, uint32_t *__restrict__ mask,
                                                uint32_t mask_words)
{
    __shared__ uint32_t s_w[4];
    __shared__ unsigned long long s_p[4];
    uint32_t j_lo, j_hi;
    round_range<ROUND>(ctl, near_count, j_lo, j_hi);
    const uint32_t nch = (j_hi - j_lo + GS_BLOCK - 1) / GS_BLOCK;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t per = (nch + GS_BLOCK - 1) / GS_BLOCK;
    const uint32_t lo = min(threadIdx.x * per, nch), hi = min(lo + per, nch);
    for (uint32_t row = blockIdx.x; row < tiles_y; row += gridDim.x) {
        if (ROUND == 0) for (uint32_t i = threadIdx.x; i < mask_words; i += GS_BLOCK) mask[row * mask_words + i] = 0u;   // read by blend<0> onwards
        uint32_t *__restrict__ rc = row_cnt + (size_t)row * rc_stride;
        uint32_t sr = 0;
        unsigned long long sp = 0;
        for (uint32_t c = lo; c < hi; c++) { const uint32_t v = rc[c]; sr += v & 0x1FFu; sp += v >> 9; }
        uint32_t total;
        uint32_t run = block_exscan(sr, s_w, lane, w, total);
        const unsigned long long tp = block_sum64(sp, s_p, lane, w);
        for (uint32_t c = lo; c < hi; c++) { const uint32_t v = rc[c]; rc[c] = run; run += v & 0x1FFu; }
        if (threadIdx.x == 0) row_tot[row] = make_uint2(total, tp > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)tp);   // (saturated: larger than any pair_cap)
    }
}

template <int ROUND>
__global__ __launch_bounds__(GS_BLOCK) void k_row_scan(uint32_t *__restrict__ row_cnt, uint2 *__restrict__ row_tot, const GsControl *ctl,
                                                       uint32_t near_count, uint32_t rc_stride, uint32_t tiles_y, uint32_t *__restrict__ mask,
                                                       uint32_t mask_words)
{
    k_row_scan_body<ROUND>(row_cnt, row_tot, ctl, near_count, rc_stride, tiles_y, mask, mask_words);
}

// The runs of one chunk of 256 sorted positions, each to its tile row's segment (see above).  The traversal is k_project's:
// splats of one or two tile rows by their own thread, of up to 16 rows by 16 lanes, beyond by a wavefront -- twice, once to build
// the occupancy words and once to write.
template <int ROUND>
__device__ __forceinline__ void k_emit_runs_body(const gsm::Projected *__restrict__ proj, const uint2 *__restrict__ rect,
                                                 const uint32_t *__restrict__ tile_count, const uint32_t *__restrict__ row_cnt,
                                                 const uint2 *__restrict__ row_tot, const GsFrameUniforms &u,
                                                 uint32_t *__restrict__ run_geom, uint32_t *__restrict__ run_ref,
                                                 const uint32_t *__restrict__ mask, const GsControl *ctl, uint32_t pair_cap)
{
    __shared__ float s_rec[GS_BLOCK][6];
    __shared__ uint32_t s_rows[GS_BLOCK], s_t[GS_BLOCK];            // queued splats: first | last << 16 tile row, thread (= position in the chunk)
    __shared__ unsigned long long s_m[GS_BLOCK][4];                 // per tile row: the chunk's positions with a run there
    __shared__ uint32_t s_rowrun[GS_BLOCK + 1], s_rowpair[GS_BLOCK + 1], s_base[GS_BLOCK];
    __shared__ uint32_t s_nbig, s_nmid;
    __shared__ unsigned long long s_p[4];
    const uint32_t tid = threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t tiles_y = (uint32_t)u.tiles_y;
    {
        const unsigned long long I = row_prefixes<false>(row_tot, tiles_y, s_rowrun, s_rowpair, nullptr, s_p);   // first run of every tile row
        // a round that does not fit the buffers (or follows one of this frame that did not) bins nothing: k_lists flags the frame
        if (I > pair_cap || (ROUND == 1 && ctl->pair_overflow)) return;
    }
    uint32_t j_lo, j_hi;
    round_range<ROUND>(ctl, u.near_count, j_lo, j_hi);
    const uint32_t nchunks = (j_hi - j_lo + GS_BLOCK - 1) / GS_BLOCK;
    for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        s_m[tid][0] = 0ull; s_m[tid][1] = 0ull; s_m[tid][2] = 0ull; s_m[tid][3] = 0ull;
        if (tid == 0) { s_nbig = 0; s_nmid = 0; }
        if (tid < tiles_y) s_base[tid] = s_rowrun[tid] + row_cnt[(size_t)tid * u.rc_stride + c];
        __syncthreads();
        const uint32_t j = j_lo + c * GS_BLOCK + tid;
        const uint32_t cnt = j < j_hi ? tile_count[j] : 0u;
        bool own = false;
        uint32_t ty0 = 0, ty1 = 0;
        gsm::Projected p;
        p.cx = p.cy = p.ax = p.ay = p.bx = p.by = 0.0f;
        uint32_t own_runs = 0, own_present = 0;                     // a splat of one or two tile rows: its runs as k_project left them
        if (cnt) {
            const uint2 rc = rect[j];
            if (rc.x & (1u << 14)) {
                own = true;
                ty0 = rc.x >> 16; ty1 = ty0 + ((rc.x >> 15) & 1u);
                own_runs = rc.y; own_present = (rc.x >> 12) & 3u;
            } else {
                const float4 *src = reinterpret_cast<const float4 *>(proj + j);
                const float4 ra = src[0]; const float2 rb = *reinterpret_cast<const float2 *>(src + 1);
                p.cx = ra.x; p.cy = ra.y; p.ax = ra.z; p.ay = ra.w; p.bx = rb.x; p.by = rb.y;
                ty0 = rc.x >> 16; ty1 = rc.y >> 16;
                const uint32_t q = (ty1 - ty0 >= 16) ? (GS_BLOCK - 1u - atomicAdd(&s_nbig, 1u)) : atomicAdd(&s_nmid, 1u);
                s_rec[q][0] = p.cx; s_rec[q][1] = p.cy; s_rec[q][2] = p.ax; s_rec[q][3] = p.ay; s_rec[q][4] = p.bx; s_rec[q][5] = p.by;
                s_rows[q] = ty0 | (ty1 << 16); s_t[q] = tid;
            }
        }
        __syncthreads();
        const uint32_t nmid = s_nmid, nbig = s_nbig;
        // FN(position in the chunk, tile row, first tile, tiles) for every run of the chunk that has a tile to bin
#define GS_RUN_ONE(PP, EE, T, TY, FN) do { uint32_t a_, n_;                                                                       \
            gsm::splat_tile_row(PP, EE, (int)(TY), u.H, u.x0, u.x1b, a_, n_);                                                     \
            if (n_ && (ROUND == 0 || mask_count(mask + (TY) * u.mask_words, a_, n_))) { FN((T), (TY), a_, n_); } } while (0)
#define GS_RUN_PASS(FN) do {                                                                                                      \
            if (own) {                                                                                                            \
                for (uint32_t ty = ty0; ty <= ty1; ty++) if ((own_present >> (ty - ty0)) & 1u) {                                  \
                    const uint32_t r_ = own_runs >> (16u * (ty - ty0)), a_ = r_ & 0xFFu, n_ = ((r_ >> 8) & 0xFFu) + 1u;           \
                    if (ROUND == 0 || mask_count(mask + ty * u.mask_words, a_, n_)) { FN(tid, ty, a_, n_); } } }                  \
            for (uint32_t mi = (uint32_t)w * 4u + ((uint32_t)lane >> 4); mi < nmid; mi += 16u) {                                  \
                gsm::Projected q;                                                                                                 \
                q.cx = s_rec[mi][0]; q.cy = s_rec[mi][1]; q.ax = s_rec[mi][2]; q.ay = s_rec[mi][3]; q.bx = s_rec[mi][4]; q.by = s_rec[mi][5]; \
                gsm::EllipseRows e; gsm::ellipse_rows_setup(q, e);                                                                \
                const uint32_t ty = (s_rows[mi] & 0xFFFFu) + ((uint32_t)lane & 15u);                                              \
                if (ty <= (s_rows[mi] >> 16)) GS_RUN_ONE(q, e, s_t[mi], ty, FN); }                                                \
            for (uint32_t bq = (uint32_t)w; bq < nbig; bq += 4u) {                                                                \
                const uint32_t bi = GS_BLOCK - 1u - bq;                                                                           \
                gsm::Projected q;                                                                                                 \
                q.cx = s_rec[bi][0]; q.cy = s_rec[bi][1]; q.ax = s_rec[bi][2]; q.ay = s_rec[bi][3]; q.bx = s_rec[bi][4]; q.by = s_rec[bi][5]; \
                gsm::EllipseRows e; gsm::ellipse_rows_setup(q, e);                                                                \
                for (uint32_t ty = (s_rows[bi] & 0xFFFFu) + (uint32_t)lane; ty <= (s_rows[bi] >> 16); ty += 64u) GS_RUN_ONE(q, e, s_t[bi], ty, FN); } \
        } while (0)
#define GS_RUN_MARK(T, TY, A, N) atomicOr(&s_m[TY][(T) >> 6], 1ull << ((T) & 63u))
        GS_RUN_PASS(GS_RUN_MARK);
        __syncthreads();
#define GS_RUN_WRITE(T, TY, A, N) do { const uint32_t wq_ = (T) >> 6;                                                             \
            uint32_t rank_ = (uint32_t)__popcll(s_m[TY][wq_] & ((1ull << ((T) & 63u)) - 1ull));                                   \
            for (uint32_t k_ = 0; k_ < wq_; k_++) rank_ += (uint32_t)__popcll(s_m[TY][k_]);                                       \
            const uint32_t slot_ = s_base[TY] + rank_;                                                                            \
            run_geom[slot_] = (A) | ((N) << 16); run_ref[slot_] = j_lo + c * GS_BLOCK + (T); } while (0)
        GS_RUN_PASS(GS_RUN_WRITE);
#undef GS_RUN_WRITE
#undef GS_RUN_MARK
#undef GS_RUN_PASS
#undef GS_RUN_ONE
        __syncthreads();                                            // the tables are rewritten by the next chunk
    }
}

template <int ROUND>
__global__ __launch_bounds__(GS_BLOCK) void k_emit_runs(const gsm::Projected *__restrict__ proj, const uint2 *__restrict__ rect,
                                                        const uint32_t *__restrict__ tile_count, const uint32_t *__restrict__ row_cnt,
                                                        const uint2 *__restrict__ row_tot, GsFrameUniforms u,
                                                        uint32_t *__restrict__ run_geom, uint32_t *__restrict__ run_ref,
                                                        const uint32_t *__restrict__ mask, const GsControl *ctl, uint32_t pair_cap)
{
    k_emit_runs_body<ROUND>(proj, rect, tile_count, row_cnt, row_tot, u, run_geom, run_ref, mask, ctl, pair_cap);
}

// Per item of k_lists -- (tile row, segment of the row's runs) -- the difference array of the segment's runs over the tile columns
// (+1 at a run's first column, -1 behind its last), written as one row of 256 ints.  k_lists then needs no pass over the row's
// runs to know how many runs cover a column before its segment and in the whole row: it adds up the rows of the segments (at most
// GS_LIST_SEGS loads per thread, all in flight) and scans once.  With every k_lists item counting its whole tile row itself the
// work was rows x segments x runs: fine for the headline pose (3 000 runs per row), not for frames of many small splats (the 6 M
// cut-out pose: 17 000 runs per row and 16 segments -- k_lists 55 us, more than the two radix passes it replaced).
template <int ROUND>
__device__ __forceinline__ void k_seg_count_body(const uint32_t *__restrict__ run_geom, const uint2 *__restrict__ row_tot, int *__restrict__ seg_diff,
                                                 const GsFrameUniforms &u, const GsControl *ctl, uint32_t pair_cap)
{
    __shared__ uint32_t s_rrun[GS_BLOCK + 1], s_rpair[GS_BLOCK + 1], s_item[GS_BLOCK + 1];
    __shared__ int s_d[GS_BLOCK + 1];
    __shared__ unsigned long long s_p[4];
    const uint32_t tid = threadIdx.x, tiles_y = (uint32_t)u.tiles_y;
    const unsigned long long I = row_prefixes<true>(row_tot, tiles_y, s_rrun, s_rpair, s_item, s_p);
    if (I > pair_cap || (ROUND == 1 && ctl->pair_overflow)) return;  // (k_lists writes empty ranges and reads none of this)
    const uint32_t NI = s_item[GS_BLOCK];
    for (uint32_t item = blockIdx.x; item < NI; item += gridDim.x) {
        uint32_t row = 0;
#pragma unroll
        for (uint32_t step = GS_BLOCK / 2; step; step >>= 1) { const uint32_t t = row + step; if (t < tiles_y && s_item[t] <= item) row = t; }
        const uint32_t seg = item - s_item[row];
        const uint32_t rb = s_rrun[row], nr = s_rrun[row + 1] - rb;
        const uint32_t S = list_seg_len(nr), s0 = seg * S, s1 = min(nr, s0 + S);
        s_d[tid] = 0;
        if (tid == 0) s_d[GS_BLOCK] = 0;
        __syncthreads();
        for (uint32_t i0 = s0; i0 < s1; i0 += GS_LIST_LOADS * GS_BLOCK) {
            uint32_t g[GS_LIST_LOADS];
#pragma unroll
            for (uint32_t k = 0; k < GS_LIST_LOADS; k++) { const uint32_t i = i0 + k * GS_BLOCK + tid; g[k] = i < s1 ? run_geom[rb + i] : 0u; }
#pragma unroll
            for (uint32_t k = 0; k < GS_LIST_LOADS; k++) {
                const uint32_t i = i0 + k * GS_BLOCK + tid;
                if (i < s1) { const uint32_t t0 = g[k] & 0xFFFFu; atomicAdd(&s_d[t0], 1); atomicSub(&s_d[t0 + (g[k] >> 16)], 1); }
            }
        }
        __syncthreads();
        seg_diff[(size_t)item * GS_BLOCK + tid] = s_d[tid];          // (a -1 at column 256 lies behind every column)
        __syncthreads();
    }
}

template <int ROUND>
__global__ __launch_bounds__(GS_BLOCK) void k_seg_count(const uint32_t *__restrict__ run_geom, const uint2 *__restrict__ row_tot, int *__restrict__ seg_diff,
                                                        GsFrameUniforms u, const GsControl *ctl, uint32_t pair_cap)
{
    k_seg_count_body<ROUND>(run_geom, row_tot, seg_diff, u, ctl, pair_cap);
}

// The tile lists of one segment of one tile row's runs (see above); block 0 also keeps the round's books.
// Per item: (1) the difference arrays of the row's segments (k_seg_count) added up into ONE array of 64-bit words -- low half: every
// segment, high half: the segments before this one -- and one 64-bit scan over the columns gives both counts per column (the sums
// such a scan passes through are counts, never negative: exact); (2) the segment in batches of 256 runs: every run ORs its
// bit into the occupancy word of each column it covers (the words of a column are 256 bits: which runs of the batch cover it),
// and the column's thread appends the batch's runs in bit order -- a column's thread takes as many steps as runs cover it, not
// as many as the batch has.
__device__ __forceinline__ unsigned long long wave_incl_scan_u64(unsigned long long v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long t = ((unsigned long long)__shfl_up((uint32_t)(v >> 32), d, 64) << 32) | __shfl_up((uint32_t)v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

template <int ROUND, bool SEGC>
__device__ __forceinline__ void k_lists_body(const uint32_t *__restrict__ run_geom, const uint32_t *__restrict__ run_ref,
                                             const uint2 *__restrict__ row_tot, const int *__restrict__ seg_diff, uint32_t *__restrict__ lists, uint2 *__restrict__ tile_range,
                                             const GsFrameUniforms &u, const uint32_t *__restrict__ mask, GsControl *ctl, uint32_t pair_cap,
                                             const uint32_t *__restrict__ part_vis, uint32_t nparts, int last_round)
{
    __shared__ uint32_t s_rrun[GS_BLOCK + 1], s_rpair[GS_BLOCK + 1], s_item[GS_BLOCK + 1];   // prefix sums over the tile rows: runs, tiles, items
    __shared__ unsigned long long s_d[GS_BLOCK + 1];                // difference array over the tile columns: all runs of the row | runs before the segment << 32
    __shared__ __attribute__((aligned(16))) unsigned long long s_m[GS_BLOCK][4];                 // per tile column: the runs of the batch that cover it
    __shared__ uint32_t s_r[GS_BLOCK];                              // the batch's sorted positions
    __shared__ uint32_t s_off[GS_BLOCK], s_on[GS_BLOCK];            // per tile column: write cursor at the segment's start, column taken (ROUND 1: unsaturated)
    __shared__ uint32_t s_w[4], s_vis;
    __shared__ unsigned long long s_p[4];
    const uint32_t tid = threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t tiles_x = (uint32_t)u.tiles_x, tiles_y = (uint32_t)u.tiles_y;
    // s_rrun / s_rpair / s_item [r]: runs, tiles (meaningful when the round fits: I <= pair_cap < 2^32), items before tile row r
    const unsigned long long I = row_prefixes<true>(row_tot, tiles_y, s_rrun, s_rpair, s_item, s_p);
    const uint32_t NI = s_item[GS_BLOCK];
    const bool overflow = I > pair_cap || (ROUND == 1 && ctl->pair_overflow);
    uint32_t j_lo, j_hi;
    round_range<ROUND>(ctl, u.near_count, j_lo, j_hi);
    if (blockIdx.x == 0) {
        // the round's bookkeeping (k_pairs_check's, for the pair records)
        if (tid == 0) s_vis = 0;
        __syncthreads();
        uint32_t v = 0;
        if (j_hi > j_lo) for (uint32_t i = tid; i < nparts; i += GS_BLOCK) v += part_vis[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if (lane == 0 && v) atomicAdd(&s_vis, v);
        __syncthreads();
        if (tid == 0) {
            const uint32_t total = I > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)I;
            ctl->vis_total = 0; ctl->n_emit_extra = 0;
            ctl->n_runs = (ROUND == 0 ? 0u : ctl->n_runs) + s_rrun[GS_BLOCK];   // runs of the frame (a hint for the next frames: k_seg_count or not)
            if (ROUND == 0) { ctl->n_visible = 0; ctl->n_pairs_frame = 0; ctl->want_frame = 0; }
            else { ctl->unsat_round0 = ctl->unsat_count; if (ctl->unsat_count) ctl->unsat_events += 1; }
            ctl->j_lo = j_lo; ctl->j_hi = j_hi;                      // blend<1> returns at once when nothing was left for round 1
            ctl->scan_total = total;
            ctl->want_frame = (ctl->want_frame + total < total) ? 0xFFFFFFFFu : ctl->want_frame + total;   // (saturating)
            if (ctl->want_frame > ctl->max_total) ctl->max_total = ctl->want_frame;
            if (total > pair_cap) { ctl->pair_overflow = 1; ctl->overflow_sticky = 1; ctl->n_pairs = 0; }
            else if (ROUND == 0) { ctl->pair_overflow = 0; ctl->n_pairs = total; }
            else ctl->n_pairs = ctl->pair_overflow ? 0u : total;
            ctl->n_visible += s_vis; ctl->n_pairs_frame += ctl->n_pairs;
            if (ROUND == 0 && u.near_count != 0xFFFFFFFFu) ctl->unsat_count = 0;   // counted by blend<0>, read by round 1
            if (last_round) {
                ctl->acc_frames += 1; ctl->acc_sorted += ctl->n_kept; ctl->acc_visible += ctl->n_visible; ctl->acc_pairs += ctl->n_pairs_frame;
            }
        }
    }
    __syncthreads();
    for (uint32_t item = blockIdx.x; item < NI; item += gridDim.x) {
        uint32_t row = 0;                                            // the largest row with s_item[row] <= item
#pragma unroll
        for (uint32_t step = GS_BLOCK / 2; step; step >>= 1) { const uint32_t t = row + step; if (t < tiles_y && s_item[t] <= item) row = t; }
        const uint32_t seg = item - s_item[row];
        const uint32_t rb = s_rrun[row], nr = overflow ? 0u : s_rrun[row + 1] - rb, pb = s_rpair[row];
        const uint32_t S = list_seg_len(nr), s0 = seg * S, s1 = min(nr, s0 + S);
        if (SEGC) {
            // this column's difference-array entries of the row's segments (k_seg_count), all loads in flight: low half = every
            // segment, high half = the segments before this one (a prefix sum of such words is exact: the sums it passes through
            // are counts, never negative)
            const uint32_t it0 = s_item[row], nsg = overflow ? 0u : s_item[row + 1] - it0;
            int dd[GS_LIST_SEGS + 1];
#pragma unroll
            for (uint32_t k = 0; k <= GS_LIST_SEGS; k++) dd[k] = k < nsg ? seg_diff[(size_t)(it0 + k) * GS_BLOCK + tid] : 0;
            long long da = 0, db = 0;
#pragma unroll
            for (uint32_t k = 0; k <= GS_LIST_SEGS; k++) { da += dd[k]; if (k < seg) db += dd[k]; }
            s_d[tid] = (unsigned long long)da + ((unsigned long long)db << 32);
        } else {
            // the item counts its tile row itself: every run of the row into the difference array, all loads in flight (rows of a
            // few thousand runs: cheaper than another launch)
            s_d[tid] = 0ull;
            if (tid == 0) s_d[GS_BLOCK] = 0ull;
            __syncthreads();
            for (uint32_t i0 = 0; i0 < nr; i0 += GS_LIST_LOADS * GS_BLOCK) {
                uint32_t g[GS_LIST_LOADS];
#pragma unroll
                for (uint32_t k = 0; k < GS_LIST_LOADS; k++) { const uint32_t i = i0 + k * GS_BLOCK + tid; g[k] = i < nr ? run_geom[rb + i] : 0u; }
#pragma unroll
                for (uint32_t k = 0; k < GS_LIST_LOADS; k++) {
                    const uint32_t i = i0 + k * GS_BLOCK + tid;
                    if (i < nr) {
                        const uint32_t t0 = g[k] & 0xFFFFu, t1 = t0 + (g[k] >> 16);
                        const unsigned long long one = i < s0 ? 0x100000001ull : 1ull;
                        atomicAdd(&s_d[t0], one); atomicAdd(&s_d[t1], 0ull - one);
                    }
                }
            }
        }
        __syncthreads();
        // runs that cover this thread's column: all of the row (low half) / those before the segment (high half)
        const unsigned long long inc = wave_incl_scan_u64(s_d[tid], lane);
        __syncthreads();
        if (lane == 63) s_p[w] = inc;
        __syncthreads();
        unsigned long long cab = inc;
#pragma unroll
        for (int k = 0; k < 4; k++) if (k < w) cab += s_p[k];
        const uint32_t ca = (uint32_t)cab, cb = (uint32_t)(cab >> 32);
        const bool on = tid < tiles_x && (ROUND == 0 || ((mask[row * u.mask_words + (tid >> 5)] >> (tid & 31u)) & 1u));
        const uint32_t cnt = on ? ca : 0u;
        uint32_t tc;
        const uint32_t first = pb + block_exscan(cnt, s_w, lane, w, tc);
        if (seg == 0 && tid < tiles_x) tile_range[row * tiles_x + tid] = overflow ? make_uint2(0u, 0u) : make_uint2(first, first + cnt);
        // A column's batch is walked by G threads (G = 2 while the strip has at most 128 tile columns: 1920 pixels), each taking a share
        // of the batch's occupancy words, four runs per step: what bounds an item is the dependent chain bit -> position -> store of the
        // busiest column (a batch of near, screen-wide splats covers every column 256 times).
        const uint32_t G = tiles_x <= 128u ? 2u : 1u, col = G == 2u ? (tid & 127u) : tid, grp = G == 2u ? (tid >> 7) : 0u;
        const uint32_t q_lo = grp * (4u / G), q_hi = q_lo + 4u / G;
        s_off[tid] = first + cb;
        s_on[tid] = on ? 1u : 0u;
        uint32_t gn = 0, rn = 0;                                     // the next batch's records: in flight while this one is walked
        if (s0 + tid < s1) { gn = run_geom[rb + s0 + tid]; rn = run_ref[rb + s0 + tid]; }
        __syncthreads();
        uint32_t off = s_off[col];
        const bool mine = col < tiles_x && s_on[col] != 0u;
        for (uint32_t b0 = s0; b0 < s1; b0 += GS_BLOCK) {
            const uint32_t g = gn;
            s_r[tid] = rn;
            s_m[tid][0] = 0ull; s_m[tid][1] = 0ull; s_m[tid][2] = 0ull; s_m[tid][3] = 0ull;
            __syncthreads();
            gn = 0;
            if (b0 + GS_BLOCK + tid < s1) { gn = run_geom[rb + b0 + GS_BLOCK + tid]; rn = run_ref[rb + b0 + GS_BLOCK + tid]; }
            {
                // word w of every column belongs to wavefront w's 64 runs.  (Measured and dropped: a wavefront that holds long runs taking
                // the columns one by one -- a ballot IS the column's word, no atomics on one address --: 21 -> 25 us alone on the headline
                // frame, 72 -> 113 us where tiles do not saturate; the same-address atomics are not what an item waits for.)
                const uint32_t t0 = g & 0xFFFFu, t1 = t0 + (g >> 16);   // (no run: t0 = t1 = 0)
                const unsigned long long bit = 1ull << (tid & 63u);
                for (uint32_t c = t0; c < t1; c++) atomicOr(&s_m[c][w], bit);
            }
            __syncthreads();
            if (mine) {
                const ulonglong2 m01 = *reinterpret_cast<const ulonglong2 *>(&s_m[col][0]), m23 = *reinterpret_cast<const ulonglong2 *>(&s_m[col][2]);
                const unsigned long long mw[4] = { m01.x, m01.y, m23.x, m23.y };
                uint32_t o = off;
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    unsigned long long bits = mw[q];
                    const uint32_t nq = (uint32_t)__popcll(bits);
                    if (q >= q_lo && q < q_hi) {
                        for (uint32_t k = 0; k < nq; k += 4u) {
                            uint32_t ix[4];
#pragma unroll
                            for (int e = 0; e < 4; e++) { ix[e] = bits ? (uint32_t)__ffsll((long long)bits) - 1u : 0u; bits &= bits - 1ull; }
                            uint32_t rr[4];
#pragma unroll
                            for (int e = 0; e < 4; e++) rr[e] = s_r[q * 64u + ix[e]];
                            const uint32_t left = nq - k;
                            lists[o + k] = rr[0];
                            if (left > 1u) lists[o + k + 1u] = rr[1];
                            if (left > 2u) lists[o + k + 2u] = rr[2];
                            if (left > 3u) lists[o + k + 3u] = rr[3];
                        }
                    }
                    o += nq;
                }
                off = o;
            }
            __syncthreads();
        }
    }
}

template <int ROUND, bool SEGC>
__global__ __launch_bounds__(GS_BLOCK) void k_lists(const uint32_t *__restrict__ run_geom, const uint32_t *__restrict__ run_ref,
                                                    const uint2 *__restrict__ row_tot, const int *__restrict__ seg_diff, uint32_t *__restrict__ lists, uint2 *__restrict__ tile_range,
                                                    GsFrameUniforms u, const uint32_t *__restrict__ mask, GsControl *ctl, uint32_t pair_cap,
                                                    const uint32_t *__restrict__ part_vis, uint32_t nparts, int last_round)
{
    k_lists_body<ROUND, SEGC>(run_geom, run_ref, row_tot, seg_diff, lists, tile_range, u, mask, ctl, pair_cap, part_vis, nparts, last_round);
}

// Fragment shader + blend for one 16x16 tile, ONE wavefront per tile, four horizontally adjacent pixels per lane
// (lane l: image row l/4 of the tile, pixels 4*(l%4) .. +3).  The projected records of a batch are staged in LDS and
// broadcast-read by the whole wave: with one pixel per lane the 4 waves of a 256-thread tile each re-read every record
// and the kernel is LDS-bandwidth-bound (2 x ds_read_b128 = 8 LDS cycles per record per wave against ~20 VALU cycles);
// four pixels per lane amortise each record read over 4x the arithmetic, and the row-shared terms dy*ay, dy*by are
// computed once -- the expression tree per pixel is unchanged (frag_power), so coverage stays bit-identical.
// Traversal is FRONT-to-back (the list is back-to-front) with a per-pixel transmittance accumulator; a lane leaves the
// list once its four pixels all have T < t_eps, the wave when every lane has (ballot); one rounding to RGBA8 at the end.
// ROUND 0 starts from (T = 1, C = 0); a tile that is not saturated when its list ends saves its per-pixel state and
// sets its bit in the tile mask (if a round 1 follows).  ROUND 1 runs only for masked tiles and resumes from the state.
#ifndef GS_BLEND_BATCH
#define GS_BLEND_BATCH 64
#endif
#ifndef GS_BLEND_FACTOR
#define GS_BLEND_FACTOR 1           // the coverage test of plain frames' whole-batch walk as a factor (GS_BLEND_APPLY_W)
#endif
typedef float f2 __attribute__((ext_vector_type(2)));            // two pixels per packed-fp32 instruction (v_pk_*_f32)
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
// fma2(a, (f2)(b.x), (f2)(c.x)) / fma2(a, (f2)(b.y), (f2)(c.y)): both halves of the result take the LOW / HIGH words of b and c
__device__ __forceinline__ f2 fma2_lo12(f2 a, f2 b, f2 c)
{
    f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ f2 fma2_hi12(f2 a, f2 b, f2 c)
{
    f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// clamp(a * b + c) to [0, 1], both halves (v_pk_fma_f32 ... clamp)
__device__ __forceinline__ f2 fma2_clamp(f2 a, f2 b, f2 c)
{
    f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// fma2((f2)(a.y), b, c): both halves of the result take the HIGH word of a (v_pk_fma_f32 op_sel:[1,0,0])
__device__ __forceinline__ f2 fma2_hi0(f2 a, f2 b, f2 c)
{
    f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}


// ---- sub-tile lists (GS_OPT_SUBTILE; round 6).  One wavefront still owns a 16x16 tile, but a list entry that covers only a corner
// of it -- the camera outside the cloud: the median splat is 13 pixels across, a tile's entry passes the coverage test in 19 % of
// the tile's 256 pixels -- no longer costs all 64 lanes the entry's ~45 VALU instructions.  The tile is cut into sixteen 4x4-pixel
// blocks, each the four pixels of four lanes (lane l draws row l/4, columns 4 (l%4) .. +3: block (l%4, l/16) = lanes 16 br + 4 k
// + bc), and every staged batch of 64 entries is split into sixteen lists, one per block: the lane that stages entry s works out
// from the entry's own record which blocks its |p| <= 2 ellipse can reach (the binning's exact band arithmetic, gsm::ellipse_band_
// xrange, over the tile's four 4-row bands: a superset of the pixels that pass q <= 4, index.js:171-172), sixteen ballots turn the
// lanes' masks into the blocks' lists, and the four lanes of a block walk THEIR list -- the wave takes as many steps as the
// longest of the sixteen lists is long instead of one per entry.  An entry left out of a block's list would have failed the
// coverage test in each of the block's pixels (e = 0: T and the colours unchanged), so every pixel still sees the same operations
// in the same order: frames are bit-identical to the whole-tile walk's, fragment counts included.  A batch whose longest list is
// nearly the batch (large splats: the headline pose) is walked whole, as before.
#ifndef GS_SUBTILE_KEEP_NUM
#define GS_SUBTILE_KEEP_NUM 13u     // a batch is split when its longest block list is at most 13/16 of it ...
#endif
#define GS_SUBTILE_STRIDE 68u       // bytes per block list: 64 entries + the inert pair behind them, banks 17 g + k / 4 apart
#define GS_SUBTILE_INERT 64u        // the slot of the record no pixel passes
#define GS_SUBTILE_BANDS 4          // row bands per tile: 4x4-pixel blocks, sixteen lists (eight lists of 4 x 8 pixels: the same frames/s, docs/LAB_NOTES.md)
#define GS_SUBTILE_GROUPS (4 * GS_SUBTILE_BANDS)
#define GS_SUBTILE_ROWS (16 / GS_SUBTILE_BANDS)
__device__ __forceinline__ uint32_t subtile_mask(const float4 ra, const float bx, const float by, const float tile_x0, const int r0, const int H)
{
    gsm::Projected p;
    p.cx = ra.x; p.cy = ra.y; p.ax = ra.z; p.ay = ra.w; p.bx = bx; p.by = by;
    gsm::EllipseRows e;
    gsm::ellipse_rows_setup(p, e);
    // half height of the ellipse, as ellipse_rows_setup's half width: 2 sqrt(A / D)
    const float hh = 2.0f * gsm::fast_sqrt((0.25f * e.D4A) * gsm::fast_rcp(e.D));
    if (!(hh >= 0.0f) || !(e.pad >= 0.0f)) return (1u << GS_SUBTILE_GROUPS) - 1u;   // (no such record leaves k_project; whole tile if one did)
    uint32_t m = 0;
#pragma unroll 1
    for (int b = 0; b < GS_SUBTILE_BANDS; b++) {
        // image rows r0 + ROWS b .. + ROWS - 1 (top-down); GL pixel-centre y of image row r is (H - 1 - r) + 0.5
        const float ya = ((float)(H - 1 - (r0 + GS_SUBTILE_ROWS * b + GS_SUBTILE_ROWS - 1)) + 0.5f) - p.cy - e.pad, yb = ((float)(H - 1 - (r0 + GS_SUBTILE_ROWS * b)) + 0.5f) - p.cy + e.pad;
        if (ya <= hh && yb >= -hh) {
            float xmin, xmax;
            gsm::ellipse_band_xrange(e, ya, yb, xmin, xmax);
            const float fi0 = fmaxf(ceilf(p.cx + xmin - e.pad - 0.5f), tile_x0), fi1 = fminf(floorf(p.cx + xmax + e.pad - 0.5f), tile_x0 + 15.0f);
            if (fi0 <= fi1) {
                const uint32_t c0 = (uint32_t)(fi0 - tile_x0) >> 2, c1 = (uint32_t)(fi1 - tile_x0) >> 2;
                m |= (((2u << c1) - 1u) & ~((1u << c0) - 1u)) << (4 * b);
            }
        }
    }
    return m;
}

// SCENE: the opaque scene's depth buffer (fragment kept iff its window depth <= the buffer: depthTest LEQUAL,
// depthWrite off, index.js:179-180) and/or colour image (the destination the splats are blended over).
// SUB: with the sub-tile lists (GS_OPT_SUBTILE; a kernel of its own: the lists cost 20 vector registers, i.e. a wave per SIMD)
template <bool COUNT, int ROUND, bool SCENE, bool SUB>
__device__ __forceinline__ void k_blend_body(const uint2 *__restrict__ tile_range, const void *__restrict__ pairs,
                                             const gsm::Projected *__restrict__ proj, const GsFrameUniforms &u,
                                             uint8_t *__restrict__ out, float4 *__restrict__ state, uint32_t *__restrict__ mask,
                                             const float *__restrict__ zwin, const float *__restrict__ scene_depth,
                                             const uint32_t *__restrict__ scene_rgba, GsControl *ctl)
{
    // one batch of list entries (+1 inert slot), 48 bytes each: the projected record's geometry (cx, cy, ax, ay | bx, by, -, -)
    // and its colour converted once per record (rgb8 * alpha / 255, alpha) -- one LDS base address serves all three reads
    __shared__ float4 s_ent[3 * (GS_BLEND_BATCH + 1)];
    __shared__ float s_z[GS_BLEND_BATCH + 2];                    // their window depths (SCENE only)
    __shared__ __attribute__((aligned(16))) uint8_t s_list[SUB ? GS_SUBTILE_GROUPS * GS_SUBTILE_STRIDE : 16];   // GS_OPT_SUBTILE: per 4x4-pixel block, the batch's entries (slots) that can reach it
    const int lane = threadIdx.x;
    // (a round whose records did not fit bins nothing and this kernel draws the background: the completion word says so)
    if (blockIdx.x == 0 && lane == 0 && u.status && ctl->pair_overflow) atomicOr(u.status, 2u);
    if (ROUND == 1 && ctl->j_hi == 0) return;                      // every tile saturated in round 0
    static_assert(GS_BLEND_BATCH == 64, "the sub-tile lists name a batch's entries by the lane that staged them");
    if (SUB && lane == 0) {                                  // the record no pixel passes, behind every batch (staging never writes slot 64)
        if (SCENE) s_z[GS_SUBTILE_INERT] = 0.0f;
        s_ent[3 * GS_SUBTILE_INERT] = make_float4(-1.0e9f, -1.0e9f, 1.0f, 1.0f); s_ent[3 * GS_SUBTILE_INERT + 1] = make_float4(1.0f, 1.0f, 0.0f, 0.0f);
        s_ent[3 * GS_SUBTILE_INERT + 2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    const uint32_t ntiles = (uint32_t)u.tiles_x * (uint32_t)u.tiles_y;
    const bool span = u.rc_stride != 0u;                          // the lists hold sorted positions (span lists) / (tile, position) records
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {   // round 0: one tile per wave; round 1: small grid
    const uint32_t tx = tile % (uint32_t)u.tiles_x, ty = tile / (uint32_t)u.tiles_x;
    if (ROUND == 1 && !((mask[ty * u.mask_words + (tx >> 5)] >> (tx & 31)) & 1u)) continue;   // this tile is final already
    const int xb = u.x0 + (int)tx * GS_TILE + (lane & 3) * 4;      // first of this lane's 4 pixels
    const int r = (int)ty * GS_TILE + (lane >> 2);                 // image row, 0 = top
    const bool row_in = r < u.H;
    const float fy = (float)(u.H - 1 - r) + 0.5f;                  // pixel centre, GL window coordinates
    const bool no_early = (COUNT && !(u.flags & GS_RENDER_COUNT_EVALUATED)) || (u.flags & GS_RENDER_NO_EARLY_OUT);
    const float t_eps = no_early ? -1.0f : u.t_eps;                // T < t_eps never holds when early-out is off
    // per pixel pair: x centre, transmittance, premultiplied colour (alpha is 1 - T)
    f2 fxA = { (float)xb + 0.5f, (float)(xb + 1) + 0.5f }, fxB = { (float)(xb + 2) + 0.5f, (float)(xb + 3) + 0.5f };
    // qm: coverage threshold, 4 inside the strip (fragment kept iff q <= 4, index.js:172), -1 for pixels outside it (never
    // covered, never written; their T starts at 0 so that they do not keep the lane alive)
    // (x1b: the strip's right edge rounded up to the lane's group of 4 pixels, inside the frame: those pixels are blended --
    // not written -- so that the lane leaves its list exactly where it does when the whole frame is drawn)
    const f2 qmA = { (row_in && xb < u.x1b) ? 4.0f : -1.0f, (row_in && xb + 1 < u.x1b) ? 4.0f : -1.0f };
    const f2 qmB = { (row_in && xb + 2 < u.x1b) ? 4.0f : -1.0f, (row_in && xb + 3 < u.x1b) ? 4.0f : -1.0f };
    f2 TA = { qmA.x > 0.0f ? 1.0f : 0.0f, qmA.y > 0.0f ? 1.0f : 0.0f }, TB = { qmB.x > 0.0f ? 1.0f : 0.0f, qmB.y > 0.0f ? 1.0f : 0.0f };
    f2 crA = { 0, 0 }, crB = { 0, 0 }, cgA = { 0, 0 }, cgB = { 0, 0 }, cbA = { 0, 0 }, cbB = { 0, 0 };
    // opaque scene depth under each of the lane's 4 pixels (+inf = nothing in front of the far plane)
    float zb0 = 3.0e38f, zb1 = 3.0e38f, zb2 = 3.0e38f, zb3 = 3.0e38f;
    if (SCENE && u.has_depth && row_in) {
        const float *zr = scene_depth + (size_t)r * u.W + xb;
        if (xb < u.x1b) zb0 = zr[0];
        if (xb + 1 < u.x1b) zb1 = zr[1];
        if (xb + 2 < u.x1b) zb2 = zr[2];
        if (xb + 3 < u.x1b) zb3 = zr[3];
    }
    float4 *st = state + ((size_t)tile * 64 + lane) * 4;           // 4 x float4 per lane: T, r, g, b of its 4 pixels
    if (ROUND == 1) {
        const float4 t = st[0], c0 = st[1], c1 = st[2], c2 = st[3];
        TA = (f2){ t.x, t.y }; TB = (f2){ t.z, t.w };
        crA = (f2){ c0.x, c0.y }; crB = (f2){ c0.z, c0.w }; cgA = (f2){ c1.x, c1.y }; cgB = (f2){ c1.z, c1.w };
        cbA = (f2){ c2.x, c2.y }; cbB = (f2){ c2.z, c2.w };
    }
    // A lane leaves the list when all four of its pixels are below the threshold (checked after every list entry, so the
    // point of exit depends on the list alone -- not on batching or on the split into rounds); until then every pixel of
    // the lane keeps blending: what a pixel below the threshold still receives is < t_eps in total.
#define GS_LANE_LIVE() (fmaxf(fmaxf(TA.x, TA.y), fmaxf(TB.x, TB.y)) >= t_eps)
    f2 kbig = { 1.0e30f, 1.0e30f }, kone = { 1.0f, 1.0f };         // (GS_BLEND_APPLY_W; the empty asm keeps them in registers: as literals the
    asm volatile("" : "+v"(kbig), "+v"(kone));                     // compiler builds each pair again in front of every use, 2 moves per entry)
    bool live = GS_LANE_LIVE();
    uint32_t nfr = 0, staged = 0, evaluated = 0;
    const uint2 range = tile_range[tile];
    if (u.split_min && range.y - range.x >= u.split_min) continue;  // a long list: k_blend_px takes the tile (GS_OPT_BLEND_SPLIT)
    uint32_t j_mine = 0xFFFFFFFFu, nb_last = 0, e_l = 0;            // the sorted position this lane staged last / the size of the last batch / the last entry
                                                                    // of that batch this lane evaluated (GsControl::need_near)
    const uint32_t need_known = ctl->need_near[tile % GS_NEED_WORDS];   // (read now, compared at the end: a stale value only costs an atomic)

    for (uint32_t end = range.y; end > range.x;) {
        const uint32_t nb = min((uint32_t)GS_BLEND_BATCH, end - range.x);
        staged += nb;
        nb_last = nb; e_l = 0;
        uint32_t m16 = 0;                                          // GS_OPT_SUBTILE: the 4x4-pixel blocks of the tile this lane's entry can reach
#pragma unroll
        for (int h = 0; h < GS_BLEND_BATCH / 64; h++) {            // nearest first: reverse the back-to-front list
            const uint32_t slot = h * 64 + lane;
            if (slot < nb) {
                const uint32_t j = span ? reinterpret_cast<const uint32_t *>(pairs)[end - 1 - slot] : reinterpret_cast<const uint2 *>(pairs)[end - 1 - slot].y;
                const float4 *src = reinterpret_cast<const float4 *>(proj + j);
                const float4 ra = src[0], rb = src[1];
                if (GS_BLEND_BATCH == 64) j_mine = j;
                if (SUB) m16 = subtile_mask(ra, rb.x, rb.y, (float)(u.x0 + (int)tx * GS_TILE), (int)ty * GS_TILE, u.H);
                // (cx, cy, ax, bx | ay, by, -, -): the two coefficients a row shares with dy sit in one register pair, so that
                // dy * (ay, by) is one packed multiplication, and so do the two that multiply dx
                s_ent[3 * slot] = make_float4(ra.x, ra.y, ra.z, rb.x);
                s_ent[3 * slot + 1] = make_float4(ra.w, rb.y, 0.0f, 0.0f);
                // what every lane would otherwise redo for every list entry: unpack the colour, fold alpha / 255 into it
                const uint32_t rgba = __float_as_uint(rb.z);
                const float a255 = rb.w * (1.0f / 255.0f);
                // (-alpha: T <- T - alpha * e is one packed fma with the record's word as it stands; with +alpha and a negate modifier the
                // compiler copies the word to a register of its own first, one VALU instruction per list entry)
                s_ent[3 * slot + 2] = make_float4(-rb.w, (float)(rgba & 0xFF) * a255, (float)((rgba >> 8) & 0xFF) * a255, (float)((rgba >> 16) & 0xFF) * a255);
                if (SCENE) s_z[slot] = u.has_depth ? zwin[j] : 0.0f;
            }
        }
        if (lane == 0 && (nb & 1)) {
            if (SCENE) s_z[nb] = 0.0f;                               // pad an odd batch with a record no pixel can pass
            s_ent[3 * nb] = make_float4(-1.0e9f, -1.0e9f, 1.0f, 1.0f);   // centre far away, a = b = (1,1): q ~ 1e18 > 4
            s_ent[3 * nb + 1] = make_float4(1.0f, 1.0f, 0.0f, 0.0f);
            s_ent[3 * nb + 2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        __syncthreads();
        // ---- how the batch is walked: whole (every lane takes every entry), or split into the sixteen 4x4-pixel blocks' lists
        uint32_t cnt_max = nb;
        bool split = false;
        if (SUB && nb >= 8u) {
            // the lists: block g's k-th entry is the k-th set bit of its ballot; behind a list the inert slot, up to the longest list's
            // (even) length.  Built before it is known whether the batch will be walked by them: the ballots are the count
            uint32_t cm = 0;
            uint32_t *lw = reinterpret_cast<uint32_t *>(s_list);
            for (uint32_t i = (uint32_t)lane; i < GS_SUBTILE_GROUPS * GS_SUBTILE_STRIDE / 4u; i += 64u) lw[i] = GS_SUBTILE_INERT * 0x01010101u;
#pragma unroll
            for (int g = 0; g < GS_SUBTILE_GROUPS; g++) {
                const unsigned long long bal = __ballot((m16 >> g) & 1u);
                const uint32_t c = (uint32_t)__popcll(bal);
                cm = c > cm ? c : cm;
                const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                if ((m16 >> g) & 1u) s_list[g * GS_SUBTILE_STRIDE + pos] = (uint8_t)lane;
            }
            split = cm * 16u <= nb * GS_SUBTILE_KEEP_NUM;            // (uniform: ballots and their counts are scalars)
            if (split) {
                cnt_max = cm;
                __syncthreads();
            }
        }
        // |p|^2 of the interpolated vPosition, same expression tree per pixel as gsm::frag_power:
        //   px = dx * ax + dy * ay, py = dx * bx + dy * by, q = px * px + py * py            (-A, index.js:171)
        // EB: the entry's 48 bytes in LDS -- (cx, cy, ax, bx | ay, by, -, - | -alpha, r, g, b premultiplied)
#define GS_BLEND_Q(K, EB, qA, qB)                                                                                      \
                const float4 a_##K = *reinterpret_cast<const float4 *>(EB); const f2 ab_##K = *reinterpret_cast<const f2 *>((EB) + 16); \
                const f2 axbx_##K = { a_##K.z, a_##K.w };                                                               \
                const f2 dyab_##K = (f2)(fy - a_##K.y) * ab_##K;     /* (dy * ay, dy * by) */                          \
                const f2 dxA_##K = fxA - a_##K.x, dxB_##K = fxB - a_##K.x;                                              \
                const f2 pxA_##K = fma2_lo12(dxA_##K, axbx_##K, dyab_##K), pxB_##K = fma2_lo12(dxB_##K, axbx_##K, dyab_##K); \
                const f2 pyA_##K = fma2_hi12(dxA_##K, axbx_##K, dyab_##K), pyB_##K = fma2_hi12(dxB_##K, axbx_##K, dyab_##K); \
                const f2 qA = fma2(pxA_##K, pxA_##K, pyA_##K * pyA_##K), qB = fma2(pxB_##K, pxB_##K, pyB_##K * pyB_##K);
#define GS_BLEND_APPLY(qA, qB, EB, zz)                                                                                 \
                {                                                                                                      \
                    bool p0 = qA.x <= qmA.x, p1 = qA.y <= qmA.y, p2 = qB.x <= qmB.x, p3 = qB.y <= qmB.y;               \
                    if (SCENE) { p0 = p0 && zz <= zb0; p1 = p1 && zz <= zb1; p2 = p2 && zz <= zb2; p3 = p3 && zz <= zb3; } \
                    if (live & (p0 | p1 | p2 | p3)) {                  /* discard test, index.js:172 */                \
                        /* the colour record as two 8-byte halves: the compiler splats .x / .y of a register pair through \
                           op_sel but copies the fourth word of a 16-byte read to a register of its own first */      \
                        const float2 cl_ = *reinterpret_cast<const float2 *>((EB) + 32);                               \
                        const float2 ch_ = *reinterpret_cast<const float2 *>((EB) + 40);                               \
                        const float nalpha = cl_.x;                  /* -alpha (staging) */                          \
                        /* exp(A) (index.js:173); 0 for the pixels of this lane that the splat misses */               \
                        /* (= __expf(-q): v_exp_f32 of q * -log2(e), the four multiplications as two packed ones) */   \
                        const f2 tA = qA * (f2)(-1.44269502f), tB = qB * (f2)(-1.44269502f);                            \
                        const f2 EA = { p0 ? __builtin_amdgcn_exp2f(tA.x) : 0.0f, p1 ? __builtin_amdgcn_exp2f(tA.y) : 0.0f }; \
                        const f2 EB_ = { p2 ? __builtin_amdgcn_exp2f(tB.x) : 0.0f, p3 ? __builtin_amdgcn_exp2f(tB.y) : 0.0f }; \
                        /* fragment alpha B = exp(A)*vColor.a; its weight under what is in front: w = B*T.  T <- T - w \
                           (= T*(1-B)), colour += (rgb8 * alpha/255) * (exp(A)*T) with the bracket converted at staging */ \
                        const f2 eA = EA * TA, eB = EB_ * TB;                                                          \
                        TA = fma2((f2)(nalpha), eA, TA); TB = fma2((f2)(nalpha), eB, TB);                              \
                        crA = fma2((f2)(cl_.y), eA, crA); crB = fma2((f2)(cl_.y), eB, crB);                            \
                        cgA = fma2((f2)(ch_.x), eA, cgA); cgB = fma2((f2)(ch_.x), eB, cgB);                            \
                        /* (the compiler splats the high word of the FIRST pair through op_sel but copies the second    \
                           pair's to a fresh register: the same packed fma, the selection written out) */              \
                        const f2 chv_ = { ch_.x, ch_.y };                                                              \
                        cbA = fma2_hi0(chv_, eA, cbA); cbB = fma2_hi0(chv_, eB, cbB);                                  \
                        if (COUNT) nfr += (uint32_t)p0 + (uint32_t)p1 + (uint32_t)p2 + (uint32_t)p3;                   \
                        live = GS_LANE_LIVE();                                                                         \
                    }                                                                                                  \
                }
        // The same step with the coverage test as a FACTOR instead of four compares and four selects (plain frames, both walks: no
        // fragment count, no scene depth): w = clamp((qm - q) * 1e30 + 1) is exactly 1 where q <= qm and exactly 0 where q > qm (two
        // floats that differ differ by 2^-22 at least near 4: times 1e30 that is +-2e23), and exp(A) * w is exp(A) or +0 -- the values the
        // selects produce, so T and the colours see the same operations.  What goes is the skip of a step no lane's pixel passes
        // (rare: lane utilisation 0.85 at the headline pose, and a step of the block lists serves sixteen blocks with sixteen entries), 2 of
        // the 39 vector instructions per entry.
#define GS_BLEND_APPLY_W(qA, qB, EB)                                                                                   \
                if (live) {                                                                                            \
                    const float2 cl_ = *reinterpret_cast<const float2 *>((EB) + 32);                                   \
                    const float2 ch_ = *reinterpret_cast<const float2 *>((EB) + 40);                                   \
                    const float nalpha = cl_.x;                                                                        \
                    const f2 wA = fma2_clamp(qmA - qA, kbig, kone), wB = fma2_clamp(qmB - qB, kbig, kone);             \
                    const f2 tA = qA * (f2)(-1.44269502f), tB = qB * (f2)(-1.44269502f);                                \
                    const f2 EA = { __builtin_amdgcn_exp2f(tA.x), __builtin_amdgcn_exp2f(tA.y) };                      \
                    const f2 EB_ = { __builtin_amdgcn_exp2f(tB.x), __builtin_amdgcn_exp2f(tB.y) };                     \
                    const f2 eA = (EA * wA) * TA, eB = (EB_ * wB) * TB;                                                \
                    TA = fma2((f2)(nalpha), eA, TA); TB = fma2((f2)(nalpha), eB, TB);                                  \
                    crA = fma2((f2)(cl_.y), eA, crA); crB = fma2((f2)(cl_.y), eB, crB);                                \
                    cgA = fma2((f2)(ch_.x), eA, cgA); cgB = fma2((f2)(ch_.x), eB, cgB);                                \
                    const f2 chv_ = { ch_.x, ch_.y };                                                                  \
                    cbA = fma2_hi0(chv_, eA, cbA); cbB = fma2_hi0(chv_, eB, cbB);                                      \
                    live = GS_LANE_LIVE();                                                                             \
                }
        if (SUB && live && split) {
            // this lane's block list, two entries per step like the whole walk below (their coverage tests overlap); a lane whose
            // list is shorter than the longest one finishes on the inert record
            const uint8_t *lst = s_list + (((uint32_t)lane / (4u * GS_SUBTILE_ROWS)) * 4u + ((uint32_t)lane & 3u)) * GS_SUBTILE_STRIDE;
            uint32_t k = 0, i_last = nb - 1u;
            for (; k < cnt_max; k += 2u) {
                const uint32_t pr = *reinterpret_cast<const uint16_t *>(lst + k);
                const uint32_t i0 = pr & 0xFFu, i1 = pr >> 8;
                const char *eb0 = reinterpret_cast<const char *>(s_ent) + i0 * 48u, *eb1 = reinterpret_cast<const char *>(s_ent) + i1 * 48u;
                GS_BLEND_Q(0, eb0, qA0, qB0)
                GS_BLEND_Q(1, eb1, qA1, qB1)
                if (GS_BLEND_FACTOR && !COUNT && !SCENE) {
                    GS_BLEND_APPLY_W(qA0, qB0, eb0)
                    GS_BLEND_APPLY_W(qA1, qB1, eb1)
                } else {
                const float z0 = SCENE ? s_z[i0] : 0.0f, z1 = SCENE ? s_z[i1] : 0.0f;
                GS_BLEND_APPLY(qA0, qB0, eb0, z0)
                GS_BLEND_APPLY(qA1, qB1, eb1, z1)
                }
                if (!live) { i_last = min(i1 < GS_SUBTILE_INERT ? i1 : (i0 < GS_SUBTILE_INERT ? i0 : 0u), nb - 1u); break; }
            }
            e_l = i_last;                                            // (the entry at which the lane left the list, or the batch's last)
            if (u.record_staged == 2) evaluated += min(k + 2u, cnt_max);
        } else if (live) {
            // Two splats per step: their coverage tests are independent, so the second one's LDS read + ~13-instruction
            // dependent chain overlaps the first one's (the per-step latency, not issue bandwidth, bounds a tile that
            // runs alone in the kernel's tail).  Blending is still applied strictly in list order.
            // the LDS address of the step's first entry, kept in a VECTOR register (the empty asm hides that it is uniform): left to
            // itself the compiler keeps it in a scalar register and copies it to a vector register before each of the step's three
            // groups of reads -- 1.5 VALU instructions per list entry of the ~44.5
            uint32_t vo = 0;
            asm volatile("" : "+v"(vo));
            const uint32_t vend = nb * 48u;
            for (; vo < vend; vo += 48u * 2u) {
                const char *eb = reinterpret_cast<const char *>(s_ent) + vo;
                GS_BLEND_Q(0, eb, qA0, qB0)
                GS_BLEND_Q(1, eb + 48, qA1, qB1)                      // slot nb holds an inert record when nb is odd
                const uint32_t s = SCENE ? vo / 48u : 0u;               // (the entry's index: only the scene's depth test needs it)
                const float z0 = SCENE ? s_z[s] : 0.0f;
                if (GS_BLEND_FACTOR && !COUNT && !SCENE) {
                    GS_BLEND_APPLY_W(qA0, qB0, eb)
                    GS_BLEND_APPLY_W(qA1, qB1, eb + 48)
                } else {
                GS_BLEND_APPLY(qA0, qB0, eb, z0)
                const float z1 = SCENE ? s_z[s + 1] : 0.0f;
                GS_BLEND_APPLY(qA1, qB1, eb + 48, z1)
                }
                if (!live) break;
            }
            e_l = min(vo / 48u + 1u, nb - 1u);                        // (the step in which the lane left the list, or the batch's last entry)
            if (u.record_staged == 2) evaluated += min(vo / 48u + 2u, nb);   // list entries this lane evaluated (measurement aid)
        }
#undef GS_BLEND_APPLY_W
#undef GS_BLEND_APPLY
#undef GS_BLEND_Q
        end -= nb;
        __syncthreads();                                           // s_ent is rewritten by the next batch
        if (__all(!live)) break;
    }
    if (!COUNT && GS_BLEND_BATCH == 64 && !(u.flags & GS_RENDER_NO_EARLY_OUT)) {
        // how many of the nearest splats this tile needed (GsControl::need_near): the sorted position of the entry at which its LAST
        // lane left the list (lane k staged entry k of the batch: a tile's entries lie ~1000 sorted positions apart, "the batch" would
        // be 40 % too much)
        const bool sat = !__any(live);
        const bool more = ROUND == 0 && u.near_count < ctl->n_kept;  // (not saturated, farther splats to come: round 1 speaks for the tile, or the frame is flagged)
        if (sat || !more) {
            // the wave's maximum by data-parallel-primitive moves (no LDS round trips: as six ds_bpermute steps, each waiting for the one
            // before, it was 39 vector instructions and ~700 cycles at the end of every tile): within quads, within rows of 16, then row 0 -> 1,
            // 2 -> 3 and rows 0-1 -> 2-3; lane 63 holds the maximum
            uint32_t e_max = e_l;
            e_max = max(e_max, (uint32_t)__builtin_amdgcn_update_dpp((int)e_max, (int)e_max, 0xB1, 0xF, 0xF, false));    // quad_perm:[1,0,3,2]
            e_max = max(e_max, (uint32_t)__builtin_amdgcn_update_dpp((int)e_max, (int)e_max, 0x4E, 0xF, 0xF, false));    // quad_perm:[2,3,0,1]
            e_max = max(e_max, (uint32_t)__builtin_amdgcn_update_dpp((int)e_max, (int)e_max, 0x141, 0xF, 0xF, false));   // row_half_mirror
            e_max = max(e_max, (uint32_t)__builtin_amdgcn_update_dpp((int)e_max, (int)e_max, 0x140, 0xF, 0xF, false));   // row_mirror
            e_max = max(e_max, (uint32_t)__builtin_amdgcn_update_dpp((int)e_max, (int)e_max, 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
            e_max = max(e_max, (uint32_t)__builtin_amdgcn_update_dpp((int)e_max, (int)e_max, 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2, 3
            e_max = (uint32_t)__builtin_amdgcn_readlane((int)e_max, 63);
#ifdef GS_DEBUG_DPP_MAX
            { uint32_t chk = e_l;
              for (int m = 32; m >= 1; m >>= 1) chk = max(chk, (uint32_t)__shfl_xor((int)chk, m, 64));
              if (chk != e_max) __builtin_trap(); }
#endif
            const uint32_t jf = nb_last ? (uint32_t)__shfl((int)j_mine, (int)e_max, 64) : 0xFFFFFFFFu;
            if (lane == 0) {
                const uint32_t V = ctl->n_kept;
                const uint32_t need = !sat ? 0xFFFFFFFFu : (nb_last && jf < V ? V - jf : 0u);
                if (need > need_known) atomicMax(&ctl->need_near[tile % GS_NEED_WORDS], need);
            }
        }
    }
    if (ROUND == 0 && u.near_count < ctl->n_kept) {               // farther splats exist beyond this round
        // the nearer splats did not saturate this tile: keep the exact per-pixel state for round 1 and flag the tile
        if (__any(live)) {
            st[0] = make_float4(TA.x, TA.y, TB.x, TB.y);
            st[1] = make_float4(crA.x, crA.y, crB.x, crB.y); st[2] = make_float4(cgA.x, cgA.y, cgB.x, cgB.y);
            st[3] = make_float4(cbA.x, cbA.y, cbB.x, cbB.y);
            if (lane == 0) {
                atomicOr(&mask[ty * u.mask_words + (tx >> 5)], 1u << (tx & 31)); atomicAdd(&ctl->unsat_count, 1u);
                if (u.skip_round1) { ctl->round1_missed = 1; if (u.status) atomicOr(u.status, 1u); }   // nobody will come for this tile unless the host notices
            }
        }
    }
    if (row_in && xb < u.x1) {
        // dst <- src.rgb*a + dst.rgb*(1-a), dst.a <- a + dst.a*(1-a), composed over the background
        const int sw = u.out_pitch;
        const int orow = (u.flags & GS_RENDER_FLIP_Y) ? (u.H - 1 - r) : r;
        const float Tk[4] = { TA.x, TA.y, TB.x, TB.y }, rk[4] = { crA.x, crA.y, crB.x, crB.y }, gk[4] = { cgA.x, cgA.y, cgB.x, cgB.y };
        const float bk[4] = { cbA.x, cbA.y, cbB.x, cbB.y };
        uint32_t px[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float b0 = u.bg[0], b1 = u.bg[1], b2 = u.bg[2], b3 = u.bg[3];
            if (SCENE && u.has_scene_rgba && xb + k < u.x1) {      // destination = the opaque scene's colour at this pixel
                const uint32_t c = scene_rgba[(size_t)r * u.W + xb + k];
                b0 = (float)(c & 0xFF) / 255.0f; b1 = (float)((c >> 8) & 0xFF) / 255.0f;
                b2 = (float)((c >> 16) & 0xFF) / 255.0f; b3 = (float)(c >> 24) / 255.0f;
            }
            const float o0 = fmaf(Tk[k], b0, rk[k]), o1 = fmaf(Tk[k], b1, gk[k]);
            // accumulated alpha = sum of the weights w = 1 - T (the weights telescope: T_k = T_{k-1} - w_k)
            const float o2 = fmaf(Tk[k], b2, bk[k]), o3 = fmaf(Tk[k], b3, 1.0f - Tk[k]);
            px[k] = (uint32_t)(fminf(fmaxf(o0, 0.0f), 1.0f) * 255.0f + 0.5f) |
                    ((uint32_t)(fminf(fmaxf(o1, 0.0f), 1.0f) * 255.0f + 0.5f) << 8) |
                    ((uint32_t)(fminf(fmaxf(o2, 0.0f), 1.0f) * 255.0f + 0.5f) << 16) |
                    ((uint32_t)(fminf(fmaxf(o3, 0.0f), 1.0f) * 255.0f + 0.5f) << 24);
        }
        uint32_t *dst = reinterpret_cast<uint32_t *>(out) + (size_t)orow * sw + (xb - u.x0);
        if (xb + 3 < u.x1 && (sw & 3) == 0) *reinterpret_cast<uint4 *>(dst) = make_uint4(px[0], px[1], px[2], px[3]);
        else {
#pragma unroll
            for (int k = 0; k < 4; k++) if (xb + k < u.x1) dst[k] = px[k];
        }
    }
    if (COUNT) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) nfr += __shfl_xor(nfr, m, 64);
        if (lane == 0 && nfr) atomicAdd(&ctl->n_frags, (unsigned long long)nfr);
    }
    if (u.record_staged == 2) {                                    // entries the wave evaluated = its longest-lived lane's
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) evaluated = max(evaluated, (uint32_t)__shfl_xor(evaluated, m, 64));
        staged = evaluated;
    }
#undef GS_LANE_LIVE
    if (u.record_staged && lane == 0) const_cast<uint2 *>(tile_range)[tile] = make_uint2(staged, range.y - range.x);   // GS_OPT_RECORD_STAGED
    __syncthreads();                                               // s_ent is reused by the next tile of this wave
    }
}

template <bool COUNT, int ROUND, bool SCENE, bool SUB = false>
__global__ __launch_bounds__(64) void k_blend(const uint2 *__restrict__ tile_range, const void *__restrict__ pairs,
                                              const gsm::Projected *__restrict__ proj, GsFrameUniforms u,
                                              uint8_t *__restrict__ out, float4 *__restrict__ state, uint32_t *__restrict__ mask,
                                              const float *__restrict__ zwin, const float *__restrict__ scene_depth,
                                              const uint32_t *__restrict__ scene_rgba, GsControl *ctl)
{
    k_blend_body<COUNT, ROUND, SCENE, SUB>(tile_range, pairs, proj, u, out, state, mask, zwin, scene_depth, scene_rgba, ctl);
}

// GS_OPT_BLEND_SPLIT: the tiles with LONG lists, four wavefronts per tile, ONE pixel per lane (wave w: tile rows 4w .. 4w+3).
// A frame in which few tiles carry long lists (a cut-out scene filling a tenth of the screen: 800 active tiles, lists of
// 4000-6000 entries of which the busiest tile evaluates 1100 before it saturates) lasts as long as ONE wavefront's serial
// walk of its list, at ~43 dependent VALU instructions per entry for its 4 pixels per lane, while most of the chip idles.
// With a pixel per lane an entry costs ~14 instructions, the four bands of a tile walk the list concurrently on different
// SIMDs, and each band stops as soon as ITS 64 pixels are saturated.  Same fragments, same per-pixel operation sequence; a
// pixel now stops exactly when it falls below the threshold instead of when its lane's four pixels have (differences
// < t_eps: within the 1 LSB tolerance, not bit-identical to k_blend's image).  Every wave stages its own batches, so the
// projected records are read four times: affordable exactly when few tiles are active.
// (Measured and dropped: splitting the LIST over 8 waves -- segments blended from fresh states and composed front to back.
// A segment that starts at T = 1 never terminates early, so the tile's work grows from the ~300-1100 entries it really needs
// to min(L, 2048): 188 -> 284 us on the 6 M cut-out frame.  The same speculation on top of THIS kernel -- segment-0 wave from
// the band's true state, one to three more waves per band walking the following 256-entry segments from fresh states,
// composed while the band is unsaturated -- changed nothing alone on the GPU (84 -> 80 us) and lost with frames overlapped
// (8613 -> 8174 frames/s with 2 segments, 6658 with 4): the kernel is not bound by one tile's chain any more.)
#ifndef GS_PX_BATCH
#define GS_PX_BATCH 128u           // list entries k_blend_px stages per batch (a multiple of 64: two dependent global loads per batch are the latency to hide)
#endif
#ifndef GS_PX_GROUP
#define GS_PX_GROUP 4u             // list entries per step of k_blend_px (independent coverage tests and exp(): instruction-level parallelism)
#endif
template <int ROUND, bool SCENE>
__device__ __forceinline__ void k_blend_px_body(const uint2 *__restrict__ tile_range, const void *__restrict__ pairs,
                                                const gsm::Projected *__restrict__ proj, const GsFrameUniforms &u,
                                                uint8_t *__restrict__ out, float4 *__restrict__ state, uint32_t *__restrict__ mask,
                                                const float *__restrict__ zwin, const float *__restrict__ scene_depth,
                                                const uint32_t *__restrict__ scene_rgba, GsControl *ctl)
{
    constexpr uint32_t PB = GS_PX_BATCH, PR = GS_PX_BATCH / 64u;   // records per batch / per lane
    // a batch in LDS, per wave, laid out for two entries per packed-fp32 instruction: pair p = entries (2p, 2p+1) holds
    // (cx0, cx1, cy0, cy1), (ax0, ax1, ay0, ay1), (bx0, bx1, by0, by1); colours (r, g, b) * alpha / 255 and alpha per entry
    __shared__ float s_pair_all[4][3][PB / 2][4];
    __shared__ float4 s_col_all[4][PB];
    __shared__ float s_z_all[4][PB];
    __shared__ uint32_t s_live[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float (*s_pair)[PB / 2][4] = s_pair_all[w];
    float4 *s_col = s_col_all[w];
    float *s_z = s_z_all[w];
    // a batch staged by the wave's lanes is read by all of them; the four waves run their own trip counts, so this is the
    // completion of the wave's own LDS operations, not a workgroup barrier.  Only lgkmcnt: a fence would also wait for the
    // global loads of the NEXT batch, which are in flight on purpose (that made every batch pay its two dependent loads)
#define GS_WAVE_LDS_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
    if (ROUND == 1 && ctl->j_hi == 0) return;                      // every tile saturated in round 0
    const uint32_t ntiles = (uint32_t)u.tiles_x * (uint32_t)u.tiles_y;
    const bool span = u.rc_stride != 0u;                          // the lists hold sorted positions (span lists) / (tile, position) records
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t tx = tile % (uint32_t)u.tiles_x, ty = tile / (uint32_t)u.tiles_x;
        if (ROUND == 1 && !((mask[ty * u.mask_words + (tx >> 5)] >> (tx & 31)) & 1u)) continue;   // this tile is final already
        const uint2 range = tile_range[tile];
        if (range.y - range.x < u.split_min) continue;                // a short list: k_blend's tile
        const int col = lane & 15, rr = w * 4 + (lane >> 4);           // pixel (col, rr) of the tile
        const int px = u.x0 + (int)tx * GS_TILE + col;
        const int r = (int)ty * GS_TILE + rr;                          // image row, 0 = top
        const bool in = r < u.H && px < u.x1;
        const float fx = (float)px + 0.5f, fy = (float)(u.H - 1 - r) + 0.5f;   // pixel centre, GL window coordinates
        const float t_eps = (u.flags & GS_RENDER_NO_EARLY_OUT) ? -1.0f : u.t_eps;
        const float qm = in ? 4.0f : -1.0f;                            // fragment kept iff q <= 4 (index.js:172); never outside the strip
        float T = in ? 1.0f : 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
        float zb = 3.0e38f;
        if (SCENE && u.has_depth && in) zb = scene_depth[(size_t)r * u.W + px];
        // the per-pixel state of k_blend's layout: lane (rr * 4 + col / 4) holds 4 x float4 (T, r, g, b) of its 4 pixels
        float *st = reinterpret_cast<float *>(state + ((size_t)tile * 64 + (uint32_t)(rr * 4 + col / 4)) * 4) + (col & 3);
        if (ROUND == 1) { T = st[0]; cr = st[4]; cg = st[8]; cb = st[12]; }
        bool live = T >= t_eps;
        // the batch after the current one is fetched (pair -> projected record: two dependent global loads) while the current
        // one is blended: a wave alone on its SIMD has nothing else to hide that latency behind
        float4 n0[PR], n1[PR];
        float nz[PR];
#define GS_PX_FETCH(END, NB) do { _Pragma("unroll") for (uint32_t h = 0; h < PR; h++) { const uint32_t slot = h * 64u + (uint32_t)lane;          \
            if (slot < (NB)) {                                                                                                               \
            const uint32_t j = span ? reinterpret_cast<const uint32_t *>(pairs)[(END) - 1 - slot] : reinterpret_cast<const uint2 *>(pairs)[(END) - 1 - slot].y; \
            const float4 *src = reinterpret_cast<const float4 *>(proj + j);                                                                  \
            n0[h] = src[0]; n1[h] = src[1];                                                                                                  \
            if (SCENE) nz[h] = u.has_depth ? zwin[j] : 0.0f; } } } while (0)
        GS_PX_FETCH(range.y, min(PB, range.y - range.x));              // nearest first: the list is back to front
        uint32_t end_w = range.y, e_l = 0;                              // this band's last batch was [.., end_w) of the list (nearest first); the last of its entries this lane evaluated
        const uint32_t need_known = ctl->need_near[tile % GS_NEED_WORDS];
        for (uint32_t end = range.y; end > range.x;) {
            const uint32_t nb = min(PB, end - range.x);
            const uint32_t nbp = (nb + GS_PX_GROUP - 1u) & ~(GS_PX_GROUP - 1u);   // whole groups
            end_w = end; e_l = 0;
#pragma unroll
            for (uint32_t h = 0; h < PR; h++) {
                const uint32_t slot = h * 64u + (uint32_t)lane;
                const uint32_t pi = slot >> 1, pk = slot & 1u;
                if (slot < nb) {
                    s_pair[0][pi][pk] = n0[h].x; s_pair[0][pi][2 + pk] = n0[h].y;      // centre
                    s_pair[1][pi][pk] = n0[h].z; s_pair[1][pi][2 + pk] = n0[h].w;      // axis a
                    s_pair[2][pi][pk] = n1[h].x; s_pair[2][pi][2 + pk] = n1[h].y;      // axis b
                    // what every lane would otherwise redo for every list entry: unpack the colour, fold alpha / 255 into it
                    const uint32_t rgba = __float_as_uint(n1[h].z);
                    const float a255 = n1[h].w * (1.0f / 255.0f);
                    // (r, -alpha, g, b: the compiler pairs (cr, T) and (cg, cb) for packed fmas -- stored like this the record's words
                    // are those pairs' multipliers as they stand; with (r, g, b, alpha) it spent a negation and two copies per entry)
                    s_col[slot] = make_float4((float)(rgba & 0xFF) * a255, -n1[h].w, (float)((rgba >> 8) & 0xFF) * a255, (float)((rgba >> 16) & 0xFF) * a255);
                    if (SCENE) s_z[slot] = nz[h];
                } else if (slot < nbp) {                              // pad the batch to whole groups with records no pixel can pass
                    if (SCENE) s_z[slot] = 0.0f;
                    s_pair[0][pi][pk] = -1.0e9f; s_pair[0][pi][2 + pk] = -1.0e9f;       // centre far away, a = b = (1,1): q ~ 1e18 > 4
                    s_pair[1][pi][pk] = 1.0f; s_pair[1][pi][2 + pk] = 1.0f;
                    s_pair[2][pi][pk] = 1.0f; s_pair[2][pi][2 + pk] = 1.0f;
                    s_col[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
            }
            if (end - nb > range.x) GS_PX_FETCH(end - nb, min(PB, end - nb - range.x));
            GS_WAVE_LDS_SYNC();
            if (live) {
                // GS_PX_GROUP list entries per step.  Coverage and exp() of the group are independent of each other and of the pixel's
                // state, so their LDS reads and ~10-instruction chains overlap; what is sequential per entry is only
                // e = E * T; T -= alpha * e; C += c * e -- the same operations in the same order as k_blend applies them.
                // A wave that runs alone on its SIMD (few active tiles) is bound by dependent-instruction latency, not issue.
                for (uint32_t s4 = 0; s4 < nb; s4 += GS_PX_GROUP) {
                    float E[GS_PX_GROUP];
                    float4 cc[GS_PX_GROUP];
#pragma unroll
                    for (uint32_t k = 0; k < GS_PX_GROUP; k += 2) {
                        const uint32_t pi = (s4 + k) >> 1;
                        const float4 R0 = *reinterpret_cast<const float4 *>(s_pair[0][pi]), R1 = *reinterpret_cast<const float4 *>(s_pair[1][pi]);
                        const float4 R2 = *reinterpret_cast<const float4 *>(s_pair[2][pi]);
                        cc[k] = s_col[s4 + k]; cc[k + 1] = s_col[s4 + k + 1];
                        // |p|^2 of the interpolated vPosition, same expression tree per pixel as gsm::frag_power / k_blend
                        const f2 dx = (f2)(fx) - (f2){ R0.x, R0.y }, dy = (f2)(fy) - (f2){ R0.z, R0.w };
                        const f2 pxv = fma2(dx, (f2){ R1.x, R1.y }, dy * (f2){ R1.z, R1.w });
                        const f2 pyv = fma2(dx, (f2){ R2.x, R2.y }, dy * (f2){ R2.z, R2.w });
                        const f2 q = fma2(pxv, pxv, pyv * pyv);
                        bool p0 = q.x <= qm, p1 = q.y <= qm;           // discard test, index.js:172
                        if (SCENE) { p0 = p0 && s_z[s4 + k] <= zb; p1 = p1 && s_z[s4 + k + 1] <= zb; }
                        E[k] = p0 ? __expf(-q.x) : 0.0f;               // exp(A) (index.js:173); 0 where the splat misses the pixel
                        E[k + 1] = p1 ? __expf(-q.y) : 0.0f;
                    }
#pragma unroll
                    for (uint32_t k = 0; k < GS_PX_GROUP; k++) {
                        const float e = E[k] * T;
                        T = fmaf(cc[k].y, e, T);
                        cr = fmaf(cc[k].x, e, cr); cg = fmaf(cc[k].z, e, cg); cb = fmaf(cc[k].w, e, cb);
                    }
                    live = T >= t_eps;                                 // (checked per group: a few entries past the threshold, < t_eps in total)
                    e_l = min(s4 + (GS_PX_GROUP - 1u), nb - 1u);
                    if (!live) break;
                }
            }
            end -= nb;
            GS_WAVE_LDS_SYNC();                                      // s_rec is rewritten by the next batch
            if (__all(!live)) break;
        }
        const bool wave_live = __any(live);
        if (lane == 0) s_live[w] = wave_live ? 1u : 0u;
        if (!(u.flags & GS_RENDER_NO_EARLY_OUT)) {
            // GsControl::need_near, per band of the tile (as k_blend per tile): the list entry at which the band's last pixel stopped
            const bool more = ROUND == 0 && u.near_count < ctl->n_kept;
            if (!wave_live || !more) {
                uint32_t e_max = e_l;
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) e_max = max(e_max, (uint32_t)__shfl_xor((int)e_max, m, 64));
                if (lane == 0) {
                    const uint32_t V = ctl->n_kept;
                    uint32_t need = 0xFFFFFFFFu;
                    if (!wave_live) {
                        need = 0u;
                        if (range.y > range.x && end_w > range.x + e_max) {
                            const uint32_t li = end_w - 1u - e_max;      // (slot s of a batch = list entry end - 1 - s)
                            const uint32_t jf = span ? reinterpret_cast<const uint32_t *>(pairs)[li] : reinterpret_cast<const uint2 *>(pairs)[li].y;
                            need = jf < V ? V - jf : 0u;
                        }
                    }
                    if (need > need_known) atomicMax(&ctl->need_near[tile % GS_NEED_WORDS], need);
                }
            }
        }
        __syncthreads();
        const bool tile_live = (s_live[0] | s_live[1] | s_live[2] | s_live[3]) != 0u;
        if (ROUND == 0 && u.near_count < ctl->n_kept && tile_live) {  // farther splats exist beyond this round and the tile wants them
            st[0] = T; st[4] = cr; st[8] = cg; st[12] = cb;
            if (threadIdx.x == 0) {
                atomicOr(&mask[ty * u.mask_words + (tx >> 5)], 1u << (tx & 31)); atomicAdd(&ctl->unsat_count, 1u);
                if (u.skip_round1) { ctl->round1_missed = 1; if (u.status) atomicOr(u.status, 1u); }
            }
        }
        if (in) {
            float b0 = u.bg[0], b1 = u.bg[1], b2 = u.bg[2], b3 = u.bg[3];
            if (SCENE && u.has_scene_rgba) {
                const uint32_t c = scene_rgba[(size_t)r * u.W + px];
                b0 = (float)(c & 0xFF) / 255.0f; b1 = (float)((c >> 8) & 0xFF) / 255.0f;
                b2 = (float)((c >> 16) & 0xFF) / 255.0f; b3 = (float)(c >> 24) / 255.0f;
            }
            const float o0 = fmaf(T, b0, cr), o1 = fmaf(T, b1, cg), o2 = fmaf(T, b2, cb), o3 = fmaf(T, b3, 1.0f - T);
            const int sw = u.out_pitch;
            const int orow = (u.flags & GS_RENDER_FLIP_Y) ? (u.H - 1 - r) : r;
            reinterpret_cast<uint32_t *>(out)[(size_t)orow * sw + (px - u.x0)] =
                (uint32_t)(fminf(fmaxf(o0, 0.0f), 1.0f) * 255.0f + 0.5f) | ((uint32_t)(fminf(fmaxf(o1, 0.0f), 1.0f) * 255.0f + 0.5f) << 8) |
                ((uint32_t)(fminf(fmaxf(o2, 0.0f), 1.0f) * 255.0f + 0.5f) << 16) | ((uint32_t)(fminf(fmaxf(o3, 0.0f), 1.0f) * 255.0f + 0.5f) << 24);
        }
        __syncthreads();                                             // s_live / the waves' LDS are reused by the next tile
    }
#undef GS_PX_FETCH
#undef GS_WAVE_LDS_SYNC
}

template <int ROUND, bool SCENE>
__global__ __launch_bounds__(256) void k_blend_px(const uint2 *__restrict__ tile_range, const void *__restrict__ pairs,
                                                  const gsm::Projected *__restrict__ proj, GsFrameUniforms u,
                                                  uint8_t *__restrict__ out, float4 *__restrict__ state, uint32_t *__restrict__ mask,
                                                  const float *__restrict__ zwin, const float *__restrict__ scene_depth,
                                                  const uint32_t *__restrict__ scene_rgba, GsControl *ctl)
{
    k_blend_px_body<ROUND, SCENE>(tile_range, pairs, proj, u, out, state, mask, zwin, scene_depth, scene_rgba, ctl);
}

int bits_for(uint32_t n) { int b = 1; while (b < 32 && (1u << b) < n) b++; return b; }

// the blend of one round over the tile lists `fpairs` (sorted positions if v.rc_stride != 0, else (tile, position) records)
template <int ROUND>
int launch_blend(gs_ctx *ctx, const GsFrameUniforms &u, GsFrameUniforms v, uint8_t *out, const void *fpairs, const gsm::Projected *bproj, const float *bzwin)
{
    const uint32_t ntiles = (uint32_t)u.tiles_x * (uint32_t)u.tiles_y;
    hipStream_t st = ctx->stream;
    const uint32_t gb = ROUND == 1 ? (ntiles < 1024 ? ntiles : 1024) : ntiles;
    const bool scene = u.has_depth || u.has_scene_rgba;
    if ((u.flags & GS_RENDER_COUNT_FRAGS) || u.record_staged) v.split_min = 0;     // measurement renders: every tile by k_blend
    if (v.split_min) {
        // the tiles with long lists first (the long pole): workgroups stride over all tiles' ranges and take the long ones
        const uint32_t gp = ntiles < 2048 ? ntiles : 2048;
        if (scene) hipLaunchKernelGGL((k_blend_px<ROUND, true>), dim3(gp), dim3(256), 0, st, ctx->tile_range, fpairs, bproj, v, out, ctx->state,
                                      ctx->unsat_mask, bzwin, ctx->scene_depth, ctx->scene_rgba, ctx->ctl);
        else hipLaunchKernelGGL((k_blend_px<ROUND, false>), dim3(gp), dim3(256), 0, st, ctx->tile_range, fpairs, bproj, v, out, ctx->state,
                                ctx->unsat_mask, bzwin, ctx->scene_depth, ctx->scene_rgba, ctx->ctl);
    }
#define GS_LAUNCH_BLEND_(C, S, B) hipLaunchKernelGGL((k_blend<C, ROUND, S, B>), dim3(gb), dim3(64), 0, st, ctx->tile_range, fpairs, bproj, v, \
                                                out, ctx->state, ctx->unsat_mask, bzwin, ctx->scene_depth, ctx->scene_rgba, ctx->ctl)
#define GS_LAUNCH_BLEND(C, S) do { if (v.subtile) GS_LAUNCH_BLEND_(C, S, true); else GS_LAUNCH_BLEND_(C, S, false); } while (0)
    if (u.flags & GS_RENDER_COUNT_FRAGS) { if (scene) GS_LAUNCH_BLEND(true, true); else GS_LAUNCH_BLEND(true, false); }
    else { if (scene) GS_LAUNCH_BLEND(false, true); else GS_LAUNCH_BLEND(false, false); }
#undef GS_LAUNCH_BLEND
#undef GS_LAUNCH_BLEND_
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// Span-list binning (GS_OPT_BINNING) for a round over at most `jrange` sorted positions: the chunk stride of its row-count table,
// or 0 = pair records + radix passes.  The tile columns and rows of the strip must each fit one workgroup (frames up to 4096 x
// 4096 pixels) and the table stay small (rows x chunks: 20 M positions at 4K would want 42 MB); a record format asked for by
// name (GS_OPT_WIDE_PAIRS) means the records.
#define GS_ROWCNT_MAX ((size_t)1 << 25)
uint32_t span_list_stride(const gs_ctx *ctx, const GsFrameUniforms &u, uint32_t jrange)
{
    const gs_ctx *P = gs_root(const_cast<gs_ctx *>(ctx));
    if (P->bin_mode == 1) return 0;
    if (u.tiles_x > GS_BLOCK || u.tiles_y > GS_BLOCK) return 0;
    const uint32_t stride = gs_div_up(jrange, GS_BLOCK) + 1u;
    if ((size_t)stride * (size_t)u.tiles_y > GS_ROWCNT_MAX) return 0;
    return stride;
}

size_t gs_row_table_entries(size_t n, uint32_t tiles_y)
{
    const size_t e = (size_t)(gs_div_up(n, GS_BLOCK) + 1u) * (size_t)tiles_y;
    return e > GS_ROWCNT_MAX ? 0 : e;                             // (beyond the table's limit such a round takes the pair records)
}

int gs_ensure_row_tables(gs_ctx *ctx, size_t entries)
{
    if (entries <= ctx->row_cnt_cap && ctx->row_tot && ctx->seg_diff) return GS_OK;
    GS_HIP(hipStreamSynchronize(ctx->stream));
    if (entries > ctx->row_cnt_cap) {
        if (ctx->row_cnt) (void)hipFree(ctx->row_cnt);
        ctx->row_cnt = nullptr; ctx->row_cnt_cap = 0;
        const size_t cap = entries + entries / 4;
        GS_HIP(hipMalloc((void **)&ctx->row_cnt, cap * sizeof(uint32_t)));
        ctx->row_cnt_cap = cap;
    }
    // (each table under its own check: a failed second allocation must not leave the first one vouching for both)
    if (!ctx->row_tot) { GS_HIP(hipMalloc((void **)&ctx->row_tot, GS_BLOCK * sizeof(uint2))); ctx->row_tot_cap = GS_BLOCK; }
    // (k_seg_count: a row of 256 ints per k_lists item; at most GS_LIST_SEGS items per tile row, at most 256 tile rows: 4.25 MB)
    if (!ctx->seg_diff) GS_HIP(hipMalloc((void **)&ctx->seg_diff, (size_t)GS_BLOCK * (GS_LIST_SEGS + 1u) * GS_BLOCK * sizeof(int)));
    return GS_OK;
}

// one round with span lists: project (+ row counts) -> row scan -> runs -> lists -> blend
template <int ROUND>
int run_round_spans(gs_ctx *ctx, const GsFrameUniforms &u, uint8_t *out, bool last_round, uint32_t g, uint32_t stride)
{
    hipStream_t st = ctx->stream;
    const int rcc = gs_ensure_row_tables(ctx, (size_t)stride * (size_t)u.tiles_y);
    if (rcc != GS_OK) return rcc;
    GsFrameUniforms v = u;
    v.rc_stride = stride;
    // the pair buffers hold the runs (geometry and sorted position of each: there are never more runs than tiles) and the lists
    uint32_t *run_geom = reinterpret_cast<uint32_t *>(ctx->pair_a), *run_ref = run_geom + ctx->pair_cap, *lists = reinterpret_cast<uint32_t *>(ctx->pair_b);
    const uint32_t pc = (uint32_t)ctx->pair_cap;
    hipLaunchKernelGGL((k_project<ROUND, true>), dim3(g), dim3(GS_BLOCK), 0, st, ctx->sorted, ctx->splat, v, ctx->proj, ctx->rect,
                       ctx->tile_count, ctx->row_cnt, ctx->part_vis, ctx->unsat_mask, ctx->zwin, ctx->ctl);
    GS_HIP(hipGetLastError());
    if (ROUND == 0) GS_PROF_RECORD(ctx, 3);
    hipLaunchKernelGGL(k_row_scan<ROUND>, dim3((uint32_t)u.tiles_y), dim3(GS_BLOCK), 0, st, ctx->row_cnt, ctx->row_tot, (const GsControl *)ctx->ctl, u.near_count,
                       stride, (uint32_t)u.tiles_y, ctx->unsat_mask, u.mask_words);
    hipLaunchKernelGGL(k_emit_runs<ROUND>, dim3(g), dim3(GS_BLOCK), 0, st, ctx->proj, ctx->rect, ctx->tile_count, ctx->row_cnt, ctx->row_tot, v,
                       run_geom, run_ref, ctx->unsat_mask, (const GsControl *)ctx->ctl, pc);
    uint32_t gl = (uint32_t)u.tiles_y * GS_LIST_SEGS; if (gl > (ROUND == 1 ? 512u : 2048u)) gl = ROUND == 1 ? 512u : 2048u;
    // frames of many runs per tile row (many small splats: a cut-out scene, the cloud seen from outside, tiles that do not saturate)
    // count their segments in a launch of their own; frames of few (the headline pose: 3 000 per row) let every k_lists item count its
    // row itself -- one launch less.  Decided from the runs of the last collected frame: a matter of speed only, the lists are the same.
    const bool segc = ROUND == 0 && __atomic_load_n(&gs_root(ctx)->run_hint, __ATOMIC_RELAXED) > GS_SEGC_RUNS_PER_ROW * (uint32_t)u.tiles_y;
    if (segc) {
        hipLaunchKernelGGL(k_seg_count<ROUND>, dim3(gl), dim3(GS_BLOCK), 0, st, (const uint32_t *)run_geom, (const uint2 *)ctx->row_tot, ctx->seg_diff, v,
                           (const GsControl *)ctx->ctl, pc);
        hipLaunchKernelGGL((k_lists<ROUND, true>), dim3(gl), dim3(GS_BLOCK), 0, st, run_geom, run_ref, ctx->row_tot, (const int *)ctx->seg_diff, lists, ctx->tile_range, v,
                           ctx->unsat_mask, ctx->ctl, pc, ctx->part_vis, g, last_round ? 1 : 0);
    } else
        hipLaunchKernelGGL((k_lists<ROUND, false>), dim3(gl), dim3(GS_BLOCK), 0, st, run_geom, run_ref, ctx->row_tot, (const int *)ctx->seg_diff, lists, ctx->tile_range, v,
                           ctx->unsat_mask, ctx->ctl, pc, ctx->part_vis, g, last_round ? 1 : 0);
    GS_HIP(hipGetLastError());
    if (ROUND == 0) GS_PROF_RECORD(ctx, 4);
    return launch_blend<ROUND>(ctx, u, v, out, lists, ctx->proj, ctx->zwin);
}

// one round: project -> offsets -> emit -> stable sort by tile -> ranges -> blend.  Round 1 usually finds nothing to
// do (every tile saturated), so it is launched on small grids: its kernels grid-stride when there is work.
template <int ROUND>
int run_round(gs_ctx *ctx, const GsFrameUniforms &u, uint8_t *out, bool last_round)
{
    const uint32_t ntiles = (uint32_t)u.tiles_x * (uint32_t)u.tiles_y;
    const uint32_t Vmax = (uint32_t)ctx->n;
    hipStream_t st = ctx->stream;
    uint32_t g = gs_div_up(Vmax, GS_BLOCK); if (g > GS_MAX_PART) g = GS_MAX_PART;
    const uint32_t small = 512;
    if (ROUND == 1 && g > small) g = small;
    // round 0 of a two-round frame covers at most near_count splats: no point in launching workgroups for the rest
    if (ROUND == 0 && u.near_count != 0xFFFFFFFFu) { const uint32_t gn = gs_div_up(u.near_count < Vmax ? u.near_count : Vmax, GS_BLOCK); if (gn < g) g = gn ? gn : 1; }
    const uint32_t pc = (uint32_t)ctx->pair_cap;
    // what the pair sort should expect (grid, one- or two-level offsets): round 1 usually finds nothing; round 0 about what
    // the last collected frames binned (0 = not known yet: the capacity)
    const uint32_t ph = ROUND == 1 ? (uint32_t)(small * GS_CHUNK_S) : __atomic_load_n(&gs_root(ctx)->pair_hint, __ATOMIC_RELAXED);
    const uint32_t jrange = ROUND == 0 ? (u.near_count != 0xFFFFFFFFu && u.near_count < Vmax ? u.near_count : Vmax) : Vmax;
    if (const uint32_t stride = span_list_stride(ctx, u, jrange)) return run_round_spans<ROUND>(ctx, u, out, last_round, g, stride);
    // (tile, position) records through two stable radix passes on the tile id: strips beyond 4096 pixels (more than 256 tile columns or
    // rows), or GS_OPT_BINNING = 1
    const int tb = bits_for(ntiles);
    GsFrameUniforms v = u;
    v.rc_stride = 0;
    hipLaunchKernelGGL((k_project<ROUND, false>), dim3(g), dim3(GS_BLOCK), 0, st, ctx->sorted, ctx->splat, u, ctx->proj, ctx->rect,
                       ctx->tile_count, ctx->spine, ctx->part_vis, ctx->unsat_mask, ctx->zwin, ctx->ctl);
    GS_HIP(hipGetLastError());
    if (ROUND == 0) GS_PROF_RECORD(ctx, 3);
    hipLaunchKernelGGL(k_pairs_check<ROUND>, dim3(1), dim3(GS_BLOCK), 0, st, ctx->ctl, (uint32_t)ctx->pair_cap, ctx->spine, ctx->part_vis, g,
                       u.near_count, last_round ? 1 : 0, ctx->unsat_mask, (uint32_t)u.tiles_y * u.mask_words, ctx->emit_extra);
    // (items = the round's chunks + the extra slices of the heavy ones: about I / GS_EMIT_PAIRS more)
    uint32_t ge = g + (ROUND == 1 ? 0u : gs_div_up(ph ? ph : pc, GS_EMIT_PAIRS)); if (ge > GS_MAX_PART) ge = GS_MAX_PART;
    hipLaunchKernelGGL(k_emit<ROUND>, dim3(ge), dim3(GS_BLOCK), 0, st, ctx->proj, ctx->rect, ctx->tile_count, ctx->spine,
                       ctx->emit_extra, v, ctx->pair_a, ctx->unsat_mask, ctx->ctl);
    GS_HIP(hipGetLastError());
    int rc;
    const uint2 *fpairs;
    if (tb <= 9) {
        rc = gs_launch_radix_pass(ctx, ctx->pair_a, GS_RADIX_PACKED, ctx->pair_b, GS_RADIX_PACKED, &ctx->ctl->n_pairs, pc, ph, 0, tb);
        if (rc != GS_OK) return rc;
        fpairs = ctx->pair_b;
    } else {
        const int b1 = (tb + 1) / 2, b2 = tb - b1;
        rc = gs_launch_radix_pass(ctx, ctx->pair_a, GS_RADIX_PACKED, ctx->pair_b, GS_RADIX_PACKED, &ctx->ctl->n_pairs, pc, ph, 0, b1);
        if (rc != GS_OK) return rc;
        rc = gs_launch_radix_pass(ctx, ctx->pair_b, GS_RADIX_PACKED, ctx->pair_a, GS_RADIX_PACKED, &ctx->ctl->n_pairs, pc, ph, b1, b2);
        if (rc != GS_OK) return rc;
        fpairs = ctx->pair_a;
    }
    hipLaunchKernelGGL(k_tile_ranges, dim3(ROUND == 1 ? small : 2048), dim3(GS_BLOCK), 0, st, fpairs, ctx->tile_range, ntiles, ROUND, ctx->ctl);
    GS_HIP(hipGetLastError());
    if (ROUND == 0) GS_PROF_RECORD(ctx, 4);
    return launch_blend<ROUND>(ctx, u, v, out, fpairs, ctx->proj, ctx->zwin);
}

template <int ROUND, bool RUNS> GS_BODY(F_project, k_project_body<ROUND, RUNS>);
template <int ROUND> GS_BODY(F_row_scan, k_row_scan_body<ROUND>);
template <int ROUND> GS_BODY(F_emit_runs, k_emit_runs_body<ROUND>);
template <int ROUND> GS_BODY(F_seg_count, k_seg_count_body<ROUND>);
template <int ROUND, bool SEGC> GS_BODY(F_lists, k_lists_body<ROUND, SEGC>);
template <int ROUND> GS_BODY(F_pairs_check, k_pairs_check_body<ROUND>);
template <int ROUND> GS_BODY(F_emit, k_emit_body<ROUND>);
GS_BODY(F_tile_ranges, k_tile_ranges_body);
template <int ROUND, bool SCENE, bool SUB> GS_BODY(F_blend, k_blend_body<false, ROUND, SCENE, SUB>);
template <int ROUND, bool SCENE> GS_BODY(F_blend_px, k_blend_px_body<ROUND, SCENE>);

// the blend of one round for two frames (launch_blend's paired form)
template <int ROUND>
int launch_blend2(gs_ctx *const S[2], const GsFrameUniforms &u, const GsFrameUniforms V[2], uint8_t *const out[2], const void *const fpairs[2],
                  const gsm::Projected *const bproj[2], const float *const bzwin[2])
{
    gs_ctx *ctx = S[0];
    hipStream_t st = ctx->stream;
    const uint32_t ntiles = (uint32_t)u.tiles_x * (uint32_t)u.tiles_y;
    const uint32_t gb = ROUND == 1 ? (ntiles < 1024 ? ntiles : 1024) : ntiles;
    const bool scene = u.has_depth || u.has_scene_rgba;
#define GS_BLENDPX2(SC) gs_twin<F_blend_px<ROUND, SC>, 256>(ntiles < 2048 ? ntiles : 2048, st,                                                          \
        gs_pack_make((const uint2 *)S[0]->tile_range, fpairs[0], bproj[0], V[0], out[0], S[0]->state, S[0]->unsat_mask,         \
                     bzwin[0], (const float *)S[0]->scene_depth, (const uint32_t *)S[0]->scene_rgba, S[0]->ctl),                        \
        gs_pack_make((const uint2 *)S[1]->tile_range, fpairs[1], bproj[1], V[1], out[1], S[1]->state, S[1]->unsat_mask,         \
                     bzwin[1], (const float *)S[1]->scene_depth, (const uint32_t *)S[1]->scene_rgba, S[1]->ctl))
    if (u.split_min) { if (scene) GS_BLENDPX2(true); else GS_BLENDPX2(false); }   // the tiles with long lists first
#undef GS_BLENDPX2
#define GS_BLEND2_(SC, SB) gs_twin<F_blend<ROUND, SC, SB>, 64>(gb, st,                                                                                          \
        gs_pack_make((const uint2 *)S[0]->tile_range, fpairs[0], bproj[0], V[0], out[0], S[0]->state, S[0]->unsat_mask,         \
                     bzwin[0], (const float *)S[0]->scene_depth, (const uint32_t *)S[0]->scene_rgba, S[0]->ctl),                        \
        gs_pack_make((const uint2 *)S[1]->tile_range, fpairs[1], bproj[1], V[1], out[1], S[1]->state, S[1]->unsat_mask,         \
                     bzwin[1], (const float *)S[1]->scene_depth, (const uint32_t *)S[1]->scene_rgba, S[1]->ctl))
#define GS_BLEND2(SC) do { if (V[0].subtile) GS_BLEND2_(SC, true); else GS_BLEND2_(SC, false); } while (0)
    if (scene) GS_BLEND2(true); else GS_BLEND2(false);
#undef GS_BLEND2
#undef GS_BLEND2_
    GS_HIP(hipGetLastError());
    return GS_OK;
}


// run_round_spans() for two frames, one launch per kernel
template <int ROUND>
int run_round_spans2(gs_ctx *const S[2], const GsFrameUniforms U[2], uint8_t *const out[2], bool last_round, uint32_t g, uint32_t stride)
{
    gs_ctx *ctx = S[0];
    const GsFrameUniforms &u = U[0];
    hipStream_t st = ctx->stream;
    for (int k = 0; k < 2; k++) { const int rcc = gs_ensure_row_tables(S[k], (size_t)stride * (size_t)u.tiles_y); if (rcc != GS_OK) return rcc; }
    GsFrameUniforms V[2] = { U[0], U[1] };
    for (int k = 0; k < 2; k++) V[k].rc_stride = stride;
    uint32_t *geom[2], *ref[2], *lists[2];
    for (int k = 0; k < 2; k++) { geom[k] = reinterpret_cast<uint32_t *>(S[k]->pair_a); ref[k] = geom[k] + S[k]->pair_cap; lists[k] = reinterpret_cast<uint32_t *>(S[k]->pair_b); }
    gs_twin<F_project<ROUND, true>, GS_BLOCK>(g, st,
        gs_pack_make((const uint32_t *)S[0]->sorted, (const uint4 *)S[0]->splat, V[0], S[0]->proj, S[0]->rect, S[0]->tile_count, S[0]->row_cnt, S[0]->part_vis,
                     (const uint32_t *)S[0]->unsat_mask, S[0]->zwin, S[0]->ctl),
        gs_pack_make((const uint32_t *)S[1]->sorted, (const uint4 *)S[1]->splat, V[1], S[1]->proj, S[1]->rect, S[1]->tile_count, S[1]->row_cnt, S[1]->part_vis,
                     (const uint32_t *)S[1]->unsat_mask, S[1]->zwin, S[1]->ctl));
    GS_HIP(hipGetLastError());
    if (ROUND == 0) GS_PROF_RECORD(ctx, 3);
    gs_twin<F_row_scan<ROUND>, GS_BLOCK>((uint32_t)u.tiles_y, st,
        gs_pack_make(S[0]->row_cnt, S[0]->row_tot, (const GsControl *)S[0]->ctl, U[0].near_count, stride, (uint32_t)u.tiles_y, S[0]->unsat_mask, u.mask_words),
        gs_pack_make(S[1]->row_cnt, S[1]->row_tot, (const GsControl *)S[1]->ctl, U[1].near_count, stride, (uint32_t)u.tiles_y, S[1]->unsat_mask, u.mask_words));
    gs_twin<F_emit_runs<ROUND>, GS_BLOCK>(g, st,
        gs_pack_make((const gsm::Projected *)S[0]->proj, (const uint2 *)S[0]->rect, (const uint32_t *)S[0]->tile_count, (const uint32_t *)S[0]->row_cnt,
                     (const uint2 *)S[0]->row_tot, V[0], geom[0], ref[0], (const uint32_t *)S[0]->unsat_mask, (const GsControl *)S[0]->ctl, (uint32_t)S[0]->pair_cap),
        gs_pack_make((const gsm::Projected *)S[1]->proj, (const uint2 *)S[1]->rect, (const uint32_t *)S[1]->tile_count, (const uint32_t *)S[1]->row_cnt,
                     (const uint2 *)S[1]->row_tot, V[1], geom[1], ref[1], (const uint32_t *)S[1]->unsat_mask, (const GsControl *)S[1]->ctl, (uint32_t)S[1]->pair_cap));
    uint32_t gl = (uint32_t)u.tiles_y * GS_LIST_SEGS; if (gl > (ROUND == 1 ? 512u : 2048u)) gl = ROUND == 1 ? 512u : 2048u;
    const bool segc = ROUND == 0 && __atomic_load_n(&gs_root(ctx)->run_hint, __ATOMIC_RELAXED) > GS_SEGC_RUNS_PER_ROW * (uint32_t)u.tiles_y;
    if (segc)
        gs_twin<F_seg_count<ROUND>, GS_BLOCK>(gl, st,
            gs_pack_make((const uint32_t *)geom[0], (const uint2 *)S[0]->row_tot, S[0]->seg_diff, V[0], (const GsControl *)S[0]->ctl, (uint32_t)S[0]->pair_cap),
            gs_pack_make((const uint32_t *)geom[1], (const uint2 *)S[1]->row_tot, S[1]->seg_diff, V[1], (const GsControl *)S[1]->ctl, (uint32_t)S[1]->pair_cap));
#define GS_LISTS2(SC) gs_twin<F_lists<ROUND, SC>, GS_BLOCK>(gl, st,                                                                                              \
        gs_pack_make((const uint32_t *)geom[0], (const uint32_t *)ref[0], (const uint2 *)S[0]->row_tot, (const int *)S[0]->seg_diff, lists[0], S[0]->tile_range, V[0], \
                     (const uint32_t *)S[0]->unsat_mask, S[0]->ctl, (uint32_t)S[0]->pair_cap, (const uint32_t *)S[0]->part_vis, g, last_round ? 1 : 0),                \
        gs_pack_make((const uint32_t *)geom[1], (const uint32_t *)ref[1], (const uint2 *)S[1]->row_tot, (const int *)S[1]->seg_diff, lists[1], S[1]->tile_range, V[1], \
                     (const uint32_t *)S[1]->unsat_mask, S[1]->ctl, (uint32_t)S[1]->pair_cap, (const uint32_t *)S[1]->part_vis, g, last_round ? 1 : 0))
    if (segc) GS_LISTS2(true); else GS_LISTS2(false);
#undef GS_LISTS2
    GS_HIP(hipGetLastError());
    if (ROUND == 0) GS_PROF_RECORD(ctx, 4);
    const void *fpairs[2] = { lists[0], lists[1] };
    const gsm::Projected *bproj[2] = { S[0]->proj, S[1]->proj };
    const float *bzwin[2] = { S[0]->zwin, S[1]->zwin };
    return launch_blend2<ROUND>(S, u, V, out, fpairs, bproj, bzwin);
}

// run_round() for two frames that take the same path: every kernel once, on a grid (x, 2) (blockIdx.y = the frame)
template <int ROUND>
int run_round2(gs_ctx *const S[2], const GsFrameUniforms U[2], uint8_t *const out[2], bool last_round)
{
    gs_ctx *ctx = S[0];
    const GsFrameUniforms &u = U[0];
    const uint32_t ntiles = (uint32_t)u.tiles_x * (uint32_t)u.tiles_y;
    const uint32_t Vmax = (uint32_t)ctx->n;
    hipStream_t st = ctx->stream;
    uint32_t g = gs_div_up(Vmax, GS_BLOCK); if (g > GS_MAX_PART) g = GS_MAX_PART;
    const uint32_t small = 512;
    if (ROUND == 1 && g > small) g = small;
    if (ROUND == 0 && u.near_count != 0xFFFFFFFFu) { const uint32_t gn = gs_div_up(u.near_count < Vmax ? u.near_count : Vmax, GS_BLOCK); if (gn < g) g = gn ? gn : 1; }
    const uint32_t pc = (uint32_t)(S[0]->pair_cap < S[1]->pair_cap ? S[0]->pair_cap : S[1]->pair_cap);
    const uint32_t ph = ROUND == 1 ? (uint32_t)(small * GS_CHUNK_S) : __atomic_load_n(&gs_root(ctx)->pair_hint, __ATOMIC_RELAXED);
    const uint32_t jrange = ROUND == 0 ? (u.near_count != 0xFFFFFFFFu && u.near_count < Vmax ? u.near_count : Vmax) : Vmax;
    if (const uint32_t stride = span_list_stride(ctx, u, jrange)) return run_round_spans2<ROUND>(S, U, out, last_round, g, stride);
    const int tb = bits_for(ntiles);
    GsFrameUniforms V[2] = { U[0], U[1] };
    V[0].rc_stride = V[1].rc_stride = 0;
    gs_twin<F_project<ROUND, false>, GS_BLOCK>(g, st,
        gs_pack_make((const uint32_t *)S[0]->sorted, (const uint4 *)S[0]->splat, U[0], S[0]->proj, S[0]->rect, S[0]->tile_count, S[0]->spine, S[0]->part_vis,
                     (const uint32_t *)S[0]->unsat_mask, S[0]->zwin, S[0]->ctl),
        gs_pack_make((const uint32_t *)S[1]->sorted, (const uint4 *)S[1]->splat, U[1], S[1]->proj, S[1]->rect, S[1]->tile_count, S[1]->spine, S[1]->part_vis,
                     (const uint32_t *)S[1]->unsat_mask, S[1]->zwin, S[1]->ctl));
    GS_HIP(hipGetLastError());
    if (ROUND == 0) GS_PROF_RECORD(ctx, 3);
    gs_twin<F_pairs_check<ROUND>, GS_BLOCK>(1, st,
        gs_pack_make(S[0]->ctl, (uint32_t)S[0]->pair_cap, S[0]->spine, (const uint32_t *)S[0]->part_vis, g, U[0].near_count, last_round ? 1 : 0, S[0]->unsat_mask,
                     (uint32_t)u.tiles_y * u.mask_words, S[0]->emit_extra),
        gs_pack_make(S[1]->ctl, (uint32_t)S[1]->pair_cap, S[1]->spine, (const uint32_t *)S[1]->part_vis, g, U[1].near_count, last_round ? 1 : 0, S[1]->unsat_mask,
                     (uint32_t)u.tiles_y * u.mask_words, S[1]->emit_extra));
    uint32_t ge = g + (ROUND == 1 ? 0u : gs_div_up(ph ? ph : pc, GS_EMIT_PAIRS)); if (ge > GS_MAX_PART) ge = GS_MAX_PART;
    gs_twin<F_emit<ROUND>, GS_BLOCK>(ge, st,
        gs_pack_make((const gsm::Projected *)S[0]->proj, (const uint2 *)S[0]->rect, (const uint32_t *)S[0]->tile_count, (const uint32_t *)S[0]->spine,
                     (const uint2 *)S[0]->emit_extra, V[0], S[0]->pair_a, (const uint32_t *)S[0]->unsat_mask, (const GsControl *)S[0]->ctl),
        gs_pack_make((const gsm::Projected *)S[1]->proj, (const uint2 *)S[1]->rect, (const uint32_t *)S[1]->tile_count, (const uint32_t *)S[1]->spine,
                     (const uint2 *)S[1]->emit_extra, V[1], S[1]->pair_a, (const uint32_t *)S[1]->unsat_mask, (const GsControl *)S[1]->ctl));
    GS_HIP(hipGetLastError());
    const int fmt = GS_RADIX_PACKED;
    const void *in[2]; void *outp[2]; const uint32_t *np[2] = { &S[0]->ctl->n_pairs, &S[1]->ctl->n_pairs };
    uint32_t *cnt[2] = { nullptr, nullptr }; const uint32_t *fill[2] = { nullptr, nullptr };
    const void *fpairs[2];
    int rc;
    if (tb <= 9) {
        for (int k = 0; k < 2; k++) { in[k] = S[k]->pair_a; outp[k] = S[k]->pair_b; }
        rc = gs_launch_radix_pass2(S, in, fmt, outp, fmt, np, pc, ph, 0, tb, false, 0xFFFFFFFFu, 0, cnt, fill);
        if (rc != GS_OK) return rc;
        fpairs[0] = S[0]->pair_b; fpairs[1] = S[1]->pair_b;
    } else {
        const int b1 = (tb + 1) / 2, b2 = tb - b1;
        for (int k = 0; k < 2; k++) { in[k] = S[k]->pair_a; outp[k] = S[k]->pair_b; }
        rc = gs_launch_radix_pass2(S, in, fmt, outp, fmt, np, pc, ph, 0, b1, false, 0xFFFFFFFFu, 0, cnt, fill);
        if (rc != GS_OK) return rc;
        for (int k = 0; k < 2; k++) { in[k] = S[k]->pair_b; outp[k] = S[k]->pair_a; }
        rc = gs_launch_radix_pass2(S, in, fmt, outp, fmt, np, pc, ph, b1, b2, false, 0xFFFFFFFFu, 0, cnt, fill);
        if (rc != GS_OK) return rc;
        fpairs[0] = S[0]->pair_a; fpairs[1] = S[1]->pair_a;
    }
    gs_twin<F_tile_ranges, GS_BLOCK>(ROUND == 1 ? small : 2048, st,
                                     gs_pack_make((const uint2 *)fpairs[0], S[0]->tile_range, ntiles, ROUND, (const GsControl *)S[0]->ctl),
                                     gs_pack_make((const uint2 *)fpairs[1], S[1]->tile_range, ntiles, ROUND, (const GsControl *)S[1]->ctl));
    GS_HIP(hipGetLastError());
    if (ROUND == 0) GS_PROF_RECORD(ctx, 4);
    const gsm::Projected *bproj[2] = { S[0]->proj, S[1]->proj };
    const float *bzwin[2] = { S[0]->zwin, S[1]->zwin };
    return launch_blend2<ROUND>(S, u, V, out, fpairs, bproj, bzwin);
}

}  // namespace

int gs_row_tables_ensure(gs_ctx *ctx, size_t entries) { return gs_ensure_row_tables(ctx, entries); }
size_t gs_row_tables_entries(size_t n, uint32_t tiles_y) { return gs_row_table_entries(n, tiles_y); }

// Two frames that take the same path, one launch per kernel (GS_OPT_FRAME_BATCH; grid (x, 2), blockIdx.y = the frame).  S[0], S[1]:
// sibling lanes on ONE stream, each with its own scratch, control block and output.  Frames that count fragments or record
// the staged depths take the per-frame path -- gs_frames_batchable() says whether two frames qualify.
bool gs_frames_batchable(const GsFrameUniforms &a, const GsFrameUniforms &b)
{
    return a.near_count == b.near_count && a.skip_round1 == b.skip_round1 && a.W == b.W && a.H == b.H && a.x0 == b.x0 && a.x1 == b.x1 &&
           a.flags == b.flags && !(a.flags & (GS_RENDER_COUNT_FRAGS | GS_RENDER_COUNT_EVALUATED)) && !a.record_staged && !b.record_staged &&
           a.split_min == b.split_min && a.subtile == b.subtile && a.has_depth == b.has_depth && a.has_scene_rgba == b.has_scene_rgba && a.t_eps == b.t_eps;
}

int gs_run_render2(gs_ctx *const S[2], const GsFrameUniforms U[2], uint8_t *const device_out[2])
{
    gs_ctx *ctx = S[0];
    const GsFrameUniforms &u = U[0];
    uint8_t *out[2] = { device_out[0] ? device_out[0] : S[0]->fb, device_out[1] ? device_out[1] : S[1]->fb };
    GS_PROF_RECORD(ctx, 2);
    const bool two_rounds = u.near_count != 0xFFFFFFFFu;
    int rc = run_round2<0>(S, U, out, !two_rounds || u.skip_round1);
    if (rc != GS_OK) return rc;
    GS_PROF_RECORD(ctx, 5);
    if (two_rounds && !u.skip_round1) {
        rc = run_round2<1>(S, U, out, true);
        if (rc != GS_OK) return rc;
    }
    GS_PROF_RECORD(ctx, 6);
    return GS_OK;
}

// a skipped round 1 turned out to be needed: run it now (mask + state of the frame are still intact)
int gs_run_round1(gs_ctx *ctx, const GsFrameUniforms &u, uint8_t *device_out)
{
    GsFrameUniforms v = u; v.skip_round1 = 0;
    return run_round<1>(ctx, v, device_out ? device_out : ctx->fb, false);
}

int gs_run_render(gs_ctx *ctx, const GsFrameUniforms &u, uint8_t *device_out)
{
    const uint32_t ntiles = (uint32_t)u.tiles_x * (uint32_t)u.tiles_y;
    const uint32_t Vmax = (uint32_t)ctx->n;
    uint8_t *out = device_out ? device_out : ctx->fb;
    hipStream_t st = ctx->stream;

    GS_PROF_RECORD(ctx, 2);
    if (u.flags & GS_RENDER_COUNT_FRAGS) GS_HIP(hipMemsetAsync(&ctx->ctl->n_frags, 0, sizeof(unsigned long long), st));
    if (Vmax && ctx->have_sort) {
        const bool two_rounds = u.near_count != 0xFFFFFFFFu;
        int rc = run_round<0>(ctx, u, out, !two_rounds || u.skip_round1);
        if (rc != GS_OK) return rc;
        GS_PROF_RECORD(ctx, 5);
        if (two_rounds && !u.skip_round1) {
            rc = run_round<1>(ctx, u, out, true);
            if (rc != GS_OK) return rc;
        }
    } else {
        // nothing resident / never sorted: the frame is the background
        GS_HIP(hipMemsetAsync(ctx->ctl, 0, sizeof(GsControl), st));
        GS_HIP(hipMemsetAsync(ctx->tile_range, 0, sizeof(uint2) * ntiles, st));
        GsFrameUniforms ub = u; ub.near_count = 0xFFFFFFFFu;
        GS_PROF_RECORD(ctx, 3); GS_PROF_RECORD(ctx, 4);
        if (ub.has_scene_rgba)
            hipLaunchKernelGGL((k_blend<false, 0, true>), dim3(ntiles), dim3(64), 0, st, ctx->tile_range, ctx->pair_a, ctx->proj, ub, out,
                               ctx->state, ctx->unsat_mask, ctx->zwin, ctx->scene_depth, ctx->scene_rgba, ctx->ctl);
        else
            hipLaunchKernelGGL((k_blend<false, 0, false>), dim3(ntiles), dim3(64), 0, st, ctx->tile_range, ctx->pair_a, ctx->proj, ub, out,
                               ctx->state, ctx->unsat_mask, ctx->zwin, ctx->scene_depth, ctx->scene_rgba, ctx->ctl);
        GS_HIP(hipGetLastError());
        GS_PROF_RECORD(ctx, 5);
    }
    GS_PROF_RECORD(ctx, 6);
    return GS_OK;
}

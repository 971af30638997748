// gs_sort.hip -- the worker's sortSplats (index.js:507-570) as HIP kernels.
//
// Bit-exact contract: view depth, min/max and the bucket scale are IEEE f64 evaluated
// left to right without FMA (library is built -ffp-contract=off); only the stored depth is rounded to
// f32; the result is ordered by (bucket, original index).  The 17-bit key (16-bit bucket + one value for dropped
// buckets) is sorted by two stable radix passes of 8 + 9 bits.  Culled splats leave in pass A (GS_RADIX_SKIP records are
// neither counted nor scattered), so pass B runs over the V kept splats.  Kept splats whose bucket falls outside
// [0,65535] (f32 rounding of the stored depth >> depth range) carry key 65536, sort behind every bucket and store 0 --
// exactly like the reference's out-of-bounds typed-array writes leave 0 in the tail [V',V) of its result.
#include "gs_internal.h"

namespace {

struct SortUniforms {                                            // widened on the host (exact): scalar registers
    double view[4]; double cutout[16]; int has_cutout;          // has_cutout 2: the matrix is affine (last row 0 0 0 1): gsm::in_cutout_affine
    int has_strip;
};
// gs_sort_for: rows 0, 1, 2 of gsModelViewMatrix, rows 0, 3 of gsProjectionMatrix, and the strip in pixels
struct StripUniforms { float mvr0[4], mvr1[4], mvr2[4], pr0[4], pr1[4], pr3[4]; float focal, norm_a, half_w, sx0, sx1, half_h, sy1; };   // (sy1 > 0: rows [0, sy1) tested too)
// the depth kernel's chunking is its own (no histogram depends on it)
#ifndef GS_DEPTH_IPT
#define GS_DEPTH_IPT 4             // items per thread and pass: 52 vector registers, 8 waves per SIMD (8 items: 92, 5 waves)
#endif
#ifndef GS_DEPTH_GRID
#define GS_DEPTH_GRID 2048u        // workgroups at most (one partial min/max/count slot each)
#endif

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, m, 64); hi = __shfl_xor(hi, m, 64);
    return ((unsigned long long)hi << 32) | lo;
}
// min / max of two doubles that are not NaN, one instruction each (the builtins first canonicalise both operands: three more)
__device__ __forceinline__ double min_f64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double max_f64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double shfl_xor_f64(double v, int m)
{
    return __longlong_as_double((long long)shfl_xor_u64((unsigned long long)__double_as_longlong(v), m));
}

// Near-only sorts (gs_run_sort with near_req): the pass also histograms the kept depths by the top 11 bits of the stored f32
// |depth| (kept depths are negative, so the sign-less bit pattern grows with the distance: exponent + 3 mantissa bits, 12 %
// steps); k_sort_bucket derives from it a bucket below which no splat can be among the nearest near_req.  Workgroups add
// their LDS histogram to one of GS_DH_COPIES global copies (same-address atomics serialise at ~11 ns: 2048 workgroups on one
// copy cost 22 us).  (Measured and dropped: the workgroup that finishes last reducing the copies to the threshold itself --
// the release / acquire fences that takes across the 8 XCDs' L2s cost 70 us per sort.)
// fill: this sort's histogram (nullptr: none wanted); zero: the one the lane's previous near-only sort filled, cleared here
// (its reader finished long ago: stream order); zero_word: the control block's count of valid buckets, recounted by k_sort_bucket.
struct DepthHist { uint32_t *fill, *zero, *zero_word; uint32_t *zero_grp; uint32_t zero_grp_words; };   // zero_grp: the MSD sort's group rows (below), cleared for k_sort_bucket<.., MSD>'s atomics
__device__ __forceinline__ uint32_t depth_bin(float d) { return (__float_as_uint(d) & 0x7FFFFFFFu) >> 20; }
__device__ __forceinline__ void depth_hist_begin(const DepthHist &dh, uint32_t *s_dh)
{
    if (dh.fill) for (uint32_t d = threadIdx.x; d < GS_DEPTH_BINS; d += GS_BLOCK) s_dh[d] = 0;
    if (blockIdx.x == 0) {
        if (dh.zero) for (uint32_t d = threadIdx.x; d < GS_DH_WORDS / 4u; d += GS_BLOCK) reinterpret_cast<uint4 *>(dh.zero)[d] = make_uint4(0, 0, 0, 0);
        if (dh.zero_word && threadIdx.x == 0) *dh.zero_word = 0;
        if (dh.zero_grp) for (uint32_t d = threadIdx.x; d < dh.zero_grp_words / 4u; d += GS_BLOCK) reinterpret_cast<uint4 *>(dh.zero_grp)[d] = make_uint4(0, 0, 0, 0);
    }
}
// (after a barrier that follows the last LDS atomic)  Two levels: the bins, and behind them sums over 32 consecutive bins each
// (GS_DEPTH_COARSE = 64 words per copy), so that a reader finds the bin it wants with ~800 loads instead of all 16 K words
__device__ __forceinline__ void depth_hist_end(const DepthHist &dh, const uint32_t *s_dh)
{
    if (!dh.fill) return;
    const uint32_t cp = blockIdx.x % GS_DH_COPIES;
    uint32_t *fine = dh.fill + cp * GS_DEPTH_BINS, *coarse = dh.fill + GS_DH_COPIES * GS_DEPTH_BINS + cp * GS_DEPTH_COARSE;
    for (uint32_t d = threadIdx.x; d < GS_DEPTH_BINS; d += GS_BLOCK) {
        const uint32_t v = s_dh[d];
        if (v) atomicAdd(&fine[d], v);
        uint32_t t = v;                                              // lanes 0..31 / 32..63 of a wave hold 32 consecutive bins
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
        if ((threadIdx.x & 31u) == 0u && t) atomicAdd(&coarse[d >> 5], t);
    }
}

// pass 1 (index.js:517-555): depth, culls, f64 min/max of survivors.  16 B/splat in, 4 B/splat out.
// Each workgroup leaves its (min, max, count) in a partial slot; no global atomics.
// strip test of gs_sort_for: can a fragment of the splat fall on pixel columns [sx0, sx1]?  Fragments live where |p| <= 2
// (index.js:171-172): an ellipse with semi-axes 2 v1, 2 v2 around the projected centre, whose half-width in x is
// 2 sqrt(v1x^2 + v2x^2) <= 2 |v1| = 2 min(sqrt(2 lambda1), 1024) (index.js:139-149), and
// lambda1 <= (|J|_2 |A|_2 sigma_max)^2 + 0.3 with J the shader's Jacobian at the splat's camera-space position
// (index.js:127-135; J J^T = (f/z)^2 [[1 + rx^2, -rx ry], [-rx ry, 1 + ry^2]], largest eigenvalue (f/z)^2 (1 + rx^2 + ry^2)).
// Conservative by construction, 1 % + 2 pixels of slack on top; NaN anywhere keeps the splat.
// (fp32 with hardware rcp / sqrt: the test only has to err on the side of keeping, and 3 % + 3 pixels of slack dwarf the
// rounding; in f64 -- divisions, a square root -- it cost more than the whole depth pass: 92 -> 158 us at 20 M splats.)
// camz comes from the STRIP's own modelView row 2, not from the sort's view row: the two coincide for a mono frame, but an XR
// eye is sorted with the head camera's view (index.js:441) and drawn with its own, possibly canted, matrix (index.js:185-187).
__device__ __forceinline__ bool strip_may_touch(const StripUniforms &s, float x, float y, float z, float sigma)
{
    const float camz = fmaf(s.mvr2[2], z, fmaf(s.mvr2[1], y, s.mvr2[0] * x)) + s.mvr2[3];
    const float camx = fmaf(s.mvr0[2], z, fmaf(s.mvr0[1], y, s.mvr0[0] * x)) + s.mvr0[3];
    const float camy = fmaf(s.mvr1[2], z, fmaf(s.mvr1[1], y, s.mvr1[0] * x)) + s.mvr1[3];
    const float cw = fmaf(s.pr3[2], camz, fmaf(s.pr3[1], camy, s.pr3[0] * camx)) + s.pr3[3];
    const float cx = fmaf(s.pr0[2], camz, fmaf(s.pr0[1], camy, s.pr0[0] * camx)) + s.pr0[3];
    if (!(cw > 0.0f)) return true;
    const float xpx = fmaf(cx, __builtin_amdgcn_rcpf(cw), 1.0f) * s.half_w;
    const float iz = __builtin_amdgcn_rcpf(camz), rx = camx * iz, ry = camy * iz;
    const float jn = fabsf(s.focal * iz) * __builtin_amdgcn_sqrtf(fmaf(rx, rx, fmaf(ry, ry, 1.0f)));
    const float sd = jn * s.norm_a * sigma;
    float ax = __builtin_amdgcn_sqrtf(2.0f * fmaf(sd, sd, 0.3f));
    if (!(ax < 1024.0f)) ax = 1024.0f;
    const float reach = fmaf(2.06f, ax, 3.0f);
    if (xpx + reach < s.sx0 || xpx - reach > s.sx1) return false;
    // the rows of the frame (round 6: a strip is as tall as its frame, and gs_sort_for over the WHOLE frame is a frustum cull): the same
    // bound serves both axes -- 2 sqrt(v1y^2 + v2y^2) <= 2 |v1| too
    if (s.sy1 > 0.0f) {
        const float cy = fmaf(s.pr1[2], camz, fmaf(s.pr1[1], camy, s.pr1[0] * camx)) + s.pr1[3];
        const float ypx = fmaf(cy, __builtin_amdgcn_rcpf(cw), 1.0f) * s.half_h;
        if (ypx + reach < 0.0f || ypx - reach > s.sy1) return false;
    }
    return true;
}

// SPEC (near-only sorts of long inputs, round 4): the pass hands the candidates on itself.  A near-only sort keeps the splats whose
// bucket reaches a threshold that is only known after the pass (it comes from this pass' depth histogram and min / max) -- so the
// depths of all N splats were written (80 MB at 20 M) and read again by k_near_stash, which recomputed 20 M buckets in f64 to keep
// 1.5 % of them.  But the threshold BIN moves slowly from frame to frame: with the bin the context's last near-only sort found
// (*bin_hint) plus GS_SPEC_PAD the pass stashes every splat whose depth falls in those bins -- (stored f32 depth, index), in index
// order, GS_SPEC_SLOT records per chunk of 1024 -- and writes no depth array at all; k_near_filter then applies the EXACT rule to
// the candidates alone (same f32 depth, same min / max, same bucket arithmetic: the same records as before) after checking that
// the candidates were a superset of what the exact threshold keeps.  If they were not (the camera jumped; a stash overflowed; the
// depth range is so short that buckets could be dropped, which only a pass over every depth can count), the frame is flagged like
// an overflowed stash: drawn again from a whole sort, and the hint is exact for the next frame.
#ifndef GS_SPEC_SLOT
#define GS_SPEC_SLOT 128u          // candidates a 1024-item chunk of the depth pass may stash (the share asked for is <= 1/32: 32 on average)
#endif
#ifndef GS_SPEC_PAD
#define GS_SPEC_PAD 2u             // bins (12 % of depth each) beyond the last sort's threshold bin that are stashed too (one is needed when the bin
                                   // has not moved: the threshold BUCKET lies inside the next bin; the second is the margin for a moving camera)
#endif
#define GS_SPEC_GROUP 32u          // chunks (of 1024 items) one k_near_filter workgroup compacts: 4096 stash slots
// the stash step of one chunk: ranks by (row, wavefront, lane) = index order
#define GS_SPEC_STASH_STEP(SP, FB, SROW, STASH, CNT, CTL, LIMREC) do {                                                                     \
        uint32_t bef_[GS_DEPTH_IPT];                                                                                             \
        _Pragma("unroll") for (int r = 0; r < GS_DEPTH_IPT; r++) {                                                                \
            const unsigned long long bal_ = __ballot(SP[r]);                                                                     \
            bef_[r] = (uint32_t)__popcll(bal_ & ((1ull << (threadIdx.x & 63)) - 1ull));                                           \
            if ((threadIdx.x & 63) == 0) SROW[r * 4 + (threadIdx.x >> 6)] = (uint32_t)__popcll(bal_);                              \
        }                                                                                                                        \
        __syncthreads();                                                                                                         \
        if (threadIdx.x < 64) {                                                                                                  \
            const uint32_t cv_ = threadIdx.x < 4u * GS_DEPTH_IPT ? SROW[threadIdx.x] : 0u;                                        \
            uint32_t inc_ = cv_;                                                                                                 \
            for (int d_ = 1; d_ < 64; d_ <<= 1) { const uint32_t t_ = __shfl_up(inc_, d_, 64); if ((int)threadIdx.x >= d_) inc_ += t_; } \
            if (threadIdx.x < 4u * GS_DEPTH_IPT) SROW[threadIdx.x] = inc_ - cv_;                                                  \
            const uint32_t total_ = __shfl(inc_, 63, 64);                                                                        \
            if (threadIdx.x == 0) {                                                                                              \
                CNT[c] = (total_ < GS_SPEC_SLOT ? total_ : GS_SPEC_SLOT) | (LIMREC << 16);   /* the bins THIS chunk was stashed by: checked per chunk */ \
                if (total_ > GS_SPEC_SLOT) CTL->spec_fail = 2u;                                                                   \
            }                                                                                                                    \
        }                                                                                                                        \
        __syncthreads();                                                                                                         \
        _Pragma("unroll") for (int r = 0; r < GS_DEPTH_IPT; r++) {                                                                \
            if (SP[r]) {                                                                                                         \
                const uint32_t slot_ = SROW[r * 4 + (threadIdx.x >> 6)] + bef_[r];                                                \
                if (slot_ < GS_SPEC_SLOT) STASH[(size_t)c * GS_SPEC_SLOT + slot_] = make_uint2(FB[r], c * DCHUNK + r * GS_BLOCK + threadIdx.x); \
            }                                                                                                                    \
        }                                                                                                                        \
        __syncthreads();                                                                                                         \
    } while (0)

template <bool STRIP, bool SPEC>                                  // (instantiations: the strip test / the stash must not cost the plain sort registers)
__device__ __forceinline__ void k_sort_depth_body(const float4 *__restrict__ rows, const float *__restrict__ bound_r, uint32_t n, const SortUniforms &u, const StripUniforms &su,
                                                  float *__restrict__ depth_out, unsigned long long *__restrict__ part_min,
                                                  unsigned long long *__restrict__ part_max, uint32_t *__restrict__ part_cnt, DepthHist dh,
                                                  uint2 *__restrict__ spec_stash, uint32_t *__restrict__ spec_cnt, const uint32_t *__restrict__ bin_hint, GsControl *ctl)
{
    __shared__ unsigned long long s_min, s_max;
    __shared__ uint32_t s_cnt;
    __shared__ uint32_t s_srow[SPEC ? 4 * GS_DEPTH_IPT : 1];
    // the bins this workgroup stashes: up to the last sort's threshold bin + pad.  The hint is one word shared by the context's lanes
    // and may change while this kernel runs: every chunk records the limit it was stashed by, and the filter checks chunk by chunk
    const uint32_t hint_ = SPEC ? *bin_hint : 0u;
    const uint32_t spec_lim = (SPEC && hint_ != 0xFFFFFFFFu) ? hint_ + GS_SPEC_PAD : 0u, spec_rec = (SPEC && hint_ != 0xFFFFFFFFu) ? spec_lim : 0xFFFFu;
    extern __shared__ uint32_t s_dh[];                           // GS_DEPTH_BINS words for a near-only sort, none otherwise (LDS the other frames' blends can use)
    if (threadIdx.x == 0) { s_min = ~0ull; s_max = 0ull; s_cnt = 0; }
    depth_hist_begin(dh, s_dh);
    __syncthreads();
    constexpr uint32_t DCHUNK = GS_DEPTH_IPT * GS_BLOCK;         // this kernel's own chunking (no histogram depends on it)
    const uint32_t nchunks = (n + DCHUNK - 1) / DCHUNK;
    // min / max of the kept depths as doubles (kept depths are finite and negative: sort_keep), encoded once per wavefront at the end
    // -- on the ordered-u64 encoding a splat cost two 64-bit compares and four selects
    double mn = INFINITY, mx = -INFINITY;
    uint32_t cnt = 0;
    for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        float4 mm[GS_DEPTH_IPT];                                     // all loads first: their latencies overlap
        float sg[GS_DEPTH_IPT];
#pragma unroll
        for (int r = 0; r < GS_DEPTH_IPT; r++) {
            const uint32_t i = c * DCHUNK + r * GS_BLOCK + threadIdx.x;
            // (unconditional, the index clamped: a conditional load put each item's wait and f64 conversions into its own branch, so the
            // four loads of a thread went out one after the other; items beyond n are dropped below)
            const uint32_t ic = i < n ? i : n - 1u;
            mm[r] = rows[ic];
            sg[r] = STRIP ? bound_r[ic] : 0.0f;
        }
        uint32_t fb[GS_DEPTH_IPT];
        bool sp[GS_DEPTH_IPT];
#pragma unroll
        for (int r = 0; r < GS_DEPTH_IPT; r++) { fb[r] = 0u; sp[r] = false; }
#pragma unroll
        for (int r = 0; r < GS_DEPTH_IPT; r++) {
            const uint32_t i = c * DCHUNK + r * GS_BLOCK + threadIdx.x;
            if (i < n) {
                const float4 m = mm[r];
                const double d = gsm::view_depth(u.view, m.x, m.y, m.z);
                const bool inside = u.has_cutout ? (u.has_cutout == 2 ? gsm::in_cutout_affine(u.cutout, m.x, m.y, m.z) : gsm::in_cutout(u.cutout, m.x, m.y, m.z)) : true;
                const bool keep = gsm::sort_keep(d, m.w, inside);
                // the bucket scale comes from EVERY splat the reference keeps (index.js:552-553), so a strip's order is the
                // reference's order restricted to the strip's splats; only those are handed on
                const bool mine = keep && (!STRIP || strip_may_touch(su, m.x, m.y, m.z, sg[r]));
                if (!SPEC) depth_out[i] = mine ? (float)d : INFINITY;
                if (mine && dh.fill) atomicAdd(&s_dh[depth_bin((float)d)], 1u);
                if (SPEC) { fb[r] = __float_as_uint((float)d); sp[r] = mine && depth_bin((float)d) <= spec_lim; }
                if (keep) {
                    mn = min_f64(mn, d); mx = max_f64(mx, d);
                    if (mine) cnt++;
                }
            }
        }
        if (SPEC) GS_SPEC_STASH_STEP(sp, fb, s_srow, spec_stash, spec_cnt, ctl, spec_rec);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {                           // wavefront butterfly, then one LDS atomic per wave
        const double omn = shfl_xor_f64(mn, m), omx = shfl_xor_f64(mx, m);
        mn = omn < mn ? omn : mn; mx = omx > mx ? omx : mx;
        cnt += __shfl_xor(cnt, m, 64);
    }
    if ((threadIdx.x & 63) == 0 && mx > -INFINITY) {              // (the wave kept something)
        atomicMin(&s_min, gsm::f64_to_ordered(mn)); atomicMax(&s_max, gsm::f64_to_ordered(mx)); atomicAdd(&s_cnt, cnt);
    }
    __syncthreads();
    if (threadIdx.x == 0) { part_min[blockIdx.x] = s_min; part_max[blockIdx.x] = s_max; part_cnt[blockIdx.x] = s_cnt; }
    depth_hist_end(dh, s_dh);
}

template <bool STRIP, bool SPEC>
__global__ __launch_bounds__(GS_BLOCK) void k_sort_depth(const float4 *__restrict__ rows, const float *__restrict__ bound_r, uint32_t n, SortUniforms u, StripUniforms su,
                                                         float *__restrict__ depth_out, unsigned long long *__restrict__ part_min,
                                                         unsigned long long *__restrict__ part_max, uint32_t *__restrict__ part_cnt, DepthHist dh,
                                                         uint2 *__restrict__ spec_stash, uint32_t *__restrict__ spec_cnt, const uint32_t *__restrict__ bin_hint, GsControl *ctl)
{
    k_sort_depth_body<STRIP, SPEC>(rows, bound_r, n, u, su, depth_out, part_min, part_max, part_cnt, dh, spec_stash, spec_cnt, bin_hint, ctl);
}

// The same pass for the two frames of a pair (GS_OPT_FRAME_BATCH) in ONE sweep over the splats: at 20 M splats the sort rows are
// 320 MB of the 400 MB this pass streams per frame, and both frames read the same rows -- one read, two view rows / cutout
// matrices, two depth arrays and two sets of partials (20 M @ 4K: 2464 -> 2606 frames/s; no difference at 1 M, where the rows
// are cache-resident).  Per frame exactly the arithmetic of k_sort_depth.
struct SpecArgs { uint2 *stash; uint32_t *cnt; const uint32_t *bin_hint; GsControl *ctl; };
#ifndef GS_DEPTH_PAIR_MIN_N
#define GS_DEPTH_PAIR_MIN_N (1u << 22)   // paired sorts of fewer splats run k_sort_depth's body twice (gs_run_sort2)
#endif
template <bool STRIP, bool SPEC>
__global__ __launch_bounds__(GS_BLOCK) void k_sort_depth_pair(const float4 *__restrict__ rows, const float *__restrict__ bound_r, uint32_t n,
                                                              SortUniforms u0, SortUniforms u1, StripUniforms su0, StripUniforms su1,
                                                              float *__restrict__ depth0, float *__restrict__ depth1,
                                                              unsigned long long *__restrict__ pmin0, unsigned long long *__restrict__ pmax0, uint32_t *__restrict__ pcnt0,
                                                              unsigned long long *__restrict__ pmin1, unsigned long long *__restrict__ pmax1, uint32_t *__restrict__ pcnt1,
                                                              DepthHist dh0, DepthHist dh1, SpecArgs sa0, SpecArgs sa1)
{
    __shared__ unsigned long long s_min[2], s_max[2];
    __shared__ uint32_t s_cnt[2];
    __shared__ uint32_t s_srow0[SPEC ? 4 * GS_DEPTH_IPT : 1], s_srow1[SPEC ? 4 * GS_DEPTH_IPT : 1];
    const uint32_t h0_ = SPEC ? *sa0.bin_hint : 0u, h1_ = SPEC ? *sa1.bin_hint : 0u;
    const uint32_t lim0 = (SPEC && h0_ != 0xFFFFFFFFu) ? h0_ + GS_SPEC_PAD : 0u, rec0 = (SPEC && h0_ != 0xFFFFFFFFu) ? lim0 : 0xFFFFu;
    const uint32_t lim1 = (SPEC && h1_ != 0xFFFFFFFFu) ? h1_ + GS_SPEC_PAD : 0u, rec1 = (SPEC && h1_ != 0xFFFFFFFFu) ? lim1 : 0xFFFFu;
    extern __shared__ uint32_t s_dh0[];                          // 2 x GS_DEPTH_BINS words for near-only sorts, none otherwise
    uint32_t *const s_dh1 = s_dh0 + GS_DEPTH_BINS;
    if (threadIdx.x < 2) { s_min[threadIdx.x] = ~0ull; s_max[threadIdx.x] = 0ull; s_cnt[threadIdx.x] = 0; }
    depth_hist_begin(dh0, s_dh0); depth_hist_begin(dh1, s_dh1);
    __syncthreads();
    constexpr uint32_t DCHUNK = GS_DEPTH_IPT * GS_BLOCK;
    const uint32_t nchunks = (n + DCHUNK - 1) / DCHUNK;
    double mn0 = INFINITY, mx0 = -INFINITY, mn1 = INFINITY, mx1 = -INFINITY;
    uint32_t cnt0 = 0, cnt1 = 0;
    for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        float4 mm[GS_DEPTH_IPT];
        float sg[GS_DEPTH_IPT];
#pragma unroll
        for (int r = 0; r < GS_DEPTH_IPT; r++) {
            const uint32_t i = c * DCHUNK + r * GS_BLOCK + threadIdx.x;
            // (unconditional, the index clamped: a conditional load put each item's wait and f64 conversions into its own branch, so the
            // four loads of a thread went out one after the other; items beyond n are dropped below)
            const uint32_t ic = i < n ? i : n - 1u;
            mm[r] = rows[ic];
            sg[r] = STRIP ? bound_r[ic] : 0.0f;
        }
        uint32_t fb0[GS_DEPTH_IPT], fb1[GS_DEPTH_IPT];
        bool sp0[GS_DEPTH_IPT], sp1[GS_DEPTH_IPT];
#pragma unroll
        for (int r = 0; r < GS_DEPTH_IPT; r++) { fb0[r] = fb1[r] = 0u; sp0[r] = sp1[r] = false; }
#pragma unroll
        for (int r = 0; r < GS_DEPTH_IPT; r++) {
            const uint32_t i = c * DCHUNK + r * GS_BLOCK + threadIdx.x;
            if (i < n) {
                const float4 m = mm[r];
#define GS_DEPTH_ONE(U, SU, OUT, MN, MX, CNT, DH, SDH, FB, SP, LIM) do {                                                \
                    const double d = gsm::view_depth(U.view, m.x, m.y, m.z);                                               \
                    const bool inside = U.has_cutout ? (U.has_cutout == 2 ? gsm::in_cutout_affine(U.cutout, m.x, m.y, m.z) : gsm::in_cutout(U.cutout, m.x, m.y, m.z)) : true; \
                    const bool keep = gsm::sort_keep(d, m.w, inside);                                                      \
                    const bool mine = keep && (!STRIP || strip_may_touch(SU, m.x, m.y, m.z, sg[r]));            \
                    if (!SPEC) OUT[i] = mine ? (float)d : INFINITY;                                                        \
                    if (mine && DH.fill) atomicAdd(&SDH[depth_bin((float)d)], 1u);                                         \
                    if (SPEC) { FB[r] = __float_as_uint((float)d); SP[r] = mine && depth_bin((float)d) <= LIM; }           \
                    if (keep) { MN = min_f64(MN, d); MX = max_f64(MX, d); if (mine) CNT++; }                                 \
                } while (0)
                GS_DEPTH_ONE(u0, su0, depth0, mn0, mx0, cnt0, dh0, s_dh0, fb0, sp0, lim0);
                GS_DEPTH_ONE(u1, su1, depth1, mn1, mx1, cnt1, dh1, s_dh1, fb1, sp1, lim1);
#undef GS_DEPTH_ONE
            }
        }
        if (SPEC) {
            GS_SPEC_STASH_STEP(sp0, fb0, s_srow0, sa0.stash, sa0.cnt, sa0.ctl, rec0);
            GS_SPEC_STASH_STEP(sp1, fb1, s_srow1, sa1.stash, sa1.cnt, sa1.ctl, rec1);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double a0 = shfl_xor_f64(mn0, m), b0 = shfl_xor_f64(mx0, m), a1 = shfl_xor_f64(mn1, m), b1 = shfl_xor_f64(mx1, m);
        mn0 = a0 < mn0 ? a0 : mn0; mx0 = b0 > mx0 ? b0 : mx0; mn1 = a1 < mn1 ? a1 : mn1; mx1 = b1 > mx1 ? b1 : mx1;
        cnt0 += __shfl_xor(cnt0, m, 64); cnt1 += __shfl_xor(cnt1, m, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (mx0 > -INFINITY) { atomicMin(&s_min[0], gsm::f64_to_ordered(mn0)); atomicMax(&s_max[0], gsm::f64_to_ordered(mx0)); atomicAdd(&s_cnt[0], cnt0); }
        if (mx1 > -INFINITY) { atomicMin(&s_min[1], gsm::f64_to_ordered(mn1)); atomicMax(&s_max[1], gsm::f64_to_ordered(mx1)); atomicAdd(&s_cnt[1], cnt1); }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        pmin0[blockIdx.x] = s_min[0]; pmax0[blockIdx.x] = s_max[0]; pcnt0[blockIdx.x] = s_cnt[0];
        pmin1[blockIdx.x] = s_min[1]; pmax1[blockIdx.x] = s_max[1]; pcnt1[blockIdx.x] = s_cnt[1];
    }
    depth_hist_end(dh0, s_dh0); depth_hist_end(dh1, s_dh1);
}

// pass 2 (index.js:558-561): 16-bit bucket of the stored depth; culled -> GS_RADIX_SKIP, dropped bucket -> GS_CULLED_KEY.
// Every workgroup first folds pass 1's partials (<= 2048 slots, L2-resident) into the global min/max.  One workgroup per
// radix chunk (geometry NW as in gs_prims.hip): it also leaves radix pass A's histogram row of the chunk.
// COMPACT (N <= 2^25): pass A sorts by the low 9 bucket bits and writes 4-byte records `bucket >> 9 << 25 | index`, pass B
// by the 7 bits on top of the index: half the bytes of (key, index) records through three of the five streaming passes.
// A 17th key bit has no room there, so kept splats with a dropped bucket leave the sort here as well; the final pass
// zero-fills their slots [V', V) behind the sorted records (the reference's never-written tail).
// NEAR (a near-only sort, COMPACT records): only the splats that can be among the nearest near_req of the order go on.
// The depth histogram gives the first bin T with (kept splats in bins <= T) >= near_req; its far edge -e is a depth, b* = its
// bucket, and a splat goes on iff its bucket >= b*: the bucket is monotonic in the stored depth, so every splat nearer than
// the edge qualifies (at least near_req of them), and the rule is a threshold on the sort key itself -- the survivors are
// exactly the last P valid positions of the whole order, in the same relative order.  The others leave like culled splats
// (GS_RADIX_SKIP).  The kernel also counts V' (valid buckets among ALL kept splats): the survivors sit at positions
// [V' - P, V') of the order (k_project subtracts V' - P).
// MSD (round 5, N <= 2^24: "the MSD sort" further down): the chunk's histogram row is that of the HIGH bucket byte -- H[chunk][bucket >> 8],
// 256 words -- and the row is also added to the row of the chunk's GROUP of GS_MSD_GROUP chunks (one atomicAdd per digit the chunk
// holds: 4096 distinct words at 1 M splats, each hit 32 times; cleared by the depth pass), so that k_msd_scatter finds the chunk's
// offsets with ~50 row reads instead of a scan launch.  NEAR + MSD: a near-only sort through the same four launches.
template <int NW, bool COMPACT, bool NEAR, bool MSD = false>
__device__ __forceinline__ void k_sort_bucket_body(const float *__restrict__ depth, uint32_t n, uint32_t *__restrict__ keys,
                                                   const unsigned long long *__restrict__ part_min,
                                                   const unsigned long long *__restrict__ part_max,
                                                   const uint32_t *__restrict__ part_cnt, uint32_t nparts,
                                                   uint32_t *__restrict__ hist, GsControl *ctl, const uint32_t *__restrict__ dhist, uint32_t near_req,
                                                   uint32_t *__restrict__ bin_hint, uint32_t *__restrict__ grp)
{
    static_assert(!NEAR || COMPACT, "near-only sorts use the compact records");
    static_assert(!MSD || COMPACT, "the MSD sort takes compact records");
    constexpr int NT = 64 * NW, IPT = 8, CH = NT * IPT;
    __shared__ unsigned long long s_min, s_max;
    __shared__ uint32_t s_cnt;
    constexpr uint32_t BINS = MSD ? 256u : (COMPACT ? 512u : 256u);
    __shared__ uint32_t s_hist[BINS];                             // low-digit histogram of this chunk = radix pass A's input
    __shared__ uint32_t s_nvalid;
    __shared__ int32_t s_bcut;
    if (threadIdx.x == 0) { s_min = ~0ull; s_max = 0ull; s_cnt = 0; s_nvalid = 0; s_bcut = 0; }
    // The depths of this workgroup's (first) chunk depend on nothing the kernel computes: they go out HERE, with the partials' loads below,
    // not behind the fold of min / max and its two barriers -- one dependent round trip where there were two (round 5: what the short
    // kernels of the sort wait for in the pipelined loop is round trips).  Unconditional, index clamped; what lies behind n is masked.
    const uint32_t nchunks = (n + CH - 1) / CH, vend = (nchunks + 7u) & ~7u;
    float dd[IPT];
    uint32_t c = 0;
    auto load_chunk = [&](uint32_t cc) {
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t i = cc * CH + r * NT + threadIdx.x;
            dd[r] = depth[i < n ? i : n - 1u];                           // (n >= 1: there is a chunk)
        }
    };
    bool preloaded = blockIdx.x < vend && gs_xcd_chunk(blockIdx.x, nchunks, c);
    if (preloaded) load_chunk(c);
    __syncthreads();
    {
        unsigned long long mn = ~0ull, mx = 0ull; uint32_t cnt = 0;
        for (uint32_t i = threadIdx.x; i < nparts; i += NT) {
            const unsigned long long a = part_min[i], b = part_max[i];
            mn = a < mn ? a : mn; mx = b > mx ? b : mx; cnt += part_cnt[i];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned long long omn = shfl_xor_u64(mn, m), omx = shfl_xor_u64(mx, m);
            mn = omn < mn ? omn : mn; mx = omx > mx ? omx : mx;
            cnt += __shfl_xor(cnt, m, 64);
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s_min, mn); atomicMax(&s_max, mx); atomicAdd(&s_cnt, cnt); }
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->min_enc = s_min; ctl->max_enc = s_max; ctl->n_kept = s_cnt; ctl->n_total = n; ctl->near_sorted = NEAR ? 1u : 0u; }
    const double mn = gsm::ordered_to_f64(s_min), mx = gsm::ordered_to_f64(s_max);
    const double inv = 65535.0 / (mx - mn);                       // (256*256-1)/(maxDepth-minDepth)
    if (NEAR) {
        // the first bin T with (kept splats in bins <= T) >= near_req, by the first wavefront: the 64 coarse sums (over the
        // copies), a scan, then the 32 bins of the coarse bin where the count is reached
        if (threadIdx.x < 64) {
            const int l = threadIdx.x;
            auto wave_scan = [&](uint32_t v) { for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(v, d, 64); if (l >= d) v += t; } return v; };
            uint32_t cs = 0;
#pragma unroll
            for (uint32_t c = 0; c < GS_DH_COPIES; c++) cs += dhist[GS_DH_COPIES * GS_DEPTH_BINS + c * GS_DEPTH_COARSE + l];
            const uint32_t cinc = wave_scan(cs);
            const unsigned long long m = __ballot(cinc >= near_req);
            uint32_t T = GS_DEPTH_BINS - 1u;                         // (fewer kept splats than near_req: all of them)
            if (m) {
                const int C = __ffsll((long long)m) - 1;
                const uint32_t before = __shfl(cinc - cs, C, 64);    // kept splats nearer than coarse bin C
                uint32_t fs = 0;
                if (l < 32) {
#pragma unroll
                    for (uint32_t c = 0; c < GS_DH_COPIES; c++) fs += dhist[c * GS_DEPTH_BINS + (uint32_t)C * 32u + l];
                }
                const uint32_t finc = wave_scan(fs);
                const unsigned long long m2 = __ballot(l < 32 && before + finc >= near_req);
                T = (uint32_t)C * 32u + (m2 ? (uint32_t)(__ffsll((long long)m2) - 1) : 31u);
            }
            if (l == 0) {
                int32_t bc = 0;
                if (T < GS_DEPTH_BINS - 1u && inv > 0.0 && inv < 1.0e300) {   // (a degenerate depth range maps everything to bucket 0: keep all)
                    const double x = ((double)(-__uint_as_float((T + 1u) << 20)) - mn) * inv;   // the far edge's place in the bucket table
                    bc = !(x >= 0.0) ? 0 : (x >= 65535.0 ? 65535 : (int32_t)x);                // (NaN / before the table: everything valid)
                }
                s_bcut = bc;
                if (blockIdx.x == 0 && bin_hint) *bin_hint = T;      // (what the next frames' speculative stash goes by)
            }
        }
        __syncthreads();
    }
    const int32_t bcut = s_bcut;
    uint32_t nvalid = 0;
    for (uint32_t v = blockIdx.x; v < vend; v += gridDim.x) {
        if (!preloaded) { if (!gs_xcd_chunk(v, nchunks, c)) continue; load_chunk(c); }   // XCD-aware chunk order (as the radix kernels); all loads first
        preloaded = false;
        for (uint32_t d = threadIdx.x; d < BINS; d += NT) s_hist[d] = 0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t i = c * CH + r * NT + threadIdx.x;
            if (i < n) {
                // culled splats leave the sort here: GS_RADIX_SKIP records are not counted and not scattered by pass A.
                // Kept splats whose bucket falls outside the table (the reference drops their writes) carry GS_CULLED_KEY:
                // they sort behind every bucket and store 0, the reference's never-written tail slots
                const float d = dd[r];
                uint32_t k = GS_RADIX_SKIP;
                if (d != INFINITY) {
                    const int32_t b = gsm::sort_bucket(d, mn, inv);
                    if (NEAR) { if (b >= 0) { nvalid++; if (b >= bcut) { k = (uint32_t)b; atomicAdd(&s_hist[MSD ? k >> 8 : k & 511u], 1u); } } }
                    else if (MSD) { if (b >= 0) { k = (uint32_t)b; atomicAdd(&s_hist[k >> 8], 1u); } }
                    else if (COMPACT) { if (b >= 0) { k = (uint32_t)b; atomicAdd(&s_hist[k & 511u], 1u); } }
                    else { k = b >= 0 ? (uint32_t)b : GS_CULLED_KEY; atomicAdd(&s_hist[k & 255u], 1u); }
                }
                keys[i] = k;
            }
        }
        __syncthreads();
        for (uint32_t d = threadIdx.x; d < BINS; d += NT) {
            const uint32_t hv = s_hist[d];
            hist[(size_t)c * BINS + d] = hv;                       // row c of hist[chunk][digit]
            if (MSD && hv) atomicAdd(&grp[(size_t)(c / GS_MSD_GROUP) * BINS + d], hv);   // ... and into the row of the chunk's group
        }
        __syncthreads();
    }
    if (NEAR) {                                                      // V' of the whole order: one global atomic per workgroup
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) nvalid += __shfl_xor(nvalid, m, 64);
        if ((threadIdx.x & 63) == 0 && nvalid) atomicAdd(&s_nvalid, nvalid);
        __syncthreads();
        if (threadIdx.x == 0 && s_nvalid) atomicAdd(&ctl->n_valid, s_nvalid);
    }
}

template <int NW, bool COMPACT, bool NEAR, bool MSD = false>
__global__ __launch_bounds__(64 * NW) void k_sort_bucket(const float *__restrict__ depth, uint32_t n, uint32_t *__restrict__ keys,
                                                         const unsigned long long *__restrict__ part_min,
                                                         const unsigned long long *__restrict__ part_max,
                                                         const uint32_t *__restrict__ part_cnt, uint32_t nparts,
                                                         uint32_t *__restrict__ hist, GsControl *ctl, const uint32_t *__restrict__ dhist, uint32_t near_req,
                                                         uint32_t *__restrict__ bin_hint, uint32_t *__restrict__ grp)
{
    k_sort_bucket_body<NW, COMPACT, NEAR, MSD>(depth, n, keys, part_min, part_max, part_cnt, nparts, hist, ctl, dhist, near_req, bin_hint, grp);
}

// Near-only sorts of LONG inputs (round 3).  k_sort_bucket<.., NEAR> still wrote a key for every one of the N splats and pass A
// still read, ranked and scanned all of them (20 M: 80 MB of keys, 5120 x 512 histogram rows) to move the 1.5 % that survive the
// threshold.  Here the bucket pass hands the survivors on directly: every 4096-item chunk stashes its survivors -- (bucket, index)
// records in INDEX order: items r * NT + t of a chunk are ranked by (r, t) through one ballot per row and a 64-entry scan of
// the (row, wave) counts -- in a fixed slot of GS_NEAR_STASH records and leaves their count; k_near_gather turns the stashes into
// one contiguous list (each workgroup sums the counts before its chunks itself: 20 KB of L2 reads at most, no scan launch), in
// chunk order, i.e. in index order, and two stable passes over the P' records (9 + 7 bucket bits) give the order the
// whole-length passes give.  A chunk with more survivors than its stash holds raises near_overflow + round1_missed: the frame
// is drawn again from a whole sort and the context stops using the stash (gs_api.hip).
template <int NW>
__device__ __forceinline__ void k_near_stash_body(const float *__restrict__ depth, uint32_t n, const unsigned long long *__restrict__ part_min,
                                                  const unsigned long long *__restrict__ part_max, const uint32_t *__restrict__ part_cnt, uint32_t nparts,
                                                  uint2 *__restrict__ stash, uint32_t *__restrict__ cnt_out, GsControl *ctl,
                                                  const uint32_t *__restrict__ dhist, uint32_t near_req, uint32_t *__restrict__ bin_hint)
{
    constexpr int NT = 64 * NW, IPT = 8, CH = NT * IPT;
    __shared__ unsigned long long s_min, s_max;
    __shared__ uint32_t s_cnt, s_nvalid;
    __shared__ int32_t s_bcut;
    __shared__ uint32_t s_row[IPT * NW];                           // survivors per (row, wave) -> their first slot
    if (threadIdx.x == 0) { s_min = ~0ull; s_max = 0ull; s_cnt = 0; s_nvalid = 0; s_bcut = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        unsigned long long mn = ~0ull, mx = 0ull; uint32_t cnt = 0;
        for (uint32_t i = threadIdx.x; i < nparts; i += NT) {
            const unsigned long long a = part_min[i], b = part_max[i];
            mn = a < mn ? a : mn; mx = b > mx ? b : mx; cnt += part_cnt[i];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned long long omn = shfl_xor_u64(mn, m), omx = shfl_xor_u64(mx, m);
            mn = omn < mn ? omn : mn; mx = omx > mx ? omx : mx;
            cnt += __shfl_xor(cnt, m, 64);
        }
        if (lane == 0) { atomicMin(&s_min, mn); atomicMax(&s_max, mx); atomicAdd(&s_cnt, cnt); }
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->min_enc = s_min; ctl->max_enc = s_max; ctl->n_kept = s_cnt; ctl->n_total = n; ctl->near_sorted = 1u; }
    const double mn = gsm::ordered_to_f64(s_min), mx = gsm::ordered_to_f64(s_max);
    const double inv = 65535.0 / (mx - mn);
    if (threadIdx.x < 64) {                                         // the threshold bucket: exactly k_sort_bucket<.., NEAR>'s rule
        const int l = threadIdx.x;
        auto wave_scan = [&](uint32_t v) { for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(v, d, 64); if (l >= d) v += t; } return v; };
        uint32_t cs = 0;
#pragma unroll
        for (uint32_t c = 0; c < GS_DH_COPIES; c++) cs += dhist[GS_DH_COPIES * GS_DEPTH_BINS + c * GS_DEPTH_COARSE + l];
        const uint32_t cinc = wave_scan(cs);
        const unsigned long long m = __ballot(cinc >= near_req);
        uint32_t T = GS_DEPTH_BINS - 1u;
        if (m) {
            const int C = __ffsll((long long)m) - 1;
            const uint32_t before = __shfl(cinc - cs, C, 64);
            uint32_t fs = 0;
            if (l < 32) {
#pragma unroll
                for (uint32_t c = 0; c < GS_DH_COPIES; c++) fs += dhist[c * GS_DEPTH_BINS + (uint32_t)C * 32u + l];
            }
            const uint32_t finc = wave_scan(fs);
            const unsigned long long m2 = __ballot(l < 32 && before + finc >= near_req);
            T = (uint32_t)C * 32u + (m2 ? (uint32_t)(__ffsll((long long)m2) - 1) : 31u);
        }
        if (l == 0) {
            int32_t bc = 0;
            if (T < GS_DEPTH_BINS - 1u && inv > 0.0 && inv < 1.0e300) {
                const double x = ((double)(-__uint_as_float((T + 1u) << 20)) - mn) * inv;
                bc = !(x >= 0.0) ? 0 : (x >= 65535.0 ? 65535 : (int32_t)x);
            }
            s_bcut = bc;
            if (blockIdx.x == 0 && bin_hint) *bin_hint = T;
        }
    }
    __syncthreads();
    const int32_t bcut = s_bcut;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t nvalid = 0;
    const uint32_t nchunks = (n + CH - 1) / CH;
    for (uint32_t v = blockIdx.x; v < ((nchunks + 7u) & ~7u); v += gridDim.x) {
        uint32_t c;
        if (!gs_xcd_chunk(v, nchunks, c)) continue;
        float dd[IPT];
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t i = c * CH + r * NT + threadIdx.x;
            dd[r] = i < n ? depth[i] : INFINITY;
        }
        int32_t bk[IPT];
        uint32_t before[IPT];
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            bk[r] = -1;
            if (dd[r] != INFINITY) {
                const int32_t b = gsm::sort_bucket(dd[r], mn, inv);
                if (b >= 0) { nvalid++; if (b >= bcut) bk[r] = b; }
            }
            const unsigned long long bal = __ballot(bk[r] >= 0);
            before[r] = (uint32_t)__popcll(bal & lt);
            if (lane == 0) s_row[r * NW + w] = (uint32_t)__popcll(bal);
        }
        __syncthreads();
        uint32_t total = 0;
        if (threadIdx.x < 64) {                                     // exclusive scan of the IPT * NW (= 64 at NW = 8) counts, row-major = index order
            uint32_t cv = threadIdx.x < IPT * NW ? s_row[threadIdx.x] : 0u, inc = cv;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
            if (threadIdx.x < IPT * NW) s_row[threadIdx.x] = inc - cv;
            total = __shfl(inc, 63, 64);
            if (threadIdx.x == 0) {
                cnt_out[c] = total < GS_NEAR_STASH ? total : GS_NEAR_STASH;
                if (total > GS_NEAR_STASH) { ctl->near_overflow = 1u; ctl->order_incomplete = 1u; ctl->round1_missed = 1u; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            if (bk[r] >= 0) {
                const uint32_t slot = s_row[r * NW + w] + before[r];
                if (slot < GS_NEAR_STASH) stash[(size_t)c * GS_NEAR_STASH + slot] = make_uint2((uint32_t)bk[r], c * CH + r * NT + threadIdx.x);
            }
        }
        __syncthreads();                                             // s_row is rewritten by the next chunk
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) nvalid += __shfl_xor(nvalid, m, 64);
    if (lane == 0 && nvalid) atomicAdd(&s_nvalid, nvalid);
    __syncthreads();
    if (threadIdx.x == 0 && s_nvalid) atomicAdd(&ctl->n_valid, s_nvalid);
}

// the stashes -> one list in chunk (= index) order; GS_GATHER_CHUNKS chunks per workgroup, which first sums the counts before them
#define GS_GATHER_CHUNKS 8u
__device__ __forceinline__ void k_near_gather_body(const uint2 *__restrict__ stash, const uint32_t *__restrict__ cnt, uint32_t n, uint32_t chunk,
                                                   uint2 *__restrict__ out, GsControl *ctl)
{
    __shared__ uint32_t s_w[4];
    const uint32_t nchunks = (n + chunk - 1) / chunk;
    const uint32_t ngroups = (nchunks + GS_GATHER_CHUNKS - 1) / GS_GATHER_CHUNKS;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const uint32_t c0 = g * GS_GATHER_CHUNKS;
        uint32_t s = 0;
        for (uint32_t i = threadIdx.x; i < c0; i += GS_BLOCK) s += cnt[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        __syncthreads();
        if (lane == 0) s_w[w] = s;
        __syncthreads();
        uint32_t base = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        for (uint32_t c = c0; c < c0 + GS_GATHER_CHUNKS && c < nchunks; c++) {
            const uint32_t k = cnt[c];
            for (uint32_t t = threadIdx.x; t < k; t += GS_BLOCK) out[base + t] = stash[(size_t)c * GS_NEAR_STASH + t];
            base += k;
        }
        if (g == ngroups - 1 && threadIdx.x == 0) ctl->n_sorted = base;    // P': what the two passes sort
    }
}

// The exact rule applied to the candidates the depth pass stashed (k_sort_depth<.., SPEC>): fold the partials, find the threshold
// bin and bucket exactly as k_near_stash does, CHECK that the candidates were a superset of what the rule keeps, and compact each
// group of GS_SPEC_GROUP chunks' survivors -- (bucket, index), in index order -- into the group's buffer.
//   superset: the first depth that was NOT stashed is the far edge of bin `lim` (= hint + pad); its bucket must lie below the
//             threshold bucket (the bucket is monotonic in the stored depth), and the exact bin must not lie beyond lim;
//   no dropped bucket anywhere: a kept splat's bucket leaves [0, 65535] only if rounding its depth to f32 moves it by a whole
//             bucket -- impossible while one f32 ulp of the largest |depth| is narrower than a bucket; V' = V then (k_near_stash
//             COUNTS V' over all depths, which this path never reads);
//   no chunk's stash overflowed (raised by the depth pass).
// A frame that fails is flagged like an overflowed stash (order_incomplete: drawn again from a whole sort); spec_fail says
// why: 1 = the hint was behind (transient: it is exact now), 2 = this scene does not suit the path (the context stops using it).
__device__ __forceinline__ void k_near_filter_body(const uint2 *__restrict__ stash, const uint32_t *__restrict__ cnt, uint32_t n,
                                                   const unsigned long long *__restrict__ part_min, const unsigned long long *__restrict__ part_max,
                                                   const uint32_t *__restrict__ part_cnt, uint32_t nparts, uint2 *__restrict__ group_out,
                                                   uint32_t *__restrict__ gcnt, GsControl *ctl, const uint32_t *__restrict__ dhist, uint32_t near_req,
                                                   uint32_t *__restrict__ bin_hint)
{
    __shared__ unsigned long long s_min, s_max;
    __shared__ uint32_t s_cnt, s_T, s_lmin, s_w[4];
    __shared__ int32_t s_bcut;
    if (threadIdx.x == 0) { s_min = ~0ull; s_max = 0ull; s_cnt = 0; s_bcut = 0; s_T = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        unsigned long long mn = ~0ull, mx = 0ull; uint32_t c = 0;
        for (uint32_t i = threadIdx.x; i < nparts; i += GS_BLOCK) {
            const unsigned long long a = part_min[i], b = part_max[i];
            mn = a < mn ? a : mn; mx = b > mx ? b : mx; c += part_cnt[i];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned long long omn = shfl_xor_u64(mn, m), omx = shfl_xor_u64(mx, m);
            mn = omn < mn ? omn : mn; mx = omx > mx ? omx : mx;
            c += __shfl_xor(c, m, 64);
        }
        if (lane == 0) { atomicMin(&s_min, mn); atomicMax(&s_max, mx); atomicAdd(&s_cnt, c); }
    }
    __syncthreads();
    const double mn = gsm::ordered_to_f64(s_min), mx = gsm::ordered_to_f64(s_max);
    const double inv = 65535.0 / (mx - mn);
    if (threadIdx.x < 64) {                                         // the threshold bin and bucket: exactly k_sort_bucket<.., NEAR>'s rule
        const int l = threadIdx.x;
        auto wave_scan = [&](uint32_t v) { for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(v, d, 64); if (l >= d) v += t; } return v; };
        uint32_t cs = 0;
#pragma unroll
        for (uint32_t c = 0; c < GS_DH_COPIES; c++) cs += dhist[GS_DH_COPIES * GS_DEPTH_BINS + c * GS_DEPTH_COARSE + l];
        const uint32_t cinc = wave_scan(cs);
        const unsigned long long m = __ballot(cinc >= near_req);
        uint32_t T = GS_DEPTH_BINS - 1u;
        if (m) {
            const int C = __ffsll((long long)m) - 1;
            const uint32_t before = __shfl(cinc - cs, C, 64);
            uint32_t fs = 0;
            if (l < 32) {
#pragma unroll
                for (uint32_t c = 0; c < GS_DH_COPIES; c++) fs += dhist[c * GS_DEPTH_BINS + (uint32_t)C * 32u + l];
            }
            const uint32_t finc = wave_scan(fs);
            const unsigned long long m2 = __ballot(l < 32 && before + finc >= near_req);
            T = (uint32_t)C * 32u + (m2 ? (uint32_t)(__ffsll((long long)m2) - 1) : 31u);
        }
        if (l == 0) {
            int32_t bc = 0;
            if (T < GS_DEPTH_BINS - 1u && inv > 0.0 && inv < 1.0e300) {
                const double x = ((double)(-__uint_as_float((T + 1u) << 20)) - mn) * inv;
                bc = !(x >= 0.0) ? 0 : (x >= 65535.0 ? 65535 : (int32_t)x);
            }
            s_bcut = bc; s_T = T;
        }
    }
    __syncthreads();
    const int32_t bcut = s_bcut;
    const uint32_t T = s_T;
    // what cannot be decided chunk by chunk (every workgroup finds the same): the depth range, a stash that overflowed
    uint32_t fail = 0;
    {
        const double big = fmax(fabs(mn), fabs(mx));
        const uint32_t eb = (__float_as_uint((float)big) >> 23) & 0xFFu;
        const double ulp = eb > 23u ? (double)__uint_as_float((eb - 23u) << 23) : 0.0;   // one f32 ulp of the largest |depth|
        if (!(inv > 0.0 && inv < 1.0e300) || !(ulp * inv < 1.0) || ctl->spec_fail == 2u) fail = 2u;
    }
    // the smallest limit a chunk may have been stashed by: the exact bin, and far enough that the first depth NOT stashed (the far
    // edge of that bin) has a bucket below the threshold's (a threshold bucket of 0 keeps every splat: never this path's case)
    if (threadIdx.x == 0) {
        uint32_t L = 0xFFFFu;
        if (!s_cnt) L = 0u;
        else if (bcut > 0 && !fail)
            for (uint32_t c = T; c < T + 8u && c < GS_DEPTH_BINS - 1u; c++)
                if (gsm::sort_bucket(-__uint_as_float((c + 1u) << 20), mn, inv) < bcut) { L = c; break; }
        s_lmin = L;
    }
    __syncthreads();
    const uint32_t lmin = s_lmin;
    if (lmin == 0xFFFFu && !fail) fail = 2u;                         // (buckets coarser than depth bins, or everything kept: the whole-length passes do this scene)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->min_enc = s_min; ctl->max_enc = s_max; ctl->n_kept = s_cnt; ctl->n_total = n; ctl->near_sorted = 2u;   // (2: by this path; read as "not 0" on the device)
        ctl->n_valid = s_cnt;                                        // V' = V: no bucket can be dropped (checked above)
        *bin_hint = T;
        if (fail) { ctl->spec_fail = 2u; ctl->order_incomplete = 1u; ctl->round1_missed = 1u; }
    }
    const uint32_t nchunks = (n + GS_DEPTH_IPT * GS_BLOCK - 1) / (GS_DEPTH_IPT * GS_BLOCK);
    const uint32_t ngroups = (nchunks + GS_SPEC_GROUP - 1) / GS_SPEC_GROUP;
    constexpr uint32_t PER = GS_SPEC_GROUP * GS_SPEC_SLOT / GS_BLOCK;   // 16 consecutive stash slots per thread: index order
    for (uint32_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const uint32_t slot0 = threadIdx.x * PER, c = g * GS_SPEC_GROUP + slot0 / GS_SPEC_SLOT, pos0 = slot0 % GS_SPEC_SLOT;
        const uint32_t cw = (!fail && c < nchunks) ? cnt[c] : 0u, cc = cw & 0xFFFFu;
        if (!fail && c < nchunks && pos0 == 0u && s_cnt && ((cw >> 16) < lmin || (cw >> 16) == 0xFFFFu)) {
            // this chunk was stashed by a limit that does not cover what the exact rule keeps (the hint was behind, or absent): the
            // frame is drawn again from a whole sort; the hint is exact for the next one
            if (ctl->spec_fail != 2u) ctl->spec_fail = 1u;
            ctl->order_incomplete = 1u; ctl->round1_missed = 1u;
            ctl->spec_dbg = (T << 16) | (cw >> 16);
        }
        uint2 rec[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) rec[k] = pos0 + k < cc ? stash[(size_t)c * GS_SPEC_SLOT + pos0 + k] : make_uint2(0u, 0u);
        uint32_t keep = 0, nk = 0;
        int32_t bk[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            bk[k] = -1;
            if (pos0 + k < cc) { const int32_t b = gsm::sort_bucket(__uint_as_float(rec[k].x), mn, inv); if (b >= bcut) { bk[k] = b; keep |= 1u << k; nk++; } }
        }
        // exclusive scan of the threads' counts (threads hold consecutive slots: index order)
        uint32_t inc = nk;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        __syncthreads();
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        uint32_t base = inc - nk, total = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { if (k < w) base += s_w[k]; total += s_w[k]; }
        uint2 *out = group_out + (size_t)g * (GS_SPEC_GROUP * GS_SPEC_SLOT);
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) if (keep & (1u << k)) out[base++] = make_uint2((uint32_t)bk[k], rec[k].y);
        if (threadIdx.x == 0) gcnt[g] = total;
    }
}

// the groups' survivors -> one list in group (= index) order; every workgroup first sums the counts of the groups before its own
__device__ __forceinline__ void k_near_gather_groups_body(const uint2 *__restrict__ group_out, const uint32_t *__restrict__ gcnt, uint32_t n,
                                                          uint2 *__restrict__ out, GsControl *ctl)
{
    __shared__ uint32_t s_w[4];
    const uint32_t nchunks = (n + GS_DEPTH_IPT * GS_BLOCK - 1) / (GS_DEPTH_IPT * GS_BLOCK);
    const uint32_t ngroups = (nchunks + GS_SPEC_GROUP - 1) / GS_SPEC_GROUP;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        uint32_t sv = 0;
        for (uint32_t i = threadIdx.x; i < g; i += GS_BLOCK) sv += gcnt[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sv += __shfl_xor(sv, m, 64);
        __syncthreads();
        if (lane == 0) s_w[w] = sv;
        __syncthreads();
        const uint32_t base = s_w[0] + s_w[1] + s_w[2] + s_w[3], k = gcnt[g];
        const uint2 *src = group_out + (size_t)g * (GS_SPEC_GROUP * GS_SPEC_SLOT);
        for (uint32_t t = threadIdx.x; t < k; t += GS_BLOCK) out[base + t] = src[t];
        if (g == ngroups - 1 && threadIdx.x == 0) ctl->n_sorted = base + k;   // P': what the two passes sort
    }
}
GS_BODY(F_near_filter, k_near_filter_body);
GS_BODY(F_near_gather_groups, k_near_gather_groups_body);
__global__ __launch_bounds__(GS_BLOCK) void k_near_filter(const uint2 *__restrict__ stash, const uint32_t *__restrict__ cnt, uint32_t n,
                                                          const unsigned long long *__restrict__ part_min, const unsigned long long *__restrict__ part_max,
                                                          const uint32_t *__restrict__ part_cnt, uint32_t nparts, uint2 *__restrict__ group_out,
                                                          uint32_t *__restrict__ gcnt, GsControl *ctl, const uint32_t *__restrict__ dhist, uint32_t near_req,
                                                          uint32_t *__restrict__ bin_hint)
{
    k_near_filter_body(stash, cnt, n, part_min, part_max, part_cnt, nparts, group_out, gcnt, ctl, dhist, near_req, bin_hint);
}
__global__ __launch_bounds__(GS_BLOCK) void k_near_gather_groups(const uint2 *__restrict__ group_out, const uint32_t *__restrict__ gcnt, uint32_t n,
                                                                 uint2 *__restrict__ out, GsControl *ctl)
{
    k_near_gather_groups_body(group_out, gcnt, n, out, ctl);
}

template <int NW> GS_BODY(F_near_stash, k_near_stash_body<NW>);
GS_BODY(F_near_gather, k_near_gather_body);

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_near_stash(const float *__restrict__ depth, uint32_t n, const unsigned long long *__restrict__ part_min,
                                                        const unsigned long long *__restrict__ part_max, const uint32_t *__restrict__ part_cnt, uint32_t nparts,
                                                        uint2 *__restrict__ stash, uint32_t *__restrict__ cnt_out, GsControl *ctl,
                                                        const uint32_t *__restrict__ dhist, uint32_t near_req, uint32_t *__restrict__ bin_hint)
{
    k_near_stash_body<NW>(depth, n, part_min, part_max, part_cnt, nparts, stash, cnt_out, ctl, dhist, near_req, bin_hint);
}
__global__ __launch_bounds__(GS_BLOCK) void k_near_gather(const uint2 *__restrict__ stash, const uint32_t *__restrict__ cnt, uint32_t n, uint32_t chunk,
                                                          uint2 *__restrict__ out, GsControl *ctl)
{
    k_near_gather_body(stash, cnt, n, chunk, out, ctl);
}

template <int NW, bool COMPACT, bool NEAR, bool MSD = false> GS_BODY(F_sort_bucket, k_sort_bucket_body<NW, COMPACT, NEAR, MSD>);
template <bool STRIP, bool SPEC> GS_BODY(F_sort_depth, k_sort_depth_body<STRIP, SPEC>);

}  // namespace

static void fill_sort_uniforms(const gs_ctx *ctx, const float view[4], const float *cutout16, const GsSortStrip *strip, SortUniforms &u, StripUniforms &su);

// what the lane's order was made from and how much of it exists (the render re-sorts in full if it turns out to need more)
void gs_remember_sort(gs_ctx *L, const float view[4], const float *cutout16, const GsSortStrip *strip, uint32_t near_req)
{
    memcpy(L->sv_view, view, sizeof L->sv_view);
    L->sv_has_cutout = cutout16 != nullptr;
    if (cutout16) memcpy(L->sv_cutout, cutout16, sizeof L->sv_cutout);
    L->sv_has_strip = strip != nullptr;
    if (strip) L->sv_strip = *strip;
    L->sort_near_req = near_req;
}

// the depth-histogram buffers of this sort: fill the next one (near-only sorts), clear the one the previous near-only sort filled
static DepthHist next_depth_hist(gs_ctx *L, bool near)
{
    DepthHist dh;
    dh.zero = L->dh_dirty; L->dh_dirty = nullptr;
    dh.fill = nullptr; dh.zero_word = nullptr; dh.zero_grp = nullptr; dh.zero_grp_words = 0;
    if (near) {
        dh.fill = L->dhist[L->dh_next]; L->dh_next ^= 1;
        L->dh_dirty = dh.fill;
        dh.zero_word = &L->ctl->n_valid;
    }
    return dh;
}

// The MSD sort (round 5; kernels and rationale in gs_prims.hip): sorts with compact records of at most GS_MSD_MAX_N splats take four
// launches -- depth, bucket (+ rows of the high bucket byte per chunk and per group of chunks), k_msd_scatter, k_seg_sort -- instead
// of seven.  A near-only sort on this path is a TAIL sort: the same four launches, k_msd_scatter drops the segments before the one
// that holds position V' - near_req (it has the segment totals in its hands anyway) and k_seg_sort sorts the rest -- no depth
// histogram, no threshold.  GS_SORT_MSD=0 in the environment keeps the two LSD passes (A/B runs; longer sorts use them anyway).
bool gs_msd_enabled() { static const bool on = []() { const char *e = getenv("GS_SORT_MSD"); return !(e && e[0] == '0'); }(); return on; }
static bool gs_msd_ok(const gs_ctx *L, uint32_t n, bool compact)
{
    return gs_msd_enabled() && compact && n <= GS_MSD_MAX_N && L->msd_grp &&
           (size_t)256 * (gs_div_up(gs_div_up(n, gs_radix_chunk(n)), GS_MSD_GROUP) + 1u) <= L->msd_grp_cap;
}
static void gs_msd_arm(const gs_ctx *L, uint32_t n, DepthHist &dh)   // the depth pass clears the group rows the bucket pass adds to
{
    dh.zero_grp = L->msd_grp;
    dh.zero_grp_words = 256u * gs_div_up(gs_div_up(n, gs_radix_chunk(n)), GS_MSD_GROUP);
}

// may this near-only sort hand its survivors on through the chunk stashes?  Long inputs only (the 4096-item geometry), a share of
// at most 1/32 of the splats (a stash holds 1/8 of its chunk), and not after a stash has overflowed on this context
static bool gs_near_stash_ok(const gs_ctx *L, uint32_t n, uint32_t near_req)
{
    return near_req && gs_radix_chunk(n) == GS_CHUNK_L && (uint64_t)near_req * 32u <= n && !__atomic_load_n(&gs_root(const_cast<gs_ctx *>(L))->near_stash_off, __ATOMIC_RELAXED) &&
           (size_t)(gs_div_up(n, GS_CHUNK_L) + 1u) * GS_NEAR_STASH <= L->scratch_cap / 2;
}
// ... and may its depth pass stash the candidates itself (k_sort_depth<.., SPEC>)?  Only once a near-only sort of this context has
// been collected (the hint exists), and not while the path is held back after a failure (gs_spec_back_off)
static bool gs_near_spec_ok(const gs_ctx *L, uint32_t n)
{
    const gs_ctx *P = gs_root(const_cast<gs_ctx *>(L));
    return __atomic_load_n(&P->near_spec, __ATOMIC_RELAXED) && !__atomic_load_n(&P->near_spec_hold, __ATOMIC_RELAXED) && P->near_spec_opt &&
           (size_t)(gs_div_up(n, (uint32_t)(GS_DEPTH_IPT * GS_BLOCK)) + GS_SPEC_GROUP) * GS_SPEC_SLOT <= L->scratch_cap / 4 &&
           (size_t)gs_div_up(n, (uint32_t)(GS_DEPTH_IPT * GS_BLOCK)) * 2u + 64u <= L->hist_cap;
}
static uint32_t gs_spec_groups(uint32_t n) { return gs_div_up(gs_div_up(n, (uint32_t)(GS_DEPTH_IPT * GS_BLOCK)), GS_SPEC_GROUP); }
static uint32_t gs_near_gather_grid(uint32_t n)
{
    const uint32_t g = gs_div_up(gs_div_up(n, GS_CHUNK_L), 8u);
    return g < 1 ? 1u : (g > 1024u ? 1024u : g);
}

// records the second pass of a near-only sort should expect (its geometry and grid: a matter of speed only)
static uint32_t near_hint(const gs_ctx *L, uint32_t n)
{
    const uint64_t h = (uint64_t)L->sort_near_req * 2u + 65536u;
    return h < n ? (uint32_t)h : n;
}

// Two frames' sorts, one launch per kernel (GS_OPT_FRAME_BATCH): S[0] and S[1] are sibling lanes on ONE stream holding the same
// resident data; each keeps its own depths, keys, tables, partial slots and control block.  Strip sorts (gs_sort_for) pair with strip sorts.
int gs_run_sort2(gs_ctx *const S[2], const float *const view[2], const float *const cutout16[2], const GsSortStrip *const strip[2], const uint32_t near_req[2])
{
    gs_ctx *ctx = S[0];
    S[0]->sort_gen++; S[1]->sort_gen++;
    const uint32_t n = (uint32_t)ctx->n;
    SortUniforms u[2];
    StripUniforms su[2];
    for (int k = 0; k < 2; k++) fill_sort_uniforms(S[k], view[k], cutout16[k], strip ? strip[k] : nullptr, u[k], su[k]);
    const bool compact = !ctx->wide_pairs && n <= (1u << 25);
    const bool near_both = compact && near_req && near_req[0] && near_req[1];   // (a pair takes one path)
    const bool msd = gs_msd_ok(S[0], n, compact) && gs_msd_ok(S[1], n, compact) &&
                     !(near_both && gs_near_stash_ok(S[0], n, near_req[0]) && gs_near_stash_ok(S[1], n, near_req[1]));   // (long near-only sorts keep their stashes)
    // a near-only sort on the MSD path is a TAIL sort (k_msd_scatter cuts the order at a segment boundary: no histogram, no threshold
    // search -- each frame of the pair for itself); `near` below is the histogram form of the longer inputs
    const uint32_t tail[2] = { msd && compact && near_req ? near_req[0] : 0u, msd && compact && near_req ? near_req[1] : 0u };
    const bool near = near_both && !msd;
    DepthHist dh[2];
    for (int k = 0; k < 2; k++) { gs_remember_sort(S[k], view[k], cutout16[k], strip ? strip[k] : nullptr, near ? near_req[k] : tail[k]); dh[k] = next_depth_hist(S[k], near); }
    if (msd) for (int k = 0; k < 2; k++) gs_msd_arm(S[k], n, dh[k]);
    const bool strips = u[0].has_strip && u[1].has_strip;
    if (!strips) u[0].has_strip = u[1].has_strip = 0;              // (a pair takes one path: both strip sorts, or both plain)
    const uint32_t g = gs_radix_grid(n);
    hipStream_t st = ctx->stream;
    GS_PROF_RECORD(ctx, 0);
    uint32_t gd = gs_div_up(n, (uint32_t)(GS_DEPTH_IPT * GS_BLOCK));
    if (gd < 1) gd = 1;
    if (gd > GS_DEPTH_GRID) gd = GS_DEPTH_GRID;
    // ONE sweep computes both frames' depths: the rows are read once (the lanes of a context alias the owner's resident arrays)
    if (S[0]->sort_rows != S[1]->sort_rows) { snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "paired sort: the two lanes hold different splat arrays"); return GS_E_STATE; }
    // near-only sorts of long inputs hand their survivors on through per-chunk stashes instead of two whole-length passes (above);
    // with a threshold hint the depth pass stashes the candidates itself and writes no depths (SPEC)
    const bool stash = near && gs_near_stash_ok(S[0], n, near_req[0]) && gs_near_stash_ok(S[1], n, near_req[1]);
    const bool spec = stash && gs_near_spec_ok(S[0], n) && gs_near_spec_ok(S[1], n);
    uint32_t *const bin_hint = &gs_root(ctx)->ctl->near_bin_hint;
    const uint32_t nch = gs_div_up(n, (uint32_t)(GS_DEPTH_IPT * GS_BLOCK));
    SpecArgs sa[2];
    for (int k = 0; k < 2; k++) { sa[k].stash = S[k]->kv_b; sa[k].cnt = S[k]->hist; sa[k].bin_hint = bin_hint; sa[k].ctl = S[k]->ctl; }
#define GS_DEPTHP(ST, SP) hipLaunchKernelGGL((k_sort_depth_pair<ST, SP>), dim3(gd), dim3(GS_BLOCK), near ? 2u * GS_DEPTH_BINS * sizeof(uint32_t) : 0u, st, (const float4 *)S[0]->sort_rows, (const float *)S[0]->bound_r, n,  \
                                         u[0], u[1], su[0], su[1], S[0]->depth, S[1]->depth, S[0]->part_min, S[0]->part_max, S[0]->part_cnt,                            \
                                         S[1]->part_min, S[1]->part_max, S[1]->part_cnt, dh[0], dh[1], sa[0], sa[1])
    // ... where one sweep saves memory traffic: inputs the caches do not hold.  Below GS_DEPTH_PAIR_MIN_N splats the rows stay resident
    // between the two frames' reads, and the one-sweep kernel pays for holding two frames' uniforms (two view rows, two cut-out matrices,
    // two strips' 31 words each: more than the scalar registers hold -- the compiler parks them in lanes of a vector register, and 39 % of
    // that kernel's vector instructions at 1 M splats were v_readlane): there the pair is k_sort_depth's body twice, blockIdx.y = the frame
    if (!near && !spec && n < GS_DEPTH_PAIR_MIN_N) {
#define GS_DEPTHT(ST) gs_twin<F_sort_depth<ST, false>, GS_BLOCK>(gd, st,                                                                                          \
        gs_pack_make((const float4 *)S[0]->sort_rows, (const float *)S[0]->bound_r, n, u[0], su[0], S[0]->depth, S[0]->part_min, S[0]->part_max, S[0]->part_cnt, \
                     dh[0], (uint2 *)nullptr, (uint32_t *)nullptr, (const uint32_t *)bin_hint, S[0]->ctl),                                                        \
        gs_pack_make((const float4 *)S[1]->sort_rows, (const float *)S[1]->bound_r, n, u[1], su[1], S[1]->depth, S[1]->part_min, S[1]->part_max, S[1]->part_cnt, \
                     dh[1], (uint2 *)nullptr, (uint32_t *)nullptr, (const uint32_t *)bin_hint, S[1]->ctl))
        if (strips) GS_DEPTHT(true); else GS_DEPTHT(false);
#undef GS_DEPTHT
    }
    else if (spec) { if (strips) GS_DEPTHP(true, true); else GS_DEPTHP(false, true); }
    else { if (strips) GS_DEPTHP(true, false); else GS_DEPTHP(false, false); }
#undef GS_DEPTHP
    if (stash) {
        uint2 *list[2] = { S[0]->kv_b + S[0]->scratch_cap / 2, S[1]->kv_b + S[1]->scratch_cap / 2 };
        if (spec) {
            uint2 *grp[2] = { S[0]->kv_b + S[0]->scratch_cap / 4, S[1]->kv_b + S[1]->scratch_cap / 4 };
            const uint32_t ng = gs_spec_groups(n);
            gs_twin<F_near_filter, GS_BLOCK>(ng, st,
                gs_pack_make((const uint2 *)S[0]->kv_b, (const uint32_t *)S[0]->hist, n, (const unsigned long long *)S[0]->part_min, (const unsigned long long *)S[0]->part_max,
                             (const uint32_t *)S[0]->part_cnt, gd, grp[0], S[0]->hist + nch, S[0]->ctl, (const uint32_t *)dh[0].fill, S[0]->sort_near_req, bin_hint),
                gs_pack_make((const uint2 *)S[1]->kv_b, (const uint32_t *)S[1]->hist, n, (const unsigned long long *)S[1]->part_min, (const unsigned long long *)S[1]->part_max,
                             (const uint32_t *)S[1]->part_cnt, gd, grp[1], S[1]->hist + nch, S[1]->ctl, (const uint32_t *)dh[1].fill, S[1]->sort_near_req, bin_hint));
            gs_twin<F_near_gather_groups, GS_BLOCK>(ng, st,
                gs_pack_make((const uint2 *)grp[0], (const uint32_t *)(S[0]->hist + nch), n, list[0], S[0]->ctl),
                gs_pack_make((const uint2 *)grp[1], (const uint32_t *)(S[1]->hist + nch), n, list[1], S[1]->ctl));
        } else {
        gs_twin<F_near_stash<8>, 512>(g, st,
            gs_pack_make((const float *)S[0]->depth, n, (const unsigned long long *)S[0]->part_min, (const unsigned long long *)S[0]->part_max, (const uint32_t *)S[0]->part_cnt, gd,
                         S[0]->kv_b, S[0]->hist, S[0]->ctl, (const uint32_t *)dh[0].fill, S[0]->sort_near_req, bin_hint),
            gs_pack_make((const float *)S[1]->depth, n, (const unsigned long long *)S[1]->part_min, (const unsigned long long *)S[1]->part_max, (const uint32_t *)S[1]->part_cnt, gd,
                         S[1]->kv_b, S[1]->hist, S[1]->ctl, (const uint32_t *)dh[1].fill, S[1]->sort_near_req, bin_hint));
        gs_twin<F_near_gather, GS_BLOCK>(gs_near_gather_grid(n), st,
            gs_pack_make((const uint2 *)S[0]->kv_b, (const uint32_t *)S[0]->hist, n, (uint32_t)GS_CHUNK_L, list[0], S[0]->ctl),
            gs_pack_make((const uint2 *)S[1]->kv_b, (const uint32_t *)S[1]->hist, n, (uint32_t)GS_CHUNK_L, list[1], S[1]->ctl));
        }
        GS_HIP(hipGetLastError());
        const void *in2[2]; void *out2[2]; const uint32_t *np2[2]; uint32_t *cnt2[2] = { nullptr, nullptr }; const uint32_t *fill2[2] = { nullptr, nullptr };
        for (int k = 0; k < 2; k++) { in2[k] = list[k]; out2[k] = S[k]->key_a; np2[k] = &S[k]->ctl->n_sorted; }
        int rc2 = gs_launch_radix_pass2(S, in2, GS_RADIX_PACKED, out2, GS_RADIX_KEYIDX, np2, n, near_hint(S[0], n), 0, 9, false, 0xFFFFFFFFu, 25, cnt2, fill2);
        if (rc2 != GS_OK) return rc2;
        for (int k = 0; k < 2; k++) { in2[k] = S[k]->key_a; out2[k] = S[k]->val_a; }
        rc2 = gs_launch_radix_pass2(S, in2, GS_RADIX_KEYIDX, out2, GS_RADIX_KEYS, np2, n, near_hint(S[0], n), 25, 7, false, 0xFFFFFFFFu, 0, cnt2, fill2);
        if (rc2 != GS_OK) return rc2;
        GS_PROF_RECORD(ctx, 1);
        for (int k = 0; k < 2; k++) { S[k]->sorted = S[k]->val_a; S[k]->have_sort = true; }
        return GS_OK;
    }
    if (msd) {
#define GS_BUCKETM(NW) gs_twin<F_sort_bucket<NW, true, false, true>, 64 * NW>(g, st,                                                                    \
        gs_pack_make((const float *)S[0]->depth, n, S[0]->key_a, (const unsigned long long *)S[0]->part_min, (const unsigned long long *)S[0]->part_max,     \
                     (const uint32_t *)S[0]->part_cnt, gd, S[0]->hist, S[0]->ctl, (const uint32_t *)dh[0].fill, S[0]->sort_near_req, bin_hint, S[0]->msd_grp), \
        gs_pack_make((const float *)S[1]->depth, n, S[1]->key_a, (const unsigned long long *)S[1]->part_min, (const unsigned long long *)S[1]->part_max,     \
                     (const uint32_t *)S[1]->part_cnt, gd, S[1]->hist, S[1]->ctl, (const uint32_t *)dh[1].fill, S[1]->sort_near_req, bin_hint, S[1]->msd_grp))
        if (gs_radix_chunk(n) == GS_CHUNK_L) GS_BUCKETM(8); else GS_BUCKETM(4);
#undef GS_BUCKETM
        GS_HIP(hipGetLastError());
        const int rcm = gs_launch_msd_sort2(S, n, tail);
        if (rcm != GS_OK) return rcm;
        GS_PROF_RECORD(ctx, 1);
        for (int k = 0; k < 2; k++) { S[k]->sorted = S[k]->val_a; S[k]->have_sort = true; }
        return GS_OK;
    }
#define GS_BUCKET2(NW, C, NR) gs_twin<F_sort_bucket<NW, C, NR>, 64 * NW>(g, st,                                                                            \
        gs_pack_make((const float *)S[0]->depth, n, S[0]->key_a, (const unsigned long long *)S[0]->part_min, (const unsigned long long *)S[0]->part_max,     \
                     (const uint32_t *)S[0]->part_cnt, gd, S[0]->hist, S[0]->ctl, (const uint32_t *)dh[0].fill, S[0]->sort_near_req, bin_hint, (uint32_t *)nullptr), \
        gs_pack_make((const float *)S[1]->depth, n, S[1]->key_a, (const unsigned long long *)S[1]->part_min, (const unsigned long long *)S[1]->part_max,     \
                     (const uint32_t *)S[1]->part_cnt, gd, S[1]->hist, S[1]->ctl, (const uint32_t *)dh[1].fill, S[1]->sort_near_req, bin_hint, (uint32_t *)nullptr))
    if (gs_radix_chunk(n) == GS_CHUNK_L) { if (near) GS_BUCKET2(8, true, true); else if (compact) GS_BUCKET2(8, true, false); else GS_BUCKET2(8, false, false); }
    else { if (near) GS_BUCKET2(4, true, true); else if (compact) GS_BUCKET2(4, true, false); else GS_BUCKET2(4, false, false); }
#undef GS_BUCKET2
    GS_HIP(hipGetLastError());
    const void *in[2]; void *out[2]; const uint32_t *np[2]; uint32_t *cnt[2]; const uint32_t *fill[2];
    int rc;
    if (compact) {
        for (int k = 0; k < 2; k++) { in[k] = S[k]->key_a; out[k] = S[k]->kv_b; np[k] = &S[k]->ctl->n_total; cnt[k] = &S[k]->ctl->n_sorted; fill[k] = nullptr; }
        rc = gs_launch_radix_pass2(S, in, GS_RADIX_KEYS, out, GS_RADIX_KEYIDX, np, n, n, 0, 9, true, 0xFFFFFFFFu, 25, cnt, fill);
        if (rc != GS_OK) return rc;
        // (a near-only sort leaves no zero tail: k_project supplies the zeros of the positions behind its records)
        for (int k = 0; k < 2; k++) { in[k] = S[k]->kv_b; out[k] = S[k]->val_a; np[k] = &S[k]->ctl->n_sorted; cnt[k] = nullptr; fill[k] = near ? nullptr : &S[k]->ctl->n_kept; }
        rc = gs_launch_radix_pass2(S, in, GS_RADIX_KEYIDX, out, GS_RADIX_KEYS, np, n, near ? near_hint(S[0], n) : n, 25, 7, false, 0xFFFFFFFFu, 0, cnt, fill);
        if (rc != GS_OK) return rc;
    } else {
        for (int k = 0; k < 2; k++) { in[k] = S[k]->key_a; out[k] = S[k]->kv_b; np[k] = &S[k]->ctl->n_total; cnt[k] = nullptr; fill[k] = nullptr; }
        rc = gs_launch_radix_pass2(S, in, GS_RADIX_KEYS, out, GS_RADIX_PACKED, np, n, n, 0, 8, true, 0xFFFFFFFFu, 0, cnt, fill);
        if (rc != GS_OK) return rc;
        for (int k = 0; k < 2; k++) { in[k] = S[k]->kv_b; out[k] = S[k]->val_a; np[k] = &S[k]->ctl->n_kept; }
        rc = gs_launch_radix_pass2(S, in, GS_RADIX_PACKED, out, GS_RADIX_KEYS, np, n, n, 8, 9, false, GS_CULLED_KEY, 0, cnt, fill);
        if (rc != GS_OK) return rc;
    }
    GS_PROF_RECORD(ctx, 1);
    for (int k = 0; k < 2; k++) { S[k]->sorted = S[k]->val_a; S[k]->have_sort = true; }
    return GS_OK;
}

// the uniforms of one sort: view row and cutout matrix widened to f64; for a strip sort (gs_sort_for) the rows of the frame's
// matrices the reach test needs and an upper bound of ||mat3(modelView)||_2
static void fill_sort_uniforms(const gs_ctx *ctx, const float view[4], const float *cutout16, const GsSortStrip *strip, SortUniforms &u, StripUniforms &su)
{
    for (int i = 0; i < 4; i++) u.view[i] = (double)view[i];
    u.has_cutout = cutout16 != nullptr;
    for (int i = 0; i < 16; i++) u.cutout[i] = cutout16 ? (double)cutout16[i] : 0.0;
    // an entity's inverse world matrix is affine: w = 1 / (0 x + 0 y + 0 z + 1) is exactly 1 for finite positions, the division and
    // the three multiplications by it change nothing (in_cutout_affine takes the general path for a position that is not finite)
    if (cutout16 && u.cutout[3] == 0.0 && u.cutout[7] == 0.0 && u.cutout[11] == 0.0 && u.cutout[15] == 1.0) u.has_cutout = 2;
    u.has_strip = 0;
    memset(&su, 0, sizeof su);
    if (strip && ctx->renderable) {
        const float *m = strip->mv, *p = strip->proj;
        for (int k = 0; k < 4; k++) { su.mvr0[k] = m[4 * k]; su.mvr1[k] = m[4 * k + 1]; su.mvr2[k] = m[4 * k + 2]; su.pr0[k] = p[4 * k]; su.pr1[k] = p[4 * k + 1]; su.pr3[k] = p[4 * k + 3]; }
        // spectral norm of A = mat3(gsModelViewMatrix): power iteration on A^T A (symmetric 3x3), bracketed from above by the
        // Frobenius norm; a rigid pose with uniform scale s gives s
        double ata[3][3], fro = 0.0;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            ata[i][j] = 0.0;
            for (int r = 0; r < 3; r++) ata[i][j] += (double)m[4 * i + r] * m[4 * j + r];
        }
        for (int i = 0; i < 3; i++) fro += ata[i][i];
        double v[3] = { 1.0, 0.7, 0.4 }, lam = fro;
        for (int it = 0; it < 64; it++) {
            double w[3] = { 0, 0, 0 }, nn = 0.0;
            for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) w[i] += ata[i][j] * v[j]; nn += w[i] * w[i]; }
            nn = sqrt(nn);
            if (!(nn > 0.0)) break;
            for (int i = 0; i < 3; i++) v[i] = w[i] / nn;
            lam = nn;
        }
        // (the iterate's Rayleigh quotient approaches the largest eigenvalue from below: 2 % on top, never above Frobenius)
        double na = sqrt(lam) * 1.02;
        if (!(na <= sqrt(fro) * 1.0001)) na = sqrt(fro) * 1.0001;
        su.norm_a = (float)na * 1.0001f;
        su.focal = strip->focal; su.half_w = 0.5f * strip->vw; su.sx0 = (float)strip->x0; su.sx1 = (float)strip->x1;
        su.half_h = 0.5f * strip->vh; su.sy1 = strip->vh > 0.0f ? strip->vh : 0.0f;
        u.has_strip = (su.norm_a == su.norm_a && su.focal > 0.0f && strip->x1 > strip->x0) ? 1 : 0;
    }

}

int gs_run_sort(gs_ctx *ctx, const float view[4], const float *cutout16, const GsSortStrip *strip, uint32_t near_req)
{
    ctx->sort_gen++;
    const uint32_t n = (uint32_t)ctx->n;
    SortUniforms u;
    StripUniforms su;
    fill_sort_uniforms(ctx, view, cutout16, strip, u, su);
    // record format of the two passes: 4 bytes while the index fits in 25 bits (GS_OPT_WIDE_PAIRS forces the general form)
    const bool compact = !ctx->wide_pairs && n <= (1u << 25);
    const bool msd = gs_msd_ok(ctx, n, compact) && !(compact && near_req && (gs_near_stash_ok(ctx, n, near_req) || ctx->no_tail_sort));   // (long near-only sorts keep their stashes)
    const uint32_t tail = msd && compact ? near_req : 0u;          // (a near-only sort on the MSD path: k_msd_scatter cuts the order at a segment boundary)
    const bool near = compact && near_req && !msd;               // (... on the longer inputs: depth histogram + threshold)
    gs_remember_sort(ctx, view, cutout16, strip, near ? near_req : tail);
    DepthHist dh = next_depth_hist(ctx, near);
    if (msd) gs_msd_arm(ctx, n, dh);

    const uint32_t g = gs_radix_grid(n);                         // same chunking as the radix kernels (pre-filled histogram rows)
    GS_PROF_RECORD(ctx, 0);
    uint32_t gd = gs_div_up(n, (uint32_t)(GS_DEPTH_IPT * GS_BLOCK));
    if (gd < 1) gd = 1;
    if (gd > GS_DEPTH_GRID) gd = GS_DEPTH_GRID;
    const size_t dlds = near ? GS_DEPTH_BINS * sizeof(uint32_t) : 0u;
    const bool stash = near && gs_near_stash_ok(ctx, n, near_req);
    const bool spec = stash && gs_near_spec_ok(ctx, n);
    uint32_t *const bin_hint = &gs_root(ctx)->ctl->near_bin_hint;
#define GS_DEPTH1(ST, SP) hipLaunchKernelGGL((k_sort_depth<ST, SP>), dim3(gd), dim3(GS_BLOCK), dlds, ctx->stream, ctx->sort_rows, ctx->bound_r, n, u, su, ctx->depth, \
                                             ctx->part_min, ctx->part_max, ctx->part_cnt, dh, ctx->kv_b, ctx->hist, (const uint32_t *)bin_hint, ctx->ctl)
    if (spec) { if (u.has_strip) GS_DEPTH1(true, true); else GS_DEPTH1(false, true); }
    else { if (u.has_strip) GS_DEPTH1(true, false); else GS_DEPTH1(false, false); }
#undef GS_DEPTH1
    if (stash) {
        // (survivors through per-chunk stashes: k_near_stash / k_near_gather above; candidates stashed by the depth pass: k_near_filter)
        uint2 *list = ctx->kv_b + ctx->scratch_cap / 2;
        if (spec) {
            uint2 *grp = ctx->kv_b + ctx->scratch_cap / 4;
            const uint32_t nch = gs_div_up(n, (uint32_t)(GS_DEPTH_IPT * GS_BLOCK)), ng = gs_spec_groups(n);
            hipLaunchKernelGGL(k_near_filter, dim3(ng), dim3(GS_BLOCK), 0, ctx->stream, (const uint2 *)ctx->kv_b, (const uint32_t *)ctx->hist, n, (const unsigned long long *)ctx->part_min,
                               (const unsigned long long *)ctx->part_max, (const uint32_t *)ctx->part_cnt, gd, grp, ctx->hist + nch, ctx->ctl, (const uint32_t *)dh.fill, ctx->sort_near_req, bin_hint);
            hipLaunchKernelGGL(k_near_gather_groups, dim3(ng), dim3(GS_BLOCK), 0, ctx->stream, (const uint2 *)grp, (const uint32_t *)(ctx->hist + nch), n, list, ctx->ctl);
        } else {
        hipLaunchKernelGGL((k_near_stash<8>), dim3(g), dim3(512), 0, ctx->stream, (const float *)ctx->depth, n, (const unsigned long long *)ctx->part_min,
                           (const unsigned long long *)ctx->part_max, (const uint32_t *)ctx->part_cnt, gd, ctx->kv_b, ctx->hist, ctx->ctl, (const uint32_t *)dh.fill, ctx->sort_near_req, bin_hint);
        hipLaunchKernelGGL(k_near_gather, dim3(gs_near_gather_grid(n)), dim3(GS_BLOCK), 0, ctx->stream, (const uint2 *)ctx->kv_b, (const uint32_t *)ctx->hist, n,
                           (uint32_t)GS_CHUNK_L, list, ctx->ctl);
        }
        GS_HIP(hipGetLastError());
        int rcs = gs_launch_radix_pass(ctx, list, GS_RADIX_PACKED, ctx->key_a, GS_RADIX_KEYIDX, &ctx->ctl->n_sorted, n, near_hint(ctx, n), 0, 9, false, 0xFFFFFFFFu, 25);
        if (rcs != GS_OK) return rcs;
        rcs = gs_launch_radix_pass(ctx, ctx->key_a, GS_RADIX_KEYIDX, ctx->val_a, GS_RADIX_KEYS, &ctx->ctl->n_sorted, n, near_hint(ctx, n), 25, 7, false, 0xFFFFFFFFu, 0, nullptr, nullptr);
        if (rcs != GS_OK) return rcs;
        GS_PROF_RECORD(ctx, 1);
        ctx->sorted = ctx->val_a;
        ctx->have_sort = true;
        return GS_OK;
    }
    if (msd) {
        // four launches: depth (above), bucket + rows of the high bucket byte, one stable scatter by that byte, one LDS sort per segment
#define GS_LAUNCH_BUCKETM(NW) hipLaunchKernelGGL((k_sort_bucket<NW, true, false, true>), dim3(g), dim3(64 * NW), 0, ctx->stream, (const float *)ctx->depth, n, ctx->key_a,     \
                                                     (const unsigned long long *)ctx->part_min, (const unsigned long long *)ctx->part_max, (const uint32_t *)ctx->part_cnt, gd, \
                                                     ctx->hist, ctx->ctl, (const uint32_t *)dh.fill, ctx->sort_near_req, bin_hint, ctx->msd_grp)
        if (gs_radix_chunk(n) == GS_CHUNK_L) GS_LAUNCH_BUCKETM(8); else GS_LAUNCH_BUCKETM(4);
#undef GS_LAUNCH_BUCKETM
        GS_HIP(hipGetLastError());
        const int rcm = gs_launch_msd_sort(ctx, n, tail);
        if (rcm != GS_OK) return rcm;
        GS_PROF_RECORD(ctx, 1);
        ctx->sorted = ctx->val_a;
        ctx->have_sort = true;
        return GS_OK;
    }
#define GS_LAUNCH_BUCKET(NW, C, NR) hipLaunchKernelGGL((k_sort_bucket<NW, C, NR>), dim3(g), dim3(64 * NW), 0, ctx->stream, ctx->depth, n, ctx->key_a, \
                                                       ctx->part_min, ctx->part_max, ctx->part_cnt, gd, ctx->hist, ctx->ctl, (const uint32_t *)dh.fill, ctx->sort_near_req, bin_hint, (uint32_t *)nullptr)
    if (gs_radix_chunk(n) == GS_CHUNK_L) { if (near) GS_LAUNCH_BUCKET(8, true, true); else if (compact) GS_LAUNCH_BUCKET(8, true, false); else GS_LAUNCH_BUCKET(8, false, false); }
    else { if (near) GS_LAUNCH_BUCKET(4, true, true); else if (compact) GS_LAUNCH_BUCKET(4, true, false); else GS_LAUNCH_BUCKET(4, false, false); }
#undef GS_LAUNCH_BUCKET
    GS_HIP(hipGetLastError());
    int rc;
    if (compact) {
        rc = gs_launch_radix_pass(ctx, ctx->key_a, GS_RADIX_KEYS, ctx->kv_b, GS_RADIX_KEYIDX, &ctx->ctl->n_total, n, n, 0, 9, /*have_hist=*/true,
                                  0xFFFFFFFFu, 25, &ctx->ctl->n_sorted);
        if (rc != GS_OK) return rc;
        // V' records are left (culled splats and dropped buckets took no slot); slots [V', V) of the result are zero-filled:
        // the reference's never-written Uint32Array tail
        // (a near-only sort leaves no zero tail: k_project supplies the zeros of the positions behind its records)
        rc = gs_launch_radix_pass(ctx, ctx->kv_b, GS_RADIX_KEYIDX, ctx->val_a, GS_RADIX_KEYS, &ctx->ctl->n_sorted, n, near ? near_hint(ctx, n) : n, 25, 7, false,
                                  0xFFFFFFFFu, 0, nullptr, near ? nullptr : &ctx->ctl->n_kept);
        if (rc != GS_OK) return rc;
    } else {
        rc = gs_launch_radix_pass(ctx, ctx->key_a, GS_RADIX_KEYS, ctx->kv_b, GS_RADIX_PACKED, &ctx->ctl->n_total, n, n, 0, 8, /*have_hist=*/true);
        if (rc != GS_OK) return rc;
        // pass A dropped the culled splats: V records are left.  Splats with a dropped bucket (key 65536) sort behind every
        // bucket and store 0: the tail [V',V) of the result is 0 like the reference's never-written Uint32Array slots
        rc = gs_launch_radix_pass(ctx, ctx->kv_b, GS_RADIX_PACKED, ctx->val_a, GS_RADIX_KEYS, &ctx->ctl->n_kept, n, n, 8, 9, false, GS_CULLED_KEY);
        if (rc != GS_OK) return rc;
    }
    GS_PROF_RECORD(ctx, 1);
    ctx->sorted = ctx->val_a;
    ctx->have_sort = true;
    return GS_OK;
}

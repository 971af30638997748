// gs_prims.hip -- stable LSD radix pass for gfx950 (wave64): digit-histogram rows, and a scatter that derives its own
// offsets from those rows (no scan launch in between).
//
// A pass over n items on a digit of `bits` bits, in chunks of GS_CHUNK items:
//   k_radix_hist     one workgroup per GROUP of GS_RADIX_SUB consecutive chunks: a row of 2^bits counts per chunk
//                    (H[chunk][digit]) and one per group (G[group][digit]).  Skipped when the kernel that PRODUCED the keys
//                    filled the rows itself (depth pass A: k_sort_bucket).
//   k_radix_scatter  one workgroup per chunk c of group g: its offset inside every digit's run = the G rows of the groups
//                    before g + the H rows of the chunks of g before c; the digit totals (-> run starts) = all G rows.  Summed
//                    straight from the tables -- some tens of KB of L2-resident 16-byte loads per workgroup, issued under the
//                    latency of the chunk's key loads: cheaper than the 5.5 us scan kernel plus the launch boundary it
//                    replaces (frames of 1 M splats are launch-bound), and with one row per 16 K items the work stays small
//                    up to 8 M items.  (One workgroup per GROUP, chunk by chunk with the offsets kept in LDS, needs no H rows
//                    and a quarter of the row sums, but a chunk takes ~4.5 us of dependent LDS steps and barriers: 18 us per
//                    scatter instead of 10 at 1 M splats.)
//   k_radix_gscan    only beyond that (more than GS_RADIX_BRUTE_ROWS G rows, where summing every row per workgroup would
//                    dominate): exclusive sums per super-group of GS_RADIX_SUPER G rows + the digit totals; the scatter
//                    then adds the G rows of its own super-group only.
// The kernels read their problem size from device memory (GsControl), so no stage of the frame needs a host round trip.
// 512-thread workgroups (8 wavefronts), 4096 items per chunk, all of a chunk's items loaded before any is processed,
// LDS match words / wavefront ballots for the stable in-wave rank, chunk reordered in LDS before the stores (no MFMA: there
// is no contraction here).
#include "gs_internal.h"

namespace {

constexpr int NW = GS_RADIX_WAVES;
constexpr int NT = GS_RADIX_THREADS;
constexpr int IPT = GS_CHUNK / GS_RADIX_THREADS;                 // items per thread and chunk = ranking rounds per wave

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread across the NT-thread workgroup; *total = workgroup sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_wave /*[NW]*/, uint32_t *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(v, lane);
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) { const uint32_t s = s_wave[k]; if (k < w) base += s; tot += s; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__device__ __forceinline__ uint32_t row_stride(uint32_t nbins) { return nbins < 4u ? 4u : nbins; }   // rows are read 16 bytes at a time

// ---------------------------------------------------------------- histogram rows

// H[chunk][digit] and G[group][digit] rows (header).  PACKED: keys are the .x of (key,val) uint2 records.
template <bool PACKED>
__global__ __launch_bounds__(NT) void k_radix_hist(const uint32_t *__restrict__ keys, const uint32_t *n_ptr, int shift,
                                                   int bits, uint32_t *__restrict__ hrows, uint32_t *__restrict__ grows)
{
    __shared__ uint32_t s_hist[GS_RADIX_MAX_BINS];
    const uint32_t n = *n_ptr;
    const uint32_t nchunks = (n + GS_CHUNK - 1) / GS_CHUNK, ngroups = (nchunks + GS_RADIX_SUB - 1) / GS_RADIX_SUB;
    const uint32_t nbins = 1u << bits, mask = nbins - 1, rs = row_stride(nbins);
    for (uint32_t v = blockIdx.x; v < ((ngroups + 7u) & ~7u); v += gridDim.x) {
        uint32_t g;
        if (!gs_xcd_chunk(v, ngroups, g)) continue;
        if (threadIdx.x < rs) s_hist[threadIdx.x] = 0;
        uint32_t kk[GS_RADIX_SUB][IPT];                              // all loads first: their latencies overlap instead of adding up
#pragma unroll
        for (int k = 0; k < GS_RADIX_SUB; k++)
#pragma unroll
            for (int r = 0; r < IPT; r++) {
                const uint32_t i = (g * GS_RADIX_SUB + k) * GS_CHUNK + r * NT + threadIdx.x;
                kk[k][r] = i < n ? keys[PACKED ? 2 * (size_t)i : i] : 0u;
            }
        uint32_t gsum = 0;                                           // this thread's digit over the group
        __syncthreads();
#pragma unroll
        for (int k = 0; k < GS_RADIX_SUB; k++) {
            const uint32_t c = g * GS_RADIX_SUB + k;
            if (c >= nchunks) break;
#pragma unroll
            for (int r = 0; r < IPT; r++) {
                const uint32_t i = c * GS_CHUNK + r * NT + threadIdx.x;
                if (i < n) atomicAdd(&s_hist[(kk[k][r] >> shift) & mask], 1u);
            }
            __syncthreads();
            if (threadIdx.x < rs) { const uint32_t h = s_hist[threadIdx.x]; hrows[(size_t)c * rs + threadIdx.x] = h; gsum += h; s_hist[threadIdx.x] = 0; }
            __syncthreads();
        }
        if (threadIdx.x < rs) grows[(size_t)g * rs + threadIdx.x] = gsum;
    }
}

// Long inputs: gpre[s][d] = sum of the G rows of all groups before super-group s (GS_RADIX_SUPER groups each), totals[d] = sum
// of all G rows.  One 1024-thread workgroup per slab of 32 digits: thread (q, r) owns 4 digits and a contiguous run of
// super-groups; two sweeps over the rows (run totals -> exclusive offsets of the runs through LDS -> exclusive sums).
__global__ __launch_bounds__(1024) void k_radix_gscan(const uint32_t *__restrict__ grows, const uint32_t *n_ptr, int bits,
                                                      uint32_t *__restrict__ gpre, uint32_t *__restrict__ totals)
{
    __shared__ uint4 s_part[128][8];
    const uint32_t n = *n_ptr;
    const uint32_t nchunks = (n + GS_CHUNK - 1) / GS_CHUNK, ngroups = (nchunks + GS_RADIX_SUB - 1) / GS_RADIX_SUB;
    const uint32_t nbins = 1u << bits, rs = row_stride(nbins);
    const uint32_t q = threadIdx.x & 7u, r = threadIdx.x >> 3;
    const uint32_t d0 = blockIdx.x * 32u + q * 4u;                  // this thread's 4 digits
    const bool dig_ok = d0 < rs;
    const uint32_t nsuper = (ngroups + GS_RADIX_SUPER - 1) / GS_RADIX_SUPER;
    const uint32_t spl = (nsuper + 127u) / 128u;                    // super-groups per row lane
    const uint32_t s_lo = min(r * spl, nsuper), s_hi = min(s_lo + spl, nsuper);
    const uint32_t row_lo = s_lo * GS_RADIX_SUPER, row_hi = min(s_hi * GS_RADIX_SUPER, ngroups);
    const uint32_t *col = grows + d0;
    uint4 run = make_uint4(0, 0, 0, 0);
    if (dig_ok) {
        uint32_t g = row_lo;
        for (; g + 8u <= row_hi; g += 8u) {                          // eight rows in flight
            uint4 h[8];
#pragma unroll
            for (int k = 0; k < 8; k++) h[k] = *reinterpret_cast<const uint4 *>(col + (size_t)(g + k) * rs);
#pragma unroll
            for (int k = 0; k < 8; k++) { run.x += h[k].x; run.y += h[k].y; run.z += h[k].z; run.w += h[k].w; }
        }
        for (; g < row_hi; g++) { const uint4 h = *reinterpret_cast<const uint4 *>(col + (size_t)g * rs); run.x += h.x; run.y += h.y; run.z += h.z; run.w += h.w; }
    }
    s_part[r][q] = run;
    __syncthreads();
    uint4 base = make_uint4(0, 0, 0, 0);
    for (uint32_t k = 0; k < r; k++) { const uint4 p = s_part[k][q]; base.x += p.x; base.y += p.y; base.z += p.z; base.w += p.w; }
    if (dig_ok) {
        for (uint32_t sg = s_lo; sg < s_hi; sg++) {
            *reinterpret_cast<uint4 *>(gpre + (size_t)sg * rs + d0) = base;
            const uint32_t g0 = sg * GS_RADIX_SUPER, g1 = min(g0 + GS_RADIX_SUPER, ngroups);
            uint32_t g = g0;
            for (; g + 8u <= g1; g += 8u) {
                uint4 h[8];
#pragma unroll
                for (int k = 0; k < 8; k++) h[k] = *reinterpret_cast<const uint4 *>(col + (size_t)(g + k) * rs);
#pragma unroll
                for (int k = 0; k < 8; k++) { base.x += h[k].x; base.y += h[k].y; base.z += h[k].z; base.w += h[k].w; }
            }
            for (; g < g1; g++) { const uint4 h = *reinterpret_cast<const uint4 *>(col + (size_t)g * rs); base.x += h.x; base.y += h.y; base.z += h.z; base.w += h.w; }
        }
        if (r == 127u) *reinterpret_cast<uint4 *>(totals + d0) = base;   // (empty runs carry the offsets through: lane 127 ends at the grand total)
    }
}

// Stable scatter, one workgroup per chunk.  Item order inside a chunk: wave w owns an eighth of the chunk, processed in IPT
// rounds of 64 consecutive items (lane = item % 64), so "earlier" == (wave, round, lane) lexicographic.
// Rank among equal digits: in-round via LDS match words (+ one ballot for a 9th digit bit), across rounds via a wave-private
// LDS counter row, across waves via an 8-way prefix.  The chunk is then REORDERED IN LDS into digit order and written out slot
// by slot, so consecutive lanes store consecutive addresses inside each digit run instead of 64 unrelated stores per
// instruction.
// Offsets: the workgroup sums histogram rows itself (header).  gpre == nullptr: G rows [0, g) and [0, ngroups) of the whole
// table; otherwise gpre[super-group of g] + the G rows of that super-group before g, digit totals from `totals`; in both
// cases + the H rows of the chunks of group g before c.
// IN_FMT:  GS_RADIX_KEYS = a key array whose value is the element index, GS_RADIX_PACKED = (key,val) uint2 records,
//          GS_RADIX_KEYONLY = 4-byte records that are their own payload (the digit is a bit field of the record).
// OUT_FMT: GS_RADIX_KEYS = the value alone (last pass of an index sort), GS_RADIX_PACKED = (key,val) uint2 records (one
//          8-byte store per item), GS_RADIX_KEYONLY = the 4-byte record.
// zero_key: items whose key equals it store 0 as their value (value-only output): the depth sort uses this so that
// splats with a dropped bucket (key 65536, which sort behind every bucket) leave zeros in the tail of the index list.
// MAXB: bins the instantiation reserves LDS for.
template <int IN_FMT, int OUT_FMT, int MAXB>
__global__ __launch_bounds__(NT) void k_radix_scatter(const void *__restrict__ in, void *__restrict__ out,
                                                      const uint32_t *n_ptr, int shift, int bits, uint32_t zero_key,
                                                      const uint32_t *__restrict__ hrows, const uint32_t *__restrict__ grows,
                                                      const uint32_t *__restrict__ gpre, const uint32_t *__restrict__ totals)
{
    constexpr bool KEYONLY = IN_FMT == GS_RADIX_KEYONLY && OUT_FMT == GS_RADIX_KEYONLY;
    __shared__ uint32_t s_cnt[NW][MAXB];                        // per-wave digit counts -> local slot bases
    constexpr int MATCHB = MAXB <= 256 ? MAXB : 256;            // match words cover the low 8 digit bits; a 9th bit is refined by a ballot
    __shared__ unsigned long long s_match[NW][MATCHB];          // per wave and (low) digit: lanes holding it in the current round
    __shared__ uint32_t s_pre[MAXB < 4 ? 4 : MAXB];             // items of this digit before the current chunk
    __shared__ uint32_t s_tot[MAXB < 4 ? 4 : MAXB];             // items of this digit in the whole input -> start of the digit's output run
    __shared__ uint32_t s_gb[MAXB];                             // global position of local slot 0 of each digit (minus slot)
    __shared__ uint32_t s_k[GS_CHUNK];                          // the chunk in digit order: keys ...
    __shared__ uint32_t s_v[KEYONLY ? 1 : GS_CHUNK];            // ... and values (not for key-only records)
    __shared__ uint32_t s_wave[NW];
    const uint32_t n = *n_ptr;
    const uint32_t nchunks = (n + GS_CHUNK - 1) / GS_CHUNK, ngroups = (nchunks + GS_RADIX_SUB - 1) / GS_RADIX_SUB;
    const uint32_t nbins = 1u << bits, mask = nbins - 1, rs = row_stride(nbins);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // row-sum geometry: thread = (quad of 4 digits, row slice); slice s adds rows lo + s, lo + s + nsl, ...
    const uint32_t nquads = rs / 4u, nsl = (uint32_t)NT / nquads;
    const uint32_t quad = threadIdx.x % nquads, slice = threadIdx.x / nquads;
#define GS_LOAD_CHUNK(KEY, VAL, c)                                                                                     \
    _Pragma("unroll") for (int r = 0; r < IPT; r++) {                                                                  \
        const uint32_t i = (c) * GS_CHUNK + w * (GS_CHUNK / NW) + r * 64 + lane;                                       \
        const bool ok = i < n;                                                                                         \
        if (IN_FMT == GS_RADIX_PACKED) {                                                                               \
            const uint2 kv = ok ? reinterpret_cast<const uint2 *>(in)[i] : make_uint2(0xFFFFFFFFu, 0u);                \
            KEY[r] = kv.x; VAL[r] = kv.y;                                                                              \
        } else {                                                                                                       \
            KEY[r] = ok ? reinterpret_cast<const uint32_t *>(in)[i] : 0xFFFFFFFFu;                                     \
            VAL[r] = i;                                                                                                \
        }                                                                                                              \
    }
    for (uint32_t v = blockIdx.x; v < ((nchunks + 7u) & ~7u); v += gridDim.x) {
        uint32_t c;
        if (!gs_xcd_chunk(v, nchunks, c)) continue;
        const uint32_t g = c / GS_RADIX_SUB;
        for (uint32_t i = threadIdx.x; i < NW * MAXB; i += NT) { (&s_cnt[0][0])[i] = 0; if (i < NW * MATCHB) (&s_match[0][0])[i] = 0ull; }
        if (threadIdx.x < rs) { s_pre[threadIdx.x] = 0; s_tot[threadIdx.x] = 0; }
        __syncthreads();
        uint32_t key[IPT], val[IPT], rank[IPT];
        GS_LOAD_CHUNK(key, val, c)                                   // all loads first: their latencies overlap
        {   // offsets from the histogram rows (under the latency of the key loads above)
            uint4 before = make_uint4(0, 0, 0, 0), all = make_uint4(0, 0, 0, 0);
            const uint32_t lo = gpre ? (g / GS_RADIX_SUPER) * GS_RADIX_SUPER : 0u, hi = gpre ? g : ngroups;
            const uint32_t *col = grows + quad * 4u;
            uint32_t r0 = lo + slice;
            for (; r0 + 3u * nsl < hi; r0 += 4u * nsl) {            // four rows in flight per thread
                const uint4 h0 = *reinterpret_cast<const uint4 *>(col + (size_t)r0 * rs);
                const uint4 h1 = *reinterpret_cast<const uint4 *>(col + (size_t)(r0 + nsl) * rs);
                const uint4 h2 = *reinterpret_cast<const uint4 *>(col + (size_t)(r0 + 2u * nsl) * rs);
                const uint4 h3 = *reinterpret_cast<const uint4 *>(col + (size_t)(r0 + 3u * nsl) * rs);
                const uint4 hh[4] = { h0, h1, h2, h3 };
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    all.x += hh[k].x; all.y += hh[k].y; all.z += hh[k].z; all.w += hh[k].w;
                    if (r0 + k * nsl < g) { before.x += hh[k].x; before.y += hh[k].y; before.z += hh[k].z; before.w += hh[k].w; }
                }
            }
            for (; r0 < hi; r0 += nsl) {
                const uint4 h = *reinterpret_cast<const uint4 *>(col + (size_t)r0 * rs);
                all.x += h.x; all.y += h.y; all.z += h.z; all.w += h.w;
                if (r0 < g) { before.x += h.x; before.y += h.y; before.z += h.z; before.w += h.w; }
            }
            if (slice < c - g * GS_RADIX_SUB) {                      // the chunks of this group before c (at most GS_RADIX_SUB - 1 <= nsl rows)
                const uint4 h = *reinterpret_cast<const uint4 *>(hrows + (size_t)(g * GS_RADIX_SUB + slice) * rs + quad * 4u);
                before.x += h.x; before.y += h.y; before.z += h.z; before.w += h.w;
            }
            if (gpre && slice == 0) {
                const uint4 p = *reinterpret_cast<const uint4 *>(gpre + (size_t)(g / GS_RADIX_SUPER) * rs + quad * 4u);
                const uint4 t = *reinterpret_cast<const uint4 *>(totals + quad * 4u);
                before.x += p.x; before.y += p.y; before.z += p.z; before.w += p.w;
                all = t;
            } else if (gpre) all = make_uint4(0, 0, 0, 0);
            if (before.x) atomicAdd(&s_pre[quad * 4u + 0], before.x);
            if (before.y) atomicAdd(&s_pre[quad * 4u + 1], before.y);
            if (before.z) atomicAdd(&s_pre[quad * 4u + 2], before.z);
            if (before.w) atomicAdd(&s_pre[quad * 4u + 3], before.w);
            if (all.x) atomicAdd(&s_tot[quad * 4u + 0], all.x);
            if (all.y) atomicAdd(&s_tot[quad * 4u + 1], all.y);
            if (all.z) atomicAdd(&s_tot[quad * 4u + 2], all.z);
            if (all.w) atomicAdd(&s_tot[quad * 4u + 3], all.w);
        }
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t i = c * GS_CHUNK + w * (GS_CHUNK / NW) + r * 64 + lane;
            bool ok = i < n;
            const uint32_t d = (key[r] >> shift) & mask;
            if (IN_FMT == GS_RADIX_KEYS) ok = ok && key[r] != GS_RADIX_SKIP;   // compaction: skipped records take no slot
            // match-any: which lanes of the wave hold the same digit this round.  Through LDS (<= 256 bins): every lane ORs
            // its bit into the 64-bit word of its digit (one ds_or_b64 for the whole wave) and reads the word back -- two LDS
            // operations instead of ~5 VALU instructions per digit bit (the scatter competes with the blend for VALU issue).
            const uint32_t dm = d & (uint32_t)(MATCHB - 1);
            if (ok) atomicOr(&s_match[w][dm], 1ull << lane);
            __builtin_amdgcn_wave_barrier();
            unsigned long long peers = ok ? s_match[w][dm] : 0ull;
            if (MAXB > MATCHB) {                                 // 9-bit digits: split the group by the top bit with one ballot
                const bool top = (d >> 8) & 1u;
                const unsigned long long vb = __ballot(ok && top);
                peers &= top ? vb : ~vb;
            }
            const uint32_t before = __popcll(peers & lt), cnt = __popcll(peers);
            const uint32_t prev = ok ? s_cnt[w][d] : 0u;
            __builtin_amdgcn_wave_barrier();                     // every peer has read before the leader bumps / clears
            if (ok && before == 0) { s_cnt[w][d] = prev + cnt; s_match[w][dm] = 0ull; }
            __builtin_amdgcn_wave_barrier();
            rank[r] = prev + before;
        }
        __syncthreads();
        uint32_t chunk_items;                                        // records of this chunk that take a slot
        {   // digit totals of the chunk -> local digit starts; digit totals of the input -> run starts (two scans over the digits)
            const uint32_t d = threadIdx.x;
            uint32_t cw[NW], t = 0;
#pragma unroll
            for (int q = 0; q < NW; q++) { cw[q] = d < nbins ? s_cnt[q][d] : 0u; t += cw[q]; }
            uint32_t tot;
            const uint32_t ls = block_excl_scan(t, s_wave, &tot);            // local slot of the digit's first item (two barriers inside)
            chunk_items = tot;
            const uint32_t dbase = block_excl_scan(d < nbins ? s_tot[d] : 0u, s_wave, &tot);   // start of the digit's output run
            if (d < nbins) {
                uint32_t run = ls;
#pragma unroll
                for (int q = 0; q < NW; q++) { s_cnt[q][d] = run; run += cw[q]; }
                s_gb[d] = dbase + s_pre[d] - ls;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t i = c * GS_CHUNK + w * (GS_CHUNK / NW) + r * 64 + lane;
            if (i < n && (IN_FMT != GS_RADIX_KEYS || key[r] != GS_RADIX_SKIP)) {
                const uint32_t d = (key[r] >> shift) & mask;
                s_k[s_cnt[w][d] + rank[r]] = key[r];
                if (!KEYONLY) s_v[s_cnt[w][d] + rank[r]] = val[r];
            }
        }
        __syncthreads();
        const uint32_t items = chunk_items;
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t slot = r * NT + threadIdx.x;
            if (slot < items) {
                const uint2 kv = make_uint2(s_k[slot], KEYONLY ? 0u : s_v[slot]);
                const uint32_t pos = s_gb[(kv.x >> shift) & mask] + slot;
                if (OUT_FMT == GS_RADIX_PACKED) reinterpret_cast<uint2 *>(out)[pos] = kv;
                else if (OUT_FMT == GS_RADIX_KEYONLY) reinterpret_cast<uint32_t *>(out)[pos] = kv.x;
                else reinterpret_cast<uint32_t *>(out)[pos] = kv.x == zero_key ? 0u : kv.y;
            }
        }
        __syncthreads();
    }
#undef GS_LOAD_CHUNK
}

uint32_t grid_for(uint32_t max_items)
{
    uint32_t g = gs_div_up(max_items, GS_CHUNK);
    if (g < 1) g = 1;
    if (g > 1024) g = 1024;                                      // (4 workgroups of 512 threads per CU at most)
    return (g + 7u) & ~7u;                                       // a multiple of 8: workgroup index mod 8 = XCD (gs_xcd_chunk)
}

}  // namespace

uint32_t gs_radix_grid(uint32_t max_n) { return grid_for(max_n); }

int gs_launch_radix_pass(gs_ctx *ctx, const void *in, int in_fmt, void *out, int out_fmt, const uint32_t *n_ptr,
                         uint32_t max_n, uint32_t hint_n, int shift, int bits, bool have_hist, uint32_t zero_key)
{
    if (bits < 1 || bits > 9) { snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "radix pass: %d-bit digit (1..9 supported)", bits); return GS_E_BADARG; }
    if (hint_n > max_n || hint_n == 0) hint_n = max_n;
    const uint32_t g = grid_for(hint_n), gh = grid_for(gs_div_up(hint_n, GS_RADIX_SUB));
    hipStream_t st = ctx->stream;
    uint32_t *hrows = ctx->hist, *grows = gs_radix_group_rows(ctx);
    // long inputs take the two-level offsets; the choice is a matter of speed only (both forms are exact for any *n_ptr)
    const bool two_level = gs_div_up(hint_n, GS_CHUNK * GS_RADIX_SUB) > GS_RADIX_BRUTE_ROWS;
    uint32_t *gpre = two_level ? ctx->radix_aux + GS_RADIX_MAX_BINS : nullptr, *totals = two_level ? ctx->radix_aux : nullptr;
    if (have_hist) { /* the producer of `in` already wrote the H and G rows */ }
    else if (in_fmt == GS_RADIX_PACKED) hipLaunchKernelGGL(k_radix_hist<true>, dim3(gh), dim3(NT), 0, st, (const uint32_t *)in, n_ptr, shift, bits, hrows, grows);
    else hipLaunchKernelGGL(k_radix_hist<false>, dim3(gh), dim3(NT), 0, st, (const uint32_t *)in, n_ptr, shift, bits, hrows, grows);
    if (two_level) hipLaunchKernelGGL(k_radix_gscan, dim3(gs_div_up(1u << bits, 32u)), dim3(1024), 0, st, grows, n_ptr, bits, gpre, totals);
#define GS_SCATTER(I, O) do { if (bits <= 7) hipLaunchKernelGGL((k_radix_scatter<I, O, 128>), dim3(g), dim3(NT), 0, st, in, out, n_ptr, shift, bits, zero_key, hrows, grows, gpre, totals); \
                              else if (bits == 8) hipLaunchKernelGGL((k_radix_scatter<I, O, 256>), dim3(g), dim3(NT), 0, st, in, out, n_ptr, shift, bits, zero_key, hrows, grows, gpre, totals); \
                              else hipLaunchKernelGGL((k_radix_scatter<I, O, GS_RADIX_MAX_BINS>), dim3(g), dim3(NT), 0, st, in, out, n_ptr, shift, bits, zero_key, hrows, grows, gpre, totals); } while (0)
    if (in_fmt == GS_RADIX_PACKED && out_fmt == GS_RADIX_PACKED) GS_SCATTER(GS_RADIX_PACKED, GS_RADIX_PACKED);
    else if (in_fmt == GS_RADIX_PACKED && out_fmt == GS_RADIX_KEYS) GS_SCATTER(GS_RADIX_PACKED, GS_RADIX_KEYS);
    else if (in_fmt == GS_RADIX_KEYS && out_fmt == GS_RADIX_PACKED) GS_SCATTER(GS_RADIX_KEYS, GS_RADIX_PACKED);
    else if (in_fmt == GS_RADIX_KEYS && out_fmt == GS_RADIX_KEYS) GS_SCATTER(GS_RADIX_KEYS, GS_RADIX_KEYS);
    else if (in_fmt == GS_RADIX_KEYONLY && out_fmt == GS_RADIX_KEYONLY) GS_SCATTER(GS_RADIX_KEYONLY, GS_RADIX_KEYONLY);
    else { snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "radix pass: unsupported record formats %d -> %d", in_fmt, out_fmt); return GS_E_BADARG; }
#undef GS_SCATTER
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// gs_prims.hip -- stable LSD radix pass for gfx950 (wave64): digit-histogram rows, a small scan over the rows, and an
// LDS-reordered scatter.
//
// A pass over n items on a digit of `bits` bits, in chunks of CH items:
//   k_radix_hist     one row of 2^bits counts per chunk: H[chunk][digit] (contiguous rows: coalesced to write and to read).
//                    Skipped when the kernel that PRODUCED the keys filled the rows itself (depth pass A: k_sort_bucket).
//   k_radix_scan     in place: H[c][d] <- sum of H[c'][d] over c' < c (the chunk's offset inside every digit's run), and the
//                    digit totals.  One workgroup (256 / 512 threads) per slab of 16 digits -- 8 to 32 workgroups, a few us.
//   k_radix_scatter  one workgroup per chunk: stable rank of every item among the items of its digit, chunk reordered in LDS,
//                    written to start-of-run + offset-of-chunk + rank.
// Two workgroup geometries, chosen by the expected input length (a matter of speed only): 256 threads x 2048 items for
// short inputs (a frame of 1 M splats is a chain of short kernels that overlap with those of the neighbouring frames:
// small workgroups find room next to the blend's waves), 512 threads x 4096 items beyond GS_RADIX_LARGE_N (at 20 M
// splats the longer pieces of each digit run written per workgroup and the halved row count matter: 210 -> 70 us per pass).
// Measured and dropped: letting every scatter workgroup sum the rows before it itself, flat or two-level (no scan launch):
// faster by 2 % for a frame alone, 5 % slower with three frames in flight (100-200 MB of extra L2 reads per frame); one
// workgroup per 4 chunks with the offsets kept in LDS (a chunk takes ~4.5 us of dependent LDS steps: 18 us instead of 10).
// The kernels read their problem size from device memory (GsControl), so no stage of the frame needs a host round trip.
// All of a chunk's items are loaded before any is processed, LDS match words / wavefront ballots give the stable in-wave
// rank (no MFMA: there is no contraction here).
#include "gs_internal.h"

namespace {

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread across a workgroup of NW wavefronts; *total = workgroup sum
template <int NW>
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

// ---------------------------------------------------------------- histogram rows

// H[chunk][digit]: one contiguous row per chunk.  PACKED: keys are the .x of (key,val) uint2 records.
template <bool PACKED, int NW>
__device__ __forceinline__ void k_radix_hist_body(const uint32_t *__restrict__ keys, const uint32_t *n_ptr, int shift,
                                                  int bits, uint32_t *__restrict__ hist)
{
    constexpr int NT = 64 * NW, IPT = 8, CH = NT * IPT;
    __shared__ uint32_t s_hist[GS_RADIX_MAX_BINS];
    const uint32_t n = *n_ptr;
    const uint32_t nchunks = (n + CH - 1) / CH;
    const uint32_t nbins = 1u << bits, mask = nbins - 1, rs = gs_radix_row_stride(nbins);
    for (uint32_t v = blockIdx.x; v < ((nchunks + 7u) & ~7u); v += gridDim.x) {
        uint32_t c;
        if (!gs_xcd_chunk(v, nchunks, c)) continue;
        for (uint32_t d = threadIdx.x; d < rs; d += NT) s_hist[d] = 0;
        __syncthreads();
        uint32_t kk[IPT];                                            // all loads first: their latencies overlap instead of adding up
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t i = c * CH + r * NT + threadIdx.x;
            kk[r] = i < n ? keys[PACKED ? 2 * (size_t)i : i] : 0u;
        }
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t i = c * CH + r * NT + threadIdx.x;
            if (i < n) atomicAdd(&s_hist[(kk[r] >> shift) & mask], 1u);
        }
        __syncthreads();
        for (uint32_t d = threadIdx.x; d < rs; d += NT) hist[(size_t)c * rs + d] = s_hist[d];
        __syncthreads();
    }
}

template <bool PACKED, int NW>
__global__ __launch_bounds__(64 * NW) void k_radix_hist(const uint32_t *__restrict__ keys, const uint32_t *n_ptr, int shift,
                                                        int bits, uint32_t *__restrict__ hist)
{
    k_radix_hist_body<PACKED, NW>(keys, n_ptr, shift, bits, hist);
}

// In place: H[c][d] <- exclusive sum over the chunks before c; totals[d] = sum over all chunks.  One workgroup of NW waves per
// slab of 16 digits (small workgroups for short inputs: the kernel runs in the gaps of other frames' kernels, and a
// 1024-thread workgroup waits until one CU has 16 free wave slots; 8 waves for long inputs: fewer rows per thread): thread
// (q, r) owns 4 digits and a contiguous run of rows, the first 16 of which stay in registers between the two sweeps (run
// totals -> exclusive offsets of the runs by shuffles + NW LDS partials -> the rows' exclusive sums), 8 rows in flight beyond.
template <int NW>
__device__ __forceinline__ void k_radix_scan_body(uint32_t *__restrict__ hist, const uint32_t *n_ptr, uint32_t chunk, int bits,
                                                  uint32_t *__restrict__ totals)
{
    constexpr int RC = 16;                                          // rows cached in registers
    constexpr uint32_t LANES = 16u * NW;                            // row lanes: 4 quads x LANES threads
    __shared__ uint4 s_part[NW][4];                                 // [wave][quad] totals
    const uint32_t n = *n_ptr;
    const uint32_t nchunks = (n + chunk - 1) / chunk;
    const uint32_t nbins = 1u << bits, rs = gs_radix_row_stride(nbins);
    const uint32_t q = threadIdx.x & 3u, r = threadIdx.x >> 2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t d0 = blockIdx.x * 16u + q * 4u;                  // this thread's 4 digits
    const bool dig_ok = d0 < rs;
    const uint32_t rpl = (nchunks + LANES - 1u) / LANES;            // rows per row lane
    const uint32_t c_lo = min(r * rpl, nchunks), c_hi = min(c_lo + rpl, nchunks);
    uint32_t *col = hist + d0;
    uint4 h[RC];
    uint4 run = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < RC; k++) {                                  // all in flight at once
        h[k] = (dig_ok && c_lo + k < c_hi) ? *reinterpret_cast<const uint4 *>(col + (size_t)(c_lo + k) * rs) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < RC; k++) { run.x += h[k].x; run.y += h[k].y; run.z += h[k].z; run.w += h[k].w; }
    if (dig_ok) {                                                   // long inputs: the rest of the run, eight rows in flight
        uint32_t c = c_lo + RC;
        for (; c + 8u <= c_hi; c += 8u) {
            uint4 t[8];
#pragma unroll
            for (int k = 0; k < 8; k++) t[k] = *reinterpret_cast<const uint4 *>(col + (size_t)(c + k) * rs);
#pragma unroll
            for (int k = 0; k < 8; k++) { run.x += t[k].x; run.y += t[k].y; run.z += t[k].z; run.w += t[k].w; }
        }
        for (; c < c_hi; c++) { const uint4 t = *reinterpret_cast<const uint4 *>(col + (size_t)c * rs); run.x += t.x; run.y += t.y; run.z += t.z; run.w += t.w; }
    }
    // inclusive scan over the 16 row lanes of this wave that share the quad (lanes q, q + 4, ...), then over the waves
    uint4 inc = run;
#pragma unroll
    for (int dlt = 4; dlt < 64; dlt <<= 1) {
        const uint32_t x = __shfl_up(inc.x, dlt, 64), y = __shfl_up(inc.y, dlt, 64), z = __shfl_up(inc.z, dlt, 64), ww = __shfl_up(inc.w, dlt, 64);
        if (lane >= dlt) { inc.x += x; inc.y += y; inc.z += z; inc.w += ww; }
    }
    if (lane >= 60) s_part[w][q] = inc;                             // the wave's total per quad (its last row lane)
    __syncthreads();
    uint4 base = make_uint4(inc.x - run.x, inc.y - run.y, inc.z - run.z, inc.w - run.w);
    uint4 grand = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const uint4 p = s_part[k][q];
        if (k < w) { base.x += p.x; base.y += p.y; base.z += p.z; base.w += p.w; }
        grand.x += p.x; grand.y += p.y; grand.z += p.z; grand.w += p.w;
    }
    if (dig_ok) {
#pragma unroll
        for (int k = 0; k < RC; k++) {
            if (c_lo + k < c_hi) *reinterpret_cast<uint4 *>(col + (size_t)(c_lo + k) * rs) = base;
            base.x += h[k].x; base.y += h[k].y; base.z += h[k].z; base.w += h[k].w;
        }
        uint32_t c = c_lo + RC;
        for (; c + 8u <= c_hi; c += 8u) {
            uint4 t[8];
#pragma unroll
            for (int k = 0; k < 8; k++) t[k] = *reinterpret_cast<const uint4 *>(col + (size_t)(c + k) * rs);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                *reinterpret_cast<uint4 *>(col + (size_t)(c + k) * rs) = base;
                base.x += t[k].x; base.y += t[k].y; base.z += t[k].z; base.w += t[k].w;
            }
        }
        for (; c < c_hi; c++) {
            const uint4 t = *reinterpret_cast<const uint4 *>(col + (size_t)c * rs);
            *reinterpret_cast<uint4 *>(col + (size_t)c * rs) = base;
            base.x += t.x; base.y += t.y; base.z += t.z; base.w += t.w;
        }
        if (r == 0) *reinterpret_cast<uint4 *>(totals + d0) = grand;
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_radix_scan(uint32_t *__restrict__ hist, const uint32_t *n_ptr, uint32_t chunk, int bits,
                                                        uint32_t *__restrict__ totals)
{
    k_radix_scan_body<NW>(hist, n_ptr, chunk, bits, totals);
}

// Stable scatter, one workgroup per chunk.  Item order inside a chunk: wave w owns a 1/NW-th of the chunk, processed in IPT
// rounds of 64 consecutive items (lane = item % 64), so "earlier" == (wave, round, lane) lexicographic.
// Rank among equal digits: in-round via LDS match words (+ one ballot for a 9th digit bit), across rounds via a wave-private
// LDS counter row, across waves via an NW-way prefix.  The chunk is then REORDERED IN LDS into digit order and written out
// slot by slot, so consecutive lanes store consecutive addresses inside each digit run instead of 64 unrelated stores per
// instruction.
// IN_FMT:  GS_RADIX_KEYS = a key array whose value is the element index, GS_RADIX_PACKED = (key,val) uint2 records,
//          GS_RADIX_KEYONLY = 4-byte records that are their own payload (the digit is a bit field of the record).
//          GS_RADIX_KEYIDX = 4-byte records `key bits << shift | index` (digit = record >> shift, value = the low bits).
// OUT_FMT: GS_RADIX_KEYS = the value alone (last pass of an index sort), GS_RADIX_PACKED = (key,val) uint2 records (one
//          8-byte store per item), GS_RADIX_KEYONLY = the 4-byte record, GS_RADIX_KEYIDX = the key bits above this pass's
//          digit on top of the element index (idx_bits wide).
// zero_key: items whose key equals it store 0 as their value (value-only output): the depth sort uses this so that
// splats with a dropped bucket (key 65536, which sort behind every bucket) leave zeros in the tail of the index list.
// MAXB: bins the instantiation reserves LDS for (128 for the <= 7-bit digits of the pair sort; with 4-byte LDS slots for
// key-only records a pair-sort workgroup of the short geometry needs 11 KiB of LDS).
template <int IN_FMT, int OUT_FMT, int MAXB, int NW>
__device__ __forceinline__ void k_radix_scatter_body(const void *__restrict__ in, void *__restrict__ out,
                                                     const uint32_t *n_ptr, int shift, int bits, uint32_t zero_key,
                                                     const uint32_t *__restrict__ hist_scanned, const uint32_t *__restrict__ totals,
                                                     int idx_bits, uint32_t *count_out, const uint32_t *fill_to)
{
    constexpr int NT = 64 * NW, IPT = 8, CH = NT * IPT;
    constexpr bool KEYONLY = (IN_FMT == GS_RADIX_KEYONLY && OUT_FMT == GS_RADIX_KEYONLY) || IN_FMT == GS_RADIX_KEYIDX;
    __shared__ uint32_t s_cnt[NW][MAXB];                        // per-wave digit counts -> local slot bases
    constexpr int MATCHB = MAXB <= 256 ? MAXB : 256;            // match words cover the low 8 digit bits; a 9th bit is refined by a ballot
    __shared__ unsigned long long s_match[NW][MATCHB];          // per wave and (low) digit: lanes holding it in the current round
    __shared__ uint32_t s_dbase[MAXB];                          // start of every digit's output run (whole array)
    __shared__ uint32_t s_gb[MAXB];                             // global position of local slot 0 of each digit (minus slot)
    __shared__ uint32_t s_k[CH];                                // the chunk in digit order: keys ...
    __shared__ uint32_t s_v[KEYONLY ? 1 : CH];                  // ... and values (not for key-only records)
    __shared__ uint32_t s_wave[NW];
    const uint32_t n = *n_ptr;
    const uint32_t nchunks = (n + CH - 1) / CH;
    const uint32_t nbins = 1u << bits, mask = nbins - 1, rs = gs_radix_row_stride(nbins);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (IN_FMT == GS_RADIX_KEYIDX && fill_to) {                  // the zero tail behind the sorted records (depth sort: dropped buckets)
        const uint32_t upto = *fill_to;
        for (uint32_t i = n + blockIdx.x * NT + threadIdx.x; i < upto; i += gridDim.x * NT) reinterpret_cast<uint32_t *>(out)[i] = 0u;
    }
    if (blockIdx.x >= ((nchunks + 7u) & ~7u) && !(count_out && blockIdx.x == 0)) return;
    {   // exclusive scan of the <= 512 digit totals (MAXB / NT per thread) -> run starts
        constexpr int DPT = (MAXB + NT - 1) / NT;
        uint32_t tv[DPT], sum = 0;
#pragma unroll
        for (int k = 0; k < DPT; k++) { const uint32_t d = threadIdx.x * DPT + k; tv[k] = d < nbins ? totals[d] : 0u; sum += tv[k]; }
        uint32_t tot;
        uint32_t ex = block_excl_scan<NW>(sum, s_wave, &tot);
#pragma unroll
        for (int k = 0; k < DPT; k++) { const uint32_t d = threadIdx.x * DPT + k; if (d < nbins) s_dbase[d] = ex; ex += tv[k]; }
        if (count_out && blockIdx.x == 0 && threadIdx.x == 0) *count_out = tot;   // records that take a slot = the next pass's input
    }
    for (uint32_t v = blockIdx.x; v < ((nchunks + 7u) & ~7u); v += gridDim.x) {
        uint32_t c;
        if (!gs_xcd_chunk(v, nchunks, c)) continue;
        for (uint32_t i = threadIdx.x; i < NW * MAXB; i += NT) { (&s_cnt[0][0])[i] = 0; if (i < NW * MATCHB) (&s_match[0][0])[i] = 0ull; }
        __syncthreads();
        uint32_t key[IPT], val[IPT], rank[IPT];
#pragma unroll
        for (int r = 0; r < IPT; r++) {                               // all loads first: their latencies overlap
            const uint32_t i = c * CH + w * (CH / NW) + r * 64 + lane;
            const bool ok = i < n;
            if (IN_FMT == GS_RADIX_PACKED) {
                const uint2 kv = ok ? reinterpret_cast<const uint2 *>(in)[i] : make_uint2(0xFFFFFFFFu, 0u);
                key[r] = kv.x; val[r] = kv.y;
            } else {
                key[r] = ok ? reinterpret_cast<const uint32_t *>(in)[i] : 0xFFFFFFFFu;
                val[r] = i;
            }
        }
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t i = c * CH + w * (CH / NW) + r * 64 + lane;
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
        {   // digit totals of the chunk -> local digit starts (exclusive scan over digits, MAXB / NT per thread)
            constexpr int DPT = (MAXB + NT - 1) / NT;
            uint32_t tv[DPT], sum = 0;
#pragma unroll
            for (int k = 0; k < DPT; k++) {
                const uint32_t d = threadIdx.x * DPT + k;
                tv[k] = 0;
                if (d < nbins) {
#pragma unroll
                    for (int q = 0; q < NW; q++) tv[k] += s_cnt[q][d];
                }
                sum += tv[k];
            }
            uint32_t tot;
            uint32_t ex = block_excl_scan<NW>(sum, s_wave, &tot);            // (two barriers inside)
            chunk_items = tot;
#pragma unroll
            for (int k = 0; k < DPT; k++) {
                const uint32_t d = threadIdx.x * DPT + k;
                if (d < nbins) {
                    uint32_t run = ex;                                       // local slot of the digit's first item
#pragma unroll
                    for (int q = 0; q < NW; q++) { const uint32_t cq = s_cnt[q][d]; s_cnt[q][d] = run; run += cq; }
                    s_gb[d] = s_dbase[d] + hist_scanned[(size_t)c * rs + d] - ex;
                }
                ex += tv[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t i = c * CH + w * (CH / NW) + r * 64 + lane;
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
                else if (OUT_FMT == GS_RADIX_KEYIDX) reinterpret_cast<uint32_t *>(out)[pos] = ((kv.x >> (shift + bits)) << idx_bits) | kv.y;
                else if (IN_FMT == GS_RADIX_KEYIDX) reinterpret_cast<uint32_t *>(out)[pos] = kv.x & ((1u << shift) - 1u);
                else reinterpret_cast<uint32_t *>(out)[pos] = kv.x == zero_key ? 0u : kv.y;
            }
        }
        __syncthreads();
    }
}

template <int IN_FMT, int OUT_FMT, int MAXB, int NW>
__global__ __launch_bounds__(64 * NW) void k_radix_scatter(const void *__restrict__ in, void *__restrict__ out,
                                                           const uint32_t *n_ptr, int shift, int bits, uint32_t zero_key,
                                                           const uint32_t *__restrict__ hist_scanned, const uint32_t *__restrict__ totals,
                                                           int idx_bits, uint32_t *count_out, const uint32_t *fill_to)
{
    k_radix_scatter_body<IN_FMT, OUT_FMT, MAXB, NW>(in, out, n_ptr, shift, bits, zero_key, hist_scanned, totals, idx_bits, count_out, fill_to);
}

// ---------------------------------------------------------------- the MSD depth sort: its last two kernels (round 5)
// The depth sort of a frame of <= 2^24 splats used to be seven launches (depth, bucket + histogram A, scan A, scatter A, histogram B,
// scan B, scatter B), most of them at the launch floor at 1 M splats.  It is four now: depth, bucket, and these two.
//   k_sort_bucket<.., MSD> (gs_sort.hip) writes the 16-bit bucket keys, the histogram rows of the HIGH bucket byte H[chunk][256] and,
//     by one atomicAdd per digit a chunk holds, the rows G[group][256] of groups of GS_MSD_GROUP chunks.
//   k_msd_scatter: one stable pass by the high byte WITHOUT a scan launch: a chunk's offset inside a digit's run is the sum of the
//     group rows before its group and of the <= 31 chunk rows of its group before it (one digit per thread, ~50 independent 4-byte
//     loads: ~12 MB of L2 reads per 1 M-splat sort, where every workgroup summing all rows before it would read 128 MB: measured and
//     dropped in round 2), the digits' run starts are the column sums of the group rows.  Records out: low bucket byte << 24 | index.
//   k_seg_sort: the array now consists of 256 segments (one per high byte), each in index order; ONE workgroup per segment sorts it by
//     the low byte, stably, inside LDS: blocks of NT x IPT records, wave-private digit counters, match words for the in-round rank --
//     the scatter's machinery with the offsets kept in LDS -- and writes the index list.  A segment longer than a block (depth
//     distributions are peaked: the busiest 1/256 of the depth range holds 1 % of a Gaussian cloud) first counts its digits, then
//     takes its blocks in order with a cursor per digit: any length is sorted correctly, a pathological one (all depths equal: one
//     segment) just slowly.  The zero tail [V', V) of dropped buckets (index.js:561-567) is filled here.
// The order is the reference's: ascending (bucket, original index) -- stable pass by the high byte, then stable by the low byte
// inside a segment whose records are in index order.
// Column dg of a table of 256-word rows summed over the rows [0, na) of table A and then [b0, b1) of table B -- BATCH loads in flight:
// every load is unconditional (its row index clamped to a row that exists), what lies beyond the end contributes nothing.  A row read
// costs an L2 miss (the group rows were written by device-scope atomics, which do not leave the line in the XCD's L2), so what matters
// is the number of dependent round trips, not the loads.
template <int BATCH>
__device__ __forceinline__ uint32_t msd_column_sum(const uint32_t *__restrict__ A, uint32_t na, const uint32_t *__restrict__ Bt, uint32_t b0, uint32_t b1, uint32_t dg)
{
    const uint32_t total = na + (b1 - b0);
    uint32_t sum = 0;
    for (uint32_t v0 = 0; v0 < total; v0 += (uint32_t)BATCH) {
        uint32_t t[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; k++) {
            const uint32_t v = v0 + (uint32_t)k < total ? v0 + (uint32_t)k : v0;       // (v0 < total: a row that exists)
            const uint32_t *row = v < na ? A + (size_t)v * 256u : Bt + (size_t)(b0 + (v - na)) * 256u;
            t[k] = row[dg];
        }
#pragma unroll
        for (int k = 0; k < BATCH; k++) sum += v0 + (uint32_t)k < total ? t[k] : 0u;
    }
    return sum;
}

// Stable rank of every item of a wavefront among the items of ITS digit, over IPT rounds of 64 consecutive items (the wave's items in
// order: round, lane): rank = (items of the digit in earlier rounds) + (lanes of this round that hold the digit, below this one).
// In-round: match-any through an LDS word per digit (every lane ORs its bit in, reads the word back).  Across rounds: the first lane
// of every group adds the group's size to the wave's counter of the digit with a RETURNING LDS atomic -- LDS operations of one
// wavefront execute in program order, so the values returned over the rounds are the running counts even though nothing waits for
// them: they are collected at the end, one bpermute per round.  One LDS round trip per round (the word read back), where
// k_radix_scatter's loop has two dependent ones (word and counter read, then the counter written before the next round may read it).
// s_match: 256 words of THIS wave, zero on entry and on exit; s_cnt: 256 counters of this wave -- zero before the first call of a
// block, the wave's digit counts afterwards (a second call continues the count: IPT = 32 as two calls of 16 keeps the registers of one).
template <int IPT>
__device__ __forceinline__ void wave_digit_ranks(const uint32_t *dig, const bool *ok, unsigned long long *s_match, uint32_t *s_cnt, uint32_t *rank, int lane)
{
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t prev[IPT], before[IPT];
    int leader[IPT];
#pragma unroll
    for (int r = 0; r < IPT; r++) {
        if (ok[r]) atomicOr(&s_match[dig[r]], 1ull << lane);
        __builtin_amdgcn_wave_barrier();
        const unsigned long long peers = ok[r] ? s_match[dig[r]] : 0ull;
        before[r] = (uint32_t)__popcll(peers & lt);
        leader[r] = peers ? __ffsll((long long)peers) - 1 : 0;
        uint32_t pv = 0;
        if (ok[r] && before[r] == 0u) { pv = atomicAdd(&s_cnt[dig[r]], (uint32_t)__popcll(peers)); s_match[dig[r]] = 0ull; }
        __builtin_amdgcn_wave_barrier();
        prev[r] = pv;
    }
#pragma unroll
    for (int r = 0; r < IPT; r++) rank[r] = (uint32_t)__shfl((int)prev[r], leader[r], 64) + before[r];
}

#ifndef GS_SEG_B
#define GS_SEG_B 4096u             // records per block of k_seg_sort (4 wavefronts x 16 rounds x 64)
#endif
#define GS_SEG_MAXBLK 8u           // a segment of more blocks is one work item: its blocks in turn, by one workgroup
template <int NW>
__device__ __forceinline__ void k_msd_scatter_body(const uint32_t *__restrict__ keys, uint32_t *__restrict__ rec, const uint32_t *n_ptr,
                                                   const uint32_t *__restrict__ rows, const uint32_t *__restrict__ grp, uint32_t *count_out,
                                                   uint32_t *__restrict__ seg_tab, uint32_t tail_req, GsControl *ctl, uint32_t n_host)
{
    constexpr int NT = 64 * NW, IPT = 8, CH = NT * IPT, NB = 256, PB = 24;
    __shared__ uint32_t s_cnt[NW][NB];
    __shared__ unsigned long long s_match[NW][NB];
    __shared__ uint32_t s_dbase[NB], s_gb[NB];
    __shared__ uint32_t s_k[CH], s_v[CH];
    __shared__ uint32_t s_wave[NW];
    __shared__ uint32_t s_cut, s_thr;
    // (the depth sort's record count is the host's N: one dependent load less in front of everything -- what this kernel waits for is
    // round trips, and every one of them is ~2 us alone and more with two other kernels in flight)
    const uint32_t n = n_host ? n_host : *n_ptr;
    const uint32_t nchunks = (n + CH - 1) / CH, ngroups = (nchunks + GS_MSD_GROUP - 1u) / GS_MSD_GROUP;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t dg = threadIdx.x < (uint32_t)NB ? threadIdx.x : 0u;   // this thread's digit (threads beyond 256 compute digit 0's sums and drop them)
    const bool has_dg = threadIdx.x < (uint32_t)NB;
    const uint32_t vend = (nchunks + 7u) & ~7u;
    if (blockIdx.x >= vend && !(count_out && blockIdx.x == 0)) return;
    // The loads of this workgroup's (first) chunk -- its keys, and the rows that give the chunk's offset inside every segment -- depend on
    // nothing the kernel computes: they go out HERE, together with the group rows of the segment totals below, so that the kernel waits
    // for ONE round trip where it waited for three (the record count, the totals, then keys + rows): 50.9 -> 42.7 us per pair of frames
    // in the pipelined loop (tools/gpu_r5aa.sh; alone 10.0 -> 9.9: there the trips are short).  Unconditional, indices clamped: all in
    // flight at once.
    uint32_t key[IPT], pt[PB];
    uint32_t c = 0;
    auto load_chunk = [&](uint32_t cc) {
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t i = cc * CH + w * (CH / NW) + r * 64 + lane;
            key[r] = keys[i < n ? i : n - 1u];                          // (n >= 1 here: there is a chunk; what lies behind n is masked below)
        }
        // This chunk's offset inside every digit's run = the rows of the groups before its own + the rows of the chunks of its group
        // before it: <= 15 + 31 rows at 1 M splats (an L2 miss each: the rows were written by another XCD)
        const uint32_t g0 = cc / GS_MSD_GROUP, nrow = g0 + (cc - g0 * GS_MSD_GROUP);
#pragma unroll
        for (int k = 0; k < PB; k++) {
            const uint32_t vv = (uint32_t)k < nrow ? (uint32_t)k : 0u;
            const uint32_t *row = (nrow == 0u || vv < g0) ? grp + (size_t)vv * 256u : rows + (size_t)(g0 * GS_MSD_GROUP + (vv - g0)) * 256u;
            pt[k] = row[dg];
        }
    };
    bool preloaded = blockIdx.x < vend && gs_xcd_chunk(blockIdx.x, nchunks, c);
    if (preloaded) load_chunk(c);
    for (uint32_t i = threadIdx.x; i < NW * NB; i += NT) (&s_match[0][0])[i] = 0ull;   // (the ranking leaves them zero)
    if (threadIdx.x == 0) { s_cut = 0u; s_thr = 0u; }
    uint32_t thr = 0u;
    {   // digit totals = column sums of the group rows -> run starts
        const uint32_t tot_d = has_dg ? msd_column_sum<16>(grp, ngroups, grp, 0u, 0u, dg) : 0u;
        uint32_t tot;
        uint32_t ex = block_excl_scan<NW>(tot_d, s_wave, &tot);
        // A TAIL sort (tail_req != 0: the frame reads the last tail_req positions of the order at most -- the nearest splats, one binning
        // round, gs_api.hip: sort_near_request): the order is cut at a SEGMENT boundary, the start of the segment that holds position
        // V' - tail_req.  The segments before it are neither scattered nor sorted; the records behind the cut are stored from slot 0
        // (k_project: near_sorted, j_base = n_valid - n_sorted).  The cut costs nothing: every workgroup has the digit totals in its hands
        // here -- where the near-only sorts of long inputs build a depth histogram and search it for a threshold (10 us at 1 M splats).
        if (tail_req && tail_req < tot) {
            const uint32_t suffix = tot - ex;                              // records of this digit and the ones behind it
            if (has_dg && suffix >= tail_req && suffix - tot_d < tail_req) { s_cut = ex; s_thr = dg; }   // (exactly one digit: its segment is not empty)
            __syncthreads();
        }
        const uint32_t cut = s_cut;
        thr = s_thr;
        ex -= ex >= cut ? cut : ex;
        if (has_dg) s_dbase[dg] = ex;
        if (count_out && blockIdx.x == 0 && threadIdx.x == 0) *count_out = tot - cut;   // the records that take a slot (V', or what a tail sort keeps)
        if (tail_req && blockIdx.x == 0 && threadIdx.x == 0) { ctl->near_sorted = 3u; ctl->n_valid = tot; }   // (3: a tail sort; read as "not 0" on the device)
        if (blockIdx.x == 0) {
            // k_seg_sort's work items (workgroup 0 writes them, every workgroup of that kernel reads its own): one per block of
            // GS_SEG_B records of every non-empty segment -- the blocks of a segment are sorted by different workgroups --, but ONE item
            // for a segment of more than GS_SEG_MAXBLK blocks (taken block after block by one workgroup: each block of a segment counts
            // the whole segment first, which is quadratic in the length).  tab[0] = items, tab[1] = V'; items from tab + 4:
            // (segment start, segment length, block, blocks | 0xFFFFFFFF = all of them in turn)
            const uint32_t nbk = (tot_d && dg >= thr) ? (tot_d + GS_SEG_B - 1u) / GS_SEG_B : 0u;
            const uint32_t ni = nbk > GS_SEG_MAXBLK ? 1u : nbk;
            uint32_t nitems;
            const uint32_t ix = block_excl_scan<NW>(has_dg ? ni : 0u, s_wave, &nitems);
            uint4 *items = reinterpret_cast<uint4 *>(seg_tab + 4);
            if (has_dg) for (uint32_t j = 0; j < ni; j++) items[ix + j] = make_uint4(ex, tot_d, j, nbk > GS_SEG_MAXBLK ? 0xFFFFFFFFu : nbk);
            if (threadIdx.x == 0) { seg_tab[0] = nitems; seg_tab[1] = tot - cut; }
        }
    }
    for (uint32_t v = blockIdx.x; v < vend; v += gridDim.x) {
        if (!preloaded) { if (!gs_xcd_chunk(v, nchunks, c)) continue; load_chunk(c); }   // (a grid smaller than the chunks: the later ones as they come)
        preloaded = false;
        for (uint32_t i = threadIdx.x; i < NW * NB; i += NT) (&s_cnt[0][0])[i] = 0;
        __syncthreads();
        uint32_t rank[IPT];
        const uint32_t g0 = c / GS_MSD_GROUP, nrow = g0 + (c - g0 * GS_MSD_GROUP);
        {
            uint32_t dig[IPT]; bool ok[IPT];
#pragma unroll
            for (int r = 0; r < IPT; r++) {                              // culled / dropped splats take no slot, nor do those before a tail sort's cut
                if (c * CH + w * (CH / NW) + r * 64 + lane >= n) key[r] = GS_RADIX_SKIP;
                dig[r] = (key[r] >> 8) & 255u;
                if (dig[r] < thr) key[r] = GS_RADIX_SKIP;
                ok[r] = key[r] != GS_RADIX_SKIP;
            }
            wave_digit_ranks<IPT>(dig, ok, &s_match[w][0], &s_cnt[w][0], rank, lane);
        }
        uint32_t pre = 0;
#pragma unroll
        for (int k = 0; k < PB; k++) pre += (uint32_t)k < nrow ? pt[k] : 0u;
        for (uint32_t v0 = PB; v0 < nrow; v0 += 16u) {                // (more than 24 rows: the later chunks of the later groups)
            uint32_t t[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t vv = v0 + (uint32_t)k < nrow ? v0 + (uint32_t)k : v0;
                const uint32_t *row = vv < g0 ? grp + (size_t)vv * 256u : rows + (size_t)(g0 * GS_MSD_GROUP + (vv - g0)) * 256u;
                t[k] = row[dg];
            }
#pragma unroll
            for (int k = 0; k < 16; k++) pre += v0 + (uint32_t)k < nrow ? t[k] : 0u;
        }
        __syncthreads();
        uint32_t chunk_items;
        {   // digit totals of the chunk -> local digit starts; global position of every digit's local slot 0
            uint32_t tv = 0;
            if (has_dg) {
#pragma unroll
                for (int q = 0; q < NW; q++) tv += s_cnt[q][dg];
            }
            uint32_t tot;
            const uint32_t ex = block_excl_scan<NW>(tv, s_wave, &tot);          // (two barriers inside)
            chunk_items = tot;
            if (has_dg) {
                uint32_t run = ex;
#pragma unroll
                for (int q = 0; q < NW; q++) { const uint32_t cq = s_cnt[q][dg]; s_cnt[q][dg] = run; run += cq; }
                s_gb[dg] = s_dbase[dg] + pre - ex;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            if (key[r] != GS_RADIX_SKIP) {
                const uint32_t d = (key[r] >> 8) & 255u;
                s_k[s_cnt[w][d] + rank[r]] = key[r];
                s_v[s_cnt[w][d] + rank[r]] = c * CH + w * (CH / NW) + r * 64 + lane;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const uint32_t slot = r * NT + threadIdx.x;
            if (slot < chunk_items) {
                const uint32_t k = s_k[slot];
                rec[s_gb[(k >> 8) & 255u] + slot] = ((k & 255u) << 24) | s_v[slot];
            }
        }
        __syncthreads();
    }
}

// one block of a segment: R rounds per wavefront, NT x R records at most (the kernel picks the smallest R that holds the block's records:
// a near-only sort's segments hold a few hundred, and every round is an LDS round trip whether it holds records or not)
template <int NW, int R>
__device__ __forceinline__ void seg_sort_block(const uint32_t *__restrict__ blk, uint32_t items, uint32_t *__restrict__ out, uint32_t (*s_cnt)[256],
                                               unsigned long long (*s_match)[256], const uint32_t *s_dbase, uint32_t *s_cur, uint32_t *s_gb, uint32_t *s_k,
                                               uint32_t *s_wave, uint32_t start, bool multi)
{
    constexpr int NT = 64 * NW, NB = 256;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t dg = threadIdx.x < (uint32_t)NB ? threadIdx.x : 0u;
    const bool has_dg = threadIdx.x < (uint32_t)NB;
    for (uint32_t i = threadIdx.x; i < NW * NB; i += NT) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    uint32_t key[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        // item order inside a block: (wave, round, lane) = position = index order.  Unconditional and unclamped: all loads in flight,
        // one base address + immediate offsets; what lies behind the block's end is other segments' records or unused scratch of the
        // same allocation (rec holds 2 x capacity words) and is ignored below
        key[r] = blk[w * (R * 64) + r * 64 + lane];
    }
    // (1) the waves' digit counts: fire-and-forget LDS atomics, nothing waits for them but the barrier
#pragma unroll
    for (int r = 0; r < R; r++)
        if ((uint32_t)(w * (R * 64) + r * 64 + lane) < items) atomicAdd(&s_cnt[w][key[r] >> 24], 1u);
    __syncthreads();
    {   // (2) counts -> the local slot at which every wave's items of every digit begin; the digit's global position
        uint32_t tv = 0;
        if (has_dg) {
#pragma unroll
            for (int q = 0; q < NW; q++) tv += s_cnt[q][dg];
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan<NW>(tv, s_wave, &tot);
        if (has_dg) {
            uint32_t run = ex;
#pragma unroll
            for (int q = 0; q < NW; q++) { const uint32_t cq = s_cnt[q][dg]; s_cnt[q][dg] = run; run += cq; }
            // global position of local slot 0 of the digit: one block = the whole segment in digit order; several = the digit's
            // run inside the segment + what the blocks before this one put there
            s_gb[dg] = multi ? start + s_dbase[dg] + s_cur[dg] - ex : start;
            if (multi) s_cur[dg] += tv;
        }
    }
    __syncthreads();
    // (3) stable ranks with the counters running on from those slots: the rank IS the item's local slot, it is placed at once and no
    // rank outlives its eight rounds (all 32 ranks in registers at a time took 255 of them)
#pragma unroll
    for (int h = 0; h < R / 8; h++) {
        uint32_t dig[8], slot[8]; bool ok[8];
#pragma unroll
        for (int r = 0; r < 8; r++) { ok[r] = (uint32_t)(w * (R * 64) + (h * 8 + r) * 64 + lane) < items; dig[r] = key[h * 8 + r] >> 24; }
        wave_digit_ranks<8>(dig, ok, &s_match[w][0], &s_cnt[w][0], slot, lane);
#pragma unroll
        for (int r = 0; r < 8; r++) if (ok[r]) s_k[slot[r]] = key[h * 8 + r];
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < R; r++) {                                     // (eight at a time: unrolled whole, the records, digit bases and addresses cost 100 registers)
        const uint32_t slot = r * NT + threadIdx.x;
        if (slot < items) { const uint32_t k = s_k[slot]; out[s_gb[k >> 24] + slot] = k & 0x00FFFFFFu; }
    }
    __syncthreads();
}

template <int NW, int IPT>
__device__ __forceinline__ void k_seg_sort_body(const uint32_t *__restrict__ rec, uint32_t *__restrict__ out, const uint32_t *__restrict__ seg_tab, const uint32_t *fill_to)
{
    constexpr int NT = 64 * NW, B = NT * IPT, NB = 256;
    static_assert(NT >= NB && IPT == 16 && B == (int)GS_SEG_B, "one digit per thread; blocks of 8 / 16 rounds");
    __shared__ uint32_t s_cnt[NW][NB];
    __shared__ unsigned long long s_match[NW][NB];
    __shared__ uint32_t s_dbase[NB], s_cur[NB], s_gb[NB];
    __shared__ uint32_t s_k[B];
    __shared__ uint32_t s_wave[NW];
    const uint32_t dg = threadIdx.x < (uint32_t)NB ? threadIdx.x : 0u;
    const bool has_dg = threadIdx.x < (uint32_t)NB;
    const uint4 *items = reinterpret_cast<const uint4 *>(seg_tab + 4);
    // (the header and the workgroup's first item in ONE round trip -- what this kernel reads was written by another XCD a moment ago,
    // every dependent load is ~2 us: the table holds at least GS_SEG_GRID items' room, an item beyond the count is read and dropped)
    const uint4 first_item = items[blockIdx.x];
    const uint32_t nitems = seg_tab[0], total = seg_tab[1];
    for (uint32_t i = threadIdx.x; i < NW * NB; i += NT) (&s_match[0][0])[i] = 0ull;   // (the ranking leaves them zero)
    if (fill_to) {  // the zero tail behind the V' sorted records: splats whose bucket the reference drops (index.js:561-567 leave their slots 0)
        const uint32_t upto = *fill_to;
        for (uint32_t i = total + blockIdx.x * NT + threadIdx.x; i < upto; i += gridDim.x * NT) out[i] = 0u;
    }
    for (uint32_t v = blockIdx.x; v < nitems; v += gridDim.x) {
        const uint4 it = v == blockIdx.x ? first_item : items[v];
        const uint32_t start = it.x, len = it.y, blk = it.z, nblk = it.w;
        const uint32_t *seg = rec + start;
        __syncthreads();
        if (nblk == 1u) {
            if (len <= (uint32_t)NT * 8u) seg_sort_block<NW, 8>(seg, len, out, s_cnt, s_match, s_dbase, s_cur, s_gb, s_k, s_wave, start, false);
            else seg_sort_block<NW, 16>(seg, len, out, s_cnt, s_match, s_dbase, s_cur, s_gb, s_k, s_wave, start, false);
            continue;
        }
        // a segment of several blocks: its digit totals (the digits' run starts inside the segment) and what the blocks before THIS one
        // hold of every digit (where this block's items of the digit go inside the run) -- counted here, by every block for itself
        const bool all = nblk == 0xFFFFFFFFu;
        const uint32_t before = all ? 0u : blk * (uint32_t)B;
        if (has_dg) { s_cur[dg] = 0; s_dbase[dg] = 0; }
        __syncthreads();
        for (uint32_t i0 = 0; i0 < len; i0 += 16u * NT) {            // sixteen loads in flight per thread (unclamped: see seg_sort_block)
            uint32_t t[16];
#pragma unroll
            for (int k = 0; k < 16; k++) t[k] = seg[i0 + (uint32_t)k * NT + threadIdx.x];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t i = i0 + (uint32_t)k * NT + threadIdx.x;
                if (i < len) { atomicAdd(&s_dbase[t[k] >> 24], 1u); if (i < before) atomicAdd(&s_cur[t[k] >> 24], 1u); }
            }
        }
        __syncthreads();
        const uint32_t tv = has_dg ? s_dbase[dg] : 0u;
        uint32_t tt;
        const uint32_t ex = block_excl_scan<NW>(tv, s_wave, &tt);
        if (has_dg) s_dbase[dg] = ex;
        __syncthreads();
        if (!all) {
            const uint32_t items_b = len - before < (uint32_t)B ? len - before : (uint32_t)B;
            seg_sort_block<NW, 16>(seg + before, items_b, out, s_cnt, s_match, s_dbase, s_cur, s_gb, s_k, s_wave, start, true);
        } else {
            for (uint32_t b = 0; b * (uint32_t)B < len; b++) {
                const uint32_t items_b = len - b * (uint32_t)B < (uint32_t)B ? len - b * (uint32_t)B : (uint32_t)B;
                seg_sort_block<NW, 16>(seg + (size_t)b * B, items_b, out, s_cnt, s_match, s_dbase, s_cur, s_gb, s_k, s_wave, start, true);
            }
        }
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW, 4) void k_msd_scatter(const uint32_t *__restrict__ keys, uint32_t *__restrict__ rec, const uint32_t *n_ptr,
                                                         const uint32_t *__restrict__ rows, const uint32_t *__restrict__ grp, uint32_t *count_out,
                                                         uint32_t *__restrict__ seg_tab, uint32_t tail_req, GsControl *ctl, uint32_t n_host)
{
    k_msd_scatter_body<NW>(keys, rec, n_ptr, rows, grp, count_out, seg_tab, tail_req, ctl, n_host);
}
template <int NW, int IPT>
__global__ __launch_bounds__(64 * NW, 4) void k_seg_sort(const uint32_t *__restrict__ rec, uint32_t *__restrict__ out, const uint32_t *__restrict__ seg_tab, const uint32_t *fill_to)
{
    k_seg_sort_body<NW, IPT>(rec, out, seg_tab, fill_to);
}
template <int NW> GS_BODY(F_msd_scatter, k_msd_scatter_body<NW>);
template <int NW, int IPT> GS_BODY(F_seg_sort, k_seg_sort_body<NW, IPT>);
#define GS_SEG_GRID 512u           // k_seg_sort: workgroups (they stride over the work items: 256 segments + the extra blocks of the long ones)
#ifndef GS_SEG_NW
#define GS_SEG_NW 4                // k_seg_sort: wavefronts per workgroup ...
#endif
#ifndef GS_SEG_IPT
#define GS_SEG_IPT 16              // ... and records per thread and block (4 x 64 x 16 = GS_SEG_B records per block)
#endif

uint32_t grid_for(uint32_t items, uint32_t chunk)
{
    uint32_t g = gs_div_up(items, chunk);
    if (g < 1) g = 1;
    const uint32_t cap = chunk == GS_CHUNK_L ? 1024u : 2048u;    // workgroups the chip holds at once, roughly
    if (g > cap) g = cap;
    return (g + 7u) & ~7u;                                       // a multiple of 8: workgroup index mod 8 = XCD (gs_xcd_chunk)
}

template <int NW>
int launch_pass(gs_ctx *ctx, const void *in, int in_fmt, void *out, int out_fmt, const uint32_t *n_ptr, uint32_t hint_n, int shift,
                int bits, bool have_hist, uint32_t zero_key, int idx_bits, uint32_t *count_out, const uint32_t *fill_to)
{
    constexpr uint32_t CH = 64 * NW * 8;
    const uint32_t g = grid_for(hint_n, CH);
    hipStream_t st = ctx->stream;
    uint32_t *totals = ctx->radix_aux;
    const dim3 G(g), B(64 * NW);
    if (have_hist) { /* the producer of `in` already wrote hist[chunk][digit] */ }
    else if (in_fmt == GS_RADIX_PACKED) hipLaunchKernelGGL((k_radix_hist<true, NW>), G, B, 0, st, (const uint32_t *)in, n_ptr, shift, bits, ctx->hist);
    else hipLaunchKernelGGL((k_radix_hist<false, NW>), G, B, 0, st, (const uint32_t *)in, n_ptr, shift, bits, ctx->hist);
    hipLaunchKernelGGL((k_radix_scan<NW>), dim3(gs_div_up(gs_radix_row_stride(1u << bits), 16u)), dim3(64 * NW), 0, st, ctx->hist, n_ptr, CH, bits, totals);
#define GS_SCATTER(I, O) do { if (bits <= 7) hipLaunchKernelGGL((k_radix_scatter<I, O, 128, NW>), G, B, 0, st, in, out, n_ptr, shift, bits, zero_key, ctx->hist, totals, idx_bits, count_out, fill_to); \
                              else if (bits == 8) hipLaunchKernelGGL((k_radix_scatter<I, O, 256, NW>), G, B, 0, st, in, out, n_ptr, shift, bits, zero_key, ctx->hist, totals, idx_bits, count_out, fill_to); \
                              else hipLaunchKernelGGL((k_radix_scatter<I, O, GS_RADIX_MAX_BINS, NW>), G, B, 0, st, in, out, n_ptr, shift, bits, zero_key, ctx->hist, totals, idx_bits, count_out, fill_to); } while (0)
    if (in_fmt == GS_RADIX_PACKED && out_fmt == GS_RADIX_PACKED) GS_SCATTER(GS_RADIX_PACKED, GS_RADIX_PACKED);
    else if (in_fmt == GS_RADIX_PACKED && out_fmt == GS_RADIX_KEYS) GS_SCATTER(GS_RADIX_PACKED, GS_RADIX_KEYS);
    else if (in_fmt == GS_RADIX_KEYS && out_fmt == GS_RADIX_PACKED) GS_SCATTER(GS_RADIX_KEYS, GS_RADIX_PACKED);
    else if (in_fmt == GS_RADIX_KEYS && out_fmt == GS_RADIX_KEYS) GS_SCATTER(GS_RADIX_KEYS, GS_RADIX_KEYS);
    else if (in_fmt == GS_RADIX_KEYS && out_fmt == GS_RADIX_KEYIDX) GS_SCATTER(GS_RADIX_KEYS, GS_RADIX_KEYIDX);
    else if (in_fmt == GS_RADIX_KEYIDX && out_fmt == GS_RADIX_KEYS) GS_SCATTER(GS_RADIX_KEYIDX, GS_RADIX_KEYS);
    else if (in_fmt == GS_RADIX_PACKED && out_fmt == GS_RADIX_KEYIDX) GS_SCATTER(GS_RADIX_PACKED, GS_RADIX_KEYIDX);
    else { snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "radix pass: unsupported record formats %d -> %d", in_fmt, out_fmt); return GS_E_BADARG; }
#undef GS_SCATTER
    GS_HIP(hipGetLastError());
    return GS_OK;
}

template <bool PACKED, int NW> GS_BODY(F_hist, k_radix_hist_body<PACKED, NW>);
template <int NW> GS_BODY(F_scan, k_radix_scan_body<NW>);
template <int I, int O, int B, int NW> GS_BODY(F_scatter, k_radix_scatter_body<I, O, B, NW>);

// the same pass for two frames in one launch per kernel (GS_OPT_FRAME_BATCH): S[k] supplies histogram rows and totals of frame k
template <int NW>
int launch_pass2(gs_ctx *const S[2], const void *const in[2], int in_fmt, void *const out[2], int out_fmt, const uint32_t *const n_ptr[2],
                 uint32_t hint_n, int shift, int bits, bool have_hist, uint32_t zero_key, int idx_bits, uint32_t *const count_out[2],
                 const uint32_t *const fill_to[2])
{
    constexpr uint32_t CH = 64 * NW * 8;
    constexpr int NT = 64 * NW;
    gs_ctx *ctx = S[0];
    const uint32_t g = grid_for(hint_n, CH);
    hipStream_t st = ctx->stream;
    if (!have_hist) {
        if (in_fmt == GS_RADIX_PACKED) { typedef F_hist<true, NW> F;
            gs_twin<F, NT>(g, st, gs_pack_make((const uint32_t *)in[0], n_ptr[0], shift, bits, S[0]->hist), gs_pack_make((const uint32_t *)in[1], n_ptr[1], shift, bits, S[1]->hist)); }
        else { typedef F_hist<false, NW> F;
            gs_twin<F, NT>(g, st, gs_pack_make((const uint32_t *)in[0], n_ptr[0], shift, bits, S[0]->hist), gs_pack_make((const uint32_t *)in[1], n_ptr[1], shift, bits, S[1]->hist)); }
    }
    { typedef F_scan<NW> F;
      gs_twin<F, NT>(gs_div_up(gs_radix_row_stride(1u << bits), 16u), st, gs_pack_make(S[0]->hist, n_ptr[0], CH, bits, S[0]->radix_aux),
                        gs_pack_make(S[1]->hist, n_ptr[1], CH, bits, S[1]->radix_aux)); }
#define GS_SCATTER2_B(I, O, B) do { typedef F_scatter<I, O, B, NW> F;                     \
        gs_twin<F, NT>(g, st, gs_pack_make(in[0], out[0], n_ptr[0], shift, bits, zero_key, (const uint32_t *)S[0]->hist, (const uint32_t *)S[0]->radix_aux,  \
                                              idx_bits, count_out[0], fill_to[0]),                                                                               \
                          gs_pack_make(in[1], out[1], n_ptr[1], shift, bits, zero_key, (const uint32_t *)S[1]->hist, (const uint32_t *)S[1]->radix_aux,         \
                                       idx_bits, count_out[1], fill_to[1])); } while (0)
#define GS_SCATTER2(I, O) do { if (bits <= 7) GS_SCATTER2_B(I, O, 128); else if (bits == 8) GS_SCATTER2_B(I, O, 256); else GS_SCATTER2_B(I, O, GS_RADIX_MAX_BINS); } while (0)
    if (in_fmt == GS_RADIX_PACKED && out_fmt == GS_RADIX_PACKED) GS_SCATTER2(GS_RADIX_PACKED, GS_RADIX_PACKED);
    else if (in_fmt == GS_RADIX_KEYS && out_fmt == GS_RADIX_KEYIDX) GS_SCATTER2(GS_RADIX_KEYS, GS_RADIX_KEYIDX);
    else if (in_fmt == GS_RADIX_KEYIDX && out_fmt == GS_RADIX_KEYS) GS_SCATTER2(GS_RADIX_KEYIDX, GS_RADIX_KEYS);
    else if (in_fmt == GS_RADIX_KEYS && out_fmt == GS_RADIX_PACKED) GS_SCATTER2(GS_RADIX_KEYS, GS_RADIX_PACKED);
    else if (in_fmt == GS_RADIX_PACKED && out_fmt == GS_RADIX_KEYS) GS_SCATTER2(GS_RADIX_PACKED, GS_RADIX_KEYS);
    else if (in_fmt == GS_RADIX_PACKED && out_fmt == GS_RADIX_KEYIDX) GS_SCATTER2(GS_RADIX_PACKED, GS_RADIX_KEYIDX);
    else { snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "radix pass: unsupported record formats %d -> %d", in_fmt, out_fmt); return GS_E_BADARG; }
#undef GS_SCATTER2
#undef GS_SCATTER2_B
    GS_HIP(hipGetLastError());
    return GS_OK;
}

}  // namespace

int gs_launch_radix_pass2(gs_ctx *const S[2], const void *const in[2], int in_fmt, void *const out[2], int out_fmt, const uint32_t *const n_ptr[2],
                          uint32_t max_n, uint32_t hint_n, int shift, int bits, bool have_hist, uint32_t zero_key, int idx_bits,
                          uint32_t *const count_out[2], const uint32_t *const fill_to[2])
{
    gs_ctx *ctx = S[0];
    if (bits < 1 || bits > 9) { snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "radix pass: %d-bit digit (1..9 supported)", bits); return GS_E_BADARG; }
    if (hint_n > max_n || hint_n == 0) hint_n = max_n;
    return gs_radix_chunk(hint_n) == GS_CHUNK_L ? launch_pass2<8>(S, in, in_fmt, out, out_fmt, n_ptr, hint_n, shift, bits, have_hist, zero_key, idx_bits, count_out, fill_to)
                                                : launch_pass2<4>(S, in, in_fmt, out, out_fmt, n_ptr, hint_n, shift, bits, have_hist, zero_key, idx_bits, count_out, fill_to);
}

int gs_launch_msd_sort(gs_ctx *ctx, uint32_t n, uint32_t tail_req)
{
    const bool near = tail_req != 0u;
    // k_seg_sort reads whole blocks: up to GS_SEG_B words past a segment's end, i.e. up to n + GS_SEG_B - 1 words into `rec` (= kv_b, 2 x scratch_cap words)
    if ((size_t)ctx->scratch_cap * 2 < (size_t)n + GS_SEG_B) { snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "msd sort: scratch of %zu splats is too small for %u records", ctx->scratch_cap, n); return GS_E_STATE; }
    const uint32_t chunk = gs_radix_chunk(n), g = grid_for(n, chunk);
    hipStream_t st = ctx->stream;
    uint32_t *rec = reinterpret_cast<uint32_t *>(ctx->kv_b);
    if (chunk == GS_CHUNK_L) hipLaunchKernelGGL((k_msd_scatter<8>), dim3(g), dim3(512), 0, st, (const uint32_t *)ctx->key_a, rec, (const uint32_t *)&ctx->ctl->n_total,
                                                (const uint32_t *)ctx->hist, (const uint32_t *)ctx->msd_grp, &ctx->ctl->n_sorted, ctx->msd_tab, tail_req, ctx->ctl, n);
    else hipLaunchKernelGGL((k_msd_scatter<4>), dim3(g), dim3(256), 0, st, (const uint32_t *)ctx->key_a, rec, (const uint32_t *)&ctx->ctl->n_total,
                            (const uint32_t *)ctx->hist, (const uint32_t *)ctx->msd_grp, &ctx->ctl->n_sorted, ctx->msd_tab, tail_req, ctx->ctl, n);
    hipLaunchKernelGGL((k_seg_sort<GS_SEG_NW, GS_SEG_IPT>), dim3(GS_SEG_GRID), dim3(64 * GS_SEG_NW), 0, st, (const uint32_t *)rec, ctx->val_a, (const uint32_t *)ctx->msd_tab,
                       near ? (const uint32_t *)nullptr : (const uint32_t *)&ctx->ctl->n_kept);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

int gs_launch_msd_sort2(gs_ctx *const S[2], uint32_t n, const uint32_t tail_req[2])
{
    gs_ctx *ctx = S[0];
    for (int k = 0; k < 2; k++) if ((size_t)S[k]->scratch_cap * 2 < (size_t)n + GS_SEG_B) { snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "msd sort: scratch of %zu splats is too small for %u records", S[k]->scratch_cap, n); return GS_E_STATE; }
    const uint32_t chunk = gs_radix_chunk(n), g = grid_for(n, chunk);
    hipStream_t st = ctx->stream;
    uint32_t *rec[2] = { reinterpret_cast<uint32_t *>(S[0]->kv_b), reinterpret_cast<uint32_t *>(S[1]->kv_b) };
#define GS_MSD_SC(NW) gs_twin_w<F_msd_scatter<NW>, 64 * NW, 4>(g, st,                                                                                      \
        gs_pack_make((const uint32_t *)S[0]->key_a, rec[0], (const uint32_t *)&S[0]->ctl->n_total, (const uint32_t *)S[0]->hist, (const uint32_t *)S[0]->msd_grp, &S[0]->ctl->n_sorted, S[0]->msd_tab, tail_req[0], S[0]->ctl, n), \
        gs_pack_make((const uint32_t *)S[1]->key_a, rec[1], (const uint32_t *)&S[1]->ctl->n_total, (const uint32_t *)S[1]->hist, (const uint32_t *)S[1]->msd_grp, &S[1]->ctl->n_sorted, S[1]->msd_tab, tail_req[1], S[1]->ctl, n))
    if (chunk == GS_CHUNK_L) GS_MSD_SC(8); else GS_MSD_SC(4);
#undef GS_MSD_SC
    typedef F_seg_sort<GS_SEG_NW, GS_SEG_IPT> FS;
    gs_twin_w<FS, 64 * GS_SEG_NW, 4>(GS_SEG_GRID, st,
        gs_pack_make((const uint32_t *)rec[0], S[0]->val_a, (const uint32_t *)S[0]->msd_tab, tail_req[0] ? (const uint32_t *)nullptr : (const uint32_t *)&S[0]->ctl->n_kept),
        gs_pack_make((const uint32_t *)rec[1], S[1]->val_a, (const uint32_t *)S[1]->msd_tab, tail_req[1] ? (const uint32_t *)nullptr : (const uint32_t *)&S[1]->ctl->n_kept));
    GS_HIP(hipGetLastError());
    return GS_OK;
}

uint32_t gs_radix_chunk(uint32_t hint_n) { return hint_n > GS_RADIX_LARGE_N ? GS_CHUNK_L : GS_CHUNK_S; }
uint32_t gs_radix_grid(uint32_t hint_n) { return grid_for(hint_n, gs_radix_chunk(hint_n)); }

int gs_launch_radix_pass(gs_ctx *ctx, const void *in, int in_fmt, void *out, int out_fmt, const uint32_t *n_ptr,
                         uint32_t max_n, uint32_t hint_n, int shift, int bits, bool have_hist, uint32_t zero_key, int idx_bits,
                         uint32_t *count_out, const uint32_t *fill_to)
{
    if (bits < 1 || bits > 9) { snprintf(GS_ERRBUF(ctx), GS_ERRLEN, "radix pass: %d-bit digit (1..9 supported)", bits); return GS_E_BADARG; }
    if (hint_n > max_n || hint_n == 0) hint_n = max_n;
    // the geometry is a matter of speed only: both forms are exact for any *n_ptr <= max_n (a producer that pre-filled the
    // histogram rows used gs_radix_chunk(hint_n) as well)
    return gs_radix_chunk(hint_n) == GS_CHUNK_L ? launch_pass<8>(ctx, in, in_fmt, out, out_fmt, n_ptr, hint_n, shift, bits, have_hist, zero_key, idx_bits, count_out, fill_to)
                                                : launch_pass<4>(ctx, in, in_fmt, out, out_fmt, n_ptr, hint_n, shift, bits, have_hist, zero_key, idx_bits, count_out, fill_to);
}

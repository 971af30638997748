// gs_prims.hip -- stable LSD radix pass for gfx950 (wave64): histogram, per-digit row scan, LDS-reordered scatter.
//
// The kernels read their problem size from device memory (GsControl), so no stage of the frame
// needs a host round trip.  They are HBM/L2-streaming kernels: 256-thread workgroups, 2048 items per
// workgroup pass, 16-byte vector loads where the access pattern allows, LDS digit histograms, and
// LDS match words / wavefront ballots for the stable in-wave rank (no MFMA: there is no contraction here).
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

// exclusive scan of one value per thread across the 256-thread workgroup; *total = workgroup sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_wave /*[4]*/, uint32_t *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(v, lane);
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t s = s_wave[k]; if (k < w) base += s; tot += s; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// ---------------------------------------------------------------- radix pass

// Per-chunk digit histogram -> hist[digit * nchunks + chunk] (digit-major: each digit's row is contiguous for
// k_radix_rowscan).  PACKED: keys are the .x of (key,val) uint2 records.
template <bool PACKED>
__global__ __launch_bounds__(GS_BLOCK) void k_radix_hist(const uint32_t *__restrict__ keys, const uint32_t *n_ptr, int shift,
                                                         int bits, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t s_hist[GS_RADIX_MAX_BINS];
    const uint32_t n = *n_ptr;
    const uint32_t nchunks = (n + GS_CHUNK - 1) / GS_CHUNK;
    const uint32_t nbins = 1u << bits, mask = nbins - 1;
    for (uint32_t v = blockIdx.x; v < ((nchunks + 7u) & ~7u); v += gridDim.x) {
        uint32_t c;
        if (!gs_xcd_chunk(v, nchunks, c)) continue;
        for (uint32_t d = threadIdx.x; d < nbins; d += GS_BLOCK) s_hist[d] = 0;
        __syncthreads();
        uint32_t kk[GS_IPT];                                         // all loads first: their latencies overlap instead of adding up
#pragma unroll
        for (int r = 0; r < GS_IPT; r++) {
            const uint32_t i = c * GS_CHUNK + r * GS_BLOCK + threadIdx.x;
            kk[r] = i < n ? keys[PACKED ? 2 * (size_t)i : i] : 0u;
        }
#pragma unroll
        for (int r = 0; r < GS_IPT; r++) {
            const uint32_t i = c * GS_CHUNK + r * GS_BLOCK + threadIdx.x;
            if (i < n) atomicAdd(&s_hist[(kk[r] >> shift) & mask], 1u);
        }
        __syncthreads();
        for (uint32_t d = threadIdx.x; d < nbins; d += GS_BLOCK) hist[d * nchunks + c] = s_hist[d];
        __syncthreads();
    }
}

// One workgroup per digit: exclusive scan of that digit's row hist[d][0..nchunks) in place (running offset of
// every chunk inside the digit's output run) and the row total -> totals[d].  Replaces a 3-kernel flat scan.
__global__ __launch_bounds__(GS_BLOCK) void k_radix_rowscan(uint32_t *__restrict__ hist, const uint32_t *n_ptr, uint32_t *__restrict__ totals)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t n = *n_ptr;
    const uint32_t nchunks = (n + GS_CHUNK - 1) / GS_CHUNK;
    uint32_t *row = hist + (size_t)blockIdx.x * nchunks;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nchunks; base += GS_SCAN_TILE) {
        const uint32_t i0 = base + threadIdx.x * 8;
        uint32_t v[8], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) { v[k] = (i0 + k < nchunks) ? row[i0 + k] : 0u; sum += v[k]; }
        uint32_t tot;
        uint32_t run = carry + block_excl_scan(sum, s_wave, &tot);
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) { if (i0 + k < nchunks) row[i0 + k] = run; run += v[k]; }
        carry += tot;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// Stable scatter.  Item order inside a chunk: wave w owns a quarter of the chunk (512 items at GS_CHUNK = 2048), processed in GS_IPT rounds
// of 64 consecutive items (lane = item % 64), so "earlier" == (wave, round, lane) lexicographic.
// Rank among equal digits: in-round via ballot match (one ballot per digit bit), across rounds via a
// wave-private LDS counter row, across waves via a 4-way prefix.  The chunk is then REORDERED IN LDS into digit
// order and written out slot by slot, so consecutive lanes store consecutive addresses inside each digit run
// (runs of ~2048/bins items) instead of 64 unrelated 8-byte stores per instruction.
// IN_FMT:  GS_RADIX_KEYS = a key array whose value is the element index, GS_RADIX_PACKED = (key,val) uint2 records,
//          GS_RADIX_KEYONLY = 4-byte records that are their own payload (the digit is a bit field of the record).
// OUT_FMT: GS_RADIX_KEYS = the value alone (last pass of an index sort), GS_RADIX_PACKED = (key,val) uint2 records (one
//          8-byte store per item), GS_RADIX_KEYONLY = the 4-byte record.
// zero_key: items whose key equals it store 0 as their value (value-only output): the depth sort uses this so that
// culled splats (key 65536, which sort behind every bucket) leave zeros in the tail of the index list.
// MAXB: bins the instantiation reserves LDS for (128 for the <= 7-bit digits of the pair sort, GS_RADIX_MAX_BINS otherwise);
// together with 4-byte LDS slots for key-only records this takes a pair-sort workgroup from 28 KiB to 11 KiB of LDS.
template <int IN_FMT, int OUT_FMT, int MAXB>
__global__ __launch_bounds__(GS_BLOCK) void k_radix_scatter(const void *__restrict__ in, void *__restrict__ out,
                                                            const uint32_t *n_ptr, int shift, int bits, uint32_t zero_key,
                                                            const uint32_t *__restrict__ hist_scanned, const uint32_t *__restrict__ totals)
{
    constexpr bool KEYONLY = IN_FMT == GS_RADIX_KEYONLY && OUT_FMT == GS_RADIX_KEYONLY;
    __shared__ uint32_t s_cnt[4][MAXB];                         // per-wave digit counts -> local slot bases
    constexpr int MATCHB = MAXB <= 256 ? MAXB : 256;            // match words cover the low 8 digit bits; a 9th bit is refined by a ballot
    __shared__ unsigned long long s_match[4][MATCHB];           // per wave and (low) digit: lanes holding it in the current round
    __shared__ uint32_t s_dbase[MAXB];                          // start of every digit's output run (whole array)
    __shared__ uint32_t s_gb[MAXB];                             // global position of local slot 0 of each digit (minus slot)
    __shared__ uint32_t s_k[GS_CHUNK];                          // the chunk in digit order: keys ...
    __shared__ uint32_t s_v[KEYONLY ? 1 : GS_CHUNK];            // ... and values (not for key-only records)
    __shared__ uint32_t s_wave[4];
    const uint32_t n = *n_ptr;
    const uint32_t nchunks = (n + GS_CHUNK - 1) / GS_CHUNK;
    const uint32_t nbins = 1u << bits, mask = nbins - 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (blockIdx.x >= ((nchunks + 7u) & ~7u)) return;
    {   // exclusive scan of the <= 512 digit totals (2 per thread)
        const uint32_t d0 = threadIdx.x * 2;
        const uint32_t v0 = d0 < nbins ? totals[d0] : 0u, v1 = d0 + 1 < nbins ? totals[d0 + 1] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(v0 + v1, s_wave, &tot);
        if (d0 < nbins) s_dbase[d0] = ex;
        if (d0 + 1 < nbins) s_dbase[d0 + 1] = ex + v0;
    }
    for (uint32_t v = blockIdx.x; v < ((nchunks + 7u) & ~7u); v += gridDim.x) {
        uint32_t c;
        if (!gs_xcd_chunk(v, nchunks, c)) continue;
        for (uint32_t i = threadIdx.x; i < 4 * MAXB; i += GS_BLOCK) { (&s_cnt[0][0])[i] = 0; if (i < 4 * MATCHB) (&s_match[0][0])[i] = 0ull; }
        __syncthreads();
        uint32_t key[GS_IPT], val[GS_IPT], rank[GS_IPT];
#pragma unroll
        for (int r = 0; r < GS_IPT; r++) {                            // all loads first: their latencies overlap
            const uint32_t i = c * GS_CHUNK + w * (GS_CHUNK / 4) + r * 64 + lane;
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
        for (int r = 0; r < GS_IPT; r++) {
            const uint32_t i = c * GS_CHUNK + w * (GS_CHUNK / 4) + r * 64 + lane;
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
        {   // digit totals of the chunk -> local digit starts (exclusive scan over digits, 2 per thread)
            const uint32_t d0 = threadIdx.x * 2;
            uint32_t t0 = 0, t1 = 0;
            if (d0 < nbins) t0 = s_cnt[0][d0] + s_cnt[1][d0] + s_cnt[2][d0] + s_cnt[3][d0];
            if (d0 + 1 < nbins) t1 = s_cnt[0][d0 + 1] + s_cnt[1][d0 + 1] + s_cnt[2][d0 + 1] + s_cnt[3][d0 + 1];
            uint32_t tot;
            const uint32_t ex = block_excl_scan(t0 + t1, s_wave, &tot);      // (two barriers inside)
            chunk_items = tot;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint32_t d = d0 + k;
                if (d < nbins) {
                    const uint32_t ls = k ? ex + t0 : ex;                    // local slot of the digit's first item
                    const uint32_t c0 = s_cnt[0][d], c1 = s_cnt[1][d], c2 = s_cnt[2][d];
                    s_cnt[0][d] = ls; s_cnt[1][d] = ls + c0; s_cnt[2][d] = ls + c0 + c1; s_cnt[3][d] = ls + c0 + c1 + c2;
                    s_gb[d] = s_dbase[d] + hist_scanned[d * nchunks + c] - ls;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < GS_IPT; r++) {
            const uint32_t i = c * GS_CHUNK + w * (GS_CHUNK / 4) + r * 64 + lane;
            if (i < n && (IN_FMT != GS_RADIX_KEYS || key[r] != GS_RADIX_SKIP)) {
                const uint32_t d = (key[r] >> shift) & mask;
                s_k[s_cnt[w][d] + rank[r]] = key[r];
                if (!KEYONLY) s_v[s_cnt[w][d] + rank[r]] = val[r];
            }
        }
        __syncthreads();
        const uint32_t items = chunk_items;
#pragma unroll
        for (int r = 0; r < GS_IPT; r++) {
            const uint32_t slot = r * GS_BLOCK + threadIdx.x;
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
}

uint32_t grid_for(uint32_t max_items)
{
    uint32_t g = gs_div_up(max_items, GS_CHUNK);
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (g + 7u) & ~7u;                                       // a multiple of 8: workgroup index mod 8 = XCD (gs_xcd_chunk)
}

}  // namespace

uint32_t gs_radix_grid(uint32_t max_n) { return grid_for(max_n); }

int gs_launch_radix_pass(gs_ctx *ctx, const void *in, int in_fmt, void *out, int out_fmt, const uint32_t *n_ptr,
                         uint32_t max_n, int shift, int bits, bool have_hist, uint32_t zero_key)
{
    const uint32_t g = grid_for(max_n);
    uint32_t *totals = ctx->spine;                               // 2^bits words; no generic scan is in flight here
    const dim3 G(g), B(GS_BLOCK);
    hipStream_t st = ctx->stream;
    if (have_hist) { /* the producer of `in` already wrote hist[digit][chunk] */ }
    else if (in_fmt == GS_RADIX_PACKED) hipLaunchKernelGGL(k_radix_hist<true>, G, B, 0, st, (const uint32_t *)in, n_ptr, shift, bits, ctx->hist);
    else hipLaunchKernelGGL(k_radix_hist<false>, G, B, 0, st, (const uint32_t *)in, n_ptr, shift, bits, ctx->hist);
    hipLaunchKernelGGL(k_radix_rowscan, dim3(1u << bits), B, 0, st, ctx->hist, n_ptr, totals);
#define GS_SCATTER(I, O) do { if (bits <= 7) hipLaunchKernelGGL((k_radix_scatter<I, O, 128>), G, B, 0, st, in, out, n_ptr, shift, bits, zero_key, ctx->hist, totals); \
                              else if (bits == 8) hipLaunchKernelGGL((k_radix_scatter<I, O, 256>), G, B, 0, st, in, out, n_ptr, shift, bits, zero_key, ctx->hist, totals); \
                              else hipLaunchKernelGGL((k_radix_scatter<I, O, GS_RADIX_MAX_BINS>), G, B, 0, st, in, out, n_ptr, shift, bits, zero_key, ctx->hist, totals); } while (0)
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

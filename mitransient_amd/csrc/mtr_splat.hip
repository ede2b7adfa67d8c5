// mtr_splat.hip — device-side PARTITION BY PIXEL in front of the row kernel of the stand-alone time-bin scatter-add
// (mtr_splat_add, variant 1, on input that does not come pixel by pixel).
//
// The reference adds every contribution with one atomic per channel wherever it lands (transient_image_block.py:79-81,
// 131-149: dr.scatter_reduce(Add) into the flat (H, W, T, C) tensor).  On gfx950 that form runs at 2 % of the HBM roof: 3 x 2^30
// f32 atomics into a 4 GiB film are bound by the L2 atomic units (20 G/s), not by bytes.  The fast form needs every
// contribution of a pixel in ONE workgroup, whose LDS row takes them with 64-bit integer atomics (k_splat_rows*).  For
// input in arbitrary order that means a sort by pixel; its traffic is what bounds the result:
//
//   k_part_hist_hi     read the pixel ids (4 B)                    -> histogram of the HIGH half of the pixel index
//   k_part_scatter_hi  read 20 B (pixel, opl, r, g, b), write 16 B -> records (pixel | bin << bits, r, g, b) grouped by high half
//   k_part_hist_lo     read the records (16 B)                     -> histogram of the full pixel index = the run starts
//   k_part_scatter_lo  read 16 B, write 16 B                       -> records (bin, r, g, b) grouped by pixel
//   k_splat_rows_rec   read 16 B + the touched film bins           -> LDS row per pixel, one flush per pixel
//
// = 88 B per contribution + the film against the 24 algorithmic bytes: at most ~ 25 % of the HBM roof even at full streaming
// speed.  Two scatter passes because one pass cannot keep 2^18 output streams coalesced: a tile of 4096 records spreads over
// <= 512 buckets (runs of ~128 B that L2 merges), not over 262144.  The order inside a pixel does not matter (the row sums
// are order-independent fixed point), so the scatters are UNSTABLE: a tile counts its records per bucket in LDS (the LDS
// atomic's return value is the record's rank), claims the bucket's next range with one global atomic per (tile, bucket),
// and writes.
#include "mtr_kernels.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace mtr {

namespace {

constexpr uint32_t kPartMaxDigits = 2048;               // LDS counters per tile (11 bits per half: films up to 2^22 pixels)
constexpr uint32_t kDropped = 0xffffffffu;

__device__ __forceinline__ unsigned long long splat_fixed(float v)
{
    long long q = __float2ll_rn(v * 4398046511104.0f);          // 2^42 (see mtr_kernels.hip: splat_to_fixed)
    if (q == 0 && v != 0.0f) q = v > 0.0f ? 1 : -1;
    return (unsigned long long)q;
}

// RANGE GUARD of the fixed-point rows (see mtr_wavefront.hip: to_fixed; mtr_kernels.hip: splat_fixed_limit)
__device__ __forceinline__ float part_fixed_limit(uint32_t n) { return 1048576.0f / (float)(n ? n : 1u); }
__device__ __forceinline__ bool part_fixed_unsafe(float r, float g, float b, float lim)
{
    return !(fabsf(r) < lim) || !(fabsf(g) < lim) || !(fabsf(b) < lim);
}

struct PartArgs {
    mtr_splat_soa s;
    Film film;
    uint32_t npix, bits_pix, bits_lo, n_hi, n_lo;
    uint32_t *hist_hi;      // [n_hi]       counts, then (k_part_scan_hi) the claim cursors of the buckets
    uint32_t *base_hi;      // [n_hi + 1]   exclusive scan; base_hi[n_hi] = number of records kept
    uint32_t *starts;       // [npix + 1]   histogram of the pixel index, then its exclusive scan = the run starts
    uint32_t *cur_lo;       // [npix]       claim cursors of the pixels
    uint4 *rec_a, *rec_b;
};

// ---- pass 1: histogram of the high half ------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_part_hist_hi(const PartArgs a)
{
    __shared__ uint32_t s_cnt[kPartMaxDigits];
    const int tid = threadIdx.x;
    for (uint32_t k = tid; k < a.n_hi; k += kBlock) s_cnt[k] = 0u;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + tid; i < a.s.n; i += stride) {
        const uint32_t px = __builtin_nontemporal_load(a.s.pixel + i);
        // (a contribution outside the time window is dropped by the scatter pass, which reads its path length anyway; here it
        // only makes its bucket's range a little too long)
        if (px < a.npix) atomicAdd(&s_cnt[px >> a.bits_lo], 1u);
    }
    __syncthreads();
    for (uint32_t k = tid; k < a.n_hi; k += kBlock) if (s_cnt[k]) atomicAdd(a.hist_hi + k, s_cnt[k]);
}

__global__ void __launch_bounds__(kBlock) k_part_scan_hi(const PartArgs a)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {          // <= 2048 entries
        uint32_t acc = 0;
        for (uint32_t k = 0; k < a.n_hi; ++k) { const uint32_t c = a.hist_hi[k]; a.base_hi[k] = acc; a.hist_hi[k] = acc; acc += c; }
        a.base_hi[a.n_hi] = acc;
    }
}

// ---- the tile step of both scatters ---------------------------------------------------------------------------------------
// A tile's records are REORDERED BY DIGIT IN LDS before they leave: every thread ranks its records inside (tile, digit) with
// the return value of an LDS atomic, the digit counters are scanned, each digit's range of the tile is claimed from the
// digit's global cursor with one atomic, the records are placed at (digit offset + rank) in a 64 KB staging area, and the
// staging area is written out front to back — consecutive lanes hold consecutive records of one digit, so a digit's 8 records
// of a 4096-record tile leave as ONE 128-byte run in ONE store instruction.  (Round 4's first version wrote every record from
// the thread that had loaded it: the 16 records of a run left in 16 different instructions at 16 different times, with two
// thousand tiles in flight the half-written lines did not survive in the 4 MB L2 slices, and the two scatters ran at a third of
// the streaming rate: 2^30 uniform contributions 41.6 ms, of which the scatters 31.)
#ifndef MTR_SPLAT_STAGE_PER
#define MTR_SPLAT_STAGE_PER 16
#endif
constexpr uint32_t kStagePer = MTR_SPLAT_STAGE_PER;      // records per thread and tile
constexpr uint32_t kStageTile = kBlock * kStagePer;      // 4096 records = 64 KB of staging
constexpr uint32_t kScanPer = kPartMaxDigits / kBlock;   // digit counters per thread in the scan (8)

struct TileLds {
    uint4 *stage;            // [kStageTile]
    uint32_t *cnt, *off, *gbase;     // [n_digits] each
    uint32_t *wsum;          // [kBlock / 64 + 1]
};
__device__ __forceinline__ TileLds tile_lds(unsigned char *smem, uint32_t n_digits)
{
    TileLds t;
    t.stage = (uint4 *)smem;
    t.cnt = (uint32_t *)(smem + (size_t)kStageTile * 16u);
    t.off = t.cnt + n_digits; t.gbase = t.off + n_digits; t.wsum = t.gbase + n_digits;
    return t;
}
static size_t tile_lds_bytes(uint32_t n_digits) { return (size_t)kStageTile * 16u + (3u * (size_t)n_digits + 16u) * 4u; }

// exclusive scan of cnt[0 .. n_digits) into off[]; one claim per non-empty digit from cursor[] is ISSUED (the atomics' return
// values stay in claim[] — the caller stores them to gbase[] with tile_claims_store once it has nothing else to do, so that
// the round trip to the memory-side atomic unit overlaps the staging of the tile); returns the tile's record count.
// Called by the whole workgroup; contains two barriers, the first of which also closes the ranking phase.
__device__ __forceinline__ uint32_t tile_scan_claim(const TileLds &t, uint32_t n_digits, uint32_t *cursor, int tid, uint32_t (&claim)[kScanPer])
{
    __syncthreads();
    const uint32_t per = (n_digits + kBlock - 1u) / kBlock;          // <= kScanPer
    uint32_t loc[kScanPer], cnt[kScanPer], sum = 0;
#pragma unroll
    for (uint32_t e = 0; e < kScanPer; ++e) {
        const uint32_t idx = tid * per + e;
        cnt[e] = (e < per && idx < n_digits) ? t.cnt[idx] : 0u;
        loc[e] = sum; sum += cnt[e];
    }
#pragma unroll
    for (uint32_t e = 0; e < kScanPer; ++e) claim[e] = cnt[e] ? atomicAdd(cursor + (tid * per + e), cnt[e]) : 0u;
    uint32_t inc = sum;
    const uint32_t wl = tid & 63u;
#pragma unroll
    for (uint32_t o = 1; o < 64u; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (wl >= o) inc += v; }
    if (wl == 63u) t.wsum[tid >> 6] = inc;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kBlock / 64u; ++w) { const uint32_t v = t.wsum[w]; total += v; if (w < (uint32_t)(tid >> 6)) base += v; }
    const uint32_t excl = base + inc - sum;
#pragma unroll
    for (uint32_t e = 0; e < kScanPer; ++e) {
        const uint32_t idx = tid * per + e;
        if (e < per && idx < n_digits) t.off[idx] = excl + loc[e];
    }
    __syncthreads();
    return total;
}
__device__ __forceinline__ void tile_claims_store(const TileLds &t, uint32_t n_digits, int tid, const uint32_t (&claim)[kScanPer])
{
    const uint32_t per = (n_digits + kBlock - 1u) / kBlock;
#pragma unroll
    for (uint32_t e = 0; e < kScanPer; ++e) {
        const uint32_t idx = tid * per + e;
        if (e < per && idx < n_digits) t.gbase[idx] = claim[e];
    }
}

// ---- pass 2: scatter by the high half; the record gets its time bin here ----------------------------------------------
// SOFTWARE-PIPELINED over tiles: the next tile's 80 loads per thread are issued right after this tile's claims and are in
// flight while it is staged and written out (two workgroups of four waves per CU: the occupancy hides nothing).
struct HiIn { uint32_t px[kStagePer], lz[kStagePer]; float opl[kStagePer], r[kStagePer], g[kStagePer], b[kStagePer]; };
__device__ __forceinline__ void hi_load(const PartArgs &a, uint64_t tile, int tid, HiIn &in)
{
#pragma unroll
    for (uint32_t k = 0; k < kStagePer; ++k) {
        const uint64_t i = tile * kStageTile + (uint64_t)k * kBlock + tid;
        in.px[k] = kDropped; in.lz[k] = 0u; in.opl[k] = 0.0f; in.r[k] = in.g[k] = in.b[k] = 0.0f;
        if (i < a.s.n) {
            in.px[k] = __builtin_nontemporal_load(a.s.pixel + i); in.opl[k] = __builtin_nontemporal_load(a.s.opl + i);
            if (a.s.laser) in.lz[k] = __builtin_nontemporal_load(a.s.laser + i);
            in.r[k] = __builtin_nontemporal_load(a.s.r + i); in.g[k] = __builtin_nontemporal_load(a.s.g + i); in.b[k] = __builtin_nontemporal_load(a.s.b + i);
        }
    }
}
__global__ void __launch_bounds__(kBlock, 2) k_part_scatter_hi(const PartArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const TileLds t = tile_lds(smem, a.n_hi);
    const int tid = threadIdx.x;
    const uint32_t pix_mask = (1u << a.bits_pix) - 1u;
    const uint64_t n_tiles = (a.s.n + kStageTile - 1) / kStageTile;
    HiIn cur;
    hi_load(a, blockIdx.x, tid, cur);
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        for (uint32_t k = tid; k < a.n_hi; k += kBlock) t.cnt[k] = 0u;
        __syncthreads();
        uint32_t key[kStagePer], rank[kStagePer];
#pragma unroll
        for (uint32_t k = 0; k < kStagePer; ++k) {
            key[k] = kDropped; rank[k] = 0u;
            const int32_t bin = film_row_bin(a.film, cur.opl[k], cur.lz[k]);
            if (cur.px[k] < a.npix && bin >= 0) {
                key[k] = cur.px[k] | ((uint32_t)bin << a.bits_pix);
                rank[k] = atomicAdd(&t.cnt[cur.px[k] >> a.bits_lo], 1u);      // the LDS atomic's return value: rank inside (tile, bucket)
            }
        }
        uint32_t claim[kScanPer];
        const uint32_t total = tile_scan_claim(t, a.n_hi, a.hist_hi, tid, claim);
        HiIn nxt;
        hi_load(a, tile + gridDim.x, tid, nxt);           // (past the last tile: every index is out of range, nothing is loaded)
#pragma unroll
        for (uint32_t k = 0; k < kStagePer; ++k)
            if (key[k] != kDropped)
                t.stage[t.off[(key[k] & pix_mask) >> a.bits_lo] + rank[k]] = make_uint4(key[k], __float_as_uint(cur.r[k]), __float_as_uint(cur.g[k]), __float_as_uint(cur.b[k]));
        tile_claims_store(t, a.n_hi, tid, claim);
        __syncthreads();
        for (uint32_t j = tid; j < total; j += kBlock) {
            const uint4 r = t.stage[j];
            const uint32_t d = (r.x & pix_mask) >> a.bits_lo;
            a.rec_a[t.gbase[d] + (j - t.off[d])] = r;
        }
        __syncthreads();
        cur = nxt;
    }
}

// ---- pass 3: histogram of the full pixel index over the bucket-grouped records ------------------------------------------
// (a bucket's range was sized with the contributions outside the time window still in it: the unused tail of a range holds
// no record — the cursors in hist_hi say where each bucket ends)
__global__ void __launch_bounds__(kBlock) k_part_hist_lo(const PartArgs a)
{
    __shared__ uint32_t s_cnt[kPartMaxDigits];
    const int tid = threadIdx.x;
    const uint32_t lo_mask = a.n_lo - 1u, pix_mask = (1u << a.bits_pix) - 1u;
    for (uint32_t hi = blockIdx.y; hi < a.n_hi; hi += gridDim.y) {
        const uint32_t b0 = a.base_hi[hi], b1 = a.hist_hi[hi];             // [start, end of what was written)
        for (uint32_t k = tid; k < a.n_lo; k += kBlock) s_cnt[k] = 0u;
        __syncthreads();
        for (uint64_t i = (uint64_t)b0 + (uint64_t)blockIdx.x * kBlock + tid; i < b1; i += (uint64_t)gridDim.x * kBlock)
            atomicAdd(&s_cnt[a.rec_a[i].x & pix_mask & lo_mask], 1u);
        __syncthreads();
        for (uint32_t k = tid; k < a.n_lo; k += kBlock) {
            const uint32_t px = (hi << a.bits_lo) | k;
            if (s_cnt[k] && px < a.npix) atomicAdd(a.starts + px, s_cnt[k]);
        }
        __syncthreads();
    }
}

// exclusive scan of starts[0 .. npix] in place, cursors = starts.  Two launches of npix / 4096 workgroups: block sums, then every
// workgroup adds up the sums of the blocks before it (at most 1024 values) and scans its own 4096 entries.  (The first
// version scanned the 2^18 entries in ONE workgroup: 0.59 ms, as long as the histogram pass over 2^28 records.)
constexpr uint32_t kScanBlock = kBlock * 16u;
__global__ void __launch_bounds__(kBlock) k_part_scan_lo_sums(const PartArgs a, uint32_t *block_sums)
{
    __shared__ uint32_t s_sum[kBlock / 64];
    const int tid = threadIdx.x;
    const uint32_t n = a.npix + 1u;
    const uint32_t lo = blockIdx.x * kScanBlock + (uint32_t)tid * 16u;
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) if (lo + k < n) acc += a.starts[lo + k];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((tid & 63) == 0) s_sum[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) { uint32_t t = 0; for (uint32_t w = 0; w < kBlock / 64; ++w) t += s_sum[w]; block_sums[blockIdx.x] = t; }
}
__global__ void __launch_bounds__(kBlock) k_part_scan_lo(const PartArgs a, const uint32_t *block_sums)
{
    __shared__ uint32_t s_sum[kBlock / 64 + 1];
    const int tid = threadIdx.x;
    const uint32_t n = a.npix + 1u;
    // sum of the blocks before this one
    uint32_t before = 0;
    for (uint32_t k = tid; k < blockIdx.x; k += kBlock) before += block_sums[k];
    for (int o = 32; o > 0; o >>= 1) before += __shfl_down(before, o);
    if ((tid & 63) == 0) s_sum[tid >> 6] = before;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < kBlock / 64; ++w) base += s_sum[w];
    __syncthreads();
    const uint32_t lo = blockIdx.x * kScanBlock + (uint32_t)tid * 16u;
    uint32_t c[16], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) { c[k] = (lo + k < n) ? a.starts[lo + k] : 0u; sum += c[k]; }
    uint32_t inc = sum;
    const uint32_t wl = tid & 63u;
#pragma unroll
    for (uint32_t o = 1; o < 64u; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (wl >= o) inc += v; }
    if (wl == 63u) s_sum[tid >> 6] = inc;
    __syncthreads();
    for (uint32_t w = 0; w < (uint32_t)(tid >> 6); ++w) base += s_sum[w];
    uint32_t acc = base + inc - sum;
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) {
        if (lo + k < n) { a.starts[lo + k] = acc; if (lo + k < a.npix) a.cur_lo[lo + k] = acc; }
        acc += c[k];
    }
}

// ---- pass 4: scatter by the low half inside every bucket (the same tile step; the digit is the pixel inside the bucket) ------
struct LoIn { uint4 rec[kStagePer]; };
__device__ __forceinline__ void lo_load(const PartArgs &a, uint32_t b0, uint32_t b1, uint32_t tile, int tid, LoIn &in)
{
#pragma unroll
    for (uint32_t k = 0; k < kStagePer; ++k) {
        const uint64_t i = (uint64_t)b0 + (uint64_t)tile * kStageTile + (uint64_t)k * kBlock + tid;
        in.rec[k] = make_uint4(kDropped, 0u, 0u, 0u);
        if (i < b1) in.rec[k] = nt_load(a.rec_a + i);
    }
}
__global__ void __launch_bounds__(kBlock, 2) k_part_scatter_lo(const PartArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const TileLds t = tile_lds(smem, a.n_lo);
    const int tid = threadIdx.x;
    const uint32_t lo_mask = a.n_lo - 1u;
    for (uint32_t hi = blockIdx.y; hi < a.n_hi; hi += gridDim.y) {
        const uint32_t b0 = a.base_hi[hi], b1 = a.hist_hi[hi];
        const uint32_t n_tiles = (b1 - b0 + kStageTile - 1u) / kStageTile;
        LoIn cur;
        lo_load(a, b0, b1, blockIdx.x, tid, cur);
        for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
            for (uint32_t k = tid; k < a.n_lo; k += kBlock) t.cnt[k] = 0u;
            __syncthreads();
            uint32_t rank[kStagePer];
#pragma unroll
            for (uint32_t k = 0; k < kStagePer; ++k) {
                rank[k] = 0u;
                if (cur.rec[k].x != kDropped) rank[k] = atomicAdd(&t.cnt[cur.rec[k].x & lo_mask], 1u);
            }
            uint32_t claim[kScanPer];
            const uint32_t total = tile_scan_claim(t, a.n_lo, a.cur_lo + ((size_t)hi << a.bits_lo), tid, claim);
            LoIn nxt;
            lo_load(a, b0, b1, tile + gridDim.x, tid, nxt);
#pragma unroll
            for (uint32_t k = 0; k < kStagePer; ++k)
                if (cur.rec[k].x != kDropped) t.stage[t.off[cur.rec[k].x & lo_mask] + rank[k]] = cur.rec[k];
            tile_claims_store(t, a.n_lo, tid, claim);
            __syncthreads();
            for (uint32_t j = tid; j < total; j += kBlock) {
                const uint4 r = t.stage[j];
                const uint32_t d = r.x & lo_mask;
                a.rec_b[t.gbase[d] + (j - t.off[d])] = make_uint4(r.x >> a.bits_pix, r.y, r.z, r.w);      // (bin, r, g, b)
            }
            __syncthreads();
            cur = nxt;
        }
    }
}

// ---- the row kernel on records: one workgroup per pixel, as k_wf_scatter ---------------------------------------------------
template <bool FIXED>
__global__ void __launch_bounds__(kBlock) k_splat_rows_rec(const PartArgs a, float *out, uint32_t film_zero, DevCounters *cnt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t s_redo;                            // (FIXED) RANGE GUARD: this pixel's row must be rebuilt in f32
    float *row = (float *)smem;
    unsigned long long *row64 = (unsigned long long *)smem;
    const uint32_t T = a.film.bins;
    const int tid = threadIdx.x;
    if (tid == 0) s_redo = 0u;
    for (uint32_t t = tid; t < 3 * T; t += kBlock) { if (FIXED) row64[t] = 0ull; else row[t] = 0.0f; }
    __syncthreads();
    uint32_t mine = 0;
    // as k_wf_scatter: eight independent 16-byte loads in flight per lane, and the first batch of the NEXT pixel is requested
    // before this pixel's row is flushed
    constexpr int kBatch = 8;
    uint4 r[kBatch];
    auto fetch = [&](uint32_t lo_, uint32_t hi_) {
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const uint32_t i = lo_ + (uint32_t)k * kBlock + tid;
            r[k] = (i < hi_) ? nt_load(a.rec_b + i) : make_uint4(kDropped, 0u, 0u, 0u);
        }
    };
    uint32_t lo_n = 0, hi_n = 0;
    if (blockIdx.x < a.npix) { lo_n = a.starts[blockIdx.x]; hi_n = a.starts[blockIdx.x + 1]; fetch(lo_n, hi_n); }
    for (uint32_t px = blockIdx.x; px < a.npix; px += gridDim.x) {
        const uint32_t lo = lo_n, hi = hi_n;
        const uint32_t px_next = px + gridDim.x;
        if (px_next < a.npix) { lo_n = a.starts[px_next]; hi_n = a.starts[px_next + 1]; }
        const float lim = part_fixed_limit(hi - lo);
        bool unsafe = false;
        for (uint32_t base = lo; base < hi || base == lo; base += kBatch * kBlock) {
            if (base != lo) fetch(base, hi);
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                if (r[k].x == kDropped) continue;
                if (FIXED) {
                    unsigned long long *p = row64 + r[k].x;
                    unsafe |= part_fixed_unsafe(__uint_as_float(r[k].y), __uint_as_float(r[k].z), __uint_as_float(r[k].w), lim);
                    atomicAdd(p, splat_fixed(__uint_as_float(r[k].y))); atomicAdd(p + T, splat_fixed(__uint_as_float(r[k].z))); atomicAdd(p + 2 * T, splat_fixed(__uint_as_float(r[k].w)));
                } else {
                    float *p = row + r[k].x;
                    __hip_atomic_fetch_add(p, __uint_as_float(r[k].y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(p + T, __uint_as_float(r[k].z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(p + 2 * T, __uint_as_float(r[k].w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                ++mine;
            }
        }
        if (px_next < a.npix) fetch(lo_n, hi_n);          // in flight across the flush below
        if (lo == hi) continue;                           // nothing landed in this row (uniform across the workgroup)
        if (FIXED && unsafe) s_redo = 1u;
        __syncthreads();
        bool as_f32 = !FIXED;
        if (FIXED && s_redo != 0u) {
            // RANGE GUARD: a value of this pixel does not fit the fixed-point row — the row again, in f32, from the records
            as_f32 = true;
            for (uint32_t t = tid; t < 6 * T; t += kBlock) row[t] = 0.0f;
            __syncthreads();
            if (tid == 0) s_redo = 0u;
            for (uint32_t i = lo + tid; i < hi; i += kBlock) {
                const uint4 q = nt_load(a.rec_b + i);
                if (q.x == kDropped) continue;
                float *p = row + q.x;
                __hip_atomic_fetch_add(p, __uint_as_float(q.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(p + T, __uint_as_float(q.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(p + 2 * T, __uint_as_float(q.w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __syncthreads();
        }
        float4 *dst = (float4 *)(out + (size_t)px * T * 4u);
        for (uint32_t t = tid; t < T; t += kBlock) {
            float vr, vg, vb; bool nz;
            if (FIXED && !as_f32) {
                const unsigned long long qr = row64[t], qg = row64[T + t], qb = row64[2 * T + t];
                nz = (qr | qg | qb) != 0ull;
                vr = __ll2float_rn((long long)qr) * 2.2737367544323206e-13f; vg = __ll2float_rn((long long)qg) * 2.2737367544323206e-13f;
                vb = __ll2float_rn((long long)qb) * 2.2737367544323206e-13f;
                if (nz) { row64[t] = 0ull; row64[T + t] = 0ull; row64[2 * T + t] = 0ull; }
            } else {
                vr = row[t]; vg = row[T + t]; vb = row[2 * T + t];
                nz = vr != 0.0f || vg != 0.0f || vb != 0.0f;
                if (nz) { row[t] = 0.0f; row[T + t] = 0.0f; row[2 * T + t] = 0.0f; }
            }
            if (film_zero) nt_store(dst + t, make_float4(vr, vg, vb, 0.0f));          // the caller vouches the film is zero: whole lines, no read
            else if (nz) { float4 v = nt_load(dst + t); v.x += vr; v.y += vg; v.z += vb; nt_store(dst + t, v); }
        }
        __syncthreads();
    }
    if (cnt && mine) atomicAdd(&cnt->splats_issued, (unsigned long long)mine);
}

} // namespace

// Can the partition path take this call?  (record key = pixel | bin << bits_pix in 32 bits; LDS counters for either half)
bool splat_partition_supported(const mtr_splat_soa &s, const Film &film)
{
    const uint64_t npix = (uint64_t)film.width * film.height;
    if (film.n_freq || s.n == 0 || s.n >= 0xffffffffull || npix < 2 || npix > (1ull << 22)) return false;
    uint32_t bits_pix = 1; while ((1ull << bits_pix) < npix) ++bits_pix;
    uint32_t bits_bin = 1; while ((1ull << bits_bin) < film.bins) ++bits_bin;
    if (bits_pix + bits_bin > 32u) return false;
    // the record key pixel | bin << bits_pix shares its value space with the sentinel kDropped = 0xffffffff: when both counts are
    // exact powers of two that fill the 32 bits, the last pixel's last bin WOULD be that key and its contribution would vanish
    if (bits_pix + bits_bin == 32u && (((uint64_t)(film.bins - 1u) << bits_pix) | (npix - 1u)) == 0xffffffffull) return false;
    return (size_t)film.bins * 12u <= 150u * 1024u;             // the row must fit LDS
}

size_t splat_partition_scratch_bytes(const mtr_splat_soa &s, const Film &film)
{
    const size_t npix = (size_t)film.width * film.height;
    return 2 * (size_t)s.n * 16u + (2 * (size_t)kPartMaxDigits + 2 * npix + npix / 4096u + 128) * 4u;
}

hipError_t launch_splat_partitioned(const mtr_splat_soa &s, const Film &film, float *film_out, bool film_zero, DevCounters *counters,
                                    void *scratch, int n_cu, hipStream_t stream)
{
    PartArgs a{};
    a.s = s; a.film = film;
    a.npix = film.width * film.height;
    a.bits_pix = 1; while ((1u << a.bits_pix) < a.npix) ++a.bits_pix;
    // (an 8 + 10 bit split — fewer cursors for the first scatter's same-address claims, 64-byte runs in the second — measured
    // the same as 9 + 9: 12.3 against 12.4 ms for 2^28; the claims are not what bounds the scatters)
    a.bits_lo = a.bits_pix / 2u;
    a.n_lo = 1u << a.bits_lo;
    a.n_hi = ((a.npix - 1u) >> a.bits_lo) + 1u;
    unsigned char *p = (unsigned char *)scratch;
    a.rec_a = (uint4 *)p; p += (size_t)s.n * 16u;
    a.rec_b = (uint4 *)p; p += (size_t)s.n * 16u;
    a.hist_hi = (uint32_t *)p; p += (size_t)kPartMaxDigits * 4u;
    a.base_hi = (uint32_t *)p; p += ((size_t)kPartMaxDigits + 16u) * 4u;
    a.starts = (uint32_t *)p; p += ((size_t)a.npix + 16u) * 4u;
    a.cur_lo = (uint32_t *)p; p += ((size_t)a.npix + 16u) * 4u;
    uint32_t *block_sums = (uint32_t *)p;                                           // [npix / 4096 + 1]
    hipError_t e = hipMemsetAsync(a.hist_hi, 0, (size_t)kPartMaxDigits * 4u, stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.starts, 0, ((size_t)a.npix + 16u) * 4u, stream);
    if (e != hipSuccess) return e;
    const uint64_t n_tiles = (s.n + kStageTile - 1) / kStageTile;
    const unsigned g_hist = (unsigned)std::min<uint64_t>((s.n + kBlock * 16ull - 1) / (kBlock * 16ull), (uint64_t)n_cu * 8u);
    hipLaunchKernelGGL(k_part_hist_hi, dim3(g_hist), dim3(kBlock), 0, stream, a);
    hipLaunchKernelGGL(k_part_scan_hi, dim3(1), dim3(kBlock), 0, stream, a);
    // the scatters hold a 64 KB staging tile per workgroup: two workgroups per CU, each thread with 16 records in flight
    const size_t lds_hi = tile_lds_bytes(a.n_hi), lds_lo = tile_lds_bytes(a.n_lo);
    const unsigned wg_cu_hi = std::max(1u, (unsigned)((160u * 1024u) / (lds_hi + 256u))), wg_cu_lo = std::max(1u, (unsigned)((160u * 1024u) / (lds_lo + 256u)));
    e = hipFuncSetAttribute((const void *)k_part_scatter_hi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hi);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_part_scatter_lo, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lo);
    if (e != hipSuccess) return e;
    const unsigned g_hi = (unsigned)std::min<uint64_t>(n_tiles, (uint64_t)n_cu * wg_cu_hi);
    hipLaunchKernelGGL(k_part_scatter_hi, dim3(g_hi), dim3(kBlock), lds_hi, stream, a);
    // per bucket: as many workgroups as an even share of the chip (buckets of uniform input are equally long)
    const unsigned per_bucket_h = std::max(1u, (unsigned)((uint64_t)n_cu * 8u / a.n_hi));
    hipLaunchKernelGGL(k_part_hist_lo, dim3(per_bucket_h, a.n_hi), dim3(kBlock), 0, stream, a);
    const unsigned n_scan = (a.npix + 1u + kScanBlock - 1u) / kScanBlock;            // <= 1025 (films up to 2^22 pixels)
    hipLaunchKernelGGL(k_part_scan_lo_sums, dim3(n_scan), dim3(kBlock), 0, stream, a, block_sums);
    hipLaunchKernelGGL(k_part_scan_lo, dim3(n_scan), dim3(kBlock), 0, stream, a, (const uint32_t *)block_sums);
    const unsigned per_bucket_s = std::max(1u, (unsigned)((uint64_t)n_cu * wg_cu_lo / a.n_hi));
    hipLaunchKernelGGL(k_part_scatter_lo, dim3(per_bucket_s, a.n_hi), dim3(kBlock), lds_lo, stream, a);
    const bool fixed = (size_t)film.bins * 24u <= 72u * 1024u;
    const size_t lds = (size_t)film.bins * (fixed ? 24u : 12u);
    int per_cu = (int)((150u * 1024u) / (lds + 64)); if (per_cu > 8) per_cu = 8; if (per_cu < 1) per_cu = 1;
    const unsigned grid = a.npix < (uint32_t)(n_cu * per_cu) ? a.npix : (unsigned)(n_cu * per_cu);
    void (*k)(const PartArgs, float *, uint32_t, DevCounters *) = fixed ? k_splat_rows_rec<true> : k_splat_rows_rec<false>;
    e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, stream, a, film_out, film_zero ? 1u : 0u, counters);
    return hipGetLastError();
}

} // namespace mtr

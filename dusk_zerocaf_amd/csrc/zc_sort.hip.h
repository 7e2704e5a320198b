// zc_sort.hip.h -- the key sort of the bucket-method MSM (zc_msm.hip.h), hand-written for gfx950.
// Not in the reference (the MSM itself is not); replaces the rocPRIM radix sort of
// rounds 1-2.
//
// What is sorted: k_msm_digits leaves, window-major, one 32-bit word per (window, point):
//     keys[w * n + i] = sign << 31 | d,   d = |digit| - 1  (0 .. 2^(c-1) - 1),  d = 2^(c-1) for a zero digit
// and the bucket sums (k_msm_runs) want the pairs (global bucket = w << (c-1) | d, point index | sign << 31)
// ordered by bucket, the zero digits behind every bucket.  Because the words are already grouped by
// window, only the c-1 digit bits need sorting: a stable LSD counting sort PER WINDOW in passes of at most
// 9 bits (two passes for c = 11 .. 19).  A pass is three steps:
//   k_msm_sort_hist    : one workgroup per column (= G consecutive tiles of 4096 keys of one window) counts
//                        the column's keys per bin in LDS -> table[window][bin][column]
//   k_scan_*           : flat exclusive scan of the table (bin-major, so the scanned entry IS the global
//                        position of the column's first key of that bin; every window holds exactly n keys,
//                        so window w's positions start at w * n by themselves)
//   k_msm_sort_scatter : one workgroup per column walks its tiles: keys into registers, per-wave histograms,
//                        stable ranks (rows of 64 keys in order, the lanes of a row by ballot matching -- an
//                        LSD sort is only correct if every pass is stable, so no rank comes from an atomic's
//                        return value), the tile is put in bin order in LDS and written out from there, so
//                        that neighbouring lanes write neighbouring pairs.
// In the last pass the zero digits take an extra bin whose table rows are ordered behind ALL windows: the
// flat scan then compacts the buckets of all windows to the front of the output and k_msm_runs never meets
// a zero digit before the end of the list.  The sort is stable throughout (equal buckets keep the point
// order), i.e. its output is exactly what the stable library sort produced: deterministic bucket sums.
#pragma once
#include "zc_kernels.hip.h"

namespace zc {

constexpr int MSM_SORT_PASS_BITS = 9;                          // at most 512 bins per pass (+ 1 in the last)
constexpr int MSM_SORT_KPT = 16;                               // keys per thread and tile of the scatter kernel
constexpr int MSM_SORT_TILE = ZC_BLOCK * MSM_SORT_KPT;         // 4096 keys
constexpr int MSM_SORT_BINS_PAD = 520;                         // >= 513
constexpr u32 MSM_SORT_NONE = 0xFFFFFFFFu;                     // no key (a key never has all bits set)

struct msm_sort_pass {
    u32 n;        // keys per window
    u32 W;        // windows
    u32 G;        // tiles per column
    u32 ncols;    // columns per window = ceil(n / (G * TILE))
    u32 shift;    // first digit bit of this pass
    u32 bits;     // digit bits of this pass
    u32 last;     // 1: the pass that takes the top bits (bin = d >> shift, zero digits -> bin 2^bits)
    u32 c;        // window width
};

ZC_DI u32 msm_sort_bins(const msm_sort_pass& p) { return (1u << p.bits) + p.last; }
ZC_DI u32 msm_sort_bin(const msm_sort_pass& p, u32 key)
{
    const u32 d = (key & 0x7FFFFFFFu) >> p.shift;
    return p.last ? d : d & ((1u << p.bits) - 1);
}
// table row order = output order: (window, bin) for the buckets, then the zero digits of every window
ZC_DI size_t msm_sort_slot(const msm_sort_pass& p, u32 w, u32 bin, u32 col)
{
    const u32 nb = 1u << p.bits;
    const size_t row = bin < nb ? (size_t)w * nb + bin : (size_t)p.W * nb + w;
    return row * p.ncols + col;
}

// exclusive scan of one value per thread over the workgroup (256 threads); `wsum` = 4 words of LDS
ZC_DI u32 block_exclusive_scan(u32 v, u32* __restrict__ wsum, u32* __restrict__ total = nullptr)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u32 incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 x = __shfl_up(incl, d);
        if (lane >= d) incl += x;
    }
    __syncthreads();                                            // wsum may still be read from the previous call
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    u32 base = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i < wv) base += wsum[i];
    }
    if (total) *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return base + incl - v;
}

// ---------------------------------------------------------------- flat exclusive scan of a u32 array
// reduce-then-scan in blocks of 4096 entries; the array is padded (zeros) to a whole number of blocks
constexpr int SCAN_BLOCK_ELEMS = ZC_BLOCK * 16;
ZC_KERNEL void k_scan_reduce(const u32* a, u32* sums)
{
    __shared__ u32 wsum[4];
    const uint4* v = reinterpret_cast<const uint4*>(a + (size_t)blockIdx.x * SCAN_BLOCK_ELEMS);
    u32 s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint4 x = v[j * ZC_BLOCK + threadIdx.x];
        s += x.x + x.y + x.z + x.w;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// one workgroup: sums[0 .. nblk) -> exclusive prefix, in place
ZC_KERNEL void k_scan_sums(u32* sums, u32 nblk)
{
    __shared__ u32 wsum[4];
    const u32 per = (nblk + ZC_BLOCK - 1) / ZC_BLOCK;
    const u32 lo = threadIdx.x * per, hi = lo + per < nblk ? lo + per : nblk;
    u32 s = 0;
    for (u32 i = lo; i < hi; i++) s += sums[i];
    u32 run = block_exclusive_scan(s, wsum);
    for (u32 i = lo; i < hi; i++) {
        const u32 x = sums[i];
        sums[i] = run;
        run += x;
    }
}
ZC_KERNEL void k_scan_apply(u32* a, const u32* sums)
{
    __shared__ u32 wsum[4];
    uint4* v = reinterpret_cast<uint4*>(a + (size_t)blockIdx.x * SCAN_BLOCK_ELEMS) + 4 * threadIdx.x;   // 16 entries per thread
    uint4 x[4];
    u32 s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        x[j] = v[j];
        s += x[j].x + x[j].y + x[j].z + x[j].w;
    }
    u32 run = block_exclusive_scan(s, wsum) + sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint4 o;
        o.x = run; run += x[j].x;
        o.y = run; run += x[j].y;
        o.z = run; run += x[j].z;
        o.w = run; run += x[j].w;
        v[j] = o;
    }
}

// ---------------------------------------------------------------- one counting-sort pass
// PAIRS: the input is the previous pass's output (key, value) pairs; otherwise the digit words of k_msm_digits
template <bool PAIRS>
ZC_DI void msm_sort_hist_body(const u32* __restrict__ in, u32* __restrict__ table, const msm_sort_pass& p)
{
    __shared__ u32 h[MSM_SORT_BINS_PAD];
    const u32 nbins = msm_sort_bins(p);
    for (u32 b = threadIdx.x; b < nbins; b += ZC_BLOCK) h[b] = 0;
    __syncthreads();
    const u32 w = blockIdx.x / p.ncols, col = blockIdx.x % p.ncols;
    const u64 lo = (u64)col * p.G * MSM_SORT_TILE;
    const u64 left = p.n - lo;
    const u32 cnt = left < (u64)p.G * MSM_SORT_TILE ? (u32)left : p.G * MSM_SORT_TILE;
    const u32* src = in + ((size_t)w * p.n + lo) * (PAIRS ? 2 : 1);
    for (u32 e0 = 0; e0 < cnt; e0 += 8 * ZC_BLOCK) {
        u32 k[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u32 e = e0 + j * ZC_BLOCK + threadIdx.x;
            k[j] = e < cnt ? src[PAIRS ? 2 * (size_t)e : (size_t)e] : MSM_SORT_NONE;
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (k[j] != MSM_SORT_NONE) atomicAdd(&h[msm_sort_bin(p, k[j])], 1u);
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < nbins; b += ZC_BLOCK) table[msm_sort_slot(p, w, b, col)] = h[b];
}
ZC_KERNEL void k_msm_sort_hist(const u32* in, u32* table, msm_sort_pass p) { msm_sort_hist_body<false>(in, table, p); }
ZC_KERNEL void k_msm_sort_hist_pairs(const u32* in, u32* table, msm_sort_pass p) { msm_sort_hist_body<true>(in, table, p); }

// `table` holds the scanned positions.  Output pairs: (key, value) as they came in, or in the last pass
// (global bucket | 0xFFFFFFFF for a zero digit, point index | sign << 31).
template <bool PAIRS>
ZC_DI void msm_sort_scatter_body(const u32* __restrict__ in, uint2* __restrict__ out, const u32* __restrict__ table, const msm_sort_pass& p)
{
    constexpr int KPT = MSM_SORT_KPT, TILE = MSM_SORT_TILE;
    __shared__ u32 cnt[4][MSM_SORT_BINS_PAD];              // per wave: histogram of the tile, then the next LDS position per bin
    __shared__ u32 gbase[MSM_SORT_BINS_PAD];               // global position of the column's next pair of each bin
    __shared__ u32 goff[MSM_SORT_BINS_PAD];                // global position - LDS position for the pairs of the current tile
    __shared__ u32 wsum[4];
    __shared__ uint2 stage[TILE];
    const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
    const u32 nbins = msm_sort_bins(p), nbits = p.bits + p.last;
    const u32 w = blockIdx.x / p.ncols, col = blockIdx.x % p.ncols;
    for (u32 b = t; b < nbins; b += ZC_BLOCK) gbase[b] = table[msm_sort_slot(p, w, b, col)];
    const u64 col_lo = (u64)col * p.G * TILE;
    for (u32 g = 0; g < p.G; g++) {
        const u64 lo = col_lo + (u64)g * TILE;
        if (lo >= p.n) break;
        const u32 count = p.n - lo < (u64)TILE ? (u32)(p.n - lo) : (u32)TILE;
        // (1) wave wv owns entries [wv * 64 KPT, (wv + 1) * 64 KPT) of the tile, row r = 64 consecutive entries
        const u32* src = in + ((size_t)w * p.n + lo) * (PAIRS ? 2 : 1);
        u32 key[KPT], val[KPT];
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const u32 e = wv * 64 * KPT + r * 64 + lane;
            if (PAIRS) {
                const uint2 kv = e < count ? reinterpret_cast<const uint2*>(src)[e] : make_uint2(MSM_SORT_NONE, 0);
                key[r] = kv.x;
                val[r] = kv.y;
            } else {
                key[r] = e < count ? src[e] : MSM_SORT_NONE;
                val[r] = (u32)lo + e;                           // first pass: the position in the window is the point index
            }
        }
        // (2) per-wave histogram of the tile
        for (u32 b = lane; b < nbins; b += 64) cnt[wv][b] = 0;
#pragma unroll
        for (int r = 0; r < KPT; r++)
            if (key[r] != MSM_SORT_NONE) atomicAdd(&cnt[wv][msm_sort_bin(p, key[r])], 1u);
        __syncthreads();
        // (3) bins -> LDS positions (exclusive scan over the bins; thread t owns bins 3t .. 3t + 2), wave by wave
        u32 c4[3][4], tot[3], s = 0;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const u32 b = 3 * t + j;
            tot[j] = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                c4[j][q] = b < nbins ? cnt[q][b] : 0;
                tot[j] += c4[j][q];
            }
            s += tot[j];
        }
        u32 start = block_exclusive_scan(s, wsum);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const u32 b = 3 * t + j;
            if (b < nbins) {
                goff[b] = gbase[b] - start;                     // modulo 2^32: positions stay below 2^32
                gbase[b] += tot[j];
                u32 q0 = start;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    cnt[q][b] = q0;
                    q0 += c4[j][q];
                }
            }
            start += tot[j];
        }
        __syncthreads();
        // (4) stable ranks: the rows of a wave in order, the lanes of a row by matching bins with ballots
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const bool valid = key[r] != MSM_SORT_NONE;
            const u32 bin = valid ? msm_sort_bin(p, key[r]) : 0;
            u64 peers = __ballot(valid);
            for (u32 b = 0; b < nbits; b++) {
                const bool bit = ((bin >> b) & 1u) != 0;
                const u64 vote = __ballot(bit);
                peers &= bit ? vote : ~vote;
            }
            const u32 rank = __popcll(peers & (((u64)1 << lane) - 1));
            if (valid) {
                const u32 pos = cnt[wv][bin] + rank;
                stage[pos] = make_uint2(key[r], val[r]);
                if (rank == 0) cnt[wv][bin] = pos + __popcll(peers);
            }
        }
        __syncthreads();
        // (5) out, in bin order: neighbouring lanes write neighbouring pairs
        for (u32 j = t; j < count; j += ZC_BLOCK) {
            uint2 kv = stage[j];
            const u32 bin = msm_sort_bin(p, kv.x);
            const u32 gpos = goff[bin] + j;
            if (p.last) {
                kv.y |= kv.x & 0x80000000u;
                kv.x = bin >> p.bits ? MSM_SORT_NONE : (w << (p.c - 1)) | (kv.x & 0x7FFFFFFFu);
            }
            out[gpos] = kv;
        }
        __syncthreads();
    }
}
ZC_KERNEL void k_msm_sort_scatter(const u32* in, uint2* out, const u32* table, msm_sort_pass p) { msm_sort_scatter_body<false>(in, out, table, p); }
ZC_KERNEL void k_msm_sort_scatter_pairs(const u32* in, uint2* out, const u32* table, msm_sort_pass p) { msm_sort_scatter_body<true>(in, out, table, p); }

}  // namespace zc

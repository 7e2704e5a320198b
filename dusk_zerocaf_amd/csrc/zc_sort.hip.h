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
//                        stable ranks (rows of 64 keys in order; the lanes of a row find their equals through
//                        a per-wave mask table in LDS plus three ballots and rank by lane number -- an LSD
//                        sort is only correct if every pass is stable, so no rank comes from an atomic's
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
constexpr int MSM_SORT_KPT = 16;                               // keys per thread and tile of the scatter kernel: tiles of 4096 keys,
constexpr int MSM_SORT_KPT_BIG = 32;                           // or of 8192 (two-word records of large batches: a bin's share of a
                                                               // tile is then a whole 128-byte line even with 512 bins)
constexpr int MSM_SORT_BINS_PAD = 514;                         // >= 513
constexpr u32 MSM_SORT_NONE = 0xFFFFFFFFu;                     // no key (a key never has all bits set)
constexpr int MSM_SORT_MATCH_BITS = 7;                         // bin bits matched through the LDS mask table when ranking

struct msm_sort_pass {
    u32 n;        // keys per window
    u32 W;        // windows
    u32 tile;     // keys per tile (256 x keys per thread of the scatter kernel)
    u32 G;        // tiles per column
    u32 ncols;    // columns per window = ceil(n / (G * TILE))
    u32 shift;    // first digit bit of this pass
    u32 bits;     // digit bits of this pass
    u32 last;     // 1: the pass that takes the top bits (bin = d >> shift, zero digits -> bin 2^bits)
    u32 c;        // window width
    u32 idx_bits; // PACKED records: bits of the point index field
    u32 w0;       // the table's first window (a sort over a GROUP of windows: W of them from w0 on, positions relative to the group's start)
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
// Three record formats travel between the passes:
//   DIGITS : the words of k_msm_digits (sign << 31 | d), the point index is the position in the window
//   PAIRS  : (sign << 31 | d, point index) -- the general intermediate, and with global bucket keys the final output
//   PACKED : sign << 31 | (d >> bits of pass 1) << idx_bits | point index, ONE word -- the intermediate of a two-pass
//            sort whenever those fields fit 32 bits (up to 2^22 points for c <= 17: config 5's per-GPU shard), which
//            halves pass 1's writes and pass 2's reads.  The low digit bits are not stored: after pass 1 an entry's
//            position says which bin it is in, and pass 2 reads them off the scanned table of pass 1.
enum { SORT_DIGITS = 0, SORT_PAIRS = 1, SORT_PACKED = 2 };

ZC_DI void lds_barrier()          // LDS-only synchronisation: outstanding global loads (the next tile's prefetch) keep flying
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int IN>
ZC_DI u32 msm_sort_bin_of(const msm_sort_pass& p, u32 word)
{
    if (IN == SORT_PACKED) return (word & 0x7FFFFFFFu) >> p.idx_bits;       // only ever the last pass of two
    return msm_sort_bin(p, word);
}

template <int IN>
ZC_DI void msm_sort_hist_body(const u32* __restrict__ in, u32* __restrict__ table, const msm_sort_pass& p)
{
    __shared__ u32 h[MSM_SORT_BINS_PAD];
    const u32 nbins = msm_sort_bins(p);
    for (u32 b = threadIdx.x; b < nbins; b += ZC_BLOCK) h[b] = 0;
    __syncthreads();
    const u32 w = blockIdx.x / p.ncols, col = blockIdx.x % p.ncols;
    const u64 lo = (u64)col * p.G * p.tile;
    const u64 left = p.n - lo;
    const u32 cnt = left < (u64)p.G * p.tile ? (u32)left : p.G * p.tile;
    constexpr int STRIDE = IN == SORT_PAIRS ? 2 : 1;
    const u32* src = in + ((size_t)w * p.n + lo) * STRIDE;
    for (u32 e0 = 0; e0 < cnt; e0 += 8 * ZC_BLOCK) {
        u32 k[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u32 e = e0 + j * ZC_BLOCK + threadIdx.x;
            k[j] = e < cnt ? src[STRIDE * (size_t)e] : MSM_SORT_NONE;
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (k[j] != MSM_SORT_NONE) atomicAdd(&h[msm_sort_bin_of<IN>(p, k[j])], 1u);
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < nbins; b += ZC_BLOCK) table[msm_sort_slot(p, w, b, col)] = h[b];
}
ZC_KERNEL void k_msm_sort_hist(const u32* in, u32* table, msm_sort_pass p) { msm_sort_hist_body<SORT_DIGITS>(in, table, p); }
ZC_KERNEL void k_msm_sort_hist_pairs(const u32* in, u32* table, msm_sort_pass p) { msm_sort_hist_body<SORT_PAIRS>(in, table, p); }
ZC_KERNEL void k_msm_sort_hist_packed(const u32* in, u32* table, msm_sort_pass p) { msm_sort_hist_body<SORT_PACKED>(in, table, p); }

template <int OUT> struct msm_sort_rec { typedef uint2 type; };
template <> struct msm_sort_rec<SORT_PACKED> { typedef u32 type; };

// `table` holds the scanned positions of this pass; `prev_table` (IN = PACKED) those of pass 1.
// In the last pass the output pairs are (global bucket | 0xFFFFFFFF for a zero digit, point index | sign << 31).
template <int IN, int OUT, int KPT, int MATCH_BITS>
ZC_DI void msm_sort_scatter_body(const u32* __restrict__ in, void* __restrict__ out_, const u32* __restrict__ table,
                                 const u32* __restrict__ prev_table, const msm_sort_pass& p)
{
    constexpr int TILE = ZC_BLOCK * KPT, MATCH = 1 << MATCH_BITS;
    typedef typename msm_sort_rec<OUT>::type rec_t;
    rec_t* out = reinterpret_cast<rec_t*>(out_);
    __shared__ u32 cnt[4][MSM_SORT_BINS_PAD];              // per wave: histogram of the tile, then the next LDS position per bin
    __shared__ u32 gbase[MSM_SORT_BINS_PAD];               // global position of the column's next record of each bin
    __shared__ u32 goff[MSM_SORT_BINS_PAD];                // global position - LDS position for the records of the current tile
    __shared__ u32 start1[IN == SORT_PACKED ? MSM_SORT_BINS_PAD : 1];   // IN = PACKED: first position of every bin of pass 1 in this window
    __shared__ u32 wsum[4];
    __shared__ unsigned long long match[4][MATCH];  // per wave: lane masks by low bin bits (all zero between rows)
    __shared__ rec_t stage[TILE];
    __shared__ unsigned short sbin[OUT == SORT_PACKED ? TILE : 1];     // OUT = PACKED: the record no longer holds this pass's bits
    const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
    const u32 nbins = msm_sort_bins(p);
    const u32 w = blockIdx.x / p.ncols, col = blockIdx.x % p.ncols;
    for (u32 b = t; b < nbins; b += ZC_BLOCK) gbase[b] = table[msm_sort_slot(p, w, b, col)];
    for (u32 b = lane; b < MATCH; b += 64) match[wv][b] = 0;
    const u32 nb1 = 1u << p.shift;                             // IN = PACKED: pass 1 sorted the low `shift` bits
    if (IN == SORT_PACKED) {
        for (u32 b = t; b <= nb1; b += ZC_BLOCK) {
            const size_t row = (size_t)w * nb1 + b;            // row W * nb1 would be the end of the list
            start1[b] = row < (size_t)p.W * nb1 ? prev_table[row * p.ncols] : p.W * p.n;
        }
    }
    constexpr int STRIDE = IN == SORT_PAIRS ? 2 : 1;
    const u64 col_lo = (u64)col * p.G * TILE;
    const u32 mine = wv * 64 * KPT + lane;                     // wave wv owns entries [wv * 64 KPT, (wv + 1) * 64 KPT) of a tile, row r = 64 consecutive entries
    u32 key[KPT], val[KPT], nkey[KPT], nval[KPT];
    auto load = [&](u64 lo, u32 (&k)[KPT], u32 (&v)[KPT]) {
        const u32 count = p.n - lo < (u64)TILE ? (u32)(p.n - lo) : (u32)TILE;
        const u32* src = in + ((size_t)w * p.n + lo) * STRIDE;
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const u32 e = mine + r * 64;
            if (IN == SORT_PAIRS) {
                const uint2 kv = e < count ? reinterpret_cast<const uint2*>(src)[e] : make_uint2(MSM_SORT_NONE, 0);
                k[r] = kv.x;
                v[r] = kv.y;
            } else {
                k[r] = e < count ? src[e] : MSM_SORT_NONE;
                v[r] = (u32)lo + e;                             // DIGITS: the position in the window is the point index
            }
        }
    };
    if (col_lo < p.n) load(col_lo, key, val);
    for (u32 g = 0; g < p.G; g++) {
        const u64 lo = col_lo + (u64)g * TILE;
        if (lo >= p.n) break;
        const u32 count = p.n - lo < (u64)TILE ? (u32)(p.n - lo) : (u32)TILE;
        const bool more = g + 1 < p.G && lo + TILE < p.n;
        if (more) load(lo + TILE, nkey, nval);                  // in flight while this tile is ranked
        // (2) per-wave histogram of the tile
        for (u32 b = lane; b < nbins; b += 64) cnt[wv][b] = 0;
#pragma unroll
        for (int r = 0; r < KPT; r++)
            if (key[r] != MSM_SORT_NONE) atomicAdd(&cnt[wv][msm_sort_bin_of<IN>(p, key[r])], 1u);
        lds_barrier();
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
        u32 incl = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 x = __shfl_up(incl, d);
            if (lane >= d) incl += x;
        }
        if (lane == 63) wsum[wv] = incl;
        lds_barrier();
        u32 start = incl - s;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (i < wv) start += wsum[i];
        }
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
        lds_barrier();
        // (4) stable ranks: the rows of a wave in order, the lanes of a row by matching bins with ballots
        u32 b1[IN == SORT_PACKED ? KPT : 1];                    // IN = PACKED: the pass-1 bin (= low digit bits) of this lane's entries
        if (IN == SORT_PACKED) {
            const u32 pos0 = w * p.n + (u32)lo + mine;
            u32 lo_b = 0, hi_b = nb1;                           // largest b with start1[b] <= pos0
            while (hi_b - lo_b > 1) {
                const u32 mid = (lo_b + hi_b) >> 1;
                if (start1[mid] <= pos0) lo_b = mid; else hi_b = mid;
            }
#pragma unroll
            for (int r = 0; r < KPT; r++) {                     // the lane's entries are 64 positions apart: walk on from bin to bin
                const u32 gpos = pos0 + r * 64;
                while (lo_b + 1 < nb1 && start1[lo_b + 1] <= gpos) lo_b++;
                b1[r] = lo_b;
            }
        }
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const bool valid = key[r] != MSM_SORT_NONE;
            const u32 bin = valid ? msm_sort_bin_of<IN>(p, key[r]) : 0;
            // peers = the lanes of this row with the same bin.  The low MATCH_BITS bin bits are matched through LDS: every lane
            // ORs its lane bit into the wave's mask table at [bin & 127], reads the mask back and clears it again (LDS
            // operations of one wave execute in order, so the three instructions need no waiting in between and the
            // table is all zero again before the next row); the remaining three or four bits by ballots.
            u64 peers = 0;
            if (valid) {
                unsigned long long* slot = &match[wv][bin & (MATCH - 1)];
                __hip_atomic_fetch_or(slot, (unsigned long long)1 << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                peers = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_store(slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
#pragma unroll
            for (u32 b = MATCH_BITS; b < MSM_SORT_PASS_BITS + 1; b++) {
                const bool bit = ((bin >> b) & 1u) != 0;
                const u64 vote = __ballot(bit);
                peers &= bit ? vote : ~vote;
            }
            const u32 rank = __popcll(peers & (((u64)1 << lane) - 1));
            if (valid) {
                const u32 pos = cnt[wv][bin] + rank;
                if (rank == 0) cnt[wv][bin] = pos + __popcll(peers);
                if (IN == SORT_PACKED) {                        // (only ever the last pass) rebuild the digit, emit the final pair
                    const u32 idx = key[r] & ((1u << p.idx_bits) - 1);
                    const u32 k = bin >> p.bits ? MSM_SORT_NONE : ((w + p.w0) << (p.c - 1)) | (bin << p.shift) | b1[r];
                    reinterpret_cast<uint2*>(stage)[pos] = make_uint2(k, idx | (key[r] & 0x80000000u));
                } else if (OUT == SORT_PACKED) {                // pass 1 of two: drop the low bits, keep sign | high bits | index
                    const u32 d = key[r] & 0x7FFFFFFFu;
                    reinterpret_cast<u32*>(stage)[pos] = (key[r] & 0x80000000u) | ((d >> p.bits) << p.idx_bits) | val[r];
                    sbin[pos] = (unsigned short)bin;
                } else if (p.last) {
                    const u32 k = bin >> p.bits ? MSM_SORT_NONE : ((w + p.w0) << (p.c - 1)) | (key[r] & 0x7FFFFFFFu);
                    reinterpret_cast<uint2*>(stage)[pos] = make_uint2(k, val[r] | (key[r] & 0x80000000u));
                } else {
                    reinterpret_cast<uint2*>(stage)[pos] = make_uint2(key[r], val[r]);
                }
            }
        }
        lds_barrier();
        // (5) out, in bin order: neighbouring lanes write neighbouring records; global position = goff[bin] + LDS position
        for (int r0 = 0; r0 < KPT; r0 += 8) {                   // eight records per thread at a time: their LDS reads overlap
            rec_t rec[8];
            u32 gp[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const u32 j = t + (r0 + r) * ZC_BLOCK;
                if (j < count) rec[r] = stage[j];
                if (OUT == SORT_PACKED) gp[r] = j < count ? sbin[j] : 0;
            }
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const u32 j = t + (r0 + r) * ZC_BLOCK;
                u32 bin = 0;
                if (OUT == SORT_PACKED) {
                    bin = gp[r];
                } else if (j < count) {
                    const u32 k = reinterpret_cast<const uint2&>(rec[r]).x;
                    if (p.last) bin = k == MSM_SORT_NONE ? (1u << p.bits) : (k & ((1u << (p.c - 1)) - 1)) >> p.shift;
                    else bin = msm_sort_bin(p, k);
                }
                gp[r] = goff[bin] + j;
            }
#pragma unroll
            for (int r = 0; r < 8; r++)
                if (t + (r0 + r) * ZC_BLOCK < count) out[gp[r]] = rec[r];
        }
        lds_barrier();
        if (more) {
#pragma unroll
            for (int r = 0; r < KPT; r++) {
                key[r] = nkey[r];
                val[r] = nval[r];
            }
        }
    }
}
#define ZC_SORT_ARGS const u32* in, uint2* out, const u32* table, msm_sort_pass p
ZC_KERNEL_3W void k_msm_sort_scatter(ZC_SORT_ARGS) { msm_sort_scatter_body<SORT_DIGITS, SORT_PAIRS, MSM_SORT_KPT, 7>(in, out, table, nullptr, p); }
ZC_KERNEL_3W void k_msm_sort_scatter_pairs(ZC_SORT_ARGS) { msm_sort_scatter_body<SORT_PAIRS, SORT_PAIRS, MSM_SORT_KPT, 7>(in, out, table, nullptr, p); }
ZC_KERNEL_2W void k_msm_sort_scatter_big(ZC_SORT_ARGS) { msm_sort_scatter_body<SORT_DIGITS, SORT_PAIRS, MSM_SORT_KPT_BIG, 6>(in, out, table, nullptr, p); }
ZC_KERNEL_2W void k_msm_sort_scatter_pairs_big(ZC_SORT_ARGS) { msm_sort_scatter_body<SORT_PAIRS, SORT_PAIRS, MSM_SORT_KPT_BIG, 6>(in, out, table, nullptr, p); }
ZC_KERNEL_3W void k_msm_sort_scatter_pack(const u32* in, u32* out, const u32* table, msm_sort_pass p) { msm_sort_scatter_body<SORT_DIGITS, SORT_PACKED, MSM_SORT_KPT, 7>(in, out, table, nullptr, p); }
ZC_KERNEL_3W void k_msm_sort_scatter_unpack(const u32* in, uint2* out, const u32* table, const u32* prev_table, msm_sort_pass p) { msm_sort_scatter_body<SORT_PACKED, SORT_PAIRS, MSM_SORT_KPT, 7>(in, out, table, prev_table, p); }
#undef ZC_SORT_ARGS

}  // namespace zc

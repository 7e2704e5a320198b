// zc_msm.cuh -- bucket-method (Pippenger) multi-scalar multiplication kernels.
// Not in the reference: sum_i k_i * P_i is specified through the reference's
// own ops (Mul<Scalar> then Add) and compared as a group element.  Pipeline per GPU:
//   1. k_msm_digits   : c-bit window digits -> (key = window << c | digit, value = point index)
//   2. rocPRIM radix sort of the n*W pairs by key (c + log2 W bits)
//   3. k_msm_bounds   : [start, end) of every bucket in the sorted list
//   0. k_msm_prepare  : points -> cached form (Y-X, Y+X, Z, 2dT), Montgomery domain, packed to
//                       one 128-byte cache line per point
//   4. k_msm_counts + rocPRIM sort of the bucket ids by population (descending), then
//      k_msm_accumulate: one lane per bucket, lanes of a wave get equally full buckets; each
//      point costs one 8-multiplication a = -1 addition against the cached form; the next
//      record is prefetched straight into LDS (global_load_lds), four waves per SIMD
//   5. k_msm_segments : running-sum reduction of each window in segments of SEG buckets
//                       (sum_seg = sum (d - lo + 1) B_d, acc_seg = sum B_d)
//   6. existing kernels: (lo - 1) * acc_seg via k_ed_scalar_mul, k_ed_add, k_ed_fold_pairs down
//      to one point per window
//   7. k_msm_window_combine : sum_w 2^(c w) S_w by Horner's rule, one quad of lanes per doubling
#pragma once
#include "zc_kernels.cuh"
#include "zc_quad.cuh"

namespace zc {

constexpr int MSM_SEG = 16;   // buckets per reduction segment

// digit w of the 260-bit scalar (5 x 52-bit limbs), c <= 16
ZC_DI u32 scalar_digit(const u64 (&l)[5], int w, int c)
{
    const int bit = w * c;
    const int idx = bit / 52, sh = bit % 52;
    if (idx >= 5) return 0;
    u64 x = (l[idx] & M52) >> sh;
    if (sh + c > 52 && idx + 1 < 5) x |= (l[idx + 1] & M52) << (52 - sh);
    return (u32)x & ((1u << c) - 1);
}

// largest scalar bit length of the shard (wave reduce, at most one atomicMax per wave): windows above it
// hold only zero digits and are not generated at all
ZC_KERNEL void k_msm_maxbits(const u64* k, int* maxbits, size_t n)
{
    const size_t i = gid();
    int bits = 0;
    if (i < n) {
        u64 l[5];
        load_scalar(l, k + 5 * i);
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const u64 x = l[j] & M52;
            if (x) bits = 52 * j + (64 - __builtin_clzll(x));
        }
    }
    bits = wave_max_i32(bits);
    // almost every wave sees the batch maximum: look before the (serialised) atomic
    if ((threadIdx.x & 63) == 0 && bits > __atomic_load_n(maxbits, __ATOMIC_RELAXED)) atomicMax(maxbits, bits);
}

ZC_KERNEL void k_msm_digits(const u64* k, u32* keys, u32* vals, size_t n, int c, int W)
{
    const size_t i = gid();
    if (i >= n) return;
    u64 l[5];
    load_scalar(l, k + 5 * i);
    for (int w = 0; w < W; w++) {
        keys[(size_t)w * n + i] = ((u32)w << c) | scalar_digit(l, w, c);
        vals[(size_t)w * n + i] = (u32)i;
    }
}

// start[key] / end[key] for every key present in the sorted list (arrays pre-zeroed)
ZC_KERNEL void k_msm_bounds(const u32* keys, u32* start, u32* end, size_t m)
{
    const size_t j = gid();
    if (j >= m) return;
    const u32 key = keys[j];
    if (j == 0 || keys[j - 1] != key) start[key] = (u32)j;
    if (j + 1 == m || keys[j + 1] != key) end[key] = (u32)(j + 1);
}

// Cached ("projective Niels") form of an input point for the bucket sums: (Y-X, Y+X, Z, 2dT),
// Montgomery domain.  MSM results are compared as group elements,
// so the bucket sums are free to use the cheaper dedicated a = -1 addition (HWCD'08 sec. 3.1,
// 8 multiplications against a cached operand) instead of the reference's 10-multiplication
// sequence; the sum is the same group element.
ZC_KERNEL void k_msm_prepare(const u64* points, u32* cached, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    niels_store(cached + 32 * i, niels_from_pt(pt_load(points + 20 * i)));
}
// population of every bucket (0 for digit 0, which carries no weight) and its id
ZC_KERNEL void k_msm_counts(const u32* start, const u32* end, u32* count, u32* ids, size_t nbuckets, int c)
{
    const size_t b = gid();
    if (b >= nbuckets) return;
    count[b] = ((b & ((1u << c) - 1)) != 0) ? end[b] - start[b] : 0;
    ids[b] = (u32)b;
}

// buckets[b] = sum of the points whose (window, digit) == b.  `order` lists the bucket ids by
// descending population so the 64 lanes of a wave run (almost) the same trip count.
// The (random) 128-byte gather of point e+1 is in flight while point e is being added, and it
// lands in LDS, not in registers: `global_load_lds_dwordx4` copies 16 bytes per lane straight into
// the wave's staging area (piece j of all 64 lanes contiguous: M0 = base + j KB), so the prefetch
// holds no VGPRs and the kernel fits four waves per SIMD (32 KB of LDS per block, 4 blocks per CU).
// A lane reads and overwrites only its own slots: lgkmcnt(0) before the next copy is issued,
// vmcnt(0) before the slots are read.
extern "C" __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))
void k_msm_accumulate(const u32* cached, const u32* vals, const u32* start, const u32* end, const u32* order,
                      u64* buckets, size_t nbuckets, int c)
{
    __shared__ uint4 stage[8 * ZC_BLOCK];
    const int lane = threadIdx.x & 63;
    uint4* base = stage + (threadIdx.x >> 6) * (8 * 64);
    const size_t j = gid();
    const bool valid = j < nbuckets;
    const size_t b = valid ? order[j] : 0;
    u32 lo = 0, hi = 0;
    if (valid && (b & ((1u << c) - 1)) != 0) {
        lo = start[b];
        hi = end[b];
    }
    auto fetch = [&](u32 idx) {
        const uint4* src = reinterpret_cast<const uint4*>(cached + 32 * (size_t)idx);
#pragma unroll
        for (int q = 0; q < 8; q++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q),
                                             (__attribute__((address_space(3))) void*)(base + q * 64), 16, 0, 0);
    };
    pt acc = pt_identity();
    if (lo < hi) fetch(vals[lo]);
    u32 inext = (lo + 1 < hi) ? vals[lo + 1] : 0;
    for (u32 e = lo; e < hi; e++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        niels cur;
        cur.ymx = unpack256(base[0 * 64 + lane], base[1 * 64 + lane]);
        cur.ypx = unpack256(base[2 * 64 + lane], base[3 * 64 + lane]);
        cur.z = unpack256(base[4 * 64 + lane], base[5 * 64 + lane]);
        cur.t2d = unpack256(base[6 * 64 + lane], base[7 * 64 + lane]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (e + 1 < hi) fetch(inext);
        inext = (e + 2 < hi) ? vals[e + 2] : 0;
        acc = pt_add_cached<true>(acc, cur);
    }
    if (valid) pt_store(buckets + 20 * b, acc);
}

// One lane per segment of MSM_SEG consecutive buckets [lo, lo + SEG) of one window:
//   acc = sum_d B_d,  sum = sum_d (d - lo + 1) B_d   (running sums from the top bucket down)
// so that  sum_d d * B_d = sum + (lo - 1) * acc.  Emits sum, acc and the scalar (lo - 1) >= 0.
ZC_KERNEL void k_msm_segments(const u64* buckets, u64* seg_sum, u64* seg_acc, u64* seg_scalar, size_t nseg_total, int c)
{
    const size_t s = gid();
    if (s >= nseg_total) return;
    const size_t first = s * MSM_SEG;                     // global bucket index of the segment start
    const u32 lo = (u32)(first & ((1u << c) - 1));        // digit value of the first bucket
    pt acc = pt_identity(), sum = pt_identity();
    // the lo == 0 segment stops above bucket 0 (digit 0 carries no weight): its running sum is
    // already sum_d d * B_d and its scalar is 0
    const int jmin = (lo == 0) ? 1 : 0;
    for (int j = MSM_SEG - 1; j >= jmin; j--) {
        acc = pt_add<true>(acc, pt_load(buckets + 20 * (first + j)));
        sum = pt_add<true>(sum, acc);
    }
    pt_store(seg_sum + 20 * s, sum);
    pt_store(seg_acc + 20 * s, acc);
    u64* k = seg_scalar + 5 * s;
    k[0] = (lo == 0) ? 0 : lo - 1;
    k[1] = 0; k[2] = 0; k[3] = 0; k[4] = 0;
}

// ---------------------------------------------------------------- window combination
// S = sum_w 2^(c w) S_w by Horner's rule: c doublings and one addition per window, about 250
// dependent doublings on ONE point, so the step is pure latency.  A quad of lanes shares each
// doubling: phase 1 squares X, Y, Z, X+Y on lanes 0..3, phase 2 forms E*F, G*H, F*G, E*H
// (dbl-2008-hwcd, a = -1), and DPP quad broadcasts hand the four results round; a doubling then
// costs one squaring plus one multiplication of latency instead of eight.  Result compared as a
// group element (zc_msm contract), so the dedicated doubling is admissible here.
ZC_DI pt pt_double_quad(const pt& p, int role)
{
    // one wave, nothing to overlap with: the independent-chain multiplier has the shorter latency
    const fe sq = mont_sqr_ilp<FP>(fe_by_role(role, p.X, p.Y, p.Z, fe_add(p.X, p.Y)));
    const fe A = quad_bcast<0>(sq), B = quad_bcast<1>(sq), ZZ = quad_bcast<2>(sq), S = quad_bcast<3>(sq);
    const fe E = fp_sub(fp_sub(S, A), B);
    const fe G = fp_sub(B, A);
    const fe F = fp_sub(fp_sub(G, ZZ), ZZ);
    const fe H = fp_sub(fp_neg(A), B);
    const fe m = mont_mul_ilp<FP>(fe_by_role(role, E, G, F, E), fe_by_role(role, F, H, G, H));
    pt r;
    r.X = quad_bcast<0>(m);
    r.Y = quad_bcast<1>(m);
    r.Z = quad_bcast<2>(m);
    r.T = quad_bcast<3>(m);
    return r;
}
// one wave; windows[w] = S_w (W points); out = sum_w 2^(c w) S_w
ZC_KERNEL void k_msm_window_combine(const u64* windows, u64* out, int W, int c)
{
    const int role = threadIdx.x & 3;
    pt Q = pt_load(windows + 20 * (size_t)(W - 1));
    for (int w = W - 2; w >= 0; w--) {
        for (int i = 0; i < c; i++) Q = pt_double_quad(Q, role);
        Q = pt_add<true>(Q, pt_load(windows + 20 * (size_t)w));
    }
    if (threadIdx.x == 0) pt_store(out, Q);
}

}  // namespace zc

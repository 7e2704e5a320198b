// zc_msm.hip.h -- bucket-method (Pippenger) multi-scalar multiplication kernels.
// Not in the reference: sum_i k_i * P_i is specified through the reference's
// own ops (Mul<Scalar> then Add) and compared as a group element.  Pipeline per GPU on the caller's stream (the
// normalisation forks onto a second stream beside the sort, the chains of the upper window groups onto a third;
// events join them), no host synchronisation anywhere:
//   1. k_msm_digits   : SIGNED c-bit window digits d in (-2^(c-1), 2^(c-1)] of the effective scalar
//                       (scalar_effective: double_and_add's termination rule), window-major:
//                       keys[w n + i] = sign << 31 | (|d| - 1), or 2^(c-1) for a zero digit.
//                       W = ceil(261 / c) windows hold any 260-bit pattern plus the last carry, so the
//                       window count never depends on the data (empty windows cost empty buckets only).
//   2. zc_sort.hip.h  : per-window LSD counting sort of the digit words -> pairs (global bucket =
//                       window << (c-1) | |d| - 1, point index | sign << 31) ordered by bucket, zero
//                       digits (key 0xFFFFFFFF) behind every bucket
//   0. k_msm_prepare  : points -> cached form, Montgomery domain.  Large batches: AFFINE records
//                       (y-x, y+x, 2dxy) as 27 limb words (gathered as 112 bytes) at a 128-byte stride (one record per cache line), one
//                       division-step inversion per lane shared by its points (Montgomery's trick), so that a bucket
//                       addition costs 7 multiplications; small batches: (Y-X, Y+X, Z, 2dT), 128 bytes, 8 multiplications
//   3. k_msm_runs     : the bucket sums as a SEGMENTED REDUCTION of the sorted list in fixed-length
//                       runs: lane j adds the T consecutive entries [jT, (j+1)T) whatever buckets they
//                       belong to, so every lane does the same work however skewed the digit
//                       distribution is (a window whose top bits are always zero has a few buckets
//                       with n / 2^k points each; one lane per bucket would serialise them).  A bucket
//                       that lies inside one run is written out directly; a bucket that crosses run
//                       boundaries leaves one partial sum per run ("edge"), and the edge list -- again
//                       sorted by key, 2 n W / T entries -- goes through k_msm_runs_edges(_quad) level by level until
//                       one lane holds it all (list lengths shrink by 4 per level).  Each point costs one mixed
//                       addition against the cached record (negated by swapping Y-X / Y+X and negating 2dT when the
//                       digit is negative); the next record is prefetched straight into LDS (global_load_lds), four
//                       waves per SIMD.  Shards of 2^21 .. 2^22 pairs launch it once per WINDOW GROUP, top windows first
//                       (msm_runs_body below; zerocaf_hip.hip: msm_on_device), and the chain of kernels behind a
//                       group's launch runs on a side stream beside the next group's bucket sums
//   4. k_msm_segments : one lane per segment of SEG buckets: running sums (acc = sum B, sum = sum (j + 1) B_{first+j}),
//                       then the product first' * acc and the final addition in the same lane
//                       (k_msm_segments_quad: four lanes per segment for launches of few segments)
//   5. k_msm_fold_groups : the segment sums of every window down to one point per window (LDS tree, two launches)
//   6. k_msm_window_combine : sum_w 2^(c w) S_w by Horner's rule, one quad of lanes per doubling, continued from the
//                       result of the window group above (the lowest group: from k_msm_shift's pre-multiplied copy of it)
#pragma once
#include "zc_kernels.hip.h"
#include "zc_quad.hip.h"
#include "zc_sort.hip.h"

// Compile-time variants of the MSM pipeline (defaults = what was measured best; DESIGN-EXPERIMENTS.md has the numbers):
#ifndef ZC_MSM_TAIL_SIDE
#define ZC_MSM_TAIL_SIDE 1           // the window groups' chains on the side stream beside the next group's bucket sums (0: in line)
#endif
#ifndef ZC_MSM_TAIL_PRIO
#define ZC_MSM_TAIL_PRIO 1           // that side stream at the highest stream priority (0: default priority)
#endif
#ifndef ZC_MSM_FOLD_QUAD
#define ZC_MSM_FOLD_QUAD 1           // the fold tree with four lanes per addition (0: one lane, 512 points per workgroup)
#endif
#ifndef ZC_MSM_SEG_QUAD
#define ZC_MSM_SEG_QUAD 16384        // four lanes per segment in launches of at most this many segments (twice as many for the lowest of several window groups)
#endif
#ifndef ZC_MSM_LOW_SEG_HALF
#define ZC_MSM_LOW_SEG_HALF 0        // 1: the lowest window group reduces its buckets in segments of half the length.  Round 6, measured and not
#endif                               // taken: 39 instead of 54 dependent additions, but twice the quads -- k_msm_segments_quad 150 -> 194 us at 2^21 pairs
#ifndef ZC_MSM_EDGES_QUAD
#define ZC_MSM_EDGES_QUAD 65536      // four lanes per run in the levels of the segmented reduction with at most this many runs (0: one lane; A/B)
#endif
#ifndef ZC_MSM_CARRY_PRESHIFT
#define ZC_MSM_CARRY_PRESHIFT 1      // the lowest group's Horner rule takes the carry of the groups above already multiplied by 2^(c nw) (0: A/B)
#endif
#ifndef ZC_MSM_GROUP_LANES
#define ZC_MSM_GROUP_LANES 17        // log2 of the lanes a window group's bucket-sum launch keeps busy
#endif
#ifndef ZC_MSM_GROUP_WGS
#define ZC_MSM_GROUP_WGS 3           // workgroups per CU of the bucket-sum launches that run beside a chain (0: no limit)
#endif
#ifndef ZC_MSM_AFF_LIMBS
#define ZC_MSM_AFF_LIMBS 1           // affine records hold the kernels' own 29-bit limbs (3 x 9 words = 108 bytes, gathered as 112) instead of three packed
#endif                               // 256-bit words (96 bytes): the bucket sums skip 3 x 9 limb extractions (64-bit shifts) per addition; same cache line
#ifndef ZC_MSM_REC_STRIDE
#define ZC_MSM_REC_STRIDE 128        // stride of the affine records: 128 = one per cache line, 96 = packed (packed 256-bit-word records only)
#endif
static_assert(ZC_MSM_REC_STRIDE == 128 || (ZC_MSM_REC_STRIDE == 96 && !ZC_MSM_AFF_LIMBS), "ZC_MSM_REC_STRIDE");
#ifndef ZC_MSM_ACC_ILP
#define ZC_MSM_ACC_ILP false   // bucket sums on the column-ordered multiplier: with fixed-length runs every wave has
                               // three neighbours to overlap with (2^24 pairs: 25.1 -> 24.7 ms against the
                               // independent-chain form, which paid off when one lane owned one bucket)
#endif

namespace zc {

// Buckets per reduction segment (one lane each): short segments keep enough lanes busy when there are few
// buckets, long ones spend fewer doublings' worth of work on the (first mod 2^(c-1)) * acc products.
// Measured (2^16 / 2^18 / 2^20 / 2^21 / 2^24 pairs, ms): 8: 1.10 / 1.50 / 2.81 / 4.53 / 24.75,
// 16: 1.17 / 1.55 / 2.77 / 4.42 / 24.40, 32: 1.30 / 1.68 / 2.93 / 4.56 / 24.05.
inline int msm_segment_buckets(size_t nbuckets)
{
    return nbuckets <= ((size_t)1 << 18) ? 8 : nbuckets >= ((size_t)1 << 21) ? 32 : 16;
}
constexpr int MSM_SCALAR_BITS = 261;   // 260-bit limb patterns + the carry of the signed recoding
constexpr int MSM_MIN_C = 5, MSM_MAX_C = 22;

// c bits of the 260-bit scalar starting at bit `bit` (5 x 52-bit limbs), c <= 32
ZC_DI u32 scalar_bits(const u64 (&l)[5], int bit, int c)
{
    const int idx = bit / 52, sh = bit % 52;
    if (idx >= 5) return 0;
    u64 x = l[idx] >> sh;
    if (sh + c > 52 && idx + 1 < 5) x |= l[idx + 1] << (52 - sh);
    return (u32)x & (u32)(((u64)1 << c) - 1);
}

ZC_KERNEL void k_msm_digits(const u64* k, u32* keys, size_t n, int c, int W)
{
    const size_t i = gid();
    if (i >= n) return;
    u64 l[5];
    load_scalar(l, k + 5 * i);
    const u32 half = 1u << (c - 1);
    u32 carry = 0;
    for (int w = 0; w < W; w++) {
        const u32 raw = scalar_bits(l, w * c, c) + carry;              // 0 .. 2^c
        carry = raw > half ? 1u : 0u;                                    // digit = raw - carry * 2^c
        const u32 mag = carry ? (1u << c) - raw : raw;                   // |digit| in 0 .. 2^(c-1)
        keys[(size_t)w * n + i] = (mag ? mag - 1 : half) | (carry << 31);
    }
}

// Cached ("projective Niels") form of an input point for the bucket sums: (Y-X, Y+X, Z, 2dT),
// Montgomery domain.  MSM results are compared as group elements,
// so the bucket sums are free to use the cheaper dedicated a = -1 addition (HWCD'08 sec. 3.1,
// 8 multiplications against a cached operand) instead of the reference's 10-multiplication
// sequence; the sum is the same group element.
ZC_KERNEL void k_msm_prepare_lane(const u64* points, u32* cached, size_t n)          // input array not 16-byte aligned
{
    const size_t i = gid();
    if (i >= n) return;
    niels_store(cached + 32 * i, niels_from_pt(pt_load(points + 20 * i)));
}
// The pass is bandwidth-bound (160 bytes in, 128 out, five multiplications): the workgroup's 256 point
// records (40 KB, contiguous) come in through LDS with coalesced 16-byte loads -- a lane reading its own
// 160-byte record from global memory touches twenty 8-byte words on two or three cache lines it shares
// with nobody in its wave -- and every lane writes one whole 128-byte line.
ZC_KERNEL void k_msm_prepare(const u64* points, u32* cached, size_t n)
{
    __shared__ __attribute__((aligned(16))) u64 sp[ZC_BLOCK * 20];
    const size_t base = (size_t)blockIdx.x * ZC_BLOCK;
    const int cnt = (int)((n - base < (size_t)ZC_BLOCK) ? (n - base) : (size_t)ZC_BLOCK);
    coop_load40<false>(sp, points + 20 * base, cnt * 4);         // a point = four 40-byte records
    __syncthreads();
    const int t = threadIdx.x;
    if (t < cnt) niels_store(cached + 32 * (base + t), niels_from_pt(pt_load(sp + 20 * t)));
}
// AFFINE records (large batches): (y - x, y + x, 2 d x y), Montgomery domain, as 3 x 9 limb words (ZC_MSM_AFF_LIMBS; rounds 3-5: as 3 x 256-bit words = 96 bytes); the
// bucket additions then skip Z Z' (pt_add_cached<., AFFINE>: 7 multiplications instead of 8) and gather 96
// bytes instead of 128.  One lane normalises the points lo, lo + stride, ... (at most c; stride = number of
// lanes, so a wave touches neighbouring records in every pass) with ONE division-step inversion: Montgomery's
// trick over their Z coordinates, the prefix products parked in the first 36 bytes of the records about to be
// written (as ed_to_affine_chunk: plain limbs serve as Montgomery residues, fp_inverse_of_register returns the
// plain inverse of the register value).  A wave whose points all have Z = 1 (decompressed or already affine
// inputs) skips the inversion.  Z = 0 (no point of the curve) takes the neutral value: garbage in, garbage out.
constexpr int MSM_AFF_WORDS = ZC_MSM_AFF_LIMBS ? 28 : 24;      // 32-bit words of a record as it is gathered (limb records: 27 used)
constexpr int MSM_AFF_PIECES = MSM_AFF_WORDS / 4;              // ... in 16-byte pieces
// All global traffic of the pass is coalesced: the workgroup's consecutive point records of a step come in through LDS with
// 16-byte loads (a lane reading its own 160-byte record from global memory issues twenty 8-byte loads on two or three cache
// lines nobody else in its wave shares), the prefix products go out and come back as one block (parked in the records about
// to be written), and the finished records leave through the same LDS buffer.
//
// WORKGROUP = ONE WAVE (round 6).  Rounds 3-5 ran this pass in 256-thread workgroups: 49 KB of LDS each, three per CU = 150 of
// the CU's 160 KB -- while they were resident no workgroup of the key sort's scatter kernels (41-51 KB) got on that CU, and the
// two sides of the MSM's front matter took turns on the chip instead of sharing it (the limiter is LDS, not
// the 86 VGPRs).  A wave stages its own 64 records (10 KB + 2.3 KB of prefixes = 12.3 KB), needs no workgroup barrier, and
// the dispatcher can place normalisation waves beside one or two scatter workgroups on every CU.
#ifndef ZC_MSM_PREP_BLOCK
#define ZC_MSM_PREP_BLOCK 64         // threads per workgroup of k_msm_prepare_affine (A/B knob: 256 = rounds 3-5)
#endif
constexpr int MSM_PREP_BLOCK = ZC_MSM_PREP_BLOCK;
ZC_DI void coop_copy16(uint4* __restrict__ dst, const uint4* __restrict__ src, int nvec)
{
    for (int v = threadIdx.x; v < nvec; v += MSM_PREP_BLOCK) dst[v] = src[v];
}
// `rec_words`: the record stride in 32-bit words -- 24 (records packed, 96 bytes: three of four straddle two 128-byte lines)
// or 32 (one record per 128-byte line, a quarter of the array unused).
// PREFETCH (round 6): a step is "load 10 KB, wait, a handful of multiplications, store" and a CU holds at most twelve such waves
// (LDS), so the waves spent half their cycles waiting for their own loads.  The NEXT step's vectors are now requested into
// registers (ten 16-byte vectors per lane + three of the prefix block) before the current step's arithmetic and moved to LDS
// at the top of the next step; the first backward step is requested before the lane's inversion.
#ifndef ZC_MSM_PREP_PREFETCH
#define ZC_MSM_PREP_PREFETCH 1       // 0: A/B build that loads every step when it needs it (rounds 3-5)
#endif
extern "C" __global__ __launch_bounds__(ZC_MSM_PREP_BLOCK) __attribute__((amdgpu_waves_per_eu(3)))
void k_msm_prepare_affine(const u64* points, u32* recs, size_t n, int c, u32 rec_words)
{
    constexpr int NT = MSM_PREP_BLOCK;
    constexpr int PV = (NT * 10 + NT - 1) / NT;                              // 16-byte vectors per lane of a block of NT point records (10)
    constexpr int QV = (NT * 9 / 4 + NT - 1) / NT;                           // ... of a block of NT prefix products (3)
    __shared__ __attribute__((aligned(16))) u64 sp[NT * 20];                // NT point records in, NT affine records out
    __shared__ __attribute__((aligned(16))) u32 spre[NT * 9];               // NT prefix products
    const int t = threadIdx.x;
    const size_t stride = (size_t)gridDim.x * NT;                            // lane g owns points g, g + stride, g + 2 stride, ...
    const size_t first = (size_t)blockIdx.x * NT;
    const fe neutral = fe_one_m<FP>();
    fe acc = neutral;
    bool all_one = true;
    int steps = 0;                                                           // steps of this workgroup (the same for all its lanes)
    auto count_at = [&](size_t base) { return (int)(n - base < (size_t)NT ? n - base : (size_t)NT); };
    // a block of point records: global -> registers (request) and registers -> LDS (land)
    u64x2 pv[PV];
    static_assert(QV == 3, "three prefix vectors per lane");
    uint4 qv0, qv1, qv2;                                                     // (named: an array here ends up in scratch memory)
    auto request_points = [&](size_t base, int cnt) {
        const u64x2* gv = reinterpret_cast<const u64x2*>(points + 20 * base);
#pragma unroll
        for (int i = 0; i < PV; i++)
            if (t + i * NT < cnt * 10) pv[i] = gv[t + i * NT];
    };
    auto land_points = [&](int cnt) {
        u64x2* lv = reinterpret_cast<u64x2*>(sp);
#pragma unroll
        for (int i = 0; i < PV; i++)
            if (t + i * NT < cnt * 10) lv[t + i * NT] = pv[i];
    };
    auto request_prefix = [&](size_t base, int cnt) {
        const uint4* gv = reinterpret_cast<const uint4*>(recs + (size_t)rec_words * base);
        const int nvec = (cnt * 9 + 3) / 4;
        if (t < nvec) qv0 = gv[t];
        if (t + NT < nvec) qv1 = gv[t + NT];
        if (t + 2 * NT < nvec) qv2 = gv[t + 2 * NT];
    };
    auto land_prefix = [&](int cnt) {
        uint4* lv = reinterpret_cast<uint4*>(spre);
        const int nvec = (cnt * 9 + 3) / 4;
        if (t < nvec) lv[t] = qv0;
        if (t + NT < nvec) lv[t + NT] = qv1;
        if (t + 2 * NT < nvec) lv[t + 2 * NT] = qv2;
    };
    if (ZC_MSM_PREP_PREFETCH && first < n) request_points(first, count_at(first));
    for (int j = 0; j < c; j++) {
        const size_t base = first + (size_t)j * stride;
        if (base >= n) break;
        steps = j + 1;
        const int cnt = count_at(base);
        __syncthreads();                                                     // (one wave per workgroup: no s_barrier is emitted, LDS operations of a wave execute in order)
        if (ZC_MSM_PREP_PREFETCH) {
            land_points(cnt);
            const size_t nbase = base + stride;
            if (j + 1 < c && nbase < n) request_points(nbase, count_at(nbase));     // in flight during this step's arithmetic
        } else {
            coop_load40<false, NT>(sp, points + 20 * base, cnt * 4);
        }
        __syncthreads();
        if (t < cnt) {
            u64 l[5];
            load5(l, sp + 20 * t + 10);
            all_one = all_one && l[0] == 1 && (l[1] | l[2] | l[3] | l[4]) == 0;
            const fe z = fe_select(limbs52_all_zero(l), neutral, fe_from_limbs52(l));
#pragma unroll
            for (int w = 0; w < 9; w++) spre[9 * t + w] = acc.v[w];
            acc = fp_mul(acc, z);
        }
        __syncthreads();
        // the step's prefix products wait at the start of the region the step's records will occupy (whole 16-byte
        // vectors: the tail of a partial block may run a few words into the next lane-less slots of the same region)
        coop_copy16(reinterpret_cast<uint4*>(recs + (size_t)rec_words * base), reinterpret_cast<const uint4*>(spre), (cnt * 9 + 3) / 4);
    }
    // one inversion per lane (the lanes of a wave run it in lock step: it is shared by the lane's own points, not across lanes);
    // a workgroup whose points all have Z = 1 needs none
    __shared__ int s_any;
    if (t == 0) s_any = 0;
    __syncthreads();
    if (!all_one) s_any = 1;
    __syncthreads();
    const bool skip = s_any == 0;
    if (ZC_MSM_PREP_PREFETCH && steps > 0) {                                 // the last block again (the backward sweep starts there), under the inversion
        const size_t base = first + (size_t)(steps - 1) * stride;
        request_points(base, count_at(base));
        if (!skip) request_prefix(base, count_at(base));
    }
    fe inv = neutral;
    if (!skip) inv = fp_inverse_of_register(acc) , inv = fp_mul(inv, fe_const<FP>(ModP::R3));   // carries R^3 from here on: inv * pre = R^2 / Z
    const fe d2 = fe_const<FP>(ModP::D2_M);
    for (int j = steps - 1; j >= 0; j--) {
        const size_t base = first + (size_t)j * stride;
        const int cnt = count_at(base);
        __syncthreads();
        if (ZC_MSM_PREP_PREFETCH) {
            land_points(cnt);
            if (!skip) land_prefix(cnt);
            if (j > 0) {                                                     // the block below: always a full one
                request_points(base - stride, NT);
                if (!skip) request_prefix(base - stride, NT);
            }
        } else {
            coop_load40<false, NT>(sp, points + 20 * base, cnt * 4);
            if (!skip) coop_copy16(reinterpret_cast<uint4*>(spre), reinterpret_cast<const uint4*>(recs + (size_t)rec_words * base), (cnt * 9 + 3) / 4);
        }
        __syncthreads();
        fe ymx, ypx, t2d;
        if (t < cnt) {
            u64 lx[5], ly[5], lz[5], lt[5];
            load5(lx, sp + 20 * t);
            load5(ly, sp + 20 * t + 5);
            load5(lz, sp + 20 * t + 10);
            load5(lt, sp + 20 * t + 15);
            fe zi2 = fe_const<FP>(ModP::RR);                    // (1 / Z) R^2
            if (!skip) {
                const fe z = fe_select(limbs52_all_zero(lz), neutral, fe_from_limbs52(lz));
                fe pre;
#pragma unroll
                for (int w = 0; w < 9; w++) pre.v[w] = spre[9 * t + w];
                zi2 = fp_mul(inv, pre);
                inv = fp_mul(inv, z);
            }
            const fe xm = fp_mul(fe_from_limbs52(lx), zi2);     // x R
            const fe ym = fp_mul(fe_from_limbs52(ly), zi2);     // y R
            const fe tm = fp_mul(fe_from_limbs52(lt), zi2);     // (T / Z) R = x y R
            ypx = fe_add(ym, xm);
            fe_carry(ypx);
            ymx = fp_sub(ym, xm);                               // normalized, < 7N < 2^256
            t2d = fp_mul(tm, d2);
        }
        __syncthreads();                                        // every lane has read its point: the buffer takes the records
        if (t < cnt) {
            u32* o = reinterpret_cast<u32*>(sp) + MSM_AFF_WORDS * t;
            if (ZC_MSM_AFF_LIMBS) {                             // the limbs as they are: ymx / ypx normalized, t2d R-class (every limb < 2^29 ... 2^30)
#pragma unroll
                for (int w = 0; w < 9; w++) {
                    o[w] = ymx.v[w];
                    o[9 + w] = ypx.v[w];
                    o[18 + w] = t2d.v[w];
                }
                o[27] = 0;
            } else {
                pack256(o, ymx);
                pack256(o + 8, ypx);
                pack256(o + 16, t2d);
            }
        }
        __syncthreads();
        {
            uint4* dst = reinterpret_cast<uint4*>(recs + (size_t)rec_words * base);
            const uint4* src = reinterpret_cast<const uint4*>(sp);
            const u32 rv = rec_words / 4;                       // 16-byte pieces per record slot: 6 (packed) or 8
            for (int v = threadIdx.x; v < cnt * MSM_AFF_PIECES; v += NT) dst[(size_t)(v / MSM_AFF_PIECES) * rv + (v % MSM_AFF_PIECES)] = src[v];
        }
    }
}
// Bucket sums are kept in the kernels' own number system between k_msm_runs and k_msm_segments:
// 36 x u32 (X, Y, Z, T as nine 29-bit Montgomery limbs each, R-class), 144 bytes per bucket.  A
// one-byte flag per bucket (zero-filled per call: 144 times less than the records) says whether the
// record was written; an unwritten bucket is the identity.  In the edge lists an all-zero record
// stands for the identity (Z == 0 never occurs for a point).
constexpr int MSM_RAW_WORDS = 36;
ZC_DI void pt_store_raw(u32* __restrict__ o, const pt& p)
{
    uint4* v = reinterpret_cast<uint4*>(o);               // 144-byte records: 16-byte aligned
    const u32 w[36] = {p.X.v[0], p.X.v[1], p.X.v[2], p.X.v[3], p.X.v[4], p.X.v[5], p.X.v[6], p.X.v[7], p.X.v[8],
                       p.Y.v[0], p.Y.v[1], p.Y.v[2], p.Y.v[3], p.Y.v[4], p.Y.v[5], p.Y.v[6], p.Y.v[7], p.Y.v[8],
                       p.Z.v[0], p.Z.v[1], p.Z.v[2], p.Z.v[3], p.Z.v[4], p.Z.v[5], p.Z.v[6], p.Z.v[7], p.Z.v[8],
                       p.T.v[0], p.T.v[1], p.T.v[2], p.T.v[3], p.T.v[4], p.T.v[5], p.T.v[6], p.T.v[7], p.T.v[8]};
#pragma unroll
    for (int q = 0; q < 9; q++) v[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}
ZC_DI pt pt_load_raw(const u32* __restrict__ o)
{
    const uint4* v = reinterpret_cast<const uint4*>(o);
    u32 w[36];
#pragma unroll
    for (int q = 0; q < 9; q++) {
        const uint4 x = v[q];
        w[4 * q] = x.x; w[4 * q + 1] = x.y; w[4 * q + 2] = x.z; w[4 * q + 3] = x.w;
    }
    pt p;
    u32 zor = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        p.X.v[i] = w[i]; p.Y.v[i] = w[9 + i]; p.Z.v[i] = w[18 + i]; p.T.v[i] = w[27 + i];
        zor |= w[18 + i];
    }
    return pt_select(zor == 0, pt_identity(), p);
}

ZC_DI void raw_store_zero(u32* __restrict__ o)
{
    uint4* v = reinterpret_cast<uint4*>(o);
#pragma unroll
    for (int q = 0; q < 9; q++) v[q] = make_uint4(0, 0, 0, 0);
}
// End of a segment (maximal stretch of one key inside a run): where does its sum go?
//   * a segment that neither starts the run with the previous run's last key nor ends it with the
//     next run's first key is a whole bucket -> buckets_raw[key];
//   * otherwise it becomes an edge of the next level, in the same raw format: slot 2j (open to the
//     left) or 2j + 1 (open to the right only); a run that is one segment open on both sides fills
//     both slots (the second with an all-zero record = identity) so that no unused slot ever
//     separates two edges of one bucket.  Unused slots keep the sentinel key 0xFFFFFFFF >= nbuckets
//     every lane writes into its two slots first; zero digits carry a sentinel key too and are
//     dropped here.
ZC_DI void msm_flush(u32 key, const pt& sum, bool seg_first, bool seg_last, u32 prev_key, u32 next_key, u32 j, u32 nbuckets,
                     u32* __restrict__ buckets_raw, uint8_t* __restrict__ present, u32* __restrict__ next_keys, u32* __restrict__ next_recs)
{
    if (key >= nbuckets) return;
    const bool open_left = seg_first && key == prev_key;
    const bool open_right = seg_last && key == next_key;
    u32* dst = buckets_raw + MSM_RAW_WORDS * (size_t)key;
    if (!(open_left || open_right)) present[key] = 1;
    if (open_left || open_right) {
        const size_t slot = open_left ? 2 * (size_t)j : 2 * (size_t)j + 1;
        next_keys[slot] = key;
        dst = next_recs + MSM_RAW_WORDS * slot;
        if (open_left && open_right) {
            next_keys[slot + 1] = key;
            raw_store_zero(dst + MSM_RAW_WORDS);
        }
    }
    pt_store_raw(dst, sum);
}

// Level 0 of the segmented reduction (see the file header): the sorted (key, point index | sign << 31)
// pairs, gathering the prepared 128-byte cached records.  Lane j owns the run [j T, (j + 1) T).
// The (random) 128-byte gather of entry e+1 is in flight while entry e is being added, and it
// lands in LDS, not in registers: `global_load_lds_dwordx4` copies 16 bytes per lane straight into
// the wave's staging area (piece q of all 64 lanes contiguous: M0 = base + q KB), so the prefetch
// holds no VGPRs and the kernel fits four waves per SIMD (32 KB of LDS per block, 4 blocks per CU).
// A lane reads and overwrites only its own slots: lgkmcnt(0) before the next copy is issued,
// vmcnt(0) before the slots are read.
#ifndef ZC_MSM_WAVES
#define ZC_MSM_WAVES 4         // resident waves per SIMD the bucket-sum kernel is compiled for (A/B knob)
#endif
#ifndef ZC_MSM_RUN_BLOCK
#define ZC_MSM_RUN_BLOCK 256   // threads per workgroup of k_msm_runs (its waves never synchronise: A/B knob)
#endif
constexpr int MSM_RUN_BLOCK = ZC_MSM_RUN_BLOCK;
// Everything behind the bucket sums is a chain of dependent point operations on few waves.  When the windows go through
// the pipeline in groups (below, and zerocaf_hip.hip: msm_on_device) a group's chain runs BESIDE the bucket sums of the groups below it, whose
// four waves per SIMD would otherwise win every issue slot (older waves first): the chain's kernels raise their wave
// priority, so that the arbiter serves them first -- they need few slots, but they need them now.
ZC_DI void msm_tail_priority() { __builtin_amdgcn_s_setprio(3); }
// WINDOW GROUPS.  A launch covers one group of windows: a part of the sorted list (the list is ordered by window; where a
// window starts is known on the device only -- the key sort's last scan table, zc_sort.hip.h: msm_sort_slot).  Lane j owns
// the run [*range_lo + j T, *range_lo + (j + 1) T) cut at *range_end (nullptr: the whole list [0, len)); the host sizes the
// launch for the longest the part can be, lanes behind its end idle.  The neighbours a lane looks at for its open ends may
// lie in another group (another window: another key).  `sj`: the lane's number in the edge arrays (groups share them).
template <bool AFFINE>
ZC_DI void msm_runs_body(const uint2* __restrict__ pairs, const u32* __restrict__ recs, u32 len, u32 T, u32 nbuckets,
                         u32* buckets_raw, uint8_t* present, u32* next_keys, u32* next_recs, const u32* range_lo, const u32* range_end, u32 j, u32 sj, u32 rec_words)
{
    constexpr int PIECES = AFFINE ? MSM_AFF_PIECES : 8;        // 16-byte pieces of a cached record
    __shared__ uint4 stage[PIECES * MSM_RUN_BLOCK];
    const int lane = threadIdx.x & 63;
    uint4* base = stage + (threadIdx.x >> 6) * (PIECES * 64);
    const u32 none = 0xFFFFFFFFu;
    next_keys[2 * (size_t)sj] = none;                          // this lane's two edge slots (lane sj of the edge arrays): unused until msm_flush says otherwise
    next_keys[2 * (size_t)sj + 1] = none;
    const u32 first_pos = range_lo ? *range_lo : 0u;
    const u32 end_pos = range_end ? *range_end : len;
    const u64 lo64 = (u64)first_pos + (u64)j * T;
    if (lo64 >= end_pos) return;
    const u32 lo = (u32)lo64;
    const u32 hi = (end_pos - lo > T) ? lo + T : end_pos;
    const uint2 cur = pairs[lo];
    u32 cur_key = cur.x;
    if (cur_key >= nbuckets) return;                           // zero digits sort behind every bucket: nothing to add
    // neighbours outside the launch's own part of the list belong to another window (another key)
    const u32 prev_key = lo > first_pos ? pairs[lo - 1].x : none;
    const u32 next_key = hi < end_pos ? pairs[hi].x : none;
    auto fetch = [&](u32 v) {
        const uint4* src = reinterpret_cast<const uint4*>(recs + (size_t)rec_words * (size_t)(v & 0x7FFFFFFFu));
#pragma unroll
        for (int q = 0; q < PIECES; q++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q),
                                             (__attribute__((address_space(3))) void*)(base + q * 64), 16, 0, 0);
    };
    pt acc = pt_identity();
    u32 vcur = cur.y;
    fetch(vcur);
    uint2 nxt = (lo + 1 < hi) ? pairs[lo + 1] : make_uint2(none, 0);
    bool first = true;
    for (u32 e = lo; e < hi; e++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        niels q;
        if (AFFINE && ZC_MSM_AFF_LIMBS) {
            u32 w[28];
#pragma unroll
            for (int p7 = 0; p7 < 7; p7++) {
                const uint4 x = base[p7 * 64 + lane];
                w[4 * p7] = x.x; w[4 * p7 + 1] = x.y; w[4 * p7 + 2] = x.z; w[4 * p7 + 3] = x.w;
            }
#pragma unroll
            for (int k = 0; k < 9; k++) {
                q.ymx.v[k] = w[k];
                q.ypx.v[k] = w[9 + k];
                q.t2d.v[k] = w[18 + k];
            }
            q.z = fe_zero();                                   // unused: Z' = 1
        } else {
        q.ymx = unpack256(base[0 * 64 + lane], base[1 * 64 + lane]);
        q.ypx = unpack256(base[2 * 64 + lane], base[3 * 64 + lane]);
        if (AFFINE) {
            q.z = fe_zero();                                   // unused: Z' = 1
            q.t2d = unpack256(base[4 * 64 + lane], base[5 * 64 + lane]);
        } else {
            q.z = unpack256(base[4 * 64 + lane], base[5 * 64 + lane]);
            q.t2d = unpack256(base[(PIECES - 2) * 64 + lane], base[(PIECES - 1) * 64 + lane]);
        }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const bool neg = (vcur >> 31) != 0;
        const u32 knext = nxt.x;
        if (e + 1 < hi) fetch(nxt.y);
        vcur = nxt.y;
        nxt = (e + 2 < hi) ? pairs[e + 2] : make_uint2(none, 0);
        acc = pt_add_cached<ZC_MSM_ACC_ILP, AFFINE>(acc, niels_cond_neg(neg, q));
        const bool last = e + 1 == hi;
        if (last || knext != cur_key) {                        // the segment ends with this entry
            msm_flush(cur_key, acc, first, last, prev_key, next_key, sj, nbuckets, buckets_raw, present, next_keys, next_recs);
            acc = pt_identity();
            first = false;
            cur_key = knext;
            if (cur_key >= nbuckets) break;                    // only zero digits follow
        }
    }
}
extern "C" __global__ __launch_bounds__(ZC_MSM_RUN_BLOCK) __attribute__((amdgpu_waves_per_eu(ZC_MSM_WAVES)))
void k_msm_runs(const uint2* pairs, const u32* recs, u32 len, u32 T, u32 nbuckets,
                u32* buckets_raw, uint8_t* present, u32* next_keys, u32* next_recs, const u32* range_lo, const u32* range_end, u32 nlanes, u32 slot0, u32 rec_words)
{
    const u32 j = blockIdx.x * MSM_RUN_BLOCK + threadIdx.x;
    if (j < nlanes) msm_runs_body<false>(pairs, recs, len, T, nbuckets, buckets_raw, present, next_keys, next_recs, range_lo, range_end, j, slot0 + j, rec_words);
}
extern "C" __global__ __launch_bounds__(ZC_MSM_RUN_BLOCK) __attribute__((amdgpu_waves_per_eu(ZC_MSM_WAVES)))
void k_msm_runs_affine(const uint2* pairs, const u32* recs, u32 len, u32 T, u32 nbuckets,
                       u32* buckets_raw, uint8_t* present, u32* next_keys, u32* next_recs, const u32* range_lo, const u32* range_end, u32 nlanes, u32 slot0, u32 rec_words)
{
    const u32 j = blockIdx.x * MSM_RUN_BLOCK + threadIdx.x;
    if (j < nlanes) msm_runs_body<true>(pairs, recs, len, T, nbuckets, buckets_raw, present, next_keys, next_recs, range_lo, range_end, j, slot0 + j, rec_words);
}

// Levels >= 1: the edge list of the level above (raw 144-byte records in list order, sentinel keys in
// unused slots), same run / segment rules, full unified additions.  A few per cent of level 0's work.
// Runs are shifted by one entry (run 0 = [0, T + 1), run j = [j T + 1, (j + 1) T + 1)): a bucket cut once
// at the level above left its two edges in slots 2j + 1 and 2j + 2, and with an even T that pair always
// lies inside one run here, so everything but the buckets longer than a run is finished at level 1.
// QUAD: four lanes per run (pt_add_quad: three multiplication latencies per addition instead of nine, the same field values).
// A level is a handful of dependent additions per lane on a list that fills a fraction of the chip: pure latency, and every
// level of every window group lies on a chain -- the lowest group's on the call's critical path, the others' beside the
// bucket sums of the groups below, which they hold up for as long as they run.  Every lane of a quad walks the same run
// (quad-uniform control flow), lane 0 writes.
template <bool QUAD>
ZC_DI void msm_runs_edges_body(const u32* keys, const u32* recs, u32 len, u32 T, u32 nbuckets,
                               u32* buckets_raw, uint8_t* present, u32* next_keys, u32* next_recs)
{
    msm_tail_priority();
    const int role = QUAD ? (int)(threadIdx.x & 3) : 0;
    const u32 j = QUAD ? blockIdx.x * (ZC_BLOCK / 4) + (threadIdx.x >> 2) : blockIdx.x * ZC_BLOCK + threadIdx.x;
    const u64 lo64 = j ? (u64)j * T + 1 : 0;
    if (lo64 >= len) return;
    if (role == 0) {
        next_keys[2 * (size_t)j] = 0xFFFFFFFFu;
        next_keys[2 * (size_t)j + 1] = 0xFFFFFFFFu;
    }
    const u32 lo = (u32)lo64;
    const u64 hi64 = (u64)(j + 1) * T + 1;
    const u32 hi = hi64 < len ? (u32)hi64 : len;
    const u32 none = 0xFFFFFFFFu;
    {
        // Most runs of most levels hold unused slots only (a uniform batch closes nearly every cut bucket at level 1): look at
        // the run's keys eight at a time -- independent loads, one wait -- before walking it entry by entry (a lane that walks
        // eight sentinel keys one dependent load after the other takes 40 us; measured on the levels behind the first).
        bool any = false;
        for (u32 b = lo; b < hi; b += 8) {
            u32 kk[8];
#pragma unroll
            for (int i = 0; i < 8; i++) kk[i] = b + i < hi ? keys[b + i] : none;
#pragma unroll
            for (int i = 0; i < 8; i++) any = any || kk[i] < nbuckets;
        }
        if (!any) return;
    }
    const u32 prev_key = lo > 0 ? keys[lo - 1] : none;
    const u32 next_key = hi < len ? keys[hi] : none;
    pt acc = pt_identity();
    u32 cur_key = keys[lo];
    bool first = true;
    for (u32 e = lo; e < hi; e++) {
        if (cur_key < nbuckets) {
            const pt rec = pt_load_raw(recs + MSM_RAW_WORDS * (size_t)e);
            acc = QUAD ? pt_add_quad(acc, rec, role) : pt_add<true>(acc, rec);
        }
        const bool last = e + 1 == hi;
        const u32 knext = last ? none : keys[e + 1];
        if (last || knext != cur_key) {
            if (role == 0) msm_flush(cur_key, acc, first, last, prev_key, next_key, j, nbuckets, buckets_raw, present, next_keys, next_recs);
            acc = pt_identity();
            first = false;
            cur_key = knext;
        }
    }
}
ZC_KERNEL void k_msm_runs_edges(const u32* keys, const u32* recs, u32 len, u32 T, u32 nbuckets,
                                u32* buckets_raw, uint8_t* present, u32* next_keys, u32* next_recs)
{
    msm_runs_edges_body<false>(keys, recs, len, T, nbuckets, buckets_raw, present, next_keys, next_recs);
}
ZC_KERNEL void k_msm_runs_edges_quad(const u32* keys, const u32* recs, u32 len, u32 T, u32 nbuckets,
                                     u32* buckets_raw, uint8_t* present, u32* next_keys, u32* next_recs)
{
    msm_runs_edges_body<true>(keys, recs, len, T, nbuckets, buckets_raw, present, next_keys, next_recs);
}

// One lane per segment of `seg` consecutive buckets [first, first + SEG) of one window (bucket
// index b of a window holds the digit magnitude b + 1):
//   acc = sum_j B_{first+j},  sum = sum_j (j + 1) B_{first+j}   (running sums from the top bucket down)
// so that  sum_j (first' + j + 1) B_{first+j} = sum + first' * acc  with first' = first mod 2^(c-1).
// The lane then finishes its segment on the spot: out = sum + first' * acc, the product by the reference's own
// LSB-first double-and-add on the unified addition (scalar_mul_unified; first' < 2^(c-1) is one word) -- no round trip
// through memory and no further launches between the running sums and the fold (rounds 1-3: k_msm_segments ->
// k_ed_scalar_mul_small -> k_ed_add, three launches and two canonical store / load pairs per segment).
// The arrays are the launch's own part of the bucket array (a window group starts at a window boundary).
ZC_KERNEL void k_msm_segments(const u32* buckets_raw, const uint8_t* present, u64* seg_out, size_t nseg_total, int c, int seg)
{
    __shared__ u32 sk[ZC_BLOCK];
    msm_tail_priority();
    const size_t s = gid();
    const bool valid = s < nseg_total;
    const size_t first = (valid ? s : 0) * (size_t)seg;   // bucket index of the segment start (relative to a window boundary)
    pt acc = pt_identity(), sum = pt_identity();
    if (valid) {
        for (int j = seg - 1; j >= 0; j--) {
            pt b = pt_identity();
            if (present[first + j]) b = pt_load_raw(buckets_raw + MSM_RAW_WORDS * (first + j));
            acc = pt_add<true>(acc, b);
            sum = pt_add<true>(sum, acc);
        }
    }
    const u32 k = valid ? (u32)(first & (((size_t)1 << (c - 1)) - 1)) : 0u;
    sk[threadIdx.x] = k;
    const int nbits = k ? 32 - __builtin_clz(k) : 0;
    const pt prod = scalar_mul_unified<true>(acc, sk + threadIdx.x, ZC_BLOCK, nbits);     // nbits = 0: the identity
    if (valid) pt_store(seg_out + 20 * s, pt_add<true>(sum, prod));
}

// The same with four lanes per segment (ptm_add_quad, zc_quad.hip.h: three multiplication latencies per addition instead
// of nine, the same field values): for launches of so few segments that every SIMD holds a single wave anyway -- the
// lowest window group, whose chain nothing hides, and small shards.
ZC_KERNEL void k_msm_segments_quad(const u32* buckets_raw, const uint8_t* present, u64* seg_out, size_t nseg_total, int c, int seg)
{
    msm_tail_priority();
    const int role = threadIdx.x & 3;
    const size_t s = (size_t)blockIdx.x * (ZC_BLOCK / 4) + (threadIdx.x >> 2);
    const bool valid = s < nseg_total;
    const size_t first = (valid ? s : 0) * (size_t)seg;
    const ptm ident = ptm_from_pt(pt_identity());
    ptm acc = ident, sum = ident;
    for (int j = seg - 1; j >= 0; j--) {
        pt b = pt_identity();
        if (valid && present[first + j]) b = pt_load_raw(buckets_raw + MSM_RAW_WORDS * (first + j));
        acc = ptm_add_quad(acc, ptm_from_pt(b), role);
        sum = ptm_add_quad(sum, acc, role);
    }
    const u32 k = valid ? (u32)(first & (((size_t)1 << (c - 1)) - 1)) : 0u;
    const int nbits = k ? 32 - __builtin_clz(k) : 0;
    ptm N = acc, Q = ident;                                  // k * acc: the unified-step loop of scalar_mul_unified, quad-uniform
    int pos = 0;
    bool pend = (k & 1) != 0;
    bool active = nbits > 0;
    while (active) {
        const ptm lhs = ptm_select(pend, Q, N);
        const ptm r = ptm_add_quad(lhs, N, role);
        if (pend) {
            Q = r;
            pend = false;
            active = pos < nbits - 1;
        } else {
            N = r;
            pos++;
            pend = ((k >> pos) & 1) != 0;
        }
    }
    sum = ptm_add_quad(sum, Q, role);
    if (valid && role == 0) pt_store(seg_out + 20 * s, ptm_to_pt(sum));
}

#if !ZC_MSM_FOLD_QUAD
// One workgroup folds `g` consecutive points (g a power of two, 2 <= g <= 512) into one: pairs on
// load, then a tree in LDS (raw 144-byte records).  Two launches take the 2^(c-5) segment sums of
// every window down to one point per window, where the pairwise kernel needed c - 5 launches of
// a few microseconds of work each.  Groups never straddle windows (both counts are powers of two).
ZC_KERNEL void k_msm_fold_groups(const u64* in, u64* out, u32 g)
{
    __shared__ uint4 sraw[9 * ZC_BLOCK];
    msm_tail_priority();
    u32* mine = reinterpret_cast<u32*>(sraw) + MSM_RAW_WORDS * threadIdx.x;
    const u32 t = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * g;
    u32 live = g / 2;
    if (t < live) pt_store_raw(mine, pt_add<true>(pt_load(in + 20 * (base + 2 * t)), pt_load(in + 20 * (base + 2 * t + 1))));
    while (live > 1) {
        __syncthreads();
        live >>= 1;
        pt s;
        if (t < live) s = pt_add<true>(pt_load_raw(mine), pt_load_raw(mine + MSM_RAW_WORDS * live));
        __syncthreads();
        if (t < live) pt_store_raw(mine, s);
    }
    if (t == 0) pt_store(out + 20 * (size_t)blockIdx.x, pt_load_raw(mine));
}

#endif
// The same with four lanes per addition (ptm_add_quad: three multiplication latencies per level instead of nine): 64 quads
// per workgroup, g <= 128 points.  The tree is pure latency -- a few dozen workgroups, one addition per level -- so the quad
// form takes a third of the time (2^21 pairs in three window groups: 3.46 -> 3.34 ms, four alternating rounds on one box;
// other sizes unchanged); the intermediate sums stay in the loop's own (Y-X, Y+X, Z, T) form in LDS.
ZC_DI void ptm_store_raw(u32* __restrict__ o, const ptm& p)
{
#pragma unroll
    for (int i = 0; i < 9; i++) {
        o[i] = p.Ym.v[i]; o[9 + i] = p.Yp.v[i]; o[18 + i] = p.Z.v[i]; o[27 + i] = p.T.v[i];
    }
}
ZC_DI ptm ptm_load_raw(const u32* __restrict__ o)
{
    ptm p;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        p.Ym.v[i] = o[i]; p.Yp.v[i] = o[9 + i]; p.Z.v[i] = o[18 + i]; p.T.v[i] = o[27 + i];
    }
    return p;
}
ZC_KERNEL void k_msm_fold_groups_quad(const u64* in, u64* out, u32 g)
{
    __shared__ u32 sraw[MSM_RAW_WORDS * (ZC_BLOCK / 4)];
    msm_tail_priority();
    const int role = threadIdx.x & 3;
    const u32 q = threadIdx.x >> 2;
    u32* mine = sraw + MSM_RAW_WORDS * q;
    const size_t base = (size_t)blockIdx.x * g;
    u32 live = g / 2;
    ptm s = ptm_from_pt(pt_identity());
    if (q < live) {
        s = ptm_add_quad(ptm_from_pt(pt_load(in + 20 * (base + 2 * q))), ptm_from_pt(pt_load(in + 20 * (base + 2 * q + 1))), role);
        if (role == 0) ptm_store_raw(mine, s);
    }
    while (live > 1) {
        __syncthreads();
        live >>= 1;
        if (q < live) s = ptm_add_quad(ptm_load_raw(mine), ptm_load_raw(mine + MSM_RAW_WORDS * live), role);
        __syncthreads();
        if (q < live && role == 0) ptm_store_raw(mine, s);
    }
    if (threadIdx.x == 0) pt_store(out + 20 * (size_t)blockIdx.x, ptm_to_pt(s));
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
// one wave; windows[w] = S_w (W points); out = 2^(c W) * carry + sum_w 2^(c w) S_w  (carry: nullptr = none).
// Window groups: Horner's rule runs top window first across the groups -- the group below continues where the group above
// stopped (`carry` = that group's result), so no group pays doublings for windows that are not its own.
// Leading terms that are the literal identity (0 : y : y : 0) -- the windows above the longest scalar of the batch -- need no
// doublings: the rule starts at the first non-trivial one.
ZC_DI bool msm_is_literal_identity(const u64* s)
{
    u64 nz = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) nz |= s[j] | s[15 + j] | (s[5 + j] ^ s[10 + j]);
    return nz == 0;
}
// `preshifted` != 0: `carry` already holds 2^(c W) * (the result of the groups above) -- k_msm_shift ran on the side stream as
// soon as that result existed, beside the lowest group's bucket sums -- so the rule runs over this group's own windows only
// and the carry is added last: c (W - 1) doublings on the call's critical path instead of c W.
ZC_KERNEL void k_msm_window_combine(const u64* windows, u64* out, int W, int c, const u64* carry, int preshifted)
{
    msm_tail_priority();
    const int role = threadIdx.x & 3;
    pt Q = pt_identity();
    bool started = false;
    if (carry && !preshifted && !msm_is_literal_identity(carry)) {
        Q = pt_load(carry);
        started = true;
    }
    for (int w = W - 1; w >= 0; w--) {
        const u64* sw = windows + 20 * (size_t)w;
        if (started)
            for (int i = 0; i < c; i++) Q = pt_double_quad(Q, role);
        if (!msm_is_literal_identity(sw)) {
            const pt S = pt_load(sw);
            Q = started ? pt_add<true>(Q, S) : S;
            started = true;
        }
    }
    if (carry && preshifted && !msm_is_literal_identity(carry)) {
        const pt S = pt_load(carry);
        Q = started ? pt_add<true>(Q, S) : S;
    }
    if (threadIdx.x == 0) pt_store(out, Q);
}
// out = 2^k * in (k doublings on one quad of lanes; the literal identity stays what it is)
ZC_KERNEL void k_msm_shift(const u64* in, u64* out, int k)
{
    msm_tail_priority();
    const int role = threadIdx.x & 3;
    pt Q = pt_load(in);
    if (!msm_is_literal_identity(in))
        for (int i = 0; i < k; i++) Q = pt_double_quad(Q, role);
    if (threadIdx.x == 0) pt_store(out, Q);
}

}  // namespace zc

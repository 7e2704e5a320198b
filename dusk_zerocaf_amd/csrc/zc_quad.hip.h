// zc_quad.hip.h -- one group-law evaluation shared by a quad of lanes (DPP quad broadcasts).
// For work that is pure latency: the MSM window combination (zc_msm.hip.h) and strict scalar
// multiplications of batches so small that every SIMD holds a single wave anyway.
#pragma once
#include "zc_kernels.hip.h"

namespace zc {

template <int J>
ZC_DI fe quad_bcast(const fe& x)
{
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)x.v[i], J * 0x55, 0xF, 0xF, false);
    return r;
}
ZC_DI fe fe_by_role(int role, const fe& a, const fe& b, const fe& c, const fe& d)
{
    return fe_select(role < 2, fe_select(role == 0, a, b), fe_select(role == 2, c, d));
}
// The unified addition of the scalar-mul loop (ptm_add, zc_curve.hip.h) on four lanes that all hold
// the same operands: phase 1 forms M, P, d*T1 and D on lanes 0..3, lane 2 finishes C = (d T1) T2,
// phase 2 forms X3 = E F, Y3 = G H, Z3 = F G, T3 = E H.  Three multiplication latencies per step
// instead of nine; every field value is the one ptm_add computes.
ZC_DI ptm ptm_add_quad(const ptm& p, const ptm& q, int role)
{
    fe m1 = mont_mul_ilp<FP>(fe_by_role(role, p.Ym, p.Yp, p.T, p.Z), fe_by_role(role, q.Ym, q.Yp, fe_const<FP>(ModP::D_M), q.Z));
    const fe c2 = mont_mul_ilp<FP>(m1, q.T);                 // used by lane 2 only
    m1 = fe_select(role == 2, c2, m1);
    const fe M = quad_bcast<0>(m1), P = quad_bcast<1>(m1), C = quad_bcast<2>(m1), D = quad_bcast<3>(m1);
    const fe E = fe_sub_half<FP>(P, M);
    const fe H = fe_sub_lazy<FP>(P, E);
    const fe F = fe_sub_lazy<FP>(D, C);
    const fe G = fe_add(D, C);
    const fe m2 = mont_mul_ilp<FP>(fe_by_role(role, E, G, F, E), fe_by_role(role, F, H, G, H));
    const fe X3 = quad_bcast<0>(m2), Y3 = quad_bcast<1>(m2);
    ptm r;
    r.Ym = fp_sub(Y3, X3);
    r.Yp = fe_add(Y3, X3);
    r.Z = quad_bcast<2>(m2);
    r.T = quad_bcast<3>(m2);
    return r;
}

// The same step on (X, Y, Z, T) operands and with an (X, Y, Z, T) result -- pt_add's values (edwards.rs:465-489) -- for chains
// whose intermediate sums live in memory in that form (the MSM's edge records, zc_msm.hip.h).
ZC_DI pt pt_add_quad(const pt& p, const pt& q, int role)
{
    fe m1 = mont_mul_ilp<FP>(fe_by_role(role, fp_sub(p.Y, p.X), fe_add(p.Y, p.X), p.T, p.Z),
                             fe_by_role(role, fp_sub(q.Y, q.X), fe_add(q.Y, q.X), fe_const<FP>(ModP::D_M), q.Z));
    const fe c2 = mont_mul_ilp<FP>(m1, q.T);                 // used by lane 2 only
    m1 = fe_select(role == 2, c2, m1);
    const fe M = quad_bcast<0>(m1), P = quad_bcast<1>(m1), C = quad_bcast<2>(m1), D = quad_bcast<3>(m1);
    const fe E = fe_sub_half<FP>(P, M);
    const fe H = fp_sub(P, E);
    const fe F = fp_sub(D, C);
    const fe G = fe_add(D, C);
    const fe m2 = mont_mul_ilp<FP>(fe_by_role(role, E, G, F, E), fe_by_role(role, F, H, G, H));
    pt r;
    r.X = quad_bcast<0>(m2);
    r.Y = quad_bcast<1>(m2);
    r.Z = quad_bcast<2>(m2);
    r.T = quad_bcast<3>(m2);
    return r;
}

// Strict scalar multiplication, four lanes per element (64 elements per workgroup): same loop and
// same values as scalar_mul_unified, for batches of at most 2^14 elements.
ZC_KERNEL void k_ed_scalar_mul_quad(const u64* p, const u64* k, u64* out, size_t n)
{
    __shared__ u32 sk[9 * (ZC_BLOCK / 4)];
    const int tid = threadIdx.x, role = tid & 3, slot = tid >> 2;
    const size_t i = (size_t)blockIdx.x * (ZC_BLOCK / 4) + slot;
    const bool valid = i < n;
    const size_t ii = valid ? i : 0;
    u64 l[5];
    load_scalar(l, k + 5 * ii);
    int nbits;
    {
        u32 w[9];
        scalar_to_words(w, 1, l, nbits);                     // every lane of the quad derives the same words
        if (role == 0) {
#pragma unroll
            for (int j = 0; j < 9; j++) sk[j * (ZC_BLOCK / 4) + slot] = w[j];
        }
    }
    __syncthreads();
    if (!valid) nbits = 0;
    const u32* words = sk + slot;
    const int stride = ZC_BLOCK / 4;
    ptm N = ptm_from_pt(pt_load(p + 20 * ii)), Q = ptm_from_pt(pt_identity());
    int pos = 0;
    u32 cur = words[0];
    bool pend = (cur & 1) != 0;
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
            if ((pos & 31) == 0) cur = words[(pos >> 5) * stride];
            pend = ((cur >> (pos & 31)) & 1) != 0;
        }
    }
    if (valid && role == 0) pt_store(out + 20 * i, ptm_to_pt(Q));
}

// ((p0 + p1) + p2) + ... in index order with the unified addition (edwards.rs:465-489), one launch: the exchange step
// of a sharded MSM folds the gathered per-rank partials with this, so every rank ends with identical limbs.  `count` is
// small (one point per GPU) and the chain is pure latency: ONE quad of lanes shares every addition (ptm_add_quad: three
// multiplication latencies instead of nine, the same field values, hence the same canonical limbs) and every load (lane r
// converts coordinate r: one Montgomery conversion of latency per point instead of four).  `extra` (optional) is added last.
// Eight partials: 75 -> about 25 us behind the all-gather of an 8-GPU MSM.
ZC_KERNEL void k_ed_fold_ordered(const u64* parts, size_t count, const u64* extra, u64* out)
{
    if (blockIdx.x != 0 || threadIdx.x >= 4) return;
    const int role = threadIdx.x & 3;
    auto load = [&](const u64* p) {
        const fe c = fe_load_mont<FP>(p + 5 * role);
        pt r;
        r.X = quad_bcast<0>(c);
        r.Y = quad_bcast<1>(c);
        r.Z = quad_bcast<2>(c);
        r.T = quad_bcast<3>(c);
        return r;
    };
    ptm acc = ptm_from_pt(count ? load(parts) : pt_identity());
    for (size_t i = 1; i < count; i++) acc = ptm_add_quad(acc, ptm_from_pt(load(parts + 20 * i)), role);
    if (extra) acc = ptm_add_quad(acc, ptm_from_pt(load(extra)), role);
    if (role == 0) pt_store(out, ptm_to_pt(acc));
}

}  // namespace zc

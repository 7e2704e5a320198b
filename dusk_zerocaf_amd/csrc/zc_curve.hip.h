// zc_curve.hip.h -- device-side Edwards / Ristretto group law and codecs on top of
// zc_arith.hip.h.  One point per lane; coordinates live in VGPRs in Montgomery form
// (R = 2^261, radix 2^29).  Every function cites the reference lines whose
// *values* it reproduces (paths relative to the reference checkout).
#pragma once
#include "zc_arith.hip.h"

namespace zc {

typedef ModP FP;

struct pt {
    fe X, Y, Z, T;
};

// Fixed exponents as sliding-window programs (window 3: a, a^3, a^5, a^7), generated and
// checked by gen_constants.py; read through the scalar cache (wave-uniform loads).
// entry = nsq | idx << 8: acc <- acc^(2^nsq) * a^(2 idx + 1), idx 255 = squarings only.
template <int N>
struct exp_program {
    u32 e[N];
};
template <int N>
constexpr exp_program<N> make_program(const u32 (&src)[N])
{
    exp_program<N> p{};
    for (int i = 0; i < N; i++) p.e[i] = src[i];
    return p;
}
__device__ __constant__ exp_program<ModP::PROG_INV_LEN> ZC_PROG_INV = make_program(ModP::PROG_INV);
__device__ __constant__ exp_program<ModP::PROG_P58_LEN> ZC_PROG_P58 = make_program(ModP::PROG_P58);

// Launches of at most one workgroup per CU leave a single wave on every SIMD; it cannot hide the
// latency of the column-ordered multiplier's serial chain, so the long fixed exponentiations switch
// (wave-uniformly, at run time) to the independent-chain multiplier there: 2^14..2^16 elements
// 25-35 % faster, 2^20 unchanged.
ZC_DI bool zc_small_launch()
{
#if defined(__HIP_DEVICE_COMPILE__)
    return gridDim.x <= 256;
#else
    return false;
#endif
}
// a^e on a fixed schedule (uniform branches): same value as any other evaluation order.
ZC_DI fe fp_pow_program(const fe& a, const u32* __restrict__ prog, int len)
{
    const fe a2 = mont_sqr<FP>(a);
    const fe a3 = mont_mul<FP>(a2, a), a5 = mont_mul<FP>(a3, a2), a7 = mont_mul<FP>(a5, a2);
    const u32 first = prog[0] >> 8;
    fe acc = first == 0 ? a : first == 1 ? a3 : first == 2 ? a5 : a7;
    if (zc_small_launch()) {
        for (int k = 1; k < len; k++) {
            const u32 entry = prog[k];
            for (u32 s = entry & 0xFF; s > 0; s--) acc = mont_sqr_ilp<FP>(acc);
            const u32 idx = entry >> 8;
            if (idx < 4) acc = mont_mul_ilp<FP>(acc, idx == 0 ? a : idx == 1 ? a3 : idx == 2 ? a5 : a7);
        }
        return acc;
    }
    for (int k = 1; k < len; k++) {
        const u32 entry = prog[k];
        for (u32 s = entry & 0xFF; s > 0; s--) acc = mont_sqr<FP>(acc);
        switch (entry >> 8) {
            case 0: acc = mont_mul<FP>(acc, a); break;
            case 1: acc = mont_mul<FP>(acc, a3); break;
            case 2: acc = mont_mul<FP>(acc, a5); break;
            case 3: acc = mont_mul<FP>(acc, a7); break;
            default: break;
        }
    }
    return acc;
}
ZC_DI fe fp_pow_inv(const fe& a) { return fp_pow_program(a, ZC_PROG_INV.e, ModP::PROG_INV_LEN); }      // a^(p-2)
ZC_DI fe fp_pow_p58(const fe& a) { return fp_pow_program(a, ZC_PROG_P58.e, ModP::PROG_P58_LEN); }      // a^((p-5)/8)

ZC_DI fe fp_mul(const fe& a, const fe& b) { return mont_mul<FP>(a, b); }
ZC_DI fe fp_sqr(const fe& a) { return mont_sqr<FP>(a); }
ZC_DI fe fp_sub(const fe& a, const fe& b) { return fe_sub<FP>(a, b); }
ZC_DI fe fp_neg(const fe& a) { return fe_neg<FP>(a); }
ZC_DI fe fp_canon(const fe& a) { return fe_canon_from_mont<FP>(a); }   // plain canonical value
ZC_DI bool fp_is_zero(const fe& a) { return fe_is_zero_canon(fp_canon(a)); }
// a == b (mod p); b must be R-class
ZC_DI bool fp_eq(const fe& a, const fe& b) { return fp_is_zero(fp_sub(a, b)); }

// ---------------------------------------------------------------- modular inverse by division steps
// a^-1 mod N through Bernstein-Yang division steps ("safegcd", the half-delta variant with its fixed
// schedule): 20 rounds of 30 division steps on the low words of (f, g) = (N, a), each round followed by
// one 2x2 integer matrix applied to the full-width (f, g) and, modulo N, to (d, e) = (0, 1); after 600
// steps (590 suffice for 256-bit operands) g = 0, f = +-1 and d = +-a^-1.  Signed 30-bit limbs in
// 32-bit registers; a round costs ~350 32-bit ALU instructions and ~110 multiplier-class ones (signed
// 32 x 32 + 64 multiply-accumulates), i.e. an inversion costs about as much as 45 field
// multiplications where the Fermat power a^(N-2) costs 290 -- and its dependent chain is as much
// shorter, which is what bounds a launch of one wave per SIMD.  Uniform control flow (no lane ever
// leaves the schedule); a = 0 returns 0, like a^(N-2).  Same value as the reference's Savas-Koc
// inverse (field.rs:854-925): the inverse is unique.
constexpr int32_t M30 = 0x3fffffff;
struct sgcd_mat {
    int32_t u, v, q, r;
};
// 30 division steps on the low 32 bits of f (odd) and g; returns the new zeta and the matrix t with
// t * [f, g] = 2^30 * [f', g']
ZC_DI int32_t sgcd_divsteps30(int32_t zeta, u32 f, u32 g, sgcd_mat& t)
{
    u32 u = 1, v = 0, q = 0, r = 1;
#pragma unroll 6
    for (int i = 0; i < 30; i++) {
        u32 mask1 = (u32)(zeta >> 31);                         // zeta < 0
        const u32 mask2 = 0u - (g & 1u);                       // g odd
        const u32 x = (f ^ mask1) - mask1, y = (u ^ mask1) - mask1, z = (v ^ mask1) - mask1;
        g += x & mask2;
        q += y & mask2;
        r += z & mask2;
        mask1 &= mask2;
        zeta = (int32_t)((u32)zeta ^ mask1) - 1;
        f += g & mask1;
        u += q & mask1;
        v += r & mask1;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return zeta;
}
// A 32-bit value the optimiser knows nothing about any more: without it the compiler remembers that a
// masked limb is a zero-extended 30-bit quantity and multiplies it as 64 x 64 (one unsigned
// multiply-accumulate plus two v_mul_lo_u32 and an add) instead of one signed v_mad_i64_i32.
#if defined(__HIP_DEVICE_COMPILE__)
#define ZC_PIN32(x) asm("" : "+v"(x))
#else
#define ZC_PIN32(x) asm("" : "+r"(x))
#endif
// the modulus in 30-bit limbs, as compile-time constants (N = 2^k + c: limbs 5..7 are zero)
template <class F>
struct sgcd_mod {
    static constexpr int32_t limb(int j)
    {
        const int s = 30 * j, idx = s / 29, off = s % 29;
        u64 w = idx < 9 ? (u64)F::N[idx] >> off : 0;
        if (idx + 1 < 9) w |= (u64)F::N[idx + 1] << (29 - off);
        if (idx + 2 < 9) w |= (u64)F::N[idx + 2] << (58 - off);
        return (int32_t)(w & 0x3fffffffu);
    }
};
// (f, g) <- t * (f, g) / 2^30 (exact)
ZC_DI void sgcd_update_fg(int32_t (&f)[9], int32_t (&g)[9], const sgcd_mat& t)
{
    int64_t cf = (int64_t)t.u * f[0] + (int64_t)t.v * g[0];
    int64_t cg = (int64_t)t.q * f[0] + (int64_t)t.r * g[0];
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cf += (int64_t)t.u * f[i] + (int64_t)t.v * g[i];
        cg += (int64_t)t.q * f[i] + (int64_t)t.r * g[i];
        f[i - 1] = (int32_t)cf & M30;
        g[i - 1] = (int32_t)cg & M30;
        ZC_PIN32(f[i - 1]);
        ZC_PIN32(g[i - 1]);
        cf >>= 30;
        cg >>= 30;
    }
    f[8] = (int32_t)cf;
    g[8] = (int32_t)cg;
    ZC_PIN32(f[8]);
    ZC_PIN32(g[8]);
}
// (d, e) <- t * (d, e) / 2^30 mod N, both kept in (-2N, N): a multiple of N makes the low 30 bits vanish
template <class F, int I>
ZC_DI void sgcd_de_limb(int32_t (&d)[9], int32_t (&e)[9], const sgcd_mat& t, int32_t md, int32_t me, int64_t& cd, int64_t& ce)
{
    constexpr int32_t mi = sgcd_mod<F>::limb(I);
    cd += (int64_t)t.u * d[I] + (int64_t)t.v * e[I];
    ce += (int64_t)t.q * d[I] + (int64_t)t.r * e[I];
    if constexpr (mi != 0) {
        cd += (int64_t)mi * md;
        ce += (int64_t)mi * me;
    }
    d[I - 1] = (int32_t)cd & M30;
    e[I - 1] = (int32_t)ce & M30;
    ZC_PIN32(d[I - 1]);
    ZC_PIN32(e[I - 1]);
    cd >>= 30;
    ce >>= 30;
    if constexpr (I < 8) sgcd_de_limb<F, I + 1>(d, e, t, md, me, cd, ce);
}
template <class F>
ZC_DI void sgcd_update_de(int32_t (&d)[9], int32_t (&e)[9], const sgcd_mat& t, u32 m_inv30)
{
    constexpr int32_t m0 = sgcd_mod<F>::limb(0);
    const int32_t sd = d[8] >> 31, se = e[8] >> 31;
    int32_t md = (t.u & sd) + (t.v & se);
    int32_t me = (t.q & sd) + (t.r & se);
    int64_t cd = (int64_t)t.u * d[0] + (int64_t)t.v * e[0];
    int64_t ce = (int64_t)t.q * d[0] + (int64_t)t.r * e[0];
    md -= (int32_t)((m_inv30 * (u32)cd + (u32)md) & (u32)M30);
    me -= (int32_t)((m_inv30 * (u32)ce + (u32)me) & (u32)M30);
    ZC_PIN32(md);
    ZC_PIN32(me);
    cd += (int64_t)m0 * md;
    ce += (int64_t)m0 * me;
    cd >>= 30;
    ce >>= 30;
    sgcd_de_limb<F, 1>(d, e, t, md, me, cd, ce);
    d[8] = (int32_t)cd;
    e[8] = (int32_t)ce;
    ZC_PIN32(d[8]);
    ZC_PIN32(e[8]);
}
// nine 29-bit limbs (canonical value) -> nine 30-bit limbs and back
ZC_DI void sgcd_pack30(int32_t (&o)[9], const fe& c)
{
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const int s = 30 * j, idx = s / 29, off = s % 29;
        u64 w = idx < 9 ? (u64)c.v[idx] >> off : 0;
        if (idx + 1 < 9) w |= (u64)c.v[idx + 1] << (29 - off);
        if (idx + 2 < 9) w |= (u64)c.v[idx + 2] << (58 - off);
        o[j] = (int32_t)(w & (u64)M30);
    }
}
ZC_DI fe sgcd_unpack30(const int32_t (&d)[9])
{
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int s = 29 * i, idx = s / 30, off = s % 30;
        u64 w = (u64)(u32)d[idx] >> off;
        if (idx + 1 < 9) w |= (u64)(u32)d[idx + 1] << (30 - off);
        r.v[i] = (u32)w & M29;
    }
    return r;
}
// plain canonical a (< N) -> plain canonical a^-1 mod N (0 for a = 0)
template <class F>
ZC_DI fe fe_inverse_divsteps(const fe& a)
{
    int32_t f[9], g[9], d[9], e[9];
    sgcd_pack30(g, a);
#pragma unroll
    for (int i = 0; i < 9; i++) {
        f[i] = sgcd_mod<F>::limb(i);
        d[i] = 0;
        e[i] = 0;
    }
    e[0] = 1;
    // N^-1 mod 2^30 from -N^-1 mod 2^29 (F::NP) by one Newton step
    constexpr u32 n0 = (u32)sgcd_mod<F>::limb(0) | ((u32)sgcd_mod<F>::limb(1) << 30);
    u32 inv = 0u - F::NP;
    inv *= 2u - n0 * inv;
    const u32 m_inv30 = inv & (u32)M30;
    int32_t zeta = -1;
#pragma unroll 1
    for (int round = 0; round < 20; round++) {
        sgcd_mat t;
        zeta = sgcd_divsteps30(zeta, (u32)f[0] | ((u32)f[1] << 30), (u32)g[0] | ((u32)g[1] << 30), t);
        sgcd_update_de<F>(d, e, t, m_inv30);
        sgcd_update_fg(f, g, t);
    }
    // d = +-a^-1 in (-2N, N) with the sign of f: add N if negative, negate if f < 0, add N again if negative
    const int32_t neg = f[8] >> 31;
    int32_t add = d[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = ((d[i] + (sgcd_mod<F>::limb(i) & add)) ^ neg) - neg;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        d[i + 1] += d[i] >> 30;
        d[i] &= M30;
    }
    add = d[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] += sgcd_mod<F>::limb(i) & add;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        d[i + 1] += d[i] >> 30;
        d[i] &= M30;
    }
    return sgcd_unpack30(d);
}

// a^-1 in the Montgomery domain (a R -> a^-1 R; 0 -> 0): same value as the reference's Savas-Koc
// inverse (field.rs:854-925) and as a^(p-2) (fp_pow_inv; the A/B against it: tools/debug/probes/)
ZC_DI fe fp_invert(const fe& a)
{
    return mont_to<FP>(fe_inverse_divsteps<FP>(fp_canon(a)));
}

// |x| by the reference's sign rule: negate when canonical value > (p-1)/2
// (field.rs:552-557 + subtle conditional_negate).  Returns R-class Montgomery value.
ZC_DI fe fp_abs(const fe& x)
{
    const bool pos = fe_is_positive_canon<FP>(fp_canon(x));
    return fe_select(pos, x, fe_reduce<FP>(fp_neg(x)));
}
ZC_DI bool fp_is_positive(const fe& x) { return fe_is_positive_canon<FP>(fp_canon(x)); }

// sqrt_ratio_i (field.rs:462-503): returns was_square and the non-negative root of
// u/v (square case) or of i*u/v (non-square case), (1,0) for u == 0, (0,0) for v == 0.
// One fixed exponentiation (p = 5 mod 8) instead of two inversions + Legendre +
// Tonelli-Shanks; the decision rules pick the same value.  u, v R-class.
ZC_DI bool fp_sqrt_ratio_i(fe& out, const fe& u, const fe& v)
{
    const fe i_m = fe_const<FP>(ModP::SQRT_M1_M);
    const fe v2 = fp_sqr(v);
    const fe v3 = fp_mul(v2, v);
    const fe v7 = fp_mul(fp_sqr(v3), v);
    const fe uv3 = fp_mul(u, v3);
    fe r = fp_mul(uv3, fp_pow_p58(fp_mul(u, v7)));
    const fe check = fp_mul(v, fp_sqr(r));
    const fe ui = fp_mul(u, i_m);
    const fe cc = fp_canon(check), uc = fp_canon(u);
    const bool correct = fe_eq_canon(cc, uc);
    const bool flipped = fe_is_zero_canon(fp_canon(fe_add(check, u)));        // check == -u
    const bool flipped_i = fe_is_zero_canon(fp_canon(fe_add(check, ui)));     // check == -u*i
    const fe ri = fp_mul(r, i_m);
    r = fe_select(flipped || flipped_i, ri, r);
    out = fp_abs(r);
    return correct || flipped;
}

// Tonelli-Shanks value for p - 1 = 4q with non-residue 6 (field.rs:357-441):
// x = a^((q+1)/2), times 6^q when a^q == -1; None when a^q is not +-1.  a R-class.
ZC_DI bool fp_ts_sqrt(fe& x, const fe& a)
{
    const fe w = fp_pow_p58(a);      // a^((q-1)/2)
    const fe x0 = fp_mul(a, w);                                   // a^((q+1)/2)
    const fe t = fp_canon(fp_mul(x0, w));                         // a^q, plain canonical
    fe one = fe_zero();
    one.v[0] = 1;
    const bool t_is_one = fe_eq_canon(t, one);
    const bool t_is_m1 = fe_eq_canon(t, fe_n_minus_canon<FP>(one));
    const bool a_zero = fp_is_zero(a);
    x = fe_select(t_is_m1, fp_mul(x0, fe_const<FP>(ModP::SIX_POW_Q_M)), x0);
    return t_is_one || t_is_m1 || a_zero;                         // a == 0 -> Some(0)
}
// mod_sqrt(a, sign) (field.rs:378-440): sign = 1 selects p - x_TS
ZC_DI bool fp_mod_sqrt(fe& x, const fe& a, bool sign)
{
    fe r;
    const bool ok = fp_ts_sqrt(r, a);
    x = fe_select(sign, fe_reduce<FP>(fp_neg(r)), r);
    return ok;
}

ZC_DI void load5(u64 (&l)[5], const u64* __restrict__ p)
{
#pragma unroll
    for (int i = 0; i < 5; i++) l[i] = p[i];
}
ZC_DI void store5(u64* __restrict__ p, const u64 (&l)[5])
{
#pragma unroll
    for (int i = 0; i < 5; i++) p[i] = l[i];
}

ZC_DI bool limbs52_all_zero(const u64 (&l)[5]) { return (l[0] | l[1] | l[2] | l[3] | l[4]) == 0; }

// plain inverse of a register value (R-class, i.e. < 3N): acc -> acc^-1 mod N, no Montgomery factor
ZC_DI fe fp_inverse_of_register(const fe& acc) { return fe_inverse_divsteps<FP>(fe_cond_sub_n<FP>(fe_cond_sub_n<FP>(acc))); }

// One lane's share of the batched inversion (Montgomery's trick over the up to `c` elements
// lo, lo + stride, lo + 2 stride, ...; see k_fe_invert_chunked in zc_kernels.hip.h: stride = number of
// lanes, so that the lanes of a wave touch neighbouring records in every pass).
// `num` != nullptr turns it into a batched division: out_j = num_j / a_j (Div, field.rs:277-300).
// ILP: the chunk's multiplications on the independent-chain multiplier -- for launches that leave a single wave on a SIMD (2^20
// elements at 16 per lane: BASELINE configs[1]), where nothing hides the column-ordered multiplier's serial chain: 0.0877 ->
// 0.0815 ms per 2^20 inversions; with more waves per SIMD the column-ordered form is the faster one (2^24: 0.994 against 1.004
// ms), so the host picks per launch (profiles/r06_experiments/fe_invert_ab.md).
template <bool ILP = false>
ZC_DI void fe_invert_chunk(const u64* a, u64* out, uint8_t* ok, size_t n, size_t lo, size_t stride, int c, const u64* num = nullptr)
{
    auto fp_mul = [](const fe& x, const fe& y) { return ILP ? mont_mul_ilp<FP>(x, y) : mont_mul<FP>(x, y); };
    const size_t avail = (n - lo + stride - 1) / stride;
    const int cnt = (int)(avail < (size_t)c ? avail : (size_t)c);
    const fe neutral = fe_one_m<FP>();
    fe acc = neutral;
    for (int j = 0; j < cnt; j++) {
        u64 l[5];
        load5(l, a + 5 * (lo + (size_t)j * stride));
        const fe x = fe_select(limbs52_all_zero(l), neutral, fe_from_limbs52(l));
        u32* slot = reinterpret_cast<u32*>(out + 5 * (lo + (size_t)j * stride));
#pragma unroll
        for (int w = 0; w < 9; w++) slot[w] = acc.v[w];    // acc_{j-1} (R mod p for j = 0)
        acc = fp_mul(acc, x);
    }
    fe inv = fp_inverse_of_register(acc);                  // plain inverse of the register value
    for (int j = cnt - 1; j >= 0; j--) {
        u64 l[5], r[5];
        load5(l, a + 5 * (lo + (size_t)j * stride));
        const bool z = limbs52_all_zero(l);
        const fe x = fe_select(z, neutral, fe_from_limbs52(l));
        const u32* slot = reinterpret_cast<const u32*>(out + 5 * (lo + (size_t)j * stride));
        fe pre;
#pragma unroll
        for (int w = 0; w < 9; w++) pre.v[w] = slot[w];
        fe res = fp_mul(inv, pre);                          // a_j^-1, plain, < 3p
        inv = fp_mul(inv, x);
        if (num) {
            u64 ln[5];
            load5(ln, num + 5 * (lo + (size_t)j * stride));
            res = fp_mul(fe_from_limbs52(ln), mont_to<FP>(res));
        }
        fe_to_limbs52(r, fe_cond_sub_n<FP>(fe_cond_sub_n<FP>(res)));
        if (z) r[0] = r[1] = r[2] = r[3] = r[4] = 0;
        store5(out + 5 * (lo + (size_t)j * stride), r);
        if (ok) ok[lo + (size_t)j * stride] = z ? 0 : 1;
    }
}

// a^e for a per-lane exponent e given as plain canonical limbs (Pow, field.rs:325-355):
// fixed 253-step left-to-right ladder with a selected multiply, same value as the reference's
// data-dependent loop.  a R-class Montgomery, e as nine 29-bit plain limbs.
ZC_DI fe fp_pow_var(const fe& a, const fe& e)
{
    fe acc = fe_one_m<FP>();
    for (int i = 260; i >= 0; i--) {
        acc = fp_sqr(acc);
        const bool bit = ((e.v[i / 29] >> (i % 29)) & 1) != 0;
        acc = fe_select(bit, fp_mul(acc, a), acc);
    }
    return acc;
}
// legendre_symbol (field.rs:703-706): Choice(0) iff a^((p-1)/2) == -1 (so 0 maps to 1).
// The reference's way, an exponentiation: a^q = a w^2 with w = a^((q-1)/2), p - 1 = 4 q, then (a^q)^2.
ZC_DI bool fp_legendre_pow(const fe& a)
{
    const fe w = fp_pow_p58(a);      // a^((q-1)/2)
    const fe t = fp_mul(fp_mul(a, w), w);                         // a^q
    fe one = fe_zero();
    one.v[0] = 1;
    return !fe_eq_canon(fp_canon(fp_sqr(t)), fe_n_minus_canon<FP>(one));
}
// The same bit as the JACOBI symbol (a / p) on the division-step machinery of the inversion above -- no field
// multiplication at all.  Positive division steps (f, g >= 0 throughout, so the symbol's sign rules stay the textbook
// ones):  g odd and eta < 0: swap f, g (sign flips when both are 3 mod 4);  g odd: g += f;  then g /= 2 (sign flips when
// f is 3 or 5 mod 8), eta -= 1.  30 steps on the low words, one 2x2 matrix applied to the full-width (f, g) -- only f
// and g: the symbol needs no Bezout coefficients, so a round is half an inversion round.  The walk ends when f = 1:
// (g / 1) = 1 and the collected sign is the symbol.  Unlike the signed steps of the inversion this variant has no proven
// step bound (random 252-bit inputs: 23-28 rounds, 831 steps at most in 3 x 10^4 trials), so the loop runs until every lane
// of the wave is done, at most `max_rounds` rounds, and a lane that is not falls back to the exponentiation (tests force
// that path with a small bound).  a = 0 (gcd p, f never 1) is answered up front.
#if defined(__HIP_DEVICE_COMPILE__)
ZC_DI bool wave_all(bool x) { return __all(x) != 0; }
ZC_DI bool wave_any(bool x) { return __any(x) != 0; }
#else
ZC_DI bool wave_all(bool x) { return x; }
ZC_DI bool wave_any(bool x) { return x; }
#endif
ZC_DI int32_t sgcd_posdivsteps30(int32_t eta, u32 f, u32 g, sgcd_mat& t, u32& jac)
{
    u32 u = 1, v = 0, q = 0, r = 1;
#pragma unroll 6
    for (int i = 0; i < 30; i++) {
        const u32 godd = 0u - (g & 1u);
        const u32 swap = godd & (u32)(eta >> 31);                // g odd and eta < 0
        jac ^= swap & ((f & g) >> 1);                            // both 3 mod 4 (bit 0 of jac is the sign)
        u32 x = (f ^ g) & swap;
        f ^= x; g ^= x;
        x = (u ^ q) & swap;
        u ^= x; q ^= x;
        x = (v ^ r) & swap;
        v ^= x; r ^= x;
        eta = (int32_t)(((u32)eta ^ swap) - swap);               // negated on a swap
        g += f & godd;
        q += u & godd;
        r += v & godd;
        g >>= 1;
        u <<= 1;
        v <<= 1;
        eta -= 1;
        jac ^= (f >> 1) ^ (f >> 2);                              // (2 / f) = -1 for f = 3, 5 mod 8
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return eta;
}
constexpr int JACOBI_MAX_ROUNDS = 40;
// a: R-class Montgomery value.  true unless (a / p) = -1.
ZC_DI bool fp_legendre(const fe& a, int max_rounds = JACOBI_MAX_ROUNDS)
{
    const fe c = fp_canon(a);
    int32_t f[9], g[9];
    sgcd_pack30(g, c);
#pragma unroll
    for (int i = 0; i < 9; i++) f[i] = sgcd_mod<FP>::limb(i);
    bool done = fe_is_zero_canon(c), res = true;
    int32_t eta = -1;
    u32 jac = 0;
#pragma unroll 1
    for (int round = 0; round < max_rounds; round++) {
        if (wave_all(done)) break;
        sgcd_mat t;
        eta = sgcd_posdivsteps30(eta, (u32)f[0] | ((u32)f[1] << 30), (u32)g[0] | ((u32)g[1] << 30), t, jac);
        sgcd_update_fg(f, g, t);
        const int32_t rest = f[1] | f[2] | f[3] | f[4] | f[5] | f[6] | f[7] | f[8];
        if (!done && f[0] == 1 && rest == 0) {                   // later rounds of the wave's other lanes walk on from here: keep the answer
            done = true;
            res = (jac & 1u) == 0;
        }
    }
    if (wave_any(!done)) {
        const bool slow = fp_legendre_pow(a);
        if (!done) res = slow;
    }
    return res;
}

// ---------------------------------------------------------------- points
ZC_DI pt pt_identity()                                            // edwards.rs:381-391
{
    pt r;
    r.X = fe_zero();
    r.Y = fe_one_m<FP>();
    r.Z = fe_one_m<FP>();
    r.T = fe_zero();
    return r;
}
ZC_DI pt pt_select(bool c, const pt& a, const pt& b)
{
    pt r;
    r.X = fe_select(c, a.X, b.X);
    r.Y = fe_select(c, a.Y, b.Y);
    r.Z = fe_select(c, a.Z, b.Z);
    r.T = fe_select(c, a.T, b.T);
    return r;
}
// HWCD'08 unified addition, a = -1 (edwards.rs:465-489).  The reference forms
//   A = X1 X2, B = Y1 Y2, E = (X1+Y1)(X2+Y2) - A - B = X1 Y2 + Y1 X2, H = B + A
// with three multiplications; the same two field values come out of two:
//   M = (Y1-X1)(Y2-X2) = H - E,  P = (Y1+X1)(Y2+X2) = H + E,  E = (P - M)/2 (exact halving),
//   H = P - E
// so a step costs 9 multiplications and X3 = E F, Y3 = G H, Z3 = F G, T3 = E H are the
// reference's values limb for limb.  Inputs R-class.
template <bool ILP = false>
ZC_DI pt pt_add(const pt& p, const pt& q)
{
    auto fp_mul = [](const fe& x, const fe& y) { return ILP ? mont_mul_ilp<FP>(x, y) : mont_mul<FP>(x, y); };
    const fe M = fp_mul(fp_sub(p.Y, p.X), fp_sub(q.Y, q.X));          // operands < 7N: M < 2N
    const fe P = fp_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));          // operands < 6N: P < 2N
    const fe C = fp_mul(fp_mul(fe_const<FP>(ModP::D_M), p.T), q.T);
    const fe D = fp_mul(p.Z, q.Z);
    const fe E = fe_sub_half<FP>(P, M);
    const fe H = fp_sub(P, E);
    const fe F = fp_sub(D, C);
    const fe G = fe_add(D, C);
    pt r;
    r.X = fp_mul(E, F);
    r.Y = fp_mul(G, H);
    r.Z = fp_mul(F, G);
    r.T = fp_mul(E, H);
    return r;
}
// The stand-alone point kernels (k_ed_add / sub / double / coset4) skip the Montgomery conversions: with PLAIN
// canonical coordinates and the constant d in Montgomery form, every first-level product of pt_add comes out
// with a factor 1/R (M, P, D directly; C = (dR * T1 / R) * T2 / R) and the linear steps keep it.  The four outputs are
// the products E F, G H, F G, E H: E and G each stand in two of them, one on every output, so scaling those TWO values by
// R^4 (one multiplication each) makes all four outputs plain: (E R^4 / R) F / R = E_true F_true.  11 multiplications per
// addition (rounds 2-4: 13, one multiplication by R^4 per OUTPUT; before that 8 into the domain + 9 + 4 out of it); the
// canonical results are the same limbs.
template <bool ILP = false>
ZC_DI pt pt_add_plain(const pt& p, const pt& q)         // plain R-class coordinates in, plain R-class coordinates out
{
    auto fp_mul = [](const fe& x, const fe& y) { return ILP ? mont_mul_ilp<FP>(x, y) : mont_mul<FP>(x, y); };
    const fe M = fp_mul(fp_sub(p.Y, p.X), fp_sub(q.Y, q.X));
    const fe P = fp_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
    const fe C = fp_mul(fp_mul(fe_const<FP>(ModP::D_M), p.T), q.T);
    const fe D = fp_mul(p.Z, q.Z);
    const fe E = fe_sub_half<FP>(P, M);
    const fe H = fp_sub(P, E);
    const fe F = fp_sub(D, C);
    const fe r4 = fe_const<FP>(ModP::R4);
    const fe Es = fp_mul(E, r4);                          // E_true R^2
    const fe Gs = fp_mul(fe_add(D, C), r4);               // G_true R^2
    pt r;
    r.X = fp_mul(Es, F);
    r.Y = fp_mul(Gs, H);
    r.Z = fp_mul(F, Gs);
    r.T = fp_mul(Es, H);
    return r;
}
// pt_add_plain(p, p) with the four products of equal operands as squarings (45 instead of 81 limb products each: 144 of the
// 1485 multiply-adds).  Every intermediate is the same residue mod p as in pt_add_plain(p, p) and the outputs are canonicalised
// at the store, so every output limb is the one pt_add_plain(p, p) gives (the reference's Double is `self + self`, edwards.rs:579-592).
template <bool ILP = false>
ZC_DI pt pt_double_plain(const pt& p)
{
    auto fp_mul = [](const fe& x, const fe& y) { return ILP ? mont_mul_ilp<FP>(x, y) : mont_mul<FP>(x, y); };
    auto fp_sqr = [](const fe& x) { return ILP ? mont_sqr_ilp<FP>(x) : mont_sqr<FP>(x); };
    const fe M = fp_sqr(fp_sub(p.Y, p.X));
    const fe P = fp_sqr(fe_add(p.Y, p.X));
    const fe C = fp_mul(fe_const<FP>(ModP::D_M), fp_sqr(p.T));           // d T^2 (pt_add_plain: (d T) T -- the same residue)
    const fe D = fp_sqr(p.Z);
    const fe E = fe_sub_half<FP>(P, M);
    const fe H = fp_sub(P, E);
    const fe F = fp_sub(D, C);
    const fe r4 = fe_const<FP>(ModP::R4);
    const fe Es = fp_mul(E, r4);
    const fe Gs = fp_mul(fe_add(D, C), r4);
    pt r;
    r.X = fp_mul(Es, F);
    r.Y = fp_mul(Gs, H);
    r.Z = fp_mul(F, Gs);
    r.T = fp_mul(Es, H);
    return r;
}
ZC_DI void pt_store_plain(u64* __restrict__ o, const pt& p)                   // plain R-class coordinates -> canonical limbs
{
    u64 l[5];
    fe_to_limbs52(l, fe_cond_sub_n<FP>(fe_cond_sub_n<FP>(p.X))); store5(o, l);
    fe_to_limbs52(l, fe_cond_sub_n<FP>(fe_cond_sub_n<FP>(p.Y))); store5(o + 5, l);
    fe_to_limbs52(l, fe_cond_sub_n<FP>(fe_cond_sub_n<FP>(p.Z))); store5(o + 10, l);
    fe_to_limbs52(l, fe_cond_sub_n<FP>(fe_cond_sub_n<FP>(p.T))); store5(o + 15, l);
}
// A plain coordinate as it comes from memory.  The plain-domain formulas need it R-class (< 3p: fe_sub's subtrahend);
// canonical limbs are.  Any other 5 x 52-bit pattern (value up to 2^260) is first brought there -- into the Montgomery
// domain and back, two multiplications -- on a branch canonical data never takes (top limb < 2^44 means value < 2^252),
// so the kernels answer for the value mod p whatever the limbs, as the Montgomery-domain loads always did.
ZC_DI fe fe_load_plain(const u64* __restrict__ p)
{
    u64 l[5];
    load5(l, p);
    fe r = fe_from_limbs52(l);
    if (l[4] >> 44) r = mont_from<FP>(mont_to<FP>(r));
    return r;
}
ZC_DI pt pt_load_plain(const u64* __restrict__ p)
{
    pt r;
    r.X = fe_load_plain(p);
    r.Y = fe_load_plain(p + 5);
    r.Z = fe_load_plain(p + 10);
    r.T = fe_load_plain(p + 15);
    return r;
}
// The scalar-multiplication loop keeps its two points as (Y-X, Y+X, Z, T): the unified addition
// consumes exactly these combinations of BOTH operands, so forming them once per result instead
// of once per use saves a subtraction and an addition per step.  F and H skip the carry pass
// (fe_sub_lazy): their partners G = D + C and E have limbs < 2^30.  X and Y come back at the end
// through one multiplication by 1/2 each; every field value is the reference's.
struct ptm {
    fe Ym, Yp, Z, T;     // Y - X normalized (< 7N), Y + X lazy (< 6N), Z and T R-class
};
ZC_DI ptm ptm_from_pt(const pt& p)
{
    ptm r;
    r.Ym = fp_sub(p.Y, p.X);
    r.Yp = fe_add(p.Y, p.X);
    r.Z = p.Z;
    r.T = p.T;
    return r;
}
ZC_DI ptm ptm_select(bool c, const ptm& a, const ptm& b)
{
    ptm r;
    r.Ym = fe_select(c, a.Ym, b.Ym);
    r.Yp = fe_select(c, a.Yp, b.Yp);
    r.Z = fe_select(c, a.Z, b.Z);
    r.T = fe_select(c, a.T, b.T);
    return r;
}
// ILP: the multiplier with independent column chains, for launches too small to keep more than
// one wave per SIMD busy (a lone wave cannot hide the serial chain's latency: 15 % faster there).
template <bool ILP = false>
ZC_DI ptm ptm_add(const ptm& p, const ptm& q)          // same values as pt_add
{
    auto fp_mul = [](const fe& x, const fe& y) { return ILP ? mont_mul_ilp<FP>(x, y) : mont_mul<FP>(x, y); };
    const fe M = fp_mul(p.Ym, q.Ym);
    const fe P = fp_mul(p.Yp, q.Yp);
    const fe C = fp_mul(fp_mul(fe_const<FP>(ModP::D_M), p.T), q.T);
    const fe D = fp_mul(p.Z, q.Z);
    const fe E = fe_sub_half<FP>(P, M);
    const fe H = fe_sub_lazy<FP>(P, E);
    const fe F = fe_sub_lazy<FP>(D, C);
    const fe G = fe_add(D, C);
    const fe X3 = fp_mul(E, F), Y3 = fp_mul(G, H);
    ptm r;
    r.Ym = fp_sub(Y3, X3);
    r.Yp = fe_add(Y3, X3);
    r.Z = fp_mul(F, G);
    r.T = fp_mul(E, H);
    return r;
}
ZC_DI pt ptm_to_pt(const ptm& p)
{
    const fe half = fe_const<FP>(ModP::INV2_M);
    pt r;
    r.X = fp_mul(fp_sub(p.Yp, fe_reduce<FP>(p.Ym)), half);     // Ym < 7N is no subtrahend as it stands
    r.Y = fp_mul(fe_add(p.Yp, p.Ym), half);
    r.Z = p.Z;
    r.T = p.T;
    return r;
}
ZC_DI pt pt_neg(const pt& p)                                      // edwards.rs:440-455
{
    pt r;
    r.X = fe_reduce<FP>(fp_neg(p.X));
    r.Y = p.Y;
    r.Z = p.Z;
    r.T = fe_reduce<FP>(fp_neg(p.T));
    return r;
}
ZC_DI pt pt_load(const u64* __restrict__ p)
{
    pt r;
    r.X = fe_load_mont<FP>(p);
    r.Y = fe_load_mont<FP>(p + 5);
    r.Z = fe_load_mont<FP>(p + 10);
    r.T = fe_load_mont<FP>(p + 15);
    return r;
}
ZC_DI void pt_store(u64* __restrict__ o, const pt& p)
{
    fe_store_canon<FP>(o, p.X);
    fe_store_canon<FP>(o + 5, p.Y);
    fe_store_canon<FP>(o + 10, p.Z);
    fe_store_canon<FP>(o + 15, p.T);
}

// ---------------------------------------------------------------- scalar multiplication
// The scalar double_and_add actually multiplies by.  Its loop test `n != Scalar::zero()`
// (edwards.rs:111) compares 32-byte encodings (scalar.rs:78-91) and to_bytes drops limb bits
// >= 256 (backend scalar.rs:477-516), while is_even / half_without_mod (:346, :562-574) walk the
// whole 260-bit pattern v.  The loop therefore runs T iterations, T = the smallest i with
// (v >> i) mod 2^256 == 0, and returns (v mod 2^T) * P.  For v < 2^256 that is bitlen(v), i.e.
// all of v.  For v >= 2^256 with hi = v >> 256 (4 bits): a window [i, i + 256) with i <= 256
// that misses every set bit of hi needs i <= ctz(hi) and low256 < 2^i, so the loop stops early
// -- at T = bitlen(low256), leaving low256 * P -- exactly when low256 < 2^ctz(hi); otherwise it
// runs to bitlen(v) and every bit counts.  Limb bits >= 52 are outside the contract and masked.
ZC_DI void scalar_effective(u64 (&l)[5])
{
#pragma unroll
    for (int j = 0; j < 5; j++) l[j] &= M52;
    const u64 low48 = ((u64)1 << 48) - 1;
    const u32 hi = (u32)(l[4] >> 48);
    if (hi) {
        const int tz = __builtin_ctz(hi);
        const bool stops_early = (l[1] | l[2] | l[3] | (l[4] & low48)) == 0 && l[0] < ((u64)1 << tz);
        if (stops_early) l[4] &= low48;
    }
}
// Scalar operand of Mul<Scalar> / double_and_add semantics: limbs as given, then the rule above.
ZC_DI void load_scalar(u64 (&l)[5], const u64* __restrict__ p)
{
    load5(l, p);
    scalar_effective(l);
}

// Scalar words are read through (sk, stride): word w of this lane is sk[w * stride]
// (LDS with stride = block size in the kernels, a local array with stride 1 on the host).
ZC_DI void scalar_to_words(u32* __restrict__ sk, int stride, const u64 (&l)[5], int& nbits)
{
    u32 w[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int bit = 32 * k, idx = bit / 52, sh = bit % 52;
        u64 x = (idx < 5) ? ((l[idx] & M52) >> sh) : 0;
        if (sh + 32 > 52 && idx + 1 < 5) x |= (l[idx + 1] & M52) << (52 - sh);
        w[k] = (u32)x;
    }
    w[8] &= 0xFu;                                          // 260 bits in total
    nbits = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        if (w[k]) nbits = 32 * k + (32 - __builtin_clz(w[k]));
        sk[k * stride] = w[k];
    }
}

// double_and_add (edwards.rs:102-120), unified-step form: each iteration of the loop below
// evaluates the HWCD formula once for this lane, either Q + N (pending set bit) or N + N.
// Under SIMT the lanes of a wave run the loop in lock step, so a wave performs
// max_lane(bitlen - 1 + popcount) formula evaluations.
template <bool ILP = false>
ZC_DI pt scalar_mul_unified(const pt& P, const u32* __restrict__ sk, int stride, int nbits)
{
    ptm N = ptm_from_pt(P), Q = ptm_from_pt(pt_identity());
    int pos = 0;
    u32 cur = sk[0];
    bool pend = (cur & 1) != 0;
    bool active = nbits > 0;
    while (active) {
        const ptm lhs = ptm_select(pend, Q, N);
        const ptm r = ptm_add<ILP>(lhs, N);
        if (pend) {
            Q = r;
            pend = false;
            active = pos < nbits - 1;
        } else {
            N = r;
            pos++;
            if ((pos & 31) == 0) cur = sk[(pos >> 5) * stride];
            pend = ((cur >> (pos & 31)) & 1) != 0;
        }
    }
    return ptm_to_pt(Q);
}

// Left-to-right variants of the reference, same unified-step machinery with
// Q = Q + Q or Q = Q + (+-P):
//   MODE 1  ltr_bin_mul     (edwards.rs:122-134): for i = 248..0 { Q = 2Q; if bit_i: Q += P }
//   MODE 2  binary_naf_mul  (edwards.rs:136-153): for i = 249..0 { Q = 2Q; +-1 digit: Q +-= P }
// Doubling the literal identity (0,1,1,0) returns the same limbs, so the leading doublings are
// skipped exactly; the first addition identity + (+-P) is performed literally.  `pos_bits` /
// `neg_bits` are bit strings (8 words each): positions where P is added / subtracted.
ZC_DI pt scalar_mul_ltr(const pt& P, const u32* __restrict__ pos_bits, const u32* __restrict__ neg_bits, int stride, int top)
{
    pt Q = pt_identity();
    pt Pn = P;                                             // -P (edwards.rs:440-455), for NAF digits -1
    Pn.X = fe_reduce<FP>(fp_neg(P.X));
    Pn.T = fe_reduce<FP>(fp_neg(P.T));
    int pos = top;
    bool active = top >= 0;
    bool pend = true;
    bool neg = active ? (((neg_bits[(top >> 5) * stride] >> (top & 31)) & 1) != 0) : false;
    while (active) {
        pt rhs = pt_select(neg, Pn, P);
        rhs = pt_select(pend, rhs, Q);
        Q = pt_add(Q, rhs);
        if (pend) {
            pend = false;
            active = pos > 0;
        } else {
            pos--;
            const u32 pw = pos_bits[(pos >> 5) * stride], nw = neg_bits[(pos >> 5) * stride];
            neg = ((nw >> (pos & 31)) & 1) != 0;
            pend = neg || (((pw >> (pos & 31)) & 1) != 0);
            active = pend || pos > 0;
        }
    }
    return Q;
}
// bit strings for the two left-to-right modes from the 5x52 limbs; returns the top index or -1
template <int MODE>
ZC_DI int ltr_digits(u32* __restrict__ pos_bits, u32* __restrict__ neg_bits, int stride, const u64 (&l)[5])
{
    int top = -1;
    if (MODE == 1) {
        // scalar.into_bits() works on to_bytes(): 256 bits; the loop reads bits 248..0 only
        u32 w[9];
        int nb;
        scalar_to_words(w, 1, l, nb);
        w[7] &= 0x01FFFFFFu;                               // bits 224..248
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (w[k]) top = 32 * k + (31 - __builtin_clz(w[k]));
            pos_bits[k * stride] = w[k];
            neg_bits[k * stride] = 0;
        }
        return top;
    }
    // compute_NAF (backend scalar.rs:370-389) step for step on the 260-bit pattern: odd k emits
    // ki = 2 - (k mod 4) and continues with k - Scalar::from(ki), where Scalar::from(-1) = L - 1
    // (:67-82) and Sub adds L back only on a borrow (:210-237) -- so k - (L - 1) is k + 1 for
    // k < L - 1 (the integer NAF: every canonical scalar) and k - L + 1 above; then
    // half_without_mod.  binary_naf_mul reads digits 249..0 only (edwards.rs:144).
    u64 k[5], m[5];
#pragma unroll
    for (int j = 0; j < 5; j++) k[j] = l[j] & M52;
    fe_to_limbs52(m, fe_const<ModL>(ModL::N));
    m[0] -= 1;                                             // L - 1 (L is odd: no borrow)
    u32 pcur = 0, ncur = 0;
    for (int i = 0; i < 256; i++) {
        const bool live = i < 250 && (k[0] | k[1] | k[2] | k[3] | k[4]) != 0;
        if (live && (k[0] & 1)) {
            if ((k[0] & 3) == 1) {
                k[0] -= 1;
                pcur |= 1u << (i & 31);
            } else {
                ncur |= 1u << (i & 31);
                int c = 0;                                 // sign of k - (L - 1), limb-lexicographic from the top
#pragma unroll
                for (int j = 0; j < 5; j++) c = (k[j] > m[j]) ? 1 : ((k[j] < m[j]) ? -1 : c);
                if (c >= 0) {
                    u64 borrow = 0;
#pragma unroll
                    for (int j = 0; j < 5; j++) {
                        borrow = k[j] - (m[j] + (borrow >> 63));
                        k[j] = borrow & M52;
                    }
                } else {
                    u64 carry = 1;
#pragma unroll
                    for (int j = 0; j < 5; j++) {
                        carry += k[j];
                        k[j] = carry & M52;
                        carry >>= 52;
                    }
                }
            }
            top = i;
        }
#pragma unroll
        for (int j = 0; j < 5; j++) k[j] = (k[j] >> 1) | (j < 4 ? (k[j + 1] & 1) << 51 : 0);
        if ((i & 31) == 31) {
            pos_bits[(i >> 5) * stride] = pcur;
            neg_bits[(i >> 5) * stride] = ncur;
            pcur = ncur = 0;
        }
    }
    return top;
}

// ---------------------------------------------------------------- cached points, fast scalar-mul
struct niels {
    fe ymx, ypx, z, t2d;
};
// A cached point is stored as 4 x 256-bit saturated words = 128 bytes = exactly one cache line
// (all four values are < 2^256): the bucket sums gather these records at random, so one line
// per record instead of 2.25 (144-byte records at arbitrary offsets) halves the gather traffic.
ZC_DI void pack256(u32* __restrict__ o, const fe& a)       // normalized limbs -> 8 x u32
{
    u64 w[4];
    fe_to_words256(w, a);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        o[2 * j] = (u32)w[j];
        o[2 * j + 1] = (u32)(w[j] >> 32);
    }
}
ZC_DI fe unpack256(const uint4 lo, const uint4 hi)
{
    const u64 w[4] = {(u64)lo.x | ((u64)lo.y << 32), (u64)lo.z | ((u64)lo.w << 32),
                      (u64)hi.x | ((u64)hi.y << 32), (u64)hi.z | ((u64)hi.w << 32)};
    return fe_from_words256(w);
}
ZC_DI niels niels_from_pt(const pt& p)                 // p R-class
{
    const fe two_d = fe_reduce<FP>(fe_add(fe_const<FP>(ModP::D_M), fe_const<FP>(ModP::D_M)));
    niels q;
    q.ymx = fp_sub(p.Y, p.X);                              // normalized, < 7N < 2^256
    q.ypx = fe_add(p.Y, p.X);
    fe_carry(q.ypx);
    q.z = p.Z;
    q.t2d = fp_mul(p.T, two_d);
    return q;
}
ZC_DI niels niels_identity()
{
    niels q;
    q.ymx = fe_one_m<FP>();
    q.ypx = fe_one_m<FP>();
    q.z = fe_one_m<FP>();
    q.t2d = fe_zero();
    return q;
}
// -q: swap (Y-X, Y+X), negate 2dT
ZC_DI niels niels_cond_neg(bool neg, const niels& q)
{
    niels r;
    r.ymx = fe_select(neg, q.ypx, q.ymx);
    r.ypx = fe_select(neg, q.ymx, q.ypx);
    r.z = q.z;
    r.t2d = fe_select(neg, fe_neg_lazy<FP>(q.t2d), q.t2d);  // q.t2d is R-class (a product); meets the normalized T
    return r;
}
ZC_DI void niels_store(u32* __restrict__ o, const niels& q)
{
    pack256(o, q.ymx);
    pack256(o + 8, q.ypx);
    pack256(o + 16, q.z);
    pack256(o + 24, q.t2d);
}
ZC_DI niels niels_load(const u32* __restrict__ c)
{
    const uint4* v = reinterpret_cast<const uint4*>(c);   // 128-byte aligned record
    niels q;
    q.ymx = unpack256(v[0], v[1]);
    q.ypx = unpack256(v[2], v[3]);
    q.z = unpack256(v[4], v[5]);
    q.t2d = unpack256(v[6], v[7]);
    return q;
}
// p + q, q cached: 8 multiplications (unified and complete for a = -1, d non-square).
// ILP selects the multiplier with independent column chains (mont_mul_ilp) for memory-latency
// bound callers.
// AFFINE: q.z == 1 (a table normalised once), so Z Z' is just Z: 7 multiplications.
template <bool ILP = false, bool AFFINE = false>
ZC_DI pt pt_add_cached(const pt& p, const niels& q)
{
    auto mul = [](const fe& x, const fe& y) { return ILP ? mont_mul_ilp<FP>(x, y) : mont_mul<FP>(x, y); };
    // carries only where a product needs a normalized side: q's fields are normalized (unpacked
    // 29-bit limbs), E and F are; Y - X, D = 2 Z Z', G and H stay lazy
    const fe A = mul(fe_sub_lazy<FP>(p.Y, p.X), q.ymx);
    const fe B = mul(fe_add(p.Y, p.X), q.ypx);
    const fe C = mul(p.T, q.t2d);
    const fe ZZ = AFFINE ? p.Z : mul(p.Z, q.z);
    const fe D = fe_add(ZZ, ZZ);
    const fe E = fp_sub(B, A);
    const fe F = fp_sub(D, C);
    const fe G = fe_add(D, C);                 // limbs < 1.5 * 2^30, meets F and H (< 2^30)
    const fe H = fe_add(B, A);
    pt r;
    r.X = mul(E, F);
    r.Y = mul(G, H);
    r.Z = mul(F, G);
    r.T = mul(E, H);
    return r;
}

// Dedicated doubling, a = -1 (HWCD'08 sec. 3.3 "dbl-2008-hwcd"): 4S + 3M, + 1M when T is wanted.
// Used only where results are compared as group elements or leave as canonical encodings.
template <bool WITH_T, bool ILP = false>
ZC_DI pt pt_double_fast(const pt& p)
{
    auto fp_mul = [](const fe& x, const fe& y) { return ILP ? mont_mul_ilp<FP>(x, y) : mont_mul<FP>(x, y); };
    auto fp_sqr = [](const fe& x) { return ILP ? mont_sqr_ilp<FP>(x) : mont_sqr<FP>(x); };
    const fe A = fp_sqr(p.X);
    const fe B = fp_sqr(p.Y);
    const fe ZZ = fp_sqr(p.Z);
    // E and G are normalized (one carry pass each); F and H skip it: each meets only E or G in a product
    const fe E = fe_sub2<FP>(fp_sqr(fe_add(p.X, p.Y)), A, B);         // 2XY
    const fe G = fp_sub(B, A);                                         // D + B with D = a*A = -A
    const fe F = fe_sub2_lazy<FP>(G, ZZ, ZZ);                          // G - 2Z^2
    const fe H = fe_sub2_lazy<FP>(fe_zero(), A, B);                    // D - B
    pt r;
    r.X = fp_mul(E, F);
    r.Y = fp_mul(G, H);
    r.Z = fp_mul(F, G);
    if (WITH_T) r.T = fp_mul(E, H);
    else r.T = p.T;
    return r;
}

// Signed radix-16 digits of the 260-bit scalar, d_i in [-8, 8), 66 digits (carry included),
// stored as bytes at dig[i * stride]; returns the index of the highest non-zero digit or -1.
ZC_DI int scalar_digits16(int8_t* __restrict__ dig, int stride, const u64 (&l)[5])
{
    u32 w[9];
    int nb;
    {
        u32 tmp[9];
        scalar_to_words(tmp, 1, l, nb);
#pragma unroll
        for (int k = 0; k < 9; k++) w[k] = tmp[k];
    }
    int carry = 0, top = -1;
    for (int i = 0; i < 66; i++) {
        const int word = i >> 3, sh = (i & 7) * 4;
        int d = (word < 9 ? (int)((w[word] >> sh) & 15u) : 0) + carry;
        carry = d >= 8;
        d -= carry << 4;
        dig[i * stride] = (int8_t)d;
        if (d != 0) top = i;
    }
    return top;
}

// Where a lane's table lives.  table_ptr: a plain per-lane pointer (host emulation).  The kernels pass a
// wave-uniform slot base instead (zc_kernels.hip.h: ring_table) and form the lane's offset at every access, so
// that no per-lane pointer stays live in vector registers across the window loop.
struct table_ptr {
    u32* p;
    ZC_DI u32* entry(int j) const { return p + 32 * j; }
};
template <bool ILP, class TABLE>
ZC_DI pt fast_window_loop(const TABLE table, const int8_t* __restrict__ dig, int stride, int top)
{
    pt Q = pt_identity();
    for (int i = top; i >= 0; i--) {
        if (i != top) {
            Q = pt_double_fast<false, ILP>(Q);
            Q = pt_double_fast<false, ILP>(Q);
            Q = pt_double_fast<false, ILP>(Q);
            Q = pt_double_fast<true, ILP>(Q);
        }
        const int d = dig[i * stride];
        const int mag = d < 0 ? -d : d;
        niels c = niels_identity();
        if (mag != 0) c = niels_load(table.entry(mag - 1));
        Q = pt_add_cached<ILP>(Q, niels_cond_neg(d < 0, c));
    }
    return Q;
}
// k * P with fixed signed 4-bit windows over a per-lane table {1P..8P} of cached points in
// global scratch (8 x 128 bytes per point, one cache line per entry).  Uniform control flow:
// every lane of a wave runs the same schedule from window `top` (wave-uniform) down to 0.
// NOT the reference's formula sequence: the result equals double_and_add's as a group element
// (identical affine coordinates / encodings), its (X:Y:Z:T) limbs differ by a projective factor.
template <class TABLE>
ZC_DI pt scalar_mul_fast(const pt& P, const TABLE table, const int8_t* __restrict__ dig, int stride, int top)
{
    // table[j] = (j + 1) P, cached form:
    // one doubling, then a chain of cached additions of P.  (Two live points instead of the four a
    // doubling tree keeps: the build costs the same 55 multiplications and no longer forces spills.)
    {
        const niels c1 = niels_from_pt(P);
        niels_store(table.entry(0), c1);
        pt q = pt_double_fast<true>(P);
        niels_store(table.entry(1), niels_from_pt(q));
#pragma unroll 1
        for (int j = 2; j < 8; j++) {
            q = pt_add_cached(q, c1);
            niels_store(table.entry(j), niels_from_pt(q));
        }
    }
    // small launches (one wave per SIMD) take the independent-chain multiplier, like the strict kernel
    return zc_small_launch() ? fast_window_loop<true>(table, dig, stride, top) : fast_window_loop<false>(table, dig, stride, top);
}

// ---------------------------------------------------------------- byte codecs
// plain 256-bit value <= (p-1)/2 ?  (is_positive on the raw decoded limbs, ristretto.rs:104-114)
ZC_DI bool words256_is_positive(const fe& raw)
{
    // raw.v[8] holds bits 232..255 (24 bits); HALF[8] = 2^19
    u32 borrow = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const u32 s = ModP::HALF[k] - raw.v[k] - borrow;
        borrow = s >> 31;
    }
    return borrow == 0;
}

// Ristretto decompress (ristretto.rs:96-154).  Returns ok; point has Z = 1.
ZC_DI bool ris_decompress(pt& out, const u64 (&w)[4])
{
    const fe raw = fe_from_words256(w);
    const bool s_ok = words256_is_positive(raw);         // also rejects every non-canonical s
    const fe one = fe_one_m<FP>();
    const fe s = mont_to<FP>(raw);
    const fe ss = fp_sqr(s);
    const fe u1 = fp_sub(one, ss);                       // 1 - s^2
    fe u2 = fe_add(one, ss);                             // 1 + s^2 (lazy)
    const fe u2sq = fp_sqr(u2);
    const fe du1sq = fp_mul(fe_const<FP>(ModP::D_M), fp_sqr(u1));
    const fe v = fp_sub(fp_neg(du1sq), u2sq);            // -(d*u1^2) - u2^2  (< 9N)
    fe I;
    const bool was_sq = fp_sqrt_ratio_i(I, one, fp_mul(v, u2sq));
    const fe Dx = fp_mul(I, u2);
    const fe Dy = fp_mul(fp_mul(I, Dx), v);
    const fe x = fp_abs(fp_mul(fe_add(s, s), Dx));
    const fe y = fp_mul(u1, Dy);
    const fe t = fp_mul(x, y);
    const bool t_pos = fp_is_positive(t);
    const bool y_zero = fp_is_zero(y);
    out.X = x;
    out.Y = y;
    out.Z = one;
    out.T = t;
    return s_ok && was_sq && t_pos && !y_zero;
}

// Ristretto compress (ristretto.rs:398-425) -> canonical s as plain limbs
ZC_DI fe ris_compress(const pt& p)
{
    const fe u1 = fp_mul(fe_add(p.Z, p.Y), fp_sub(p.Z, p.Y));
    const fe u2 = fp_mul(p.X, p.Y);
    fe I;
    (void)fp_sqrt_ratio_i(I, fe_one_m<FP>(), fp_mul(u1, fp_sqr(u2)));
    const fe D1 = fp_mul(u1, I);
    const fe D2 = fp_mul(u2, I);
    const fe Zinv = fp_mul(fp_mul(D1, D2), p.T);
    const bool rotate = !fp_is_positive(fp_mul(p.T, Zinv));
    const fe i_m = fe_const<FP>(ModP::SQRT_M1_M);
    const fe xr = fp_mul(i_m, p.Y), yr = fp_mul(i_m, p.X);
    const fe Dr = fp_mul(D1, fe_const<FP>(ModP::INV_SQRT_A_MINUS_D_M));
    const fe x = fe_select(rotate, xr, p.X);
    fe y = fe_select(rotate, yr, p.Y);
    const fe D = fe_select(rotate, Dr, D2);
    const bool negy = !fp_is_positive(fp_mul(x, Zinv));
    y = fe_select(negy, fe_reduce<FP>(fp_neg(y)), y);
    const fe s = fp_mul(fp_sub(p.Z, y), D);
    const fe sc = fp_canon(s);
    const bool pos = fe_is_positive_canon<FP>(sc);
    return fe_select(pos, sc, fe_n_minus_canon<FP>(sc));   // |s| = p - s when negative
}

// Ristretto equality (ristretto.rs:166-176)
ZC_DI bool ris_eq(const pt& a, const pt& b)
{
    const bool e1 = fp_eq(fp_mul(a.X, b.Y), fp_mul(a.Y, b.X));
    const bool e2 = fp_eq(fp_mul(a.X, b.X), fp_mul(a.Y, b.Y));
    return e1 || e2;
}

// Edwards -> affine (edwards.rs:1071-1092).  ok = Z != 0 (the reference panics there).
ZC_DI bool ed_to_affine(fe& x, fe& y, const pt& p)
{
    const fe zi = fp_invert(p.Z);
    x = fp_mul(p.X, zi);
    y = fp_mul(p.Y, zi);
    return !fp_is_zero(p.Z);
}
// One lane's share of a batched affine conversion: Montgomery's trick over the Z coordinates of
// up to `c` points lo, lo + stride, ... (as fe_invert_chunk: plain limbs are used as Montgomery residues, prefix
// products wait in the first 36 bytes of each 80-byte output record), then x = X/Z, y = Y/Z.
// Z = 0 (the reference's inverse panics) takes the neutral value and yields (0, 0) / ok = 0.
ZC_DI void ed_to_affine_chunk(const u64* p, u64* xy, uint8_t* ok, size_t n, size_t lo, size_t stride, int c)
{
    const size_t avail = (n - lo + stride - 1) / stride;
    const int cnt = (int)(avail < (size_t)c ? avail : (size_t)c);
    const fe neutral = fe_one_m<FP>();
    fe acc = neutral;
    for (int j = 0; j < cnt; j++) {
        u64 l[5];
        load5(l, p + 20 * (lo + (size_t)j * stride) + 10);
        const fe z = fe_select(limbs52_all_zero(l), neutral, fe_from_limbs52(l));
        u32* slot = reinterpret_cast<u32*>(xy + 10 * (lo + (size_t)j * stride));
#pragma unroll
        for (int w = 0; w < 9; w++) slot[w] = acc.v[w];
        acc = fp_mul(acc, z);
    }
    fe inv = fp_inverse_of_register(acc);
    for (int j = cnt - 1; j >= 0; j--) {
        u64 lx[5], ly[5], lz[5], r[5];
        load5(lx, p + 20 * (lo + (size_t)j * stride));
        load5(ly, p + 20 * (lo + (size_t)j * stride) + 5);
        load5(lz, p + 20 * (lo + (size_t)j * stride) + 10);
        const bool zero = limbs52_all_zero(lz);
        const fe z = fe_select(zero, neutral, fe_from_limbs52(lz));
        const u32* slot = reinterpret_cast<const u32*>(xy + 10 * (lo + (size_t)j * stride));
        fe pre;
#pragma unroll
        for (int w = 0; w < 9; w++) pre.v[w] = slot[w];
        const fe zinv = mont_to<FP>(fp_mul(inv, pre));      // (1/Z) R
        inv = fp_mul(inv, z);
        fe_to_limbs52(r, fe_cond_sub_n<FP>(fe_cond_sub_n<FP>(fp_mul(fe_from_limbs52(lx), zinv))));
        if (zero) r[0] = r[1] = r[2] = r[3] = r[4] = 0;
        store5(xy + 10 * (lo + (size_t)j * stride), r);
        fe_to_limbs52(r, fe_cond_sub_n<FP>(fe_cond_sub_n<FP>(fp_mul(fe_from_limbs52(ly), zinv))));
        if (zero) r[0] = r[1] = r[2] = r[3] = r[4] = 0;
        store5(xy + 10 * (lo + (size_t)j * stride) + 5, r);
        if (ok) ok[lo + (size_t)j * stride] = zero ? 0 : 1;
    }
}
// Edwards equality = affine equality (edwards.rs:360-364, 1044-1048), cross-multiplied
ZC_DI bool ed_eq(const pt& a, const pt& b)
{
    const bool ex = fp_eq(fp_mul(a.X, b.Z), fp_mul(b.X, a.Z));
    const bool ey = fp_eq(fp_mul(a.Y, b.Z), fp_mul(b.Y, a.Z));
    return ex && ey && !fp_is_zero(a.Z) && !fp_is_zero(b.Z);
}
// Edwards compress (edwards.rs:613-629 + find_xx :200-204): y | sign << 255 where
// sign = (mod_sqrt(xx, 0) != x), xx = (y^2 - 1)/(d y^2 + 1) = u/v with u = Y^2 - Z^2,
// v = d Y^2 + Z^2.  ONE exponentiation serves the square root and the inversion of Z:
//   E  = (u v^7 Z^8)^((p-5)/8) = a^((q-1)/2) (v Z)^-4          (a = u/v, q = (p-1)/4)
//   x0 = u v^3 Z^4 E = a^((q+1)/2)       -- the value mod_sqrt starts from (field.rs:410)
//   t  = a^q = v x0^2 / u in {1, -1}     -- -1: root is x0 * 6^q (Tonelli-Shanks correction)
//   E x0 = t (v Z)^-4   =>   1/Z = t E x0 v^4 Z^3
// and mod_sqrt(xx) == X/Z is tested as r Z == X.  u == 0 (x = 0: the identity and (0,-1)) has
// a^q = 0, there y = +-1 is read off Y == +-Z.  Same bytes and ok mask as the reference's
// affine conversion + find_xx + Tonelli-Shanks; failures (Z = 0, v = 0, non-residue) -> false.
ZC_DI bool ed_compress(u64 (&w)[4], const pt& p)
{
    const fe Y2 = fp_sqr(p.Y), Z2 = fp_sqr(p.Z);
    const fe u = fe_reduce<FP>(fp_sub(Y2, Z2));
    const fe v = fe_reduce<FP>(fe_add(fp_mul(fe_const<FP>(ModP::D_M), Y2), Z2));
    // everything that outlives the exponentiation is folded into two products first
    fe w1, w2, base;
    {
        const fe Z4 = fp_sqr(Z2);
        const fe v2 = fp_sqr(v), v3 = fp_mul(v2, v), v4 = fp_sqr(v2);
        w1 = fp_mul(fp_mul(u, v3), Z4);                               // x0 = w1 E
        w2 = fp_mul(v4, fp_mul(Z2, p.Z));                             // 1/Z = t E x0 w2
        base = fp_mul(fp_mul(w1, v4), Z4);                            // u v^7 Z^8
    }
    const fe E = fp_pow_p58(base);
    const fe x0 = fp_mul(w1, E);
    const fe check = fp_mul(v, fp_sqr(x0));
    const bool t_is_one = fe_eq_canon(fp_canon(check), fp_canon(u));
    const bool t_is_m1 = fe_is_zero_canon(fp_canon(fe_add(check, u)));
    const fe r = fe_select(t_is_one, x0, fp_mul(x0, fe_const<FP>(ModP::SIX_POW_Q_M)));
    fe zinv = fp_mul(fp_mul(E, x0), w2);
    zinv = fe_select(t_is_one, zinv, fe_reduce<FP>(fp_neg(zinv)));
    fe y = fp_mul(p.Y, zinv);
    const bool u_zero = fp_is_zero(u);
    y = fe_select(u_zero, fe_select(fp_eq(p.Y, p.Z), fe_one_m<FP>(), fe_const<FP>(ModP::MINUS_ONE_M)), y);
    const bool sign = !fp_eq(fp_mul(r, p.Z), p.X);
    fe_to_words256(w, fp_canon(y));
    w[3] |= (u64)(sign ? 1 : 0) << 63;
    return (t_is_one || t_is_m1) && !fp_is_zero(p.Z) && !fp_is_zero(v);
}
// Tonelli-Shanks value of u/v with ONE exponentiation and no inversion: for p = 5 (mod 8),
// (u v^3)(u v^7)^((p-5)/8) = (u/v)^((p+3)/8) = a^((q+1)/2) with a = u/v, q = (p-1)/4 -- exactly
// the x the reference's mod_sqrt starts from (field.rs:410) -- and a^q = x^2 v / u, so the
// correction by 6^q applies when v x^2 == -u.  Returns false where the reference returns None
// (non-residue) ; v == 0 is reported separately (the reference's Div panics).  u, v R-class.
ZC_DI bool fp_ts_sqrt_ratio(fe& x, const fe& u, const fe& v)
{
    const fe v3 = fp_mul(fp_sqr(v), v);
    const fe v7 = fp_mul(fp_sqr(v3), v);
    const fe x0 = fp_mul(fp_mul(u, v3), fp_pow_p58(fp_mul(u, v7)));
    const fe check = fp_mul(v, fp_sqr(x0));
    const bool t_is_one = fe_eq_canon(fp_canon(check), fp_canon(u));          // also u == 0 -> Some(0)
    const bool t_is_m1 = fe_is_zero_canon(fp_canon(fe_add(check, u)));
    x = fe_select(t_is_one, x0, fp_mul(x0, fe_const<FP>(ModP::SIX_POW_Q_M)));
    return t_is_one || t_is_m1;
}
// Edwards decompress (edwards.rs:313-326, :962-979, :402-417): byte 31 masked with 0x0F
ZC_DI bool ed_decompress(pt& out, const u64 (&win)[4])
{
    const bool sign = (win[3] >> 63) != 0;
    u64 w[4] = {win[0], win[1], win[2], win[3] & 0x0FFFFFFFFFFFFFFFull};
    const fe one = fe_one_m<FP>();
    const fe y = mont_to<FP>(fe_from_words256(w));
    const fe yy = fp_sqr(y);
    const fe num = fe_reduce<FP>(fp_sub(yy, one));                            // y^2 - 1
    const fe den = fe_reduce<FP>(fe_add(fp_mul(fe_const<FP>(ModP::D_M), yy), one));   // d y^2 + 1
    fe r;
    const bool have = fp_ts_sqrt_ratio(r, num, den);
    const fe x = fe_select(sign, fe_reduce<FP>(fp_neg(r)), r);                // mod_sqrt(xx, sign), field.rs:435-439
    out.X = x;
    out.Y = y;
    out.Z = one;
    out.T = fp_mul(x, y);
    return have && !fp_is_zero(den);
}

// ---------------------------------------------------------------- "next" rows
// Ristretto-flavoured Elligator map (ristretto.rs:430-471).  r0 is used as given (the
// reference's own KAT feeds a non-canonical 253-bit value), everything is value arithmetic.
ZC_DI pt ris_elligator(const fe& r0)
{
    const fe one = fe_one_m<FP>();
    const fe d = fe_const<FP>(ModP::D_M);
    const fe minus_one = fe_const<FP>(ModP::MINUS_ONE_M);
    const fe r = fp_mul(fe_const<FP>(ModP::SQRT_M1_M), fp_sqr(r0));
    const fe Ns = fp_mul(fe_add(r, one), fe_const<FP>(ModP::ONE_MINUS_D_SQ_M));
    const fe D = fp_mul(fp_sub(minus_one, fp_mul(d, r)), fe_add(r, d));
    fe s;
    const bool is_sq = fp_sqrt_ratio_i(s, Ns, D);
    fe sp = fp_mul(s, r0);
    sp = fe_select(fp_is_positive(sp), fe_reduce<FP>(fp_neg(sp)), sp);      // s' = -|s * r0|
    s = fe_select(is_sq, s, sp);
    const fe c = fe_select(is_sq, minus_one, r);
    const fe Nt = fp_sub(fp_mul(fp_mul(c, fp_sub(r, one)), fe_const<FP>(ModP::D_MINUS_ONE_SQ_M)), D);
    const fe ssq = fp_sqr(s);
    const fe W0 = fp_mul(fe_add(s, s), D);
    const fe W1 = fp_mul(Nt, fe_const<FP>(ModP::SQRT_AD_MINUS_ONE_M));
    const fe W2 = fp_sub(one, ssq);
    const fe W3 = fe_add(one, ssq);
    pt o;
    o.X = fp_mul(W0, W3);
    o.Y = fp_mul(W2, W1);
    o.Z = fp_mul(W1, W3);
    o.T = fp_mul(W0, W2);
    return o;
}
// curve equation in projective form (edwards.rs:733-748): (a X^2 + Y^2) Z^2 == Z^4 + d X^2 Y^2
ZC_DI bool ed_is_valid(const pt& p)
{
    const fe xs = fp_sqr(p.X), ys = fp_sqr(p.Y), zs = fp_sqr(p.Z);
    const fe left = fp_mul(fp_sub(ys, xs), zs);                              // a = -1
    const fe right = fe_add(fp_sqr(zs), fp_mul(fp_mul(fe_const<FP>(ModP::D_M), xs), ys));
    return fe_is_zero_canon(fp_canon(fp_sub(right, left)));
}
// ProjectivePoint (X:Y:Z) add / double (edwards.rs:809-834, :915-942)
struct ppt {
    fe X, Y, Z;
};
ZC_DI ppt proj_add(const ppt& p, const ppt& q)
{
    const fe A = fp_mul(p.Z, q.Z);
    const fe B = fp_sqr(A);
    const fe C = fp_mul(p.X, q.X);
    const fe D = fp_mul(p.Y, q.Y);
    const fe E = fp_mul(fp_mul(fe_const<FP>(ModP::D_M), C), D);
    const fe F = fp_sub(B, E);
    const fe G = fe_add(B, E);
    fe t = fp_mul(fe_add(p.X, p.Y), fe_add(q.X, q.Y));
    t = fp_sub(fp_sub(t, C), D);
    ppt r;
    r.X = fp_mul(A, fp_mul(F, t));
    r.Y = fp_mul(fp_mul(A, G), fe_add(D, C));
    r.Z = fp_mul(F, G);
    return r;
}
ZC_DI ppt proj_double(const ppt& p)
{
    const fe B = fp_sqr(fe_add(p.X, p.Y));
    const fe C = fp_sqr(p.X);
    const fe D = fp_sqr(p.Y);
    const fe F = fp_sub(D, C);                                               // E + D with E = a*C = -C
    const fe H = fp_sqr(p.Z);
    const fe J = fp_sub(fp_sub(F, H), H);                                    // F - 2H
    const fe EmD = fp_sub(fp_neg(C), D);                                     // E - D = -C - D
    ppt r;
    r.X = fp_mul(fp_sub(fp_sub(B, C), D), J);
    r.Y = fp_mul(F, EmD);
    r.Z = fp_mul(F, J);
    return r;
}

}  // namespace zc

// zc_arith.hip.h -- device-side modular arithmetic for the Sonny/Doppio field (mod p)
// and the scalar field (mod L), one element per lane.
//
// Representation: nine 29-bit limbs in 32-bit VGPRs ("radix 2^29"), Montgomery
// form with R = 2^261.  Why this shape on gfx950:
//   * the only wide integer multiplier is v_mad_u64_u32 (32x32+64 -> 64, no
//     carry-in); with 29-bit limbs a 64-bit column accumulator absorbs all nine
//     partial products AND the Montgomery correction terms without a single
//     carry instruction, so the multiplier is pure `acc = mad(a_i, b_j, acc)`;
//   * three spare bits per limb make add lazy (nine v_add_u32, no carries) and
//     261-252 = 9 spare bits of R remove every conditional subtraction from the
//     hot loops (bounds below);
//   * p = 2^252 + c and L = 2^249 + c' have five non-zero low limbs and a single
//     top bit, so a reduction step costs 6 mads;
//   * the multiplier walks the 17 columns in order and every column's chain starts from
//     the carry of the previous one (mont_mul), so carries cost one 64-bit shift each.
// The reference keeps canonical radix-2^52 limbs and does two Montgomery passes
// per Mul (src/backend/u64/field.rs:250-262, :741-813); every reference op
// returns the canonical representative, so any exact modular algorithm followed
// by canonicalisation yields identical limbs (SURVEY 8a note P).
//
// Bounds ("R-class" = value < 3N with limbs 0..7 < 2^29; "lazy" = value < 32N,
// limbs < 2^30):
//   mont_mul / mont_sqr : inputs lazy  -> output R-class   ((32N)^2/R + N < 3N)
//   fe_add              : inputs with limbs < 2^29 -> lazy (limbs < 2^30)
//   fe_sub(a, b)        : a lazy, b R-class -> normalized, value < a + 4N
// Column sums: 9 * 2^30 * 2^30 + 5 * 2^58 + 2^49 + carry < 2^63.4 < 2^64.
#pragma once
#include <hip/hip_runtime.h>
#include "zc_constants.hip.h"

namespace zc {

#define ZC_DI __device__ __forceinline__
#if defined(__HIP_DEVICE_COMPILE__)
#define ZC_OPAQUE(x) asm volatile("" : "+s"(x))   // value the optimiser cannot see through (SGPR)
#else
#define ZC_OPAQUE(x) asm volatile("" : "+r"(x))
#endif
// Pins a running 64-bit sum: LLVM's reassociation otherwise re-orders a column's additions so
// that the carry from the previous column is added last with a separate 64-bit add; with the
// sum pinned after every term each term stays one v_mad_u64_u32 on the running value.  Emits
// no instruction.
#if defined(__HIP_DEVICE_COMPILE__)
#define ZC_PIN(x) asm("" : "+v"(x))
#else
#define ZC_PIN(x) asm("" : "+r"(x))
#endif
// Invariant checks of the lazy-reduction scheme: compiled only into the host emulation of the
// test tier (tests/emul, -DZC_CHECK_BOUNDS); nothing on the device.
#if defined(ZC_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
extern "C" void zc_bound_fail(const char* what, int line);
#define ZC_ASSERT(cond) do { if (!(cond)) zc_bound_fail(#cond, __LINE__); } while (0)
#else
#define ZC_ASSERT(cond) do { } while (0)
#endif
constexpr u32 M29 = 0x1fffffffu;
constexpr u64 M52 = (1ull << 52) - 1;

struct fe {
    u32 v[9];
};

// 9 products of the largest limbs + 6 reduction terms + carry stay below 2^64 (checked builds only)
#if defined(ZC_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
inline bool fe_columns_fit(const fe& a, const fe& b)
{
    u64 ma = 0, mb = 0;
    for (int i = 0; i < 9; i++) {
        if (a.v[i] > ma) ma = a.v[i];
        if (b.v[i] > mb) mb = b.v[i];
    }
    const unsigned __int128 worst = (unsigned __int128)9 * ma * mb + ((unsigned __int128)3 << 59);
    return (worst >> 64) == 0;
}
#endif
template <class F>
ZC_DI fe fe_const(const u32 (&c)[9])
{
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = c[i];
    return r;
}
ZC_DI fe fe_zero()
{
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = 0;
    return r;
}
template <class F>
ZC_DI fe fe_one_m() { return fe_const<F>(F::ONE); }

// ---------------------------------------------------------------- Montgomery core
// Reduce the 18 column accumulators t[] (value < 2^522) to t/R mod N, R-class.
template <class F>
ZC_DI void mont_reduce_cols(fe& r, u64 (&t)[18])
{
    // N[8] = 2^TOPSHIFT enters as an opaque register so that m * N[8] stays one
    // v_mad_u64_u32 (5.1 cycles) instead of a 64-bit shift plus a 64-bit add (9.3)
    u32 ntop = 1u << F::TOPSHIFT;
    ZC_OPAQUE(ntop);
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const u32 m = ((u32)t[k] * F::NP) & M29;
        t[k] += (u64)m * F::N[0];
        t[k + 1] += (u64)m * F::N[1];
        t[k + 2] += (u64)m * F::N[2];
        t[k + 3] += (u64)m * F::N[3];
        t[k + 4] += (u64)m * F::N[4];
        t[k + 8] += (u64)m * ntop;           // N[5..7] == 0, N[8] == 1 << TOPSHIFT
        t[k + 1] += t[k] >> 29;              // low 29 bits of t[k] are now zero
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
        r.v[k - 9] = (u32)t[k] & M29;
        t[k + 1] += t[k] >> 29;
    }
    r.v[8] = (u32)t[17];
}

// Montgomery multiplication column by column (finely integrated product scanning): the carry
// out of column k is the starting value of column k + 1's multiply-accumulate chain, every term
// is one v_mad_u64_u32 on the running sum and a column costs no separate 64-bit add.
// `ZC_MONT_COLUMNS(PRODUCTS)` expands to the 17 columns; PRODUCTS(k) adds the column's partial
// products to `col` (pinned after every term, see ZC_PIN).
#define ZC_MONT_COLUMNS(PRODUCTS)                                                \
    u32 ntop = 1u << F::TOPSHIFT;                                                \
    ZC_OPAQUE(ntop);                                                             \
    u32 m[9];                                                                    \
    fe r;                                                                        \
    u64 col = 0;                                                                 \
    _Pragma("unroll") for (int k = 0; k < 17; k++) {                             \
        PRODUCTS(k)                                                              \
        _Pragma("unroll") for (int j = 0; j < 9; j++) {                          \
            if (j < k && k - j <= 4) {                                           \
                col += (u64)m[j] * F::N[k - j];                                  \
                ZC_PIN(col);                                                     \
            }                                                                    \
            if (k - j == 8) {       /* N[5..7] == 0, N[8] == 1 << TOPSHIFT */     \
                col += (u64)m[j] * ntop;                                         \
                ZC_PIN(col);                                                     \
            }                                                                    \
        }                                                                        \
        if (k < 9) {                                                             \
            m[k] = ((u32)col * F::NP) & M29;                                     \
            col += (u64)m[k] * F::N[0];                                          \
        } else {                                                                 \
            r.v[k - 9] = (u32)col & M29;                                         \
        }                                                                        \
        col >>= 29;                                                              \
    }                                                                            \
    r.v[8] = (u32)col;                                                           \
    return r;

// r = a * b / R mod N   (reference: mul_internal + montgomery_reduce, field.rs:741-813,
// scalar.rs:580-652, with R = 2^261 instead of 2^260)
template <class F>
ZC_DI fe mont_mul(const fe& a, const fe& b)
{
    ZC_ASSERT(fe_columns_fit(a, b));                                                      // columns < 2^64
    ZC_ASSERT((u64)(a.v[8] + 1) * (b.v[8] + 1) <= (1ull << 50));                          // a b < 2 R N: result < 3N
#define ZC_MUL_PRODUCTS(k)                                                       \
    _Pragma("unroll") for (int i = 0; i < 9; i++)                                \
        if (k - i >= 0 && k - i < 9) {                                           \
            col += (u64)a.v[i] * b.v[k - i];                                     \
            ZC_PIN(col);                                                         \
        }
    ZC_MONT_COLUMNS(ZC_MUL_PRODUCTS)
#undef ZC_MUL_PRODUCTS
}

// r = a * a / R mod N   (reference: square_internal, field.rs:763-777): 45 products
template <class F>
ZC_DI fe mont_sqr(const fe& a)
{
    for (int i = 0; i < 9; i++) ZC_ASSERT(a.v[i] < (1u << 30));
    ZC_ASSERT(a.v[8] < (1u << 25));
    u32 d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;   // limbs < 2^30 -> < 2^31
#define ZC_SQR_PRODUCTS(k)                                                       \
    _Pragma("unroll") for (int i = 0; i < 9; i++) {                              \
        if (k - i > i && k - i < 9) {                                            \
            col += (u64)d[i] * a.v[k - i];                                       \
            ZC_PIN(col);                                                         \
        }                                                                        \
        if (k - i == i) {                                                        \
            col += (u64)a.v[i] * a.v[i];                                         \
            ZC_PIN(col);                                                         \
        }                                                                        \
    }
    ZC_MONT_COLUMNS(ZC_SQR_PRODUCTS)
#undef ZC_SQR_PRODUCTS
}

// The same product with 18 independent column accumulators (product scanning, then
// mont_reduce_cols): more 64-bit adds, but independent chains inside one wave.  For kernels whose
// waves mostly wait on memory (the MSM bucket accumulation), where the serial column chain of
// mont_mul is exposed: measured 15 % faster there, 3 % slower in the VALU-bound kernels.
template <class F>
ZC_DI fe mont_mul_ilp(const fe& a, const fe& b)
{
    ZC_ASSERT(fe_columns_fit(a, b));
    ZC_ASSERT((u64)(a.v[8] + 1) * (b.v[8] + 1) <= (1ull << 50));
    u64 t[18];
#pragma unroll
    for (int k = 0; k < 18; k++) t[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) t[i + j] += (u64)a.v[i] * b.v[j];
    fe r;
    mont_reduce_cols<F>(r, t);
    return r;
}

template <class F>
ZC_DI fe mont_sqr_ilp(const fe& a)
{
    for (int i = 0; i < 9; i++) ZC_ASSERT(a.v[i] < (1u << 30));
    u64 t[18];
#pragma unroll
    for (int k = 0; k < 18; k++) t[k] = 0;
    u32 d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        t[2 * i] += (u64)a.v[i] * a.v[i];
#pragma unroll
        for (int j = i + 1; j < 9; j++) t[i + j] += (u64)d[i] * a.v[j];
    }
    fe r;
    mont_reduce_cols<F>(r, t);
    return r;
}

// r = a / R mod N  (from Montgomery form; reference from_montgomery, field.rs:830-836)
template <class F>
ZC_DI fe mont_from(const fe& a)
{
    u64 t[18];
#pragma unroll
    for (int k = 0; k < 9; k++) t[k] = a.v[k];
#pragma unroll
    for (int k = 9; k < 18; k++) t[k] = 0;
    fe r;
    mont_reduce_cols<F>(r, t);
    return r;
}

template <class F>
ZC_DI fe mont_to(const fe& a) { return mont_mul<F>(a, fe_const<F>(F::RR)); }

// ---------------------------------------------------------------- add / sub / normalize
ZC_DI void fe_carry(fe& a)
{
#pragma unroll
    for (int k = 0; k < 8; k++) {
        a.v[k + 1] += a.v[k] >> 29;
        a.v[k] &= M29;
    }
}
// lazy add: limbs of a, b < 2^29 (+ small top limb) -> limbs < 2^30, no carries
ZC_DI fe fe_add(const fe& a, const fe& b)
{
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
// a - b + 4N, normalized.  b must be R-class (value < 3N, limbs 0..7 < 2^29).
template <class F>
ZC_DI fe fe_sub(const fe& a, const fe& b)
{
    for (int i = 0; i < 9; i++) ZC_ASSERT(b.v[i] <= F::BIAS[i] && a.v[i] < (1u << 31));
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + (F::BIAS[i] - b.v[i]);
    fe_carry(r);
    return r;
}
// a - b + 4N with NO carry pass: limbs < 2^29 + 2^30.  Only as a multiplier operand whose partner
// has limbs < 2^30 (9 * 1.5 * 2^60 + reduction terms < 2^64).  a: limbs < 2^29; b R-class.
template <class F>
ZC_DI fe fe_sub_lazy(const fe& a, const fe& b)
{
    for (int i = 0; i < 9; i++) ZC_ASSERT(b.v[i] <= F::BIAS[i] && a.v[i] < (1u << 29) + (i == 8 ? (1u << 29) : 0));
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + (F::BIAS[i] - b.v[i]);
    return r;
}
// a - b - c + 8N with one carry pass (b, c R-class): normalized, value < a + 8N.
template <class F>
ZC_DI fe fe_sub2(const fe& a, const fe& b, const fe& c)
{
    for (int i = 0; i < 9; i++) ZC_ASSERT(b.v[i] <= F::BIAS[i] && c.v[i] <= F::BIAS[i] && a.v[i] < (1u << 30));
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + (F::BIAS[i] - b.v[i]) + (F::BIAS[i] - c.v[i]);
    fe_carry(r);
    return r;
}
// the same without the carry pass: limbs < 2^29 + 2^31.  Multiplier operand only, and only against
// a normalized partner (limbs < 2^29): 9 * 2.5 * 2^59 + reduction terms < 2^64.  a: limbs < 2^29.
template <class F>
ZC_DI fe fe_sub2_lazy(const fe& a, const fe& b, const fe& c)
{
    for (int i = 0; i < 9; i++) ZC_ASSERT(b.v[i] <= F::BIAS[i] && c.v[i] <= F::BIAS[i] && a.v[i] < (1u << 29) + (i == 8 ? (1u << 29) : 0));
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + (F::BIAS[i] - b.v[i]) + (F::BIAS[i] - c.v[i]);
    return r;
}
// 4N - b without the carry pass (limbs < 2^30): multiplier operand only.  b R-class.
template <class F>
ZC_DI fe fe_neg_lazy(const fe& b)
{
    for (int i = 0; i < 9; i++) ZC_ASSERT(b.v[i] <= F::BIAS[i]);
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = F::BIAS[i] - b.v[i];
    return r;
}
// (a - b) / 2 mod N, normalized: a - b + 4N, plus N when that is odd (N is odd, and the parity
// of the whole value is the parity of limb 0), then one exact right shift.  a, b < 2N with
// limbs 0..7 < 2^29 (products of operands below 8N): result < 3.5N, top limb below BIAS[8], so it
// may stand on either side of a later fe_sub.
template <class F>
ZC_DI fe fe_sub_half(const fe& a, const fe& b)
{
    for (int i = 0; i < 9; i++) ZC_ASSERT(b.v[i] <= F::BIAS[i] && a.v[i] < (1u << 30));
    ZC_ASSERT(a.v[8] < (2u << F::TOPSHIFT) + 2 && b.v[8] < (2u << F::TOPSHIFT) + 2);      // a, b < 2N
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + (F::BIAS[i] - b.v[i]);
    const u32 odd = 0u - (r.v[0] & 1u);
#pragma unroll
    for (int i = 0; i < 9; i++)
        if (F::N[i] != 0) r.v[i] += F::N[i] & odd;
    fe_carry(r);
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (r.v[i] >> 1) | ((r.v[i + 1] & 1u) << 28);
    r.v[8] >>= 1;
    ZC_ASSERT(r.v[8] <= F::BIAS[8]);
    return r;
}
template <class F>
ZC_DI fe fe_neg(const fe& b)
{
    for (int i = 0; i < 9; i++) ZC_ASSERT(b.v[i] <= F::BIAS[i]);
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = F::BIAS[i] - b.v[i];
    fe_carry(r);
    return r;
}
// bring any lazy value back to R-class (one multiplication by R mod N)
template <class F>
ZC_DI fe fe_reduce(const fe& a) { return mont_mul<F>(a, fe_one_m<F>()); }

ZC_DI fe fe_select(bool c, const fe& a, const fe& b)   // c ? a : b
{
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}

// ---------------------------------------------------------------- canonical form
// x normalized with value <= N+small  ->  unique representative in [0, N)
template <class F>
ZC_DI fe fe_cond_sub_n(const fe& x)
{
    fe d;
    u32 borrow = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const u32 s = x.v[k] - F::N[k] - borrow;
        borrow = s >> 31;
        d.v[k] = s & M29;
    }
    const u32 s8 = x.v[8] - F::N[8] - borrow;
    d.v[8] = s8;
    const bool neg = (s8 >> 31) != 0;
    return fe_select(neg, x, d);
}
// Montgomery-form (lazy ok) -> canonical plain value, limbs normalized
template <class F>
ZC_DI fe fe_canon_from_mont(const fe& a)
{
    fe t = a;
    fe_carry(t);
    return fe_cond_sub_n<F>(mont_from<F>(t));
}
// plain (non-Montgomery) normalized value < 2N -> canonical
template <class F>
ZC_DI fe fe_canon_plain(const fe& a) { return fe_cond_sub_n<F>(a); }

// N - c for a plain canonical c in [0, N]  (normalized limbs)
template <class F>
ZC_DI fe fe_n_minus_canon(const fe& c)
{
    fe d;
    u32 borrow = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const u32 s = F::N[k] - c.v[k] - borrow;
        borrow = s >> 31;
        d.v[k] = s & M29;
    }
    d.v[8] = F::N[8] - c.v[8] - borrow;
    return d;
}

ZC_DI bool fe_is_zero_canon(const fe& c)
{
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= c.v[i];
    return o == 0;
}
ZC_DI bool fe_eq_canon(const fe& a, const fe& b)
{
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}
// reference is_positive (field.rs:552-557): canonical value <= (N-1)/2
template <class F>
ZC_DI bool fe_is_positive_canon(const fe& c)
{
    u32 borrow = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const u32 s = F::HALF[k] - c.v[k] - borrow;
        borrow = s >> 31;
    }
    return borrow == 0;
}

// ---------------------------------------------------------------- radix 2^52 <-> 2^29
// five 52-bit limbs (reference layout, FieldElement([u64;5])) -> nine 29-bit limbs
ZC_DI fe fe_from_limbs52(const u64 (&l)[5])
{
    fe r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int bit = 29 * k;
        const int idx = bit / 52, sh = bit % 52;
        u64 x = (l[idx] & M52) >> sh;
        if (sh + 29 > 52 && idx + 1 < 5) x |= (l[idx + 1] & M52) << (52 - sh);
        r.v[k] = (k < 8) ? ((u32)x & M29) : (u32)x;       // limb 8 keeps bits 232..259
    }
    return r;
}
// canonical nine 29-bit limbs -> five 52-bit limbs
ZC_DI void fe_to_limbs52(u64 (&l)[5], const fe& c)
{
#pragma unroll
    for (int j = 0; j < 5; j++) {
        u64 acc = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int lo = 29 * k - 52 * j;               // bit position of limb k inside limb52 j
            if (lo > -29 && lo < 52) {
                if (lo >= 0) acc |= (u64)c.v[k] << lo;
                else acc |= (u64)c.v[k] >> (-lo);
            }
        }
        l[j] = (j < 4) ? (acc & M52) : acc;
    }
}

// ---------------------------------------------------------------- one-pass product (stand-alone mul / square kernels)
// The stand-alone Mul / Square kernels take plain operands and return the plain canonical product (field.rs:250-262,
// :302-315: the reference pays two Montgomery passes for that, mul_internal + montgomery_reduce twice).  Round 1-4 did the
// same on the GPU (270 / 234 v_mad_u64_u32 per product).  For a canonical operand pair the special form of the moduli
// gives the product in ONE pass: N = 2^T + c with c < 2^125, so 2^T = -c (mod N).  With one operand scaled by 2^PSHIFT
// (PSHIFT = 261 - T: free, it happens in the radix conversion) the split point sits on the limb boundary 9 x 29 = 261:
//     X' = a (b 2^S) = LO' + HI 2^261                      81 products (45 for a square), HI = limbs 9..17 as they are
//     W  = LO' - HI (c 2^S)                                 45 products, signed columns; W = WLO + WH 2^261, -2^126 < WH <= 0
//     R' = WLO - WH (c 2^S)     in [0, 1.5 2^261)           25 products; one conditional subtraction of N 2^S
// and R' / 2^S is the canonical a b mod N (every term is a multiple of 2^S): 151 / 115 multiply-adds, no domain change.
// Canonical outputs are the reference's limbs whatever algorithm produced them (SURVEY 8a note P).  Operands at or
// above 2^T (non-canonical patterns, and the 2^-127 sliver [2^T, N)) take the two-pass form (per lane: fe_mulmod_limbs52).
#ifndef ZC_MULSQ_ONEPASS
#define ZC_MULSQ_ONEPASS 1      // 0: A/B build on the two Montgomery passes of rounds 1-4
#endif
// The product form is chosen per WAVE, not per lane: if any active lane holds an operand at or above 2^TOPBIT the whole wave
// takes the two-pass form (correct for every pattern), otherwise the one-pass form.  A per-lane branch makes a mixed wave
// execute both bodies one after the other (raw 252-bit scalar patterns: 7 lanes of 8 are "non-canonical" -- 10 % slower
// than round 4); a uniform branch executes one.  On the host (tests/emul) a "wave" is the one element.
#ifndef ZC_MULSQ_WAVE_UNIFORM
#define ZC_MULSQ_WAVE_UNIFORM 1   // 0: A/B build with the per-lane branch of round 5
#endif
#if defined(__HIP_DEVICE_COMPILE__)
ZC_DI bool mulsq_wave_any(bool p) { return ZC_MULSQ_WAVE_UNIFORM ? __builtin_amdgcn_ballot_w64(p) != 0 : p; }
#else
ZC_DI bool mulsq_wave_any(bool p) { return p; }
#endif
// value * 2^sh of five 52-bit limbs -> nine normalized 29-bit limbs (value < 2^(261 - sh))
ZC_DI fe fe_from_limbs52_shl(const u64 (&l)[5], const int sh)
{
    fe r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int bit = 29 * k - sh;                          // bit of the unscaled value that lands on bit 0 of limb k
        u64 x = 0;
        if (bit < 0) {
            x = (l[0] & M52) << (-bit);
        } else {
            const int idx = bit / 52, off = bit % 52;
            if (idx < 5) {
                x = (l[idx] & M52) >> off;
                if (off + 29 > 52 && idx + 1 < 5) x |= (l[idx + 1] & M52) << (52 - off);
            }
        }
        r.v[k] = (u32)x & M29;
    }
    return r;
}
// nine limbs (limbs 0..7 normalized, limb 8 may carry bit 29) holding value * 2^sh -> five 52-bit limbs of value
ZC_DI void fe_to_limbs52_shr(u64 (&l)[5], const fe& c, const int sh)
{
#pragma unroll
    for (int j = 0; j < 5; j++) {
        u64 acc = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int lo = 29 * k - sh - 52 * j;
            if (lo > -32 && lo < 52) {
                if (lo >= 0) acc |= (u64)c.v[k] << lo;
                else acc |= (u64)c.v[k] >> (-lo);
            }
        }
        l[j] = (j < 4) ? (acc & M52) : acc;
    }
}
// x - n when that is not negative (limbs 0..7 normalized, limb 8 compared as it is)
ZC_DI fe fe_cond_sub_limbs(const fe& x, const u32 (&n)[9])
{
    fe d;
    u32 borrow = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const u32 s = x.v[k] - n[k] - borrow;
        borrow = s >> 31;
        d.v[k] = s & M29;
    }
    const u32 s8 = x.v[8] - n[8] - borrow;
    d.v[8] = s8;
    return fe_select((s8 >> 31) != 0, x, d);
}
// x[0..17]: the normalized limbs of X' = a b 2^PSHIFT (a, b < 2^TOPBIT).  Returns (a b mod N) 2^PSHIFT, canonical.
template <class F>
ZC_DI fe plain_fold_canon(const u32 (&x)[18])
{
    typedef long long i64;
    ZC_ASSERT(x[17] < (1u << (F::TOPBIT - 232)));                          // HI < 2^TOPBIT
    int32_t ndp[5];                                                        // -(c << PSHIFT), limb by limb: signed multiply-adds subtract
#pragma unroll
    for (int j = 0; j < 5; j++) ndp[j] = -(int32_t)F::DP[j];
    // W = LO' - HI (c 2^S): columns 0..12 and the signed carry out of them
    int32_t wh[5];
    u32 w[9];
    i64 col = 0;
#pragma unroll
    for (int k = 0; k < 13; k++) {
        if (k < 9) col += (i64)x[k];
#pragma unroll
        for (int i = 0; i < 9; i++)
            if (k - i >= 0 && k - i < 5) {
                col += (i64)(int32_t)x[9 + i] * (i64)ndp[k - i];
                ZC_PIN(col);
            }
        if (k < 9) w[k] = (u32)col & M29;
        else wh[k - 9] = (int32_t)((u32)col & M29);
        col >>= 29;                                                        // arithmetic: the columns are signed
    }
    ZC_ASSERT(col <= 0 && col > -(1 << 12));
    wh[4] = (int32_t)col;                                                  // WH = wh[0..3] + wh[4] 2^116 <= 0
    // R' = WLO - WH (c 2^S) >= 0: columns 0..8, limb 8 keeps what is left
    fe r;
    i64 c2 = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        c2 += (i64)w[k];
#pragma unroll
        for (int i = 0; i < 5; i++)
            if (k - i >= 0 && k - i < 5) {
                c2 += (i64)wh[i] * (i64)ndp[k - i];
                ZC_PIN(c2);
            }
        if (k < 8) {
            r.v[k] = (u32)c2 & M29;
            c2 >>= 29;
        }
    }
    ZC_ASSERT(c2 >= 0 && c2 < (3ll << 28));                                // R' < 1.5 2^261
    r.v[8] = (u32)c2;
    return fe_cond_sub_limbs(r, F::NS);                                    // R' < 2 (N 2^S): one subtraction
}
// a b mod N for plain five-limb operands, canonical five-limb result
template <class F>
ZC_DI void fe_mulmod_limbs52(u64 (&r)[5], const u64 (&xa)[5], const u64 (&xb)[5])
{
    constexpr int TOP = F::TOPBIT - 208;                                   // bits of limb 4 below 2^TOPBIT
    if (!ZC_MULSQ_ONEPASS || mulsq_wave_any((((xa[4] | xb[4]) & M52) >> TOP) != 0)) {     // an operand (of the wave) at or above 2^TOPBIT: (a R) b / R as before
        const fe am = mont_to<F>(fe_from_limbs52(xa));
        fe_to_limbs52(r, fe_cond_sub_n<F>(fe_cond_sub_n<F>(mont_mul<F>(am, fe_from_limbs52(xb)))));
        return;
    }
    const fe a = fe_from_limbs52(xa), b = fe_from_limbs52_shl(xb, F::PSHIFT);
    u32 x[18];
    u64 col = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++)
            if (k - i >= 0 && k - i < 9) {
                col += (u64)a.v[i] * b.v[k - i];
                ZC_PIN(col);
            }
        x[k] = (u32)col & M29;
        col >>= 29;
    }
    x[17] = (u32)col;
    fe_to_limbs52_shr(r, plain_fold_canon<F>(x), F::PSHIFT);
}
// a^2 mod N likewise: a^2 2^S = (1 + (S & 1)) (a 2^(S / 2))^2, so the square keeps its 45 products
template <class F>
ZC_DI void fe_sqrmod_limbs52(u64 (&r)[5], const u64 (&xa)[5])
{
    constexpr int TOP = F::TOPBIT - 208, ODD = F::PSHIFT & 1;
    if (!ZC_MULSQ_ONEPASS || mulsq_wave_any(((xa[4] & M52) >> TOP) != 0)) {
        const fe a = fe_from_limbs52(xa);                     // any 260-bit pattern: (a R) a / R, within mont_mul's bounds
        fe_to_limbs52(r, fe_cond_sub_n<F>(fe_cond_sub_n<F>(mont_mul<F>(mont_to<F>(a), a))));
        return;
    }
    const fe a = fe_from_limbs52_shl(xa, F::PSHIFT / 2);     // < 2^(TOPBIT + S / 2) < 2^261: normalized limbs
    u32 d[9], e[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        d[i] = a.v[i] << (1 + ODD);                           // cross terms: 2 a_i a_j (times 2 when S is odd), < 2^31
        e[i] = a.v[i] << ODD;
    }
    u32 x[18];
    u64 col = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            if (k - i > i && k - i < 9) {
                col += (u64)d[i] * a.v[k - i];
                ZC_PIN(col);
            }
            if (k - i == i) {
                col += (u64)e[i] * a.v[i];
                ZC_PIN(col);
            }
        }
        x[k] = (u32)col & M29;
        col >>= 29;
    }
    x[17] = (u32)col;
    fe_to_limbs52_shr(r, plain_fold_canon<F>(x), F::PSHIFT);
}

// load a reference-layout element and enter Montgomery form
template <class F>
ZC_DI fe fe_load_mont(const u64* __restrict__ p)
{
    u64 l[5];
#pragma unroll
    for (int i = 0; i < 5; i++) l[i] = p[i];
    return mont_to<F>(fe_from_limbs52(l));
}
// leave Montgomery form, canonicalise, store in reference layout
template <class F>
ZC_DI void fe_store_canon(u64* __restrict__ p, const fe& a)
{
    u64 l[5];
    fe_to_limbs52(l, fe_canon_from_mont<F>(a));
#pragma unroll
    for (int i = 0; i < 5; i++) p[i] = l[i];
}

// 32 little-endian bytes (as four u64 words) -> nine 29-bit limbs, all 256 bits kept
// (from_bytes keeps bits 208..255 in the top limb: field.rs:563-587)
ZC_DI fe fe_from_words256(const u64 (&w)[4])
{
    fe r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int bit = 29 * k, idx = bit >> 6, sh = bit & 63;
        u64 x = w[idx] >> sh;
        if (sh + 29 > 64 && idx + 1 < 4) x |= w[idx + 1] << (64 - sh);
        r.v[k] = (k < 8) ? ((u32)x & M29) : (u32)x;
    }
    return r;
}
// canonical nine 29-bit limbs -> four u64 words (to_bytes, field.rs:591-631)
ZC_DI void fe_to_words256(u64 (&w)[4], const fe& c)
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        u64 acc = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int lo = 29 * k - 64 * j;
            if (lo > -29 && lo < 64) {
                if (lo >= 0) acc |= (u64)c.v[k] << lo;
                else acc |= (u64)c.v[k] >> (-lo);
            }
        }
        w[j] = acc;
    }
}
// ---------------------------------------------------------------- fixed exponentiation
// a^e for a compile-time exponent e (wave-uniform control flow), left-to-right
// binary.  Replaces the reference's data-dependent Pow / Savas-Koc / Tonelli
// loops (field.rs:325-441, :854-925) with a fixed schedule returning the same value.
template <class F>
ZC_DI fe fe_pow_const(const fe& a, const u32 (&e)[8], int nbits)
{
    fe acc = a;                                           // top bit of e is 1
    for (int i = nbits - 2; i >= 0; i--) {
        acc = mont_sqr<F>(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = mont_mul<F>(acc, a);
    }
    return acc;
}

}  // namespace zc

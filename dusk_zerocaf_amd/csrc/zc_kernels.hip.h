// zc_kernels.hip.h -- the __global__ kernels of libzerocaf_hip (gfx950).
// Launch shape: 256-thread blocks (4 waves, one per SIMD), one element / point per
// lane, grid = ceil(n / 256).  I/O arrays are the reference's own AoS limb layout.
#pragma once
#include "zc_curve.hip.h"

namespace zc {

#define ZC_KERNEL extern "C" __global__ __launch_bounds__(256)
// same, with the register budget capped so that two waves fit on a SIMD
#define ZC_KERNEL_3W extern "C" __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))
#define ZC_KERNEL_2W extern "C" __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
constexpr int ZC_BLOCK = 256;

ZC_DI size_t gid() { return (size_t)blockIdx.x * ZC_BLOCK + threadIdx.x; }
ZC_DI int wave_max_i32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int u = __shfl_xor(v, o);
        v = u > v ? u : v;
    }
    return v;
}

// maximum over the wave of a small value (-1 <= v < 127) with seven ballots: a binary search on the
// threshold, all in scalar registers (the shuffle form keeps five lane-address VGPRs alive)
ZC_DI int wave_max_small(int v)
{
    int r = -1;
#pragma unroll
    for (int b = 6; b >= 0; b--) {
        const int t = r + (1 << b);
        if (__ballot(v >= t) != 0) r = t;
    }
    return r;
}

// the same for -1 <= v < 511 (the w-NAF's digit positions run to 255)
ZC_DI int wave_max_9bit(int v)
{
    int r = -1;
#pragma unroll
    for (int b = 8; b >= 0; b--) {
        const int t = r + (1 << b);
        if (__ballot(v >= t) != 0) r = t;
    }
    return r;
}

// ------------------------------------------------------------------ radix-2^52 add/sub
// Add/Sub/Neg are carry/borrow chains over the reference's own limbs
// (field.rs:191-240, scalar.rs:184-237); they are done directly in radix 2^52 so the
// result is the reference's for every input pattern, canonical or not.
template <class F>
ZC_DI void limbs52_of_modulus(u64 (&m)[5])
{
    fe n = fe_const<F>(F::N);
    fe_to_limbs52(m, n);
}
ZC_DI void sub52(u64 (&r)[5], const u64 (&a)[5], const u64 (&b)[5], const u64 (&m)[5])
{
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        borrow = a[i] - (b[i] + (borrow >> 63));
        r[i] = borrow & M52;
    }
    const u64 mask = 0 - (borrow >> 63);                 // all ones when the difference went negative
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        carry = (carry >> 52) + r[i] + (m[i] & mask);
        r[i] = carry & M52;
    }
}
ZC_DI void add52(u64 (&r)[5], const u64 (&a)[5], const u64 (&b)[5], const u64 (&m)[5])
{
    u64 s[5];
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        carry = a[i] + b[i] + (carry >> 52);
        s[i] = carry & M52;
    }
    sub52(r, s, m, m);
}
// ------------------------------------------------------------------ LDS-staged element I/O
// The element-wise kernels are HBM-bound (80-120 B per element).  A lane's 40-byte
// record is strided in memory, so the block's 256 records (10 KB, contiguous) are moved with
// coalesced 16-byte loads/stores through LDS and each lane reads/writes its own record there
// with five ds_read_b64 / ds_write_b64 (stride 40 B = 10 banks: conflict-free in both 32-lane
// halves).  Requires 16-byte aligned array bases (checked by the host, which otherwise uses the
// per-lane kernel).
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

template <bool NT, int THREADS = ZC_BLOCK>
ZC_DI void coop_load40(u64* __restrict__ lds, const u64* __restrict__ g, int cnt)
{
    const int nvec = (cnt * 5) >> 1;                       // 16-byte vectors
    const u64x2* gv = reinterpret_cast<const u64x2*>(g);
    u64x2* lv = reinterpret_cast<u64x2*>(lds);
    for (int v = threadIdx.x; v < nvec; v += THREADS) lv[v] = NT ? __builtin_nontemporal_load(gv + v) : gv[v];
    if ((cnt & 1) && threadIdx.x == 0) lds[cnt * 5 - 1] = g[cnt * 5 - 1];
}
template <bool NT>
ZC_DI void coop_store40(u64* __restrict__ g, const u64* __restrict__ lds, int cnt)
{
    const int nvec = (cnt * 5) >> 1;
    u64x2* gv = reinterpret_cast<u64x2*>(g);
    const u64x2* lv = reinterpret_cast<const u64x2*>(lds);
    for (int v = threadIdx.x; v < nvec; v += ZC_BLOCK) {
        if (NT) __builtin_nontemporal_store(lv[v], gv + v);
        else gv[v] = lv[v];
    }
    if ((cnt & 1) && threadIdx.x == 0) g[cnt * 5 - 1] = lds[cnt * 5 - 1];
}

// Op::apply(r, x, y): five-limb radix-2^52 in/out.  NIN = number of input arrays (1 or 2).
// STAGED: 0 = per-lane global accesses, 2 = LDS-staged with non-temporal accesses.  Measured on
// MI355X (tools/bench_ops.py, 2^24..2^26 elements): staging lifts the compute-free two-input
// ops (add/sub) from 5.1 to 6.0 TB/s; mul/square (two Montgomery passes per element) and every
// cache-resident size are as fast or faster with per-lane accesses, so the host picks per op.
template <class Op, int NIN, int STAGED>
ZC_DI void elementwise_body(const u64* a, const u64* b, u64* out, size_t n)
{
    if (STAGED) {
        __shared__ __attribute__((aligned(16))) u64 sa[ZC_BLOCK * 5];
        __shared__ __attribute__((aligned(16))) u64 sb[(NIN == 2 ? ZC_BLOCK : 1) * 5];
        const size_t base = (size_t)blockIdx.x * ZC_BLOCK;
        const int cnt = (int)((n - base < (size_t)ZC_BLOCK) ? (n - base) : (size_t)ZC_BLOCK);
        coop_load40<STAGED == 2>(sa, a + 5 * base, cnt);
        if (NIN == 2) coop_load40<STAGED == 2>(sb, b + 5 * base, cnt);
        __syncthreads();
        const int t = threadIdx.x;
        if (t < cnt) {
            u64 x[5], y[5], r[5];
#pragma unroll
            for (int j = 0; j < 5; j++) {
                x[j] = sa[t * 5 + j];
                y[j] = (NIN == 2) ? sb[t * 5 + j] : 0;
            }
            Op::apply(r, x, y);
#pragma unroll
            for (int j = 0; j < 5; j++) sa[t * 5 + j] = r[j];   // own slot only
        }
        __syncthreads();
        coop_store40<STAGED == 2>(out + 5 * base, sa, cnt);
    } else {
        const size_t i = gid();
        if (i >= n) return;
        u64 x[5], y[5], r[5];
        load5(x, a + 5 * i);
        if (NIN == 2) load5(y, b + 5 * i);
        else {
#pragma unroll
            for (int j = 0; j < 5; j++) y[j] = 0;
        }
        Op::apply(r, x, y);
        store5(out + 5 * i, r);
    }
}

template <class F> struct OpAdd {
    static ZC_DI void apply(u64 (&r)[5], const u64 (&x)[5], const u64 (&y)[5]) { u64 m[5]; limbs52_of_modulus<F>(m); add52(r, x, y, m); }
};
template <class F> struct OpSub {
    static ZC_DI void apply(u64 (&r)[5], const u64 (&x)[5], const u64 (&y)[5]) { u64 m[5]; limbs52_of_modulus<F>(m); sub52(r, x, y, m); }
};
template <class F> struct OpNeg {
    static ZC_DI void apply(u64 (&r)[5], const u64 (&x)[5], const u64 (&)[5])
    {
        u64 m[5], z[5] = {0, 0, 0, 0, 0};
        limbs52_of_modulus<F>(m);
        sub52(r, z, x, m);
    }
};
// plain operands -> a*b mod N canonical, in one pass (zc_arith.hip.h: fe_mulmod_limbs52; operands at or above 2^TOPBIT
// take the two Montgomery passes of rounds 1-4)
template <class F> struct OpMul {
    static ZC_DI void apply(u64 (&r)[5], const u64 (&x)[5], const u64 (&y)[5]) { fe_mulmod_limbs52<F>(r, x, y); }
};
template <class F> struct OpSqr {
    static ZC_DI void apply(u64 (&r)[5], const u64 (&x)[5], const u64 (&)[5]) { fe_sqrmod_limbs52<F>(r, x); }
};

#define ZC_ELEMENTWISE2(name, OP)                                                                       \
    ZC_KERNEL void name##_stream(const u64* a, const u64* b, u64* out, size_t n) { elementwise_body<OP, 2, 2>(a, b, out, n); } \
    ZC_KERNEL void name(const u64* a, const u64* b, u64* out, size_t n) { elementwise_body<OP, 2, 0>(a, b, out, n); }
#define ZC_ELEMENTWISE1(name, OP)                                                                       \
    ZC_KERNEL void name##_stream(const u64* a, u64* out, size_t n) { elementwise_body<OP, 1, 2>(a, nullptr, out, n); }      \
    ZC_KERNEL void name(const u64* a, u64* out, size_t n) { elementwise_body<OP, 1, 0>(a, nullptr, out, n); }

ZC_ELEMENTWISE2(k_fe_add, OpAdd<ModP>)
ZC_ELEMENTWISE2(k_fe_sub, OpSub<ModP>)
ZC_ELEMENTWISE1(k_fe_neg, OpNeg<ModP>)
ZC_ELEMENTWISE2(k_sc_add, OpAdd<ModL>)
ZC_ELEMENTWISE2(k_sc_sub, OpSub<ModL>)
ZC_ELEMENTWISE1(k_sc_neg, OpNeg<ModL>)
ZC_ELEMENTWISE2(k_fe_mul, OpMul<ModP>)
ZC_ELEMENTWISE2(k_sc_mul, OpMul<ModL>)
ZC_ELEMENTWISE1(k_fe_square, OpSqr<ModP>)
ZC_ELEMENTWISE1(k_sc_square, OpSqr<ModL>)

// ------------------------------------------------------------------ invert / sqrt_ratio_i
ZC_KERNEL void k_fe_invert(const u64* a, u64* out, uint8_t* ok, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    const fe x = fe_load_mont<FP>(a + 5 * i);
    const bool nz = !fp_is_zero(x);
    fe_store_canon<FP>(out + 5 * i, fp_invert(x));         // 0^(p-2) = 0
    if (ok) ok[i] = nz ? 1 : 0;
}

// Batched inversion, Montgomery's trick per lane: lane g owns the `c` consecutive elements
// [g*c, g*c+c).  The reference inverts one element at a time with ~357 field Mul each
// (field.rs:854-925); here a chunk shares ONE fixed-schedule exponentiation, so an element
// costs 3 Montgomery multiplications + 300/c.  The plain inputs are used directly as
// Montgomery residues (register value x stands for x/R): with acc_j = a_1...a_j R^-(j-1) and
// I_j = acc_j^-1 (plain), a_j^-1 = montmul(I_j, acc_{j-1}) and I_{j-1} = montmul(I_j, a_j) --
// no per-element domain conversion at all.  Prefix products are parked in the output records
// (36 of the 40 bytes) between the two sweeps, so `out` must not alias `a`.
// Zero elements (the reference panics) take the neutral value R mod p and yield 0 / ok = 0.
ZC_KERNEL void k_fe_invert_chunked(const u64* a, u64* out, uint8_t* ok, size_t n, int c)
{
    const size_t lanes = (n + (size_t)c - 1) / (size_t)c, g = gid();
    if (g < lanes) fe_invert_chunk(a, out, ok, n, g, lanes, c);
}
// the same for launches of at most one wave per SIMD (fe_invert_chunk<ILP>)
ZC_KERNEL void k_fe_invert_chunked_lone(const u64* a, u64* out, uint8_t* ok, size_t n, int c)
{
    const size_t lanes = (n + (size_t)c - 1) / (size_t)c, g = gid();
    if (g < lanes) fe_invert_chunk<true>(a, out, ok, n, g, lanes, c);
}
ZC_KERNEL void k_fe_sqrt_ratio_i(const u64* u, const u64* v, u64* out, uint8_t* was_square, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    const fe uu = fe_load_mont<FP>(u + 5 * i), vv = fe_load_mont<FP>(v + 5 * i);
    fe r;
    const bool sq = fp_sqrt_ratio_i(r, uu, vv);
    fe_store_canon<FP>(out + 5 * i, r);
    if (was_square) was_square[i] = sq ? 1 : 0;
}

// F8/F9 rows: Div, Half, Pow, legendre_symbol, ModSqrt, is_positive
ZC_KERNEL void k_fe_div(const u64* a, const u64* b, u64* out, uint8_t* ok, size_t n)   // field.rs:277-300
{
    const size_t i = gid();
    if (i >= n) return;
    const fe x = fe_load_mont<FP>(a + 5 * i), y = fe_load_mont<FP>(b + 5 * i);
    const bool nz = !fp_is_zero(y);
    fe_store_canon<FP>(out + 5 * i, fp_mul(x, fp_invert(y)));
    if (ok) ok[i] = nz ? 1 : 0;
}
ZC_KERNEL void k_fe_div_chunked(const u64* a, const u64* b, u64* out, uint8_t* ok, size_t n, int c)
{
    const size_t lanes = (n + (size_t)c - 1) / (size_t)c, g = gid();
    if (g < lanes) fe_invert_chunk(b, out, ok, n, g, lanes, c, a);
}
ZC_KERNEL void k_fe_div_chunked_lone(const u64* a, const u64* b, u64* out, uint8_t* ok, size_t n, int c)
{
    const size_t lanes = (n + (size_t)c - 1) / (size_t)c, g = gid();
    if (g < lanes) fe_invert_chunk<true>(b, out, ok, n, g, lanes, c, a);
}
ZC_KERNEL void k_fe_half(const u64* a, u64* out, size_t n)                          // field.rs:317-323
{
    const size_t i = gid();
    if (i >= n) return;
    u64 l[5], h[5];
    load5(l, a + 5 * i);
    fe_to_limbs52(h, fe_n_minus_canon<FP>(fe_const<FP>(ModP::HALF)));             // (p+1)/2 = p - (p-1)/2
    const fe p2 = mont_mul<FP>(mont_to<FP>(fe_from_limbs52(l)), fe_from_limbs52(h));
    fe_to_limbs52(h, fe_cond_sub_n<FP>(fe_cond_sub_n<FP>(p2)));
    store5(out + 5 * i, h);
}
ZC_KERNEL void k_fe_pow(const u64* a, const u64* e, u64* out, size_t n)            // field.rs:325-355
{
    const size_t i = gid();
    if (i >= n) return;
    u64 l[5];
    load5(l, e + 5 * i);
    fe_store_canon<FP>(out + 5 * i, fp_pow_var(fe_load_mont<FP>(a + 5 * i), fe_from_limbs52(l)));
}
ZC_KERNEL void k_fe_legendre(const u64* a, uint8_t* out, size_t n, int max_rounds)     // field.rs:703-706, as a Jacobi symbol
{
    const size_t i = gid();
    if (i >= n) return;
    out[i] = fp_legendre(fe_load_mont<FP>(a + 5 * i), max_rounds) ? 1 : 0;
}
ZC_KERNEL void k_fe_is_positive(const u64* a, uint8_t* out, size_t n)              // field.rs:552-557 (limb-lexicographic)
{
    const size_t i = gid();
    if (i >= n) return;
    u64 l[5], h[5];
    load5(l, a + 5 * i);
    fe_to_limbs52(h, fe_const<FP>(ModP::HALF));
    int c = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) c = (l[j] > h[j]) ? 1 : ((l[j] < h[j]) ? -1 : c);
    out[i] = (c <= 0) ? 1 : 0;
}
ZC_KERNEL void k_fe_mod_sqrt(const u64* a, int sign, u64* out, uint8_t* ok, size_t n)   // field.rs:357-441
{
    const size_t i = gid();
    if (i >= n) return;
    fe x;
    const bool have = fp_mod_sqrt(x, fe_load_mont<FP>(a + 5 * i), sign != 0);
    fe_store_canon<FP>(out + 5 * i, fe_select(have, x, fe_zero()));
    if (ok) ok[i] = have ? 1 : 0;
}

// ------------------------------------------------------------------ byte codecs
ZC_DI void load_words256(u64 (&w)[4], const uint8_t* __restrict__ p)
{
    const u64* q = reinterpret_cast<const u64*>(p);       // 32-byte records, 8-byte aligned
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = q[i];
}
ZC_DI void store_words256(uint8_t* __restrict__ p, const u64 (&w)[4])
{
    u64* q = reinterpret_cast<u64*>(p);
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = w[i];
}
// four u64 words -> five 52-bit limbs, top limb keeps 48 bits (field.rs:577-586, scalar.rs:457-463)
ZC_DI void words_to_limbs52(u64 (&l)[5], const u64 (&w)[4])
{
    l[0] = w[0] & M52;
    l[1] = ((w[0] >> 52) | (w[1] << 12)) & M52;
    l[2] = ((w[1] >> 40) | (w[2] << 24)) & M52;
    l[3] = ((w[2] >> 28) | (w[3] << 36)) & M52;
    l[4] = w[3] >> 16;
}
// (field.rs:591-631, scalar.rs:477-516): bytes of limbs, upper limb bits beyond 256 dropped
ZC_DI void limbs52_to_words(u64 (&w)[4], const u64 (&l)[5])
{
    w[0] = l[0] | (l[1] << 52);
    w[1] = (l[1] >> 12) | (l[2] << 40);
    w[2] = (l[2] >> 24) | (l[3] << 28);
    w[3] = (l[3] >> 36) | (l[4] << 16);
}
ZC_KERNEL void k_from_bytes(const uint8_t* in, u64* out, uint8_t* ok, int check_scalar_range, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    u64 w[4], l[5];
    load_words256(w, in + 32 * i);
    words_to_limbs52(l, w);
    if (check_scalar_range) {
        // assert!(s <= L - 1)  (scalar.rs:465): limb-lexicographic compare from the top; the
        // reference panics there, here the element comes back as zero with ok = 0
        u64 m[5];
        limbs52_of_modulus<ModL>(m);
        m[0] -= 1;
        int c = 0;                                         // sign of (l - m)
#pragma unroll
        for (int j = 0; j < 5; j++) c = (l[j] > m[j]) ? 1 : ((l[j] < m[j]) ? -1 : c);
        if (c > 0) l[0] = l[1] = l[2] = l[3] = l[4] = 0;
        if (ok) ok[i] = (c <= 0) ? 1 : 0;
    }
    store5(out + 5 * i, l);
}
ZC_KERNEL void k_to_bytes(const u64* in, uint8_t* out, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    u64 w[4], l[5];
    load5(l, in + 5 * i);
    limbs52_to_words(w, l);
    store_words256(out + 32 * i, w);
}

// ---- S-x / F8 rows of SURVEY 8(a) that sit beside the default path: Scalar Half / Pow / Shr, the bit and
// NAF recoders behind ltr_bin_mul / binary_naf_mul / window_naf_mul, FieldElement inv_sqrt
ZC_KERNEL void k_fe_inv_sqrt(const u64* a, u64* out, uint8_t* was_square, size_t n)  // field.rs:443-460 = sqrt_ratio_i(1, a)
{
    const size_t i = gid();
    if (i >= n) return;
    fe r;
    const bool sq = fp_sqrt_ratio_i(r, fe_one_m<FP>(), fe_load_mont<FP>(a + 5 * i));
    fe_store_canon<FP>(out + 5 * i, r);
    if (was_square) was_square[i] = sq ? 1 : 0;
}
ZC_KERNEL void k_sc_half(const u64* a, u64* out, size_t n)                           // scalar.rs:285-291 (a * INVERSE_MOD_TWO)
{
    const size_t i = gid();
    if (i >= n) return;
    u64 l[5], h[5];
    load5(l, a + 5 * i);
    fe_to_limbs52(h, fe_n_minus_canon<ModL>(fe_const<ModL>(ModL::HALF)));            // (L+1)/2 = L - (L-1)/2
    const fe p2 = mont_mul<ModL>(mont_to<ModL>(fe_from_limbs52(l)), fe_from_limbs52(h));
    fe_to_limbs52(h, fe_cond_sub_n<ModL>(fe_cond_sub_n<ModL>(p2)));
    store5(out + 5 * i, h);
}
// scalar.rs:300-322: square-and-multiply driven by halving the exponent; for a canonical exponent that is
// a^e mod L (e = 0 gives 1), computed here MSB-first on the fixed 261-bit schedule
ZC_KERNEL void k_sc_pow(const u64* a, const u64* e, u64* out, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    u64 l[5];
    load5(l, e + 5 * i);
    const fe ee = fe_from_limbs52(l);
    const fe am = fe_load_mont<ModL>(a + 5 * i);
    fe acc = fe_one_m<ModL>();
    for (int b = 260; b >= 0; b--) {
        acc = mont_sqr<ModL>(acc);
        const bool bit = ((ee.v[b / 29] >> (b % 29)) & 1) != 0;
        acc = fe_select(bit, mont_mul<ModL>(acc, am), acc);
    }
    fe_store_canon<ModL>(out + 5 * i, acc);
}
// half_without_mod (scalar.rs:562-574): the five limbs as one 260-bit integer, shifted right by one
ZC_DI void half_without_mod52(u64 (&k)[5])
{
#pragma unroll
    for (int j = 0; j < 5; j++) k[j] = (k[j] >> 1) | (j < 4 ? (k[j + 1] & 1) << 51 : 0);
}
ZC_KERNEL void k_sc_shr(const u64* a, u32 shift, u64* out, size_t n)                 // Shr<u8>, scalar.rs:165-182
{
    const size_t i = gid();
    if (i >= n) return;
    u64 k[5];
    load5(k, a + 5 * i);
    for (u32 s = 0; s < shift; s++) half_without_mod52(k);
    store5(out + 5 * i, k);
}
ZC_KERNEL void k_sc_into_bits(const u64* a, uint8_t* out, size_t n)                  // scalar.rs:352-366: the bits of to_bytes()
{
    const size_t i = gid();
    if (i >= n) return;
    u64 l[5], w[4];
    load5(l, a + 5 * i);
    limbs52_to_words(w, l);
    u32* o = reinterpret_cast<u32*>(out + 256 * i);
    for (int q = 0; q < 64; q++) {
        const u32 nib = (u32)(w[q >> 4] >> ((q & 15) * 4)) & 15u;
        o[q] = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
    }
}
// compute_NAF (scalar.rs:370-389; width == 0) and compute_window_NAF (scalar.rs:396-415; width 2..7), the
// reference's loop literally: while k >= 1 and i < 256 { if k is odd { k_i = 2 - (k mod 4) | mods(k, 2^w);
// k = k - Scalar::from(k_i) } ; k = half_without_mod(k) }, where Scalar::from of a negative digit is L - |k_i|
// (scalar.rs:68-84) and Sub adds L back only after a borrow (:210-237) -- so a scalar above L - |k_i| takes the
// reference's wrap-around, not the integer NAF.  256 digits per scalar, zeros behind the last one.
// one iteration of that loop: the digit, k updated in place (m = the limbs of L)
ZC_DI int naf_step(u64 (&k)[5], const u64 (&m)[5], u32 width)
{
    int ki = 0;
    if (((k[0] | k[1] | k[2] | k[3] | k[4]) != 0) && (k[0] & 1)) {
        if (width == 0) ki = 2 - (int)(k[0] & 3);
        else {
            const int modulus = (int)(k[0] & ((1u << width) - 1u));                      // mods_2_pow_k, scalar.rs:433-442
            ki = modulus >= (1 << (width - 1)) ? modulus - (1 << width) : modulus;
        }
        u64 t[5] = {(u64)(ki < 0 ? -ki : ki), 0, 0, 0, 0}, r[5];
        if (ki < 0) {
            const u64 z[5] = {0, 0, 0, 0, 0};
            sub52(r, z, t, m);                                                           // Neg: 0 - |k_i| mod L
#pragma unroll
            for (int j = 0; j < 5; j++) t[j] = r[j];
        }
        sub52(r, k, t, m);
#pragma unroll
        for (int j = 0; j < 5; j++) k[j] = r[j];
    }
    half_without_mod52(k);
    return ki;
}
ZC_KERNEL void k_sc_compute_naf(const u64* a, u32 width, int8_t* out, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    u64 k[5], m[5];
    load5(k, a + 5 * i);
    limbs52_of_modulus<ModL>(m);
    u32* o = reinterpret_cast<u32*>(out + 256 * i);
    u32 packed = 0;
    for (int d = 0; d < 256; d++) {
        const int ki = naf_step(k, m, width);
        packed |= ((u32)ki & 0xFFu) << (8 * (d & 3));
        if ((d & 3) == 3) {
            o[d >> 2] = packed;
            packed = 0;
        }
    }
}


// ------------------------------------------------------------------ point ops
ZC_KERNEL void k_ed_add(const u64* p, const u64* q, u64* out, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    pt_store_plain(out + 20 * i, pt_add_plain(pt_load_plain(p + 20 * i), pt_load_plain(q + 20 * i)));   // plain domain: zc_curve.hip.h
}
ZC_KERNEL void k_ed_sub(const u64* p, const u64* q, u64* out, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    // edwards.rs:503-531: add of the negated rhs with H = B - a*A == B + A (a = -1): same values
    pt_store_plain(out + 20 * i, pt_add_plain(pt_load_plain(p + 20 * i), pt_neg(pt_load_plain(q + 20 * i))));
}
#ifndef ZC_ED_DOUBLE_SQR
#define ZC_ED_DOUBLE_SQR 1              // Double with the equal-operand products as squarings (0: A/B, pt_add_plain(a, a) as in round 5)
#endif
ZC_KERNEL void k_ed_double(const u64* p, u64* out, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    const pt a = pt_load_plain(p + 20 * i);
    pt_store_plain(out + 20 * i, ZC_ED_DOUBLE_SQR ? pt_double_plain(a) : pt_add_plain(a, a));
}
// The same three with the records staged through LDS (16-byte aligned arrays).  A lane's 160-byte record is strided in
// memory: read per lane it takes twenty 8-byte loads that each touch a cache line of their own, the wave's working set
// (64 x 160 bytes per operand, three waves per SIMD) is far beyond the L1, and every line comes in from the L2 piece by
// piece.  Staged, the block's 256 records (40 KB, contiguous) move with coalesced 16-byte loads and stores, one operand
// after the other through ONE 40 KB buffer (four workgroups per CU; the kernels hold three waves per SIMD anyway), and
// the results leave the same way.  Same formulas, same limbs.
#ifndef ZC_ED_STAGED_EARLYQ
#define ZC_ED_STAGED_EARLYQ 1           // request both operands' records at once, the second waits in 20 VGPRs (0: A/B; same box ed_add 0.139 -> 0.135 ms, 2^24 1.954 -> 1.916)
#endif
#ifndef ZC_ED_STAGED_BLOCK
#define ZC_ED_STAGED_BLOCK 256          // threads per workgroup of the staged point kernels (A/B: 64 / 128 / 256)
#endif
constexpr int ED_STAGED_BLOCK = ZC_ED_STAGED_BLOCK;
template <int OP>                       // 0: add, 1: sub, 2: double
ZC_DI void ed_binop_staged(const u64* p, const u64* q, u64* out, size_t n)
{
    constexpr int B = ED_STAGED_BLOCK;
    __shared__ __attribute__((aligned(16))) u64 sp[B * 20];
    const size_t base = (size_t)blockIdx.x * B;
    const int cnt = (int)((n - base < (size_t)B) ? (n - base) : (size_t)B);
    const int t = threadIdx.x;
    const u64x2* gp = reinterpret_cast<const u64x2*>(p + 20 * base);
    u64x2* lv = reinterpret_cast<u64x2*>(sp);
#if ZC_ED_STAGED_EARLYQ
    // the second operand's pieces are requested together with the first's and wait in registers for their turn in the buffer
    u64x2 qreg[10];
    if (OP != 2) {
        const u64x2* gq = reinterpret_cast<const u64x2*>(q + 20 * base);
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const int v = t + k * B;
            if (v < cnt * 10) qreg[k] = gq[v];
        }
    }
#endif
    for (int v = t; v < cnt * 10; v += B) lv[v] = gp[v];       // a point = ten 16-byte pieces
    __syncthreads();
    pt a = pt_identity(), b;
    if (t < cnt) a = pt_load_plain(sp + 20 * t);
    if (OP != 2) {
        __syncthreads();                                     // every lane has its first operand: the buffer takes the second
#if ZC_ED_STAGED_EARLYQ
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const int v = t + k * B;
            if (v < cnt * 10) lv[v] = qreg[k];
        }
#else
        const u64x2* gq = reinterpret_cast<const u64x2*>(q + 20 * base);
        for (int v = t; v < cnt * 10; v += B) lv[v] = gq[v];
#endif
        __syncthreads();
        b = pt_identity();
        if (t < cnt) b = pt_load_plain(sp + 20 * t);
        if (OP == 1) b = pt_neg(b);                          // edwards.rs:503-531: add of the negated rhs (a = -1: same values)
    } else {
        b = a;
    }
    const pt r = OP == 2 && ZC_ED_DOUBLE_SQR ? pt_double_plain(a) : pt_add_plain(a, b);        // (double: four of the products are squarings, same limbs)
    __syncthreads();                                         // ... and then the results
    if (t < cnt) pt_store_plain(sp + 20 * t, r);
    __syncthreads();
    u64x2* go = reinterpret_cast<u64x2*>(out + 20 * base);
    for (int v = t; v < cnt * 10; v += B) go[v] = lv[v];
}
#define ZC_KERNEL_EDS extern "C" __global__ __launch_bounds__(ZC_ED_STAGED_BLOCK)
ZC_KERNEL_EDS void k_ed_add_staged(const u64* p, const u64* q, u64* out, size_t n) { ed_binop_staged<0>(p, q, out, n); }
ZC_KERNEL_EDS void k_ed_sub_staged(const u64* p, const u64* q, u64* out, size_t n) { ed_binop_staged<1>(p, q, out, n); }
ZC_KERNEL_EDS void k_ed_double_staged(const u64* p, u64* out, size_t n) { ed_binop_staged<2>(p, nullptr, out, n); }
// Neg with staged records: the workgroup's records pass through LDS once, X and T are negated in place there (the
// reference's radix-2^52 borrow chain, as k_ed_neg below), Y and Z ride along.
ZC_KERNEL_EDS void k_ed_neg_staged(const u64* p, u64* out, size_t n)
{
    constexpr int B = ED_STAGED_BLOCK;
    __shared__ __attribute__((aligned(16))) u64 sp[B * 20];
    const size_t base = (size_t)blockIdx.x * B;
    const int cnt = (int)((n - base < (size_t)B) ? (n - base) : (size_t)B);
    const int t = threadIdx.x;
    const u64x2* gp = reinterpret_cast<const u64x2*>(p + 20 * base);
    u64x2* lv = reinterpret_cast<u64x2*>(sp);
    for (int v = t; v < cnt * 10; v += B) lv[v] = gp[v];
    __syncthreads();
    if (t < cnt) {
        u64 m[5], X[5], T[5], nx[5], nt[5];
        const u64 z[5] = {0, 0, 0, 0, 0};
        limbs52_of_modulus<ModP>(m);
        load5(X, sp + 20 * t);
        load5(T, sp + 20 * t + 15);
        sub52(nx, z, X, m);
        sub52(nt, z, T, m);
        store5(sp + 20 * t, nx);
        store5(sp + 20 * t + 15, nt);
    }
    __syncthreads();
    u64x2* go = reinterpret_cast<u64x2*>(out + 20 * base);
    for (int v = t; v < cnt * 10; v += B) go[v] = lv[v];
}
ZC_KERNEL void k_ed_neg(const u64* p, u64* out, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    // Neg (edwards.rs:440-455): (-X, Y, Z, -T), the negations as the reference's own radix-2^52 borrow chain
    // (field.rs:170-189 = 0 - a), Y and Z copied: no multiplication at all
    u64 m[5], X[5], Y[5], Z[5], T[5], nx[5], nt[5];        // all loads first: `p` and `out` may alias for all the compiler knows
    const u64 z[5] = {0, 0, 0, 0, 0};
    limbs52_of_modulus<ModP>(m);
    load5(X, p + 20 * i);
    load5(Y, p + 20 * i + 5);
    load5(Z, p + 20 * i + 10);
    load5(T, p + 20 * i + 15);
    sub52(nx, z, X, m);
    sub52(nt, z, T, m);
    store5(out + 20 * i, nx);
    store5(out + 20 * i + 5, Y);
    store5(out + 20 * i + 10, Z);
    store5(out + 20 * i + 15, nt);
}

// ------------------------------------------------------------------ variable-base scalar multiplication
// Reference: double_and_add (edwards.rs:102-120): LSB-first, Q += N when the bit is
// set, N = N + N every iteration, both through the SAME unified HWCD formula.
// SIMT form: every lane walks its own op sequence  [add?] dbl [add?] dbl ... [add]
// and each wave step evaluates the formula ONCE with per-lane selected operands
// (Q+N or N+N).  Lanes therefore never pay for the adds of zero bits: a wave needs
// max_lane(bitlen - 1 + popcount) steps (~395 for random 252-bit scalars) instead
// of 2*bitlen (~502) with a predicated add per bit.  The final doubling of the
// reference loop does not influence Q and is skipped; the first add is performed
// literally (identity + N), so (X:Y:Z:T) limbs equal the reference's.
// Scalar words live in LDS (one 32-bit word per lane per refill).
// ---- lane balancing -------------------------------------------------------------------
// A wave needs max over its 64 lanes of (bitlen - 1 + popcount) unified steps.  For random
// scalars that maximum is ~6 % above the mean, so the batch is first ordered by that cost with
// a counting sort (cost < 1024 bins) and wave w processes the elements idx[64w .. 64w+63], which
// then have (nearly) equal cost.  Results are scattered back through the same index, so the
// output order is unchanged.  Three tiny kernels (~tens of microseconds at 2^20).
constexpr int ZC_COST_BINS = 1024;

ZC_DI u32 scalar_cost(const u64 (&l)[5])
{
    u32 bits = 0, pop = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const u64 x = l[j] & M52;
        pop += __builtin_popcountll(x);
        if (x) bits = 52 * j + (64 - __builtin_clzll(x));
    }
    return bits ? bits - 1 + pop : 0;                    // <= 259 + 260
}
// hist[c] = number of scalars of cost c (LDS histogram per block, one global atomic per non-empty bin)
ZC_KERNEL void k_sm_cost_hist(const u64* k, u32* hist, size_t n)
{
    __shared__ u32 h[ZC_COST_BINS];
    for (int b = threadIdx.x; b < ZC_COST_BINS; b += ZC_BLOCK) h[b] = 0;
    __syncthreads();
    const size_t i = gid();
    if (i < n) {
        u64 l[5];
        load_scalar(l, k + 5 * i);
        atomicAdd(&h[scalar_cost(l)], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < ZC_COST_BINS; b += ZC_BLOCK)
        if (h[b]) atomicAdd(&hist[b], h[b]);
}
// exclusive scan over the bins, most expensive first (long waves start early); single block
ZC_KERNEL void k_sm_cost_scan(u32* hist)
{
    __shared__ u32 part[ZC_BLOCK];
    const int t = threadIdx.x;
    u32 v[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {                        // thread t owns descending bins 1023-4t .. 1020-4t
        v[j] = hist[ZC_COST_BINS - 1 - (4 * t + j)];
        sum += v[j];
    }
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < ZC_BLOCK; d <<= 1) {
        const u32 x = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    u32 base = part[t] - sum;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        hist[ZC_COST_BINS - 1 - (4 * t + j)] = base;
        base += v[j];
    }
}
// idx[offset[cost]++] = i, with block-level aggregation of the global atomics
ZC_KERNEL void k_sm_cost_scatter(const u64* k, u32* offsets, u32* idx, size_t n)
{
    __shared__ u32 cnt[ZC_COST_BINS];
    __shared__ u32 base[ZC_COST_BINS];
    for (int b = threadIdx.x; b < ZC_COST_BINS; b += ZC_BLOCK) cnt[b] = 0;
    __syncthreads();
    const size_t i = gid();
    u32 c = 0, rank = 0;
    if (i < n) {
        u64 l[5];
        load_scalar(l, k + 5 * i);
        c = scalar_cost(l);
        rank = atomicAdd(&cnt[c], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < ZC_COST_BINS; b += ZC_BLOCK)
        if (cnt[b]) base[b] = atomicAdd(&offsets[b], cnt[b]);
    __syncthreads();
    if (i < n) idx[base[c] + rank] = (u32)i;
}

// ---- block-local lane balancing --------------------------------------------------------
// Same idea as the global counting sort above, but inside one block: the 256 elements a block
// owns are ranked by cost in LDS and lane j runs the element of rank j, so each of the four
// waves gets a cost quartile.  A block still touches exactly its own contiguous 256 records
// (HBM traffic stays algorithmic: the global permutation turns the 160-byte reads into line
// gathers) and no prepass kernels or index buffers are needed; it gives up ~1 % of the steps a
// batch-wide order saves.  Scalar words stay in the owner's LDS column, the executing lane reads
// column `e`.  Returns the in-block element index this lane executes.
ZC_DI int block_cost_rank(u32* __restrict__ skey, u32* __restrict__ sperm, u32 my_cost)
{
    const int tid = threadIdx.x;
    skey[tid] = my_cost;
    __syncthreads();
    int rank = 0;
    for (int u = 0; u < ZC_BLOCK; u++) {
        const u32 c = skey[u];
        rank += (c > my_cost || (c == my_cost && u < tid)) ? 1 : 0;
    }
    sperm[rank] = (u32)tid;
    __syncthreads();
    return (int)sperm[tid];
}

// One scalar shared by the whole batch, passed by value in the kernel arguments
// (mul_by_pow_2 / mul_by_cofactor, edwards.rs:174-191): no device copy of the scalar is needed.
struct scalar_arg {
    u64 l[5];
};
ZC_KERNEL void k_ed_scalar_mul_bcast(const u64* p, scalar_arg k, u64* out, size_t n)
{
    __shared__ u32 sk[9 * ZC_BLOCK];
    const int tid = threadIdx.x;
    const size_t i = gid();
    const bool valid = i < n;
    int nbits;
    scalar_effective(k.l);
    scalar_to_words(sk + tid, ZC_BLOCK, k.l, nbits);
    const pt Q = scalar_mul_unified(pt_load(p + 20 * (valid ? i : 0)), sk + tid, ZC_BLOCK, valid ? nbits : 0);
    if (valid) pt_store(out + 20 * i, Q);
}

// k_stride = 5 (one scalar per point).  The block's 256 scalars are ranked by cost in LDS (block_cost_rank).
template <bool ILP>
ZC_DI void ed_scalar_mul_body(const u64* p, const u64* k, size_t k_stride, u64* out, size_t n)
{
    __shared__ u32 sk[9 * ZC_BLOCK];
    __shared__ u32 skey[ZC_BLOCK];
    __shared__ u32 sperm[ZC_BLOCK];
    __shared__ int snb[ZC_BLOCK];
    const int tid = threadIdx.x;
    const size_t i = gid();
    const bool valid = i < n;
    const size_t own = valid ? i : 0;
    u64 l[5];
    load_scalar(l, k + k_stride * own);
    int nbits;
    scalar_to_words(sk + tid, ZC_BLOCK, l, nbits);
    if (!valid) nbits = 0;
    int e = tid;
    if (k_stride != 0) {
        snb[tid] = nbits;
        e = block_cost_rank(skey, sperm, valid ? scalar_cost(l) : 0);
        nbits = snb[e];
    }
    const size_t base = (size_t)blockIdx.x * ZC_BLOCK;
    const bool run = base + e < n;
    const size_t ii = run ? base + e : 0;
    const pt P = pt_load(p + 20 * ii);
    const pt Q = scalar_mul_unified<ILP>(P, sk + e, ZC_BLOCK, run ? nbits : 0);
    if (run) pt_store(out + 20 * ii, Q);
}
ZC_KERNEL void k_ed_scalar_mul(const u64* p, const u64* k, size_t k_stride, u64* out, size_t n)
{
    ed_scalar_mul_body<false>(p, k, k_stride, out, n);
}
// Persistent waves for large batches (the headline shape).  A grid of 3 workgroups per CU stays
// resident; every WAVE, on its own, pulls the next tile of 64 elements from an atomic counter until
// the batch is done.  Tiles are 64 consecutive entries of the batch-wide cost-sorted permutation
// (k_sm_cost_*, most expensive first), so
//   * the 64 lanes of a wave carry scalars of (almost) the same cost bitlen - 1 + popcount: a wave
//     performs the mean number of formula evaluations, not the maximum over a block-local quartile;
//   * no wave waits for the three other waves of a workgroup (a new workgroup needs a free slot on
//     all four SIMDs at once: with one tile per workgroup the cheapest quartile's SIMD idles until
//     the dearest quartile is done), and no partial last round of workgroups: the chip drains
//     within one tile's run time, the cheapest tiles last.
// Records are gathered through the permutation (160-byte points, 40-byte scalars: whole-line
// gathers roughly double the 210 MB read per 2^20, still ~0.3 % of the HBM roof).  Scalar words live
// in the wave's own 9 x 64-word LDS region; LDS operations of one wave execute in order, so no
// barrier is needed anywhere.  Results are the same limbs as k_ed_scalar_mul's (same per-lane loop).
ZC_KERNEL void k_ed_scalar_mul_pw(const u64* p, const u64* k, u64* out, const u32* idx, u32* counter, u32 n)
{
    __shared__ u32 sk[9 * ZC_BLOCK];
    const int lane = threadIdx.x & 63;
    u32* skw = sk + 9 * (threadIdx.x & ~63) + lane;        // word j of this lane: skw[64 j]
    const u32 ntiles = (n + 63) / 64;
    for (;;) {
        u32 t = 0;
        if (lane == 0) t = atomicAdd(counter, 1u);
        t = (u32)__builtin_amdgcn_readfirstlane((int)t);
        if (t >= ntiles) break;
        const u32 i = t * 64 + lane;
        const bool valid = i < n;
        const size_t own = valid ? (size_t)idx[i] : 0;
        u64 l[5];
        load_scalar(l, k + 5 * own);
        int nbits;
        scalar_to_words(skw, 64, l, nbits);
        const pt Q = scalar_mul_unified<false>(pt_load(p + 20 * own), skw, 64, valid ? nbits : 0);
        if (valid) pt_store(out + 20 * own, Q);
    }
}
// the same for launches of at most a few hundred workgroups (one wave per SIMD): independent-chain multiplier
ZC_KERNEL void k_ed_scalar_mul_small(const u64* p, const u64* k, size_t k_stride, u64* out, size_t n)
{
    ed_scalar_mul_body<true>(p, k, k_stride, out, n);
}

// ---- fast (non-strict) scalar multiplication ---------------------------------------------
// scalar_mul_fast (zc_curve.hip.h): fixed signed 4-bit windows, dedicated doubling, 8-mul cached
// additions, per-lane table of 8 cached multiples in global scratch (1 KB per point, one cache
// line per entry).  ~0.63x the multiplier work of the reference's formula sequence and no SIMT
// divergence at all.  The result is the same group element as double_and_add's (identical
// encodings); only its projective (X:Y:Z:T) representative differs.
// `table`: the 8 cached multiples of a lane's point, 1 KB per lane (one cache line per entry), in global
// scratch.  The scratch is a RING OF WAVE SLOTS (64 lanes x 1 KB) PER XCD, 256 MB in all however large the
// batch: a wave takes the next ticket of the XCD it runs on (HW_REG_XCC_ID; one atomic per wave), which
// names slot = ticket mod 512 and generation = ticket / 512, waits until the slot's previous holder has
// released it (practically never: 384 waves of this kernel are resident per XCD, so the holder 512 tickets
// back is long gone), builds and reads its table there and releases the slot right after its last table
// read.  One launch covers the whole batch.  (Round 2 first bounded the scratch by walking the batch in
// launches of 786432 lanes over two table areas on two streams: every launch boundary cost 0.45 ms,
// 3.7 % at 2^22.  Slot reuse inside one launch costs nothing: 61.0 ms with the ring against 61.0 ms
// with 4 GiB of per-lane scratch, 63.3 ms chunked, same box.)
//   * every lane reads only the entries it wrote itself after acquiring the slot, so no data crosses
//     waves through the table; the only inter-wave traffic is the ticket counter and the slot flags;
//   * tickets and flags of one XCD live in cache lines no other XCD touches, holder and successor of a
//     slot run on the same XCD by construction, flags are stored and polled with agent-scope accesses
//     (`sc1`: past the L1 of the CU) -- the per-XCD L2 they meet in is coherent for its own CUs;
//   * a holder never waits for a later ticket, so the waits cannot cycle; the spin is bounded anyway: a wave that
//     has polled for ~4 s GIVES UP: it sets the device's error word (pinned host memory; its device address is
//     parked behind the ring state at RING_ERR_WORD), touches no table slot, never publishes its generation but marks
//     the slot DEAD (its successors on that slot give up at their first poll: fail closed, without waiting), writes POISON into the rows it owns (limbs / bytes of
//     all ones, ok = 0) and ends.  The context stays usable: the host reports ZC_ERR_HIP at the next entry point
//     or synchronisation that touches the device (zerocaf_hip.hip: ring_check; no trap, so no sticky HIP error).
// Invariants the host side keeps (fast_ring): ONE stream per device state orders every user of the ring;
// tickets and flags are zeroed on that stream before every launch (the generation field of the parked word
// has 19 bits: a launch hands out fewer than 2^19 * slots tickets per XCD, i.e. at most slots * 2^25 lanes --
// fast_ring cuts longer batches into several launches); the error word is cleared only when it is read.
// (A persistent grid -- 768 resident workgroups walking the tiles, 192 MB of tables -- was measured
// 14 % SLOWER: co-resident waves then start together, stay in lock step and stall on their table
// loads together; short-lived workgroups drift apart and cover each other.  The ring keeps the
// short-lived workgroups.)
constexpr u32 RING_XCDS = 8;                      // HW_REG_XCC_ID is masked to this range
constexpr u32 RING_SLOTS = 512;                   // wave slots per XCD
constexpr u32 RING_TICKET_STRIDE = 32;            // one 128-byte line per XCD's ticket counter
constexpr u32 RING_STATE_WORDS = RING_XCDS * RING_TICKET_STRIDE + RING_XCDS * RING_SLOTS;   // zeroed per launch
constexpr u32 RING_ERR_WORD = RING_STATE_WORDS;    // behind them (two words, written once by the host): device address of the error word
constexpr u32 RING_ALLOC_WORDS = RING_STATE_WORDS + 32;
constexpr u32 RING_SLOT_DEAD = 0xFFFFFFFFu;       // flag word of a slot whose queue gave up (a generation count never gets there: < 2^19)
constexpr size_t RING_TABLE_BYTES = (size_t)RING_XCDS * RING_SLOTS * 64 * 1024;

// The lane's number, computed afresh wherever it is asked for: `volatile` keeps the compiler from sharing
// one v_mbcnt result between distant uses, which would pin a vector register across the whole kernel
// (the windowed-core kernels have none to spare).
ZC_DI u32 lane_id_fresh()
{
    u32 l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// A wave's slot: 64 KB starting at `base` (wave-uniform: scalar registers), lane l owns [l KB, (l + 1) KB).
struct ring_table {
    u32* base;                                    // nullptr: the wave gave up waiting (wave-uniform)
    ZC_DI u32* entry(int j) const
    {
        return base + (lane_id_fresh() * 256u + 32u * (u32)j);
    }
};
// returns the wave's slot.  What ring_release needs (the slot's flag word and the generation to
// publish, 13 + 19 bits) is parked in one LDS word per wave: the windowed core between the two calls is
// short of scalar registers as it is (their spills occupy vector registers).
ZC_DI ring_table ring_acquire(u32* __restrict__ table, u32* __restrict__ state, u32* __restrict__ hold, u32 slots)
{
    u32 xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= RING_XCDS - 1;
    const u32 lane = lane_id_fresh();
    u32 t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(state + RING_TICKET_STRIDE * xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = __builtin_amdgcn_readfirstlane(t);
#ifdef ZC_TEST_HOOKS                              // test build: the upper half of the argument shortens the spin (waves really give up)
    const u32 spin_limit = 1u << ((slots >> 16) ? (slots >> 16) : 22u);
    slots &= 0xFFFFu;
#else
    const u32 spin_limit = 1u << 22;
#endif
    const u32 slot = t % slots, gen = t / slots;          // slots <= RING_SLOTS (fewer only to exercise the waits in tests)
    const u32 flag_word = RING_XCDS * RING_TICKET_STRIDE + xcc * RING_SLOTS + slot;
    if (lane == 0) *hold = flag_word | ((gen + 1) << 13);
    if (gen) {                                    // the flag counts the generations that have released the slot
        u32 spins = 0;
        for (;;) {
            const u32 f = __builtin_amdgcn_readfirstlane(__hip_atomic_load(state + flag_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (f >= gen && f != RING_SLOT_DEAD) break;
            __builtin_amdgcn_s_sleep(16);
            if (f == RING_SLOT_DEAD || ++spins > spin_limit) {
                // give up: flag the device, take no slot (the caller writes poison and ends), and leave the slot marked DEAD so
                // that the generations queued behind this one give up at their first poll instead of spinning ~4 s each (a
                // launch of 2^24 lanes has 64 generations per slot).  ring_release publishes with an atomic MAX, so a late
                // holder's release cannot bring a dead slot back: it stays DEAD (0xFFFFFFFF is the maximum) until the host
                // zeroes the ring state before the next launch, and the error word is already set.
                if (lane == 0) {
                    u32* err = *reinterpret_cast<u32* const*>(state + RING_ERR_WORD);
                    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(state + flag_word, RING_SLOT_DEAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return ring_table{nullptr};
            }
        }
    }
    return ring_table{table + (size_t)(xcc * RING_SLOTS + slot) * (64 * 256)};
}
ZC_DI void ring_release(u32* __restrict__ state, const u32* __restrict__ hold)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every table read of every lane has returned
    if (lane_id_fresh() == 0) {
        const u32 h = *hold;
        // generations only grow, so MAX is a store for a live slot -- and leaves RING_SLOT_DEAD in place
        __hip_atomic_fetch_max(state + (h & 0x1FFFu), h >> 13, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
ZC_KERNEL_3W void k_ed_scalar_mul_fast(const u64* p, const u64* k, u32 k_stride, u64* out, u32* table, u32* ring, u32 ring_slots, u32 n)
{
    __shared__ int8_t sdig[66 * ZC_BLOCK];
    const int tid = threadIdx.x;
    const u32 i = blockIdx.x * ZC_BLOCK + tid;
    const bool valid = i < n;
    const u32 ii = valid ? i : 0;
    u64 l[5];
    load_scalar(l, k + k_stride * ii);
    int top = scalar_digits16(sdig + tid, ZC_BLOCK, l);
    if (!valid) top = -1;
    top = wave_max_small(top);
    __shared__ u32 hold[ZC_BLOCK / 64];
    const ring_table mine = ring_acquire(table, ring, hold + (tid >> 6), ring_slots);
    if (!mine.base) {                                      // the wave gave up waiting for its table slot (wave-uniform): poison, no barrier follows
        if (valid)
            for (int j = 0; j < 20; j++) out[20 * (size_t)i + j] = ~(u64)0;
        return;
    }
    const pt Q = scalar_mul_fast(pt_load(p + 20 * (size_t)ii), mine, sdig + tid, ZC_BLOCK, top);
    ring_release(ring, hold + (threadIdx.x >> 6));
    if (valid) pt_store(out + 20 * (size_t)i, Q);
}
// fused config-4 path on the fast core: the boundary is bytes in / bytes out, and a Ristretto
// encoding depends only on the group element, so the outputs stay bit-identical to the reference
ZC_KERNEL_3W void k_ris_roundtrip_mul_fast(const uint8_t* in, const u64* k, uint8_t* out, uint8_t* ok, u32* table, u32* ring, u32 ring_slots, u32 n)
{
    __shared__ int8_t sdig[66 * ZC_BLOCK];
    const int tid = threadIdx.x;
    const u32 wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u32 i = blockIdx.x * ZC_BLOCK + tid;
    const bool valid = i < n;
    const u32 ii = valid ? i : 0;
    u64 w[4], l[5];
    load_words256(w, in + 32 * (size_t)ii);
    load_scalar(l, k + 5 * (size_t)ii);
    int top = scalar_digits16(sdig + tid, ZC_BLOCK, l);
    pt P;
    const bool dec = ris_decompress(P, w);
    if (!valid || !dec) top = -1;
    top = wave_max_small(top);
    __shared__ u32 hold[ZC_BLOCK / 64];                    // the table slot is held for the multiplication only
    const ring_table mine = ring_acquire(table, ring, hold + wave_in_block, ring_slots);
    if (!mine.base) {                                      // the wave gave up waiting for its table slot (wave-uniform): poison, no barrier follows
        if (valid) {
            w[0] = w[1] = w[2] = w[3] = ~(u64)0;
            store_words256(out + 32 * (size_t)i, w);
            if (ok) ok[i] = 0;
        }
        return;
    }
    pt Q = scalar_mul_fast(P, mine, sdig + tid, ZC_BLOCK, top);
    ring_release(ring, hold + wave_in_block);
    Q = pt_select(dec, Q, pt_identity());
    fe_to_words256(w, ris_compress(Q));
    if (!dec) w[0] = w[1] = w[2] = w[3] = 0;
    // The output index is formed again here from values that occupy no vector register in between:
    // the wave's number inside the workgroup sits in an SGPR since the kernel's first instruction and
    // the lane number comes from v_mbcnt (a function of the lane position alone).  threadIdx.x itself
    // is a live-in VGPR, and anything derived from it stays live -- and gets spilled -- across the
    // three exponentiation-sized phases.
    const u32 lane_late = lane_id_fresh();
    const u32 i_late = blockIdx.x * ZC_BLOCK + wave_in_block * 64u + lane_late;
    if (i_late < n) {
        store_words256(out + 32 * (size_t)i_late, w);
        if (ok) ok[i_late] = dec ? 1 : 0;
    }
}

// ---- fixed-base multiplication of the curve basepoint ------------------------
// The reference's only fixed-base routine, window_naf_mul (edwards.rs:155-171), mis-indexes its
// odd-multiples table and its test is commented out; the key-generation half of its ECDH bench
// therefore uses the variable-base algorithms on BASEPOINT.  This is the correct fixed-base
// counterpart: a comb table T[w][j] = (j+1) * 256^w * B (33 windows x 128 cached affine points, 528 KB,
// L2-resident) and k*B = sum_w sign(d_w) * T[w][|d_w| - 1] over the signed radix-256 digits --
// 33 cached additions, no doublings (round 1 / first half of round 2: radix 16, 66 additions over a 66 KB table).  Equal to `&BASEPOINT * &k` as a group element; the fused
// variant emits Ristretto encodings, which are bit-identical to the reference's.
constexpr int ZC_BASE_WINDOWS = 33;                      // signed radix-256 digits of a 260-bit scalar (+ carry)
constexpr int ZC_BASE_ENTRIES = 128;                     // |digit| = 1 .. 128

// lane j (0..127) builds the column (j+1) * 256^w * B for w = 0..32
ZC_KERNEL void k_base_table_build(u32* table)
{
    const int j = threadIdx.x;
    if (j >= ZC_BASE_ENTRIES) return;
    pt B;
    B.X = fe_const<FP>(ModP::BASE_X_M);
    B.Y = fe_const<FP>(ModP::BASE_Y_M);
    B.Z = fe_one_m<FP>();
    B.T = fe_const<FP>(ModP::BASE_T_M);
    pt P = B;
    for (int a = 0; a < ZC_BASE_ENTRIES - 1; a++) {
        const pt s = pt_add(P, B);
        P = pt_select(a < j, s, P);                       // P = (j+1) * B
    }
    for (int w = 0; w < ZC_BASE_WINDOWS; w++) {
        // entries are normalised to Z = 1 once, here, so every later addition against them is mixed
        const fe zi = fp_invert(P.Z);
        pt A;
        A.X = fp_mul(P.X, zi);
        A.Y = fp_mul(P.Y, zi);
        A.Z = fe_one_m<FP>();
        A.T = fp_mul(A.X, A.Y);
        niels_store(table + 32 * (w * ZC_BASE_ENTRIES + j), niels_from_pt(A));
#pragma unroll 1
        for (int t = 0; t < 8; t++) P = pt_add(P, P);
    }
}
// Signed radix-256 digits of the 260-bit scalar, d_i in [-128, 128), 33 digits (carry included), stored as
// bytes at dig[i * stride]; returns the index of the highest non-zero digit or -1.
ZC_DI int scalar_digits256(int8_t* __restrict__ dig, int stride, const u64 (&l)[5])
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
    for (int i = 0; i < ZC_BASE_WINDOWS; i++) {
        int d = (int)((w[i >> 2] >> ((i & 3) * 8)) & 255u) + carry;
        carry = d >= 128;
        d -= carry << 8;
        dig[i * stride] = (int8_t)d;
        if (d != 0) top = i;
    }
    return top;
}
ZC_DI pt base_mul(const u32* __restrict__ table, const int8_t* __restrict__ dig, int stride, int top)
{
    pt Q = pt_identity();
    for (int w = top; w >= 0; w--) {
        const int d = dig[w * stride];
        const int mag = d < 0 ? -d : d;
        niels c = niels_identity();
        if (mag != 0) c = niels_load(table + 32 * (w * ZC_BASE_ENTRIES + mag - 1));
        Q = pt_add_cached<false, true>(Q, niels_cond_neg(d < 0, c));     // table entries and the identity have z = 1
    }
    return Q;
}
ZC_KERNEL void k_ed_mul_base(const u64* k, u64* out, const u32* table, size_t n)
{
    __shared__ int8_t sdig[ZC_BASE_WINDOWS * ZC_BLOCK];
    const int tid = threadIdx.x;
    const size_t i = gid();
    const bool valid = i < n;
    u64 l[5];
    load_scalar(l, k + 5 * (valid ? i : 0));
    int top = scalar_digits256(sdig + tid, ZC_BLOCK, l);
    if (!valid) top = -1;
    top = wave_max_small(top);
    const pt Q = base_mul(table, sdig + tid, ZC_BLOCK, top);
    if (valid) pt_store(out + 20 * i, Q);
}
// key generation: scalars -> compressed Ristretto public keys, (RISTRETTO_BASEPOINT * k).compress()
ZC_KERNEL void k_ris_mul_base_compress(const u64* k, uint8_t* out, const u32* table, size_t n)
{
    __shared__ int8_t sdig[ZC_BASE_WINDOWS * ZC_BLOCK];
    const int tid = threadIdx.x;
    const size_t i = gid();
    const bool valid = i < n;
    u64 l[5], w[4];
    load_scalar(l, k + 5 * (valid ? i : 0));
    int top = scalar_digits256(sdig + tid, ZC_BLOCK, l);
    if (!valid) top = -1;
    top = wave_max_small(top);
    const pt Q = base_mul(table, sdig + tid, ZC_BLOCK, top);
    fe_to_words256(w, ris_compress(Q));
    if (valid) store_words256(out + 32 * i, w);
}

// ---- fixed-base w-NAF over the odd multiples of the basepoint (third algorithm) -------------
// window_naf_mul (edwards.rs:155-171): Q = identity; for i from the top digit down { Q = 2Q; d = wNAF_w(k)[i];
// d > 0: Q += T[d]; d < 0: Q -= T[|d|] } over BASEPOINT_ODD_MULTIPLES_TABLE (constants.rs:216-972: entry 0 = identity,
// entry j = (2j - 1) B, 125 odd multiples).  The reference indexes the table with the digit itself and reads the digits
// 249..0 only (its test is commented out, :1619-1635); this is the algorithm with both repaired: entry (|d| + 1) / 2 and
// all 256 digits compute_window_NAF emits (scalar.rs:396-415, the loop literally -- naf_step above), so that the result
// is (sum_i d_i 2^i) B = k B for every canonical scalar.  ONE launch: the digits of a lane sit in LDS (256 bytes per
// lane), the table is rebuilt on the device as 125 cached AFFINE records (16 KB, L2-resident; checked entry by entry
// against the reference's table in the GPU tier), every step is a dedicated doubling and a 7-multiplication mixed
// addition under the lanes' digit mask.  Same group element as `&BASEPOINT * &k` and as the comb (zc_ed_mul_base) --
// the comb (33 additions, no doublings) stays the fast answer; this one is the reference's named algorithm.
constexpr int ZC_ODD_ENTRIES = 125;
ZC_KERNEL void k_odd_table_build(u32* table)                 // lane j: (2j + 1) B, Z = 1, cached form
{
    const int j = threadIdx.x;
    if (j >= ZC_ODD_ENTRIES) return;
    pt B;
    B.X = fe_const<FP>(ModP::BASE_X_M);
    B.Y = fe_const<FP>(ModP::BASE_Y_M);
    B.Z = fe_one_m<FP>();
    B.T = fe_const<FP>(ModP::BASE_T_M);
    const pt B2 = pt_add(B, B);
    pt P = B;
    for (int a = 0; a < ZC_ODD_ENTRIES - 1; a++) {
        const pt s = pt_add(P, B2);
        P = pt_select(a < j, s, P);
    }
    const fe zi = fp_invert(P.Z);
    pt A;
    A.X = fp_mul(P.X, zi);
    A.Y = fp_mul(P.Y, zi);
    A.Z = fe_one_m<FP>();
    A.T = fp_mul(A.X, A.Y);
    niels_store(table + 32 * j, niels_from_pt(A));
}
#ifdef ZC_TEST_HOOKS
// test build only: the cached records back as points (4x : 4y : 4 : 4xy), to be compared with the reference's table
ZC_KERNEL void k_test_odd_table_dump(const u32* table, u64* out)
{
    const int j = threadIdx.x;
    if (j >= ZC_ODD_ENTRIES) return;
    const niels c = niels_load(table + 32 * j);
    const fe two = fe_add(fe_one_m<FP>(), fe_one_m<FP>());
    pt P;
    P.X = fp_mul(fp_sub(c.ypx, c.ymx), two);
    P.Y = fp_mul(fe_add(c.ypx, c.ymx), two);
    P.Z = fp_mul(two, two);
    P.T = fp_sub(fp_mul(c.ypx, c.ypx), fp_mul(c.ymx, c.ymx));
    pt_store(out + 20 * j, P);
}
#endif
ZC_KERNEL_2W void k_ed_mul_base_wnaf(const u64* k, u32 width, u64* out, const u32* table, size_t n)
{
    __shared__ int8_t sdig[256 * ZC_BLOCK];                   // digit i of lane t: sdig[i * ZC_BLOCK + t]
    const int tid = threadIdx.x;
    const size_t i = gid();
    const bool valid = i < n;
    u64 kk[5], m[5];
    load5(kk, k + 5 * (valid ? i : 0));
    limbs52_of_modulus<ModL>(m);
    int top = -1;
    for (int d = 0; d < 256; d++) {
        const int ki = naf_step(kk, m, width);
        sdig[d * ZC_BLOCK + tid] = (int8_t)ki;
        if (ki != 0) top = d;
    }
    if (!valid) top = -1;
    top = wave_max_9bit(top);
    pt Q = pt_identity();
    for (int d = top; d >= 0; d--) {
        if (d != top) Q = pt_double_fast<true>(Q);          // doubling the identity changes nothing: the leading one is skipped
        const int ki = sdig[d * ZC_BLOCK + tid];
        const int mag = ki < 0 ? -ki : ki;
        niels c = niels_identity();
        if (mag != 0) c = niels_load(table + 32 * ((mag - 1) >> 1));                 // |d| = 2j - 1  ->  record j - 1
        Q = pt_add_cached<false, true>(Q, niels_cond_neg(ki < 0, c));             // table records and the identity have z = 1
    }
    if (valid) pt_store(out + 20 * i, Q);
}

// ltr_bin_mul (MODE 1) / binary_naf_mul (MODE 2): limbs identical to the reference's variants
template <int MODE>
ZC_DI void scalar_mul_ltr_body(const u64* p, const u64* k, u64* out, size_t n)
{
    __shared__ u32 sp[8 * ZC_BLOCK];
    __shared__ u32 sn[8 * ZC_BLOCK];
    const int tid = threadIdx.x;
    const size_t i = gid();
    const bool valid = i < n;
    const size_t ii = valid ? i : 0;
    u64 l[5];
    load5(l, k + 5 * ii);
    int top = ltr_digits<MODE>(sp + tid, sn + tid, ZC_BLOCK, l);
    if (!valid) top = -1;
    const pt Q = scalar_mul_ltr(pt_load(p + 20 * ii), sp + tid, sn + tid, ZC_BLOCK, top);
    if (valid) pt_store(out + 20 * i, Q);
}
ZC_KERNEL void k_ed_scalar_mul_ltr_bin(const u64* p, const u64* k, u64* out, size_t n) { scalar_mul_ltr_body<1>(p, k, out, n); }
ZC_KERNEL void k_ed_scalar_mul_naf(const u64* p, const u64* k, u64* out, size_t n) { scalar_mul_ltr_body<2>(p, k, out, n); }

// ------------------------------------------------------------------ affine / eq / Edwards codec
ZC_KERNEL void k_ed_to_affine(const u64* p, u64* xy, uint8_t* ok, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    fe x, y;
    const bool o = ed_to_affine(x, y, pt_load(p + 20 * i));
    fe_store_canon<FP>(xy + 10 * i, x);
    fe_store_canon<FP>(xy + 10 * i + 5, y);
    if (ok) ok[i] = o ? 1 : 0;
}
// large batches: one inversion per `c` consecutive points (ed_to_affine_chunk)
ZC_KERNEL void k_ed_to_affine_chunked(const u64* p, u64* xy, uint8_t* ok, size_t n, int c)
{
    const size_t lanes = (n + (size_t)c - 1) / (size_t)c, g = gid();
    if (g < lanes) ed_to_affine_chunk(p, xy, ok, n, g, lanes, c);
}
ZC_KERNEL void k_ed_eq(const u64* p, const u64* q, uint8_t* eq, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    eq[i] = ed_eq(pt_load_plain(p + 20 * i), pt_load_plain(q + 20 * i)) ? 1 : 0;   // cross products of plain coordinates: both sides carry 1/R
}
ZC_KERNEL_2W void k_ed_compress(const u64* p, uint8_t* out, uint8_t* ok, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    u64 w[4];
    const bool o = ed_compress(w, pt_load(p + 20 * i));
    if (!o) w[0] = w[1] = w[2] = w[3] = 0;
    store_words256(out + 32 * i, w);
    if (ok) ok[i] = o ? 1 : 0;
}
ZC_KERNEL void k_ed_decompress(const uint8_t* in, u64* out, uint8_t* ok, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    u64 w[4];
    load_words256(w, in + 32 * i);
    pt r;
    const bool o = ed_decompress(r, w);
    pt_store(out + 20 * i, pt_select(o, r, pt_identity()));
    if (ok) ok[i] = o ? 1 : 0;
}

// ------------------------------------------------------------------ Ristretto
ZC_KERNEL void k_ris_compress(const u64* p, uint8_t* out, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    u64 w[4];
    fe_to_words256(w, ris_compress(pt_load(p + 20 * i)));
    store_words256(out + 32 * i, w);
}
ZC_KERNEL void k_ris_decompress(const uint8_t* in, u64* out, uint8_t* ok, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    u64 w[4];
    load_words256(w, in + 32 * i);
    pt r;
    const bool o = ris_decompress(r, w);
    pt_store(out + 20 * i, pt_select(o, r, pt_identity()));
    if (ok) ok[i] = o ? 1 : 0;
}
ZC_KERNEL void k_ris_eq(const u64* p, const u64* q, uint8_t* eq, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    eq[i] = ris_eq(pt_load_plain(p + 20 * i), pt_load_plain(q + 20 * i)) ? 1 : 0;  // plain coordinates: both sides carry 1/R
}
// fused config-4 path: 32 B in -> registers -> 32 B out; the point never touches HBM
ZC_KERNEL_2W void k_ris_roundtrip_mul(const uint8_t* in, const u64* k, uint8_t* out, uint8_t* ok, size_t n)
{
    __shared__ u32 sk[9 * ZC_BLOCK];
    __shared__ u32 skey[ZC_BLOCK];
    __shared__ u32 sperm[ZC_BLOCK];
    __shared__ int snb[ZC_BLOCK];
    const int tid = threadIdx.x;
    const size_t i = gid();
    const bool valid = i < n;
    const size_t own = valid ? i : 0;
    u64 w[4], l[5];
    load_scalar(l, k + 5 * own);
    int nbits;
    scalar_to_words(sk + tid, ZC_BLOCK, l, nbits);
    if (!valid) nbits = 0;
    snb[tid] = nbits;
    const int e = block_cost_rank(skey, sperm, valid ? scalar_cost(l) : 0);
    nbits = snb[e];
    const size_t base = (size_t)blockIdx.x * ZC_BLOCK;
    const bool run = base + e < n;
    const size_t ii = run ? base + e : 0;
    load_words256(w, in + 32 * ii);
    pt P;
    const bool dec = ris_decompress(P, w);
    if (!run || !dec) nbits = 0;
    const pt Q = scalar_mul_unified(P, sk + e, ZC_BLOCK, nbits);
    fe_to_words256(w, ris_compress(Q));
    if (!dec) w[0] = w[1] = w[2] = w[3] = 0;
    if (run) {
        store_words256(out + 32 * ii, w);
        if (ok) ok[ii] = dec ? 1 : 0;
    }
}

// ------------------------------------------------------------------ "next" rows: validity, Elligator, ProjectivePoint
ZC_KERNEL void k_ed_is_valid(const u64* p, uint8_t* valid, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    valid[i] = ed_is_valid(pt_load_plain(p + 20 * i)) ? 1 : 0;        // plain coordinates: both sides of the equation carry 1/R^3
}
// RistrettoPoint::is_valid (ristretto.rs:205-222): (P * L == identity) & on-curve
ZC_KERNEL void k_ris_is_valid(const u64* p, uint8_t* valid, size_t n)
{
    __shared__ u32 sk[9 * ZC_BLOCK];
    const int tid = threadIdx.x;
    const size_t i = gid();
    const bool in = i < n;
    const pt P = pt_load(p + 20 * (in ? i : 0));
    u64 l[5];
    fe_to_limbs52(l, fe_const<ModL>(ModL::N));
    int nbits;
    scalar_to_words(sk + tid, ZC_BLOCK, l, nbits);
    const pt Q = scalar_mul_unified(P, sk + tid, ZC_BLOCK, in ? nbits : 0);
    const bool order_l = ed_eq(Q, pt_identity());
    if (in) valid[i] = (order_l && ed_is_valid(P)) ? 1 : 0;
}
ZC_KERNEL void k_ris_elligator(const u64* r0, u64* out, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    pt_store(out + 20 * i, ris_elligator(fe_load_mont<FP>(r0 + 5 * i)));
}
// from_uniform_bytes (ristretto.rs:493-507): two Elligator maps, one addition
ZC_KERNEL_2W void k_ris_from_uniform_bytes(const uint8_t* in, u64* out, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    u64 w[4];
    load_words256(w, in + 64 * i);
    const pt R1 = ris_elligator(mont_to<FP>(fe_from_words256(w)));
    load_words256(w, in + 64 * i + 32);
    const pt R2 = ris_elligator(mont_to<FP>(fe_from_words256(w)));
    pt_store(out + 20 * i, pt_add(R1, R2));
}
ZC_DI ppt ppt_load(const u64* __restrict__ p)
{
    ppt r;
    r.X = fe_load_mont<FP>(p);
    r.Y = fe_load_mont<FP>(p + 5);
    r.Z = fe_load_mont<FP>(p + 10);
    return r;
}
ZC_DI void ppt_store(u64* __restrict__ o, const ppt& p)
{
    fe_store_canon<FP>(o, p.X);
    fe_store_canon<FP>(o + 5, p.Y);
    fe_store_canon<FP>(o + 10, p.Z);
}
ZC_KERNEL void k_proj_add(const u64* p, const u64* q, u64* out, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    ppt_store(out + 15 * i, proj_add(ppt_load(p + 15 * i), ppt_load(q + 15 * i)));
}
ZC_KERNEL void k_proj_double(const u64* p, u64* out, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    ppt_store(out + 15 * i, proj_double(ppt_load(p + 15 * i)));
}
// From<ProjectivePoint> for EdwardsPoint (edwards.rs:402-417): (X*Z, Y*Z, Z^2, X*Y)
ZC_KERNEL void k_proj_to_extended(const u64* p, u64* out, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    const ppt a = ppt_load(p + 15 * i);
    pt r;
    r.X = fp_mul(a.X, a.Z);
    r.Y = fp_mul(a.Y, a.Z);
    r.Z = fp_sqr(a.Z);
    r.T = fp_mul(a.X, a.Y);
    pt_store(out + 20 * i, r);
}

// ---- E-x rows of SURVEY 8(a): coset4 and the ProjectivePoint operations beside add / double
// FOUR_COSET_GROUP[0..2] (backend/u64/constants.rs:141-183; coset4 never reads entry 3): X and Y limbs, Z = 1, T = 0
__device__ constexpr u64 FOUR_COSET_XY[3][2][5] = {
    {{1ull, 0ull, 0ull, 0ull, 0ull}, {0ull, 0ull, 0ull, 0ull, 0ull}},
    {{2099929430230996ull, 1464742363261928ull, 3309265759432790ull, 2285299817698826ull, 10215362715769ull}, {0ull, 0ull, 0ull, 0ull, 0ull}},
    {{0ull, 0ull, 0ull, 0ull, 0ull}, {671914833335276ull, 3916664325105025ull, 1367801ull, 0ull, 17592186044416ull}}};
// EdwardsPoint::coset4 (edwards.rs:603-610): [P, P + C0, P + C1, P + C2] through the unified addition
ZC_KERNEL void k_ed_coset4(const u64* p, u64* out4, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    const pt P = pt_load_plain(p + 20 * i);                 // plain domain (zc_curve.hip.h: pt_load_plain)
    {
        u64 l[20];
#pragma unroll
        for (int q = 0; q < 20; q++) l[q] = p[20 * i + q];
#pragma unroll
        for (int q = 0; q < 20; q++) out4[80 * i + q] = l[q];
    }
    for (int j = 0; j < 3; j++) {
        u64 x[5], y[5];
#pragma unroll
        for (int q = 0; q < 5; q++) {
            x[q] = FOUR_COSET_XY[j][0][q];
            y[q] = FOUR_COSET_XY[j][1][q];
        }
        pt C;
        C.X = fe_from_limbs52(x);
        C.Y = fe_from_limbs52(y);
        C.Z = fe_zero();
        C.Z.v[0] = 1;
        C.T = fe_zero();
        pt_store_plain(out4 + 80 * i + 20 * (j + 1), pt_add_plain(P, C));
    }
}
ZC_KERNEL void k_proj_neg(const u64* p, u64* out, size_t n)                          // edwards.rs:787-807: (-X, Y, Z)
{
    const size_t i = gid();
    if (i >= n) return;
    u64 m[5], X[5], Y[5], Z[5], nx[5];                     // as k_ed_neg: radix-2^52 negation, Y and Z copied
    const u64 z[5] = {0, 0, 0, 0, 0};
    limbs52_of_modulus<ModP>(m);
    load5(X, p + 15 * i);
    load5(Y, p + 15 * i + 5);
    load5(Z, p + 15 * i + 10);
    sub52(nx, z, X, m);
    store5(out + 15 * i, nx);
    store5(out + 15 * i + 5, Y);
    store5(out + 15 * i + 10, Z);
}
ZC_KERNEL void k_proj_sub(const u64* p, const u64* q, u64* out, size_t n)            // edwards.rs:851-879: self + (-other)
{
    const size_t i = gid();
    if (i >= n) return;
    ppt b = ppt_load(q + 15 * i);
    b.X = fe_reduce<FP>(fp_neg(b.X));
    ppt_store(out + 15 * i, proj_add(ppt_load(p + 15 * i), b));
}
// ProjectivePoint == (edwards.rs:701-711): equality of the affine images, cross-multiplied; Z = 0 (where the
// reference's inverse() panics) compares unequal
ZC_KERNEL void k_proj_eq(const u64* p, const u64* q, uint8_t* eq, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    ppt a, b;                                              // plain coordinates: both sides of each comparison carry 1/R
    a.X = fe_load_plain(p + 15 * i);
    a.Y = fe_load_plain(p + 15 * i + 5);
    a.Z = fe_load_plain(p + 15 * i + 10);
    b.X = fe_load_plain(q + 15 * i);
    b.Y = fe_load_plain(q + 15 * i + 5);
    b.Z = fe_load_plain(q + 15 * i + 10);
    const bool ex = fp_eq(fp_mul(a.X, b.Z), fp_mul(b.X, a.Z));
    const bool ey = fp_eq(fp_mul(a.Y, b.Z), fp_mul(b.Y, a.Z));
    eq[i] = (ex && ey && !fp_is_zero(a.Z) && !fp_is_zero(b.Z)) ? 1 : 0;
}
ZC_KERNEL void k_proj_is_valid(const u64* p, uint8_t* valid, size_t n)               // edwards.rs:733-748
{
    const size_t i = gid();
    if (i >= n) return;
    pt e;                                                  // plain coordinates, as k_ed_is_valid
    e.X = fe_load_plain(p + 15 * i);
    e.Y = fe_load_plain(p + 15 * i + 5);
    e.Z = fe_load_plain(p + 15 * i + 10);
    e.T = fe_zero();
    valid[i] = ed_is_valid(e) ? 1 : 0;
}
// Mul<Scalar> for ProjectivePoint (edwards.rs:881-912) = double_and_add (:102-120) over the projective
// formulas: Q = (0, 1, 1); while n != 0 { if n odd: Q = Q + N (:809-834); N = N.double() (:915-942, dedicated);
// n >>= 1 }.  The two formulas differ, so there is no unified step here: one lane per element, the wave
// runs the addition whenever any lane has its bit set.  Same limbs as the reference (first addition literal).
ZC_KERNEL void k_proj_scalar_mul(const u64* p, const u64* k, u64* out, size_t n)
{
    const size_t i = gid();
    if (i >= n) return;
    u64 l[5];
    load_scalar(l, k + 5 * i);
    ppt N = ppt_load(p + 15 * i), Q;
    Q.X = fe_zero();
    Q.Y = fe_one_m<FP>();
    Q.Z = fe_one_m<FP>();
    while ((l[0] | l[1] | l[2] | l[3] | l[4]) != 0) {
        if (l[0] & 1) Q = proj_add(Q, N);
        N = proj_double(N);
        half_without_mod52(l);
    }
    ppt_store(out + 15 * i, Q);
}

// ------------------------------------------------------------------ reduction helper for zc_msm
// out[i] = in[2i] + in[2i+1] (odd tail copied): pairwise fold of a point array
ZC_KERNEL void k_ed_fold_pairs(const u64* in, u64* out, size_t n_in)
{
    const size_t i = gid();
    const size_t n_out = (n_in + 1) / 2;
    if (i >= n_out) return;
    const pt a = pt_load(in + 20 * (2 * i));
    if (2 * i + 1 < n_in) pt_store(out + 20 * i, pt_add(a, pt_load(in + 20 * (2 * i + 1))));
    else pt_store(out + 20 * i, a);
}

// (k_ed_fold_ordered, the rank-order fold of the MSM exchange: zc_quad.hip.h)

}  // namespace zc

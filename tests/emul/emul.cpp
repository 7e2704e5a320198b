// emul.cpp -- TEST-ONLY host build of the device arithmetic headers.
// Compiles dusk_zerocaf_amd/csrc/zc_arith.hip.h + zc_curve.hip.h with g++ (the HIP
// qualifiers expand to nothing under a non-HIP compiler) so the exact limb
// algorithms the kernels run can be checked against the oracle in the CPU-only
// test tier, before any GPU time is spent.  Never shipped, never loaded by
// dusk_zerocaf_amd/: the product path has no CPU fallback.
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "../../dusk_zerocaf_amd/csrc/zc_curve.hip.h"

// -DZC_CHECK_BOUNDS build: every lazy-reduction precondition in zc_arith.hip.h is asserted
extern "C" void zc_bound_fail(const char* what, int line)
{
    std::fprintf(stderr, "zc_arith.hip.h:%d: bound violated: %s\n", line, what);
    std::abort();
}

using namespace zc;

static pt scalar_mul_seq(const pt& P, const u64 (&raw)[5])
{
    u64 l[5];
    load_scalar(l, raw);                                  // as the kernels load their scalar operand
    u32 w[9];
    int nbits;
    scalar_to_words(w, 1, l, nbits);
    return scalar_mul_unified(P, w, 1, nbits);        // the very function the kernels run per lane
}
template <int MODE>
static pt scalar_mul_ltr_seq(const pt& P, const u64 (&l)[5])
{
    u32 pb[8], nb[8];
    const int top = ltr_digits<MODE>(pb, nb, 1, l);
    return scalar_mul_ltr(P, pb, nb, 1, top);
}

template <class F>
static void store_plain(u64* o, const fe& x)
{
    u64 l[5];
    fe_to_limbs52(l, fe_cond_sub_n<F>(fe_cond_sub_n<F>(x)));
    for (int i = 0; i < 5; i++) o[i] = l[i];
}
static void ld5(u64 (&l)[5], const u64* p) { for (int i = 0; i < 5; i++) l[i] = p[i]; }

extern "C" {
void emul_fe_mul(const u64* a, const u64* b, u64* out, size_t n, int modl)
{
    for (size_t i = 0; i < n; i++) {
        u64 x[5], y[5];
        ld5(x, a + 5 * i); ld5(y, b + 5 * i);
        if (modl) store_plain<ModL>(out + 5 * i, mont_mul<ModL>(mont_to<ModL>(fe_from_limbs52(x)), fe_from_limbs52(y)));
        else store_plain<ModP>(out + 5 * i, mont_mul<ModP>(mont_to<ModP>(fe_from_limbs52(x)), fe_from_limbs52(y)));
    }
}
void emul_fe_square(const u64* a, u64* out, size_t n, int modl)
{
    for (size_t i = 0; i < n; i++) {
        u64 x[5];
        ld5(x, a + 5 * i);
        if (modl) store_plain<ModL>(out + 5 * i, mont_mul<ModL>(mont_sqr<ModL>(fe_from_limbs52(x)), fe_const<ModL>(ModL::RR)));
        else store_plain<ModP>(out + 5 * i, mont_mul<ModP>(mont_sqr<ModP>(fe_from_limbs52(x)), fe_const<ModP>(ModP::RR)));
    }
}
// what k_fe_mul / k_fe_square / k_sc_mul / k_sc_square run per lane: the one-pass product for canonical operands, the two
// Montgomery passes for patterns at or above 2^TOPBIT
void emul_mulmod(const u64* a, const u64* b, u64* prod, u64* sq, size_t n, int modl)
{
    for (size_t i = 0; i < n; i++) {
        u64 x[5], y[5], r[5];
        ld5(x, a + 5 * i); ld5(y, b + 5 * i);
        if (modl) fe_mulmod_limbs52<ModL>(r, x, y); else fe_mulmod_limbs52<ModP>(r, x, y);
        for (int j = 0; j < 5; j++) prod[5 * i + j] = r[j];
        if (modl) fe_sqrmod_limbs52<ModL>(r, x); else fe_sqrmod_limbs52<ModP>(r, x);
        for (int j = 0; j < 5; j++) sq[5 * i + j] = r[j];
    }
}
// the independent-chain multiplier pair used by small launches and the MSM bucket sums
void emul_fe_mul_square_ilp(const u64* a, const u64* b, u64* prod, u64* sq, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        u64 x[5], y[5];
        ld5(x, a + 5 * i);
        ld5(y, b + 5 * i);
        store_plain<ModP>(prod + 5 * i, mont_mul_ilp<ModP>(mont_mul_ilp<ModP>(fe_from_limbs52(x), fe_const<ModP>(ModP::RR)), fe_from_limbs52(y)));
        store_plain<ModP>(sq + 5 * i, mont_mul_ilp<ModP>(mont_sqr_ilp<ModP>(fe_from_limbs52(x)), fe_const<ModP>(ModP::RR)));
    }
}
void emul_fe_invert(const u64* a, u64* out, uint8_t* ok, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        const fe x = fe_load_mont<FP>(a + 5 * i);
        ok[i] = !fp_is_zero(x);
        fe_store_canon<FP>(out + 5 * i, fp_invert(x));
    }
}
// legendre_symbol as the Jacobi symbol on positive division steps (max_rounds rounds, then the exponentiation),
// and the exponentiation alone; `rounds_out` (may be null): not available on the host build, kept 0
void emul_fe_legendre(const u64* a, uint8_t* jac, uint8_t* pw, size_t n, int max_rounds)
{
    for (size_t i = 0; i < n; i++) {
        const fe x = fe_load_mont<FP>(a + 5 * i);
        jac[i] = fp_legendre(x, max_rounds) ? 1 : 0;
        pw[i] = fp_legendre_pow(x) ? 1 : 0;
    }
}
void emul_fe_sqrt_ratio_i(const u64* u, const u64* v, u64* out, uint8_t* sq, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        fe r;
        sq[i] = fp_sqrt_ratio_i(r, fe_load_mont<FP>(u + 5 * i), fe_load_mont<FP>(v + 5 * i));
        fe_store_canon<FP>(out + 5 * i, r);
    }
}
void emul_ed_add(const u64* p, const u64* q, u64* out, size_t n)
{ for (size_t i = 0; i < n; i++) pt_store(out + 20 * i, pt_add(pt_load(p + 20 * i), pt_load(q + 20 * i))); }
// the stand-alone kernels' plain-domain path (k_ed_add / k_ed_sub / k_ed_double: no Montgomery conversions)
void emul_ed_add_plain(const u64* p, const u64* q, u64* out, size_t n, int mode)
{
    for (size_t i = 0; i < n; i++) {
        const pt a = pt_load_plain(p + 20 * i), b = pt_load_plain(q + 20 * i);
        pt_store_plain(out + 20 * i, mode == 0 ? pt_add_plain(a, b) : mode == 1 ? pt_add_plain(a, pt_neg(b)) : pt_double_plain(a));     // (mode 2 = k_ed_double: squarings)
    }
}
void emul_ed_sub(const u64* p, const u64* q, u64* out, size_t n)
{ for (size_t i = 0; i < n; i++) pt_store(out + 20 * i, pt_add(pt_load(p + 20 * i), pt_neg(pt_load(q + 20 * i)))); }
void emul_ed_scalar_mul(const u64* p, const u64* k, u64* out, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        u64 l[5];
        ld5(l, k + 5 * i);
        pt_store(out + 20 * i, scalar_mul_seq(pt_load(p + 20 * i), l));
    }
}
// the small-launch variant: same loop on the independent-chain multiplier
void emul_ed_scalar_mul_small(const u64* p, const u64* k, u64* out, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        u64 l[5];
        load_scalar(l, k + 5 * i);
        u32 w[9];
        int nbits;
        scalar_to_words(w, 1, l, nbits);
        pt_store(out + 20 * i, scalar_mul_unified<true>(pt_load(p + 20 * i), w, 1, nbits));
    }
}
void emul_ed_to_affine(const u64* p, u64* xy, uint8_t* ok, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        fe x, y;
        ok[i] = ed_to_affine(x, y, pt_load(p + 20 * i));
        fe_store_canon<FP>(xy + 10 * i, x);
        fe_store_canon<FP>(xy + 10 * i + 5, y);
    }
}
void emul_ed_eq(const u64* p, const u64* q, uint8_t* eq, size_t n)
{ for (size_t i = 0; i < n; i++) eq[i] = ed_eq(pt_load(p + 20 * i), pt_load(q + 20 * i)); }
void emul_ed_compress(const u64* p, u64* out32, uint8_t* ok, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        u64 w[4];
        ok[i] = ed_compress(w, pt_load(p + 20 * i));
        for (int j = 0; j < 4; j++) out32[4 * i + j] = ok[i] ? w[j] : 0;
    }
}
void emul_ed_decompress(const u64* in32, u64* out, uint8_t* ok, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        u64 w[4] = {in32[4 * i], in32[4 * i + 1], in32[4 * i + 2], in32[4 * i + 3]};
        pt r;
        ok[i] = ed_decompress(r, w);
        pt_store(out + 20 * i, pt_select(ok[i], r, pt_identity()));
    }
}
void emul_ris_compress(const u64* p, u64* out32, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        u64 w[4];
        fe_to_words256(w, ris_compress(pt_load(p + 20 * i)));
        for (int j = 0; j < 4; j++) out32[4 * i + j] = w[j];
    }
}
void emul_ris_decompress(const u64* in32, u64* out, uint8_t* ok, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        u64 w[4] = {in32[4 * i], in32[4 * i + 1], in32[4 * i + 2], in32[4 * i + 3]};
        pt r;
        ok[i] = ris_decompress(r, w);
        pt_store(out + 20 * i, pt_select(ok[i], r, pt_identity()));
    }
}
void emul_ris_eq(const u64* p, const u64* q, uint8_t* eq, size_t n)
{ for (size_t i = 0; i < n; i++) eq[i] = ris_eq(pt_load(p + 20 * i), pt_load(q + 20 * i)); }
}
extern "C" void emul_fe_invert_chunked(const u64* a, u64* out, uint8_t* ok, size_t n, int c)
{
    const size_t lanes = (n + (size_t)c - 1) / (size_t)c;      // as k_fe_invert_chunked: lane g takes g, g + lanes, ...
    for (size_t g = 0; g < lanes; g++) fe_invert_chunk(a, out, ok, n, g, lanes, c);
}
// k_fe_invert_chunked_lone / k_fe_div_chunked_lone: the same on the independent-chain multiplier (launches of one wave per SIMD)
extern "C" void emul_fe_invert_chunked_lone(const u64* a, u64* out, uint8_t* ok, size_t n, int c)
{
    const size_t lanes = (n + (size_t)c - 1) / (size_t)c;
    for (size_t g = 0; g < lanes; g++) fe_invert_chunk<true>(a, out, ok, n, g, lanes, c);
}
extern "C" void emul_fe_div_chunked_lone(const u64* a, const u64* b, u64* out, uint8_t* ok, size_t n, int c)
{
    const size_t lanes = (n + (size_t)c - 1) / (size_t)c;
    for (size_t g = 0; g < lanes; g++) fe_invert_chunk<true>(b, out, ok, n, g, lanes, c, a);
}
// the MSM bucket accumulation's inner loop (zc_msm.hip.h): cached-operand additions on the
// independent-chain multiplier, through the packed 128-byte record
extern "C" void emul_bucket_sum(const u64* pts, size_t n, u64* out)
{
    pt acc = pt_identity();
    for (size_t i = 0; i < n; i++) {
        u32 rec[32];
        niels_store(rec, niels_from_pt(pt_load(pts + 20 * i)));
        acc = pt_add_cached<true>(acc, niels_load(rec));
    }
    pt_store(out, acc);
}
extern "C" void emul_ed_to_affine_chunked(const u64* pts, u64* xy, uint8_t* ok, size_t n, int c)
{
    const size_t lanes = (n + (size_t)c - 1) / (size_t)c;
    for (size_t g = 0; g < lanes; g++) ed_to_affine_chunk(pts, xy, ok, n, g, lanes, c);
}
extern "C" void emul_fe_div_chunked(const u64* a, const u64* b, u64* out, uint8_t* ok, size_t n, int c)
{
    const size_t lanes = (n + (size_t)c - 1) / (size_t)c;
    for (size_t g = 0; g < lanes; g++) fe_invert_chunk(b, out, ok, n, g, lanes, c, a);
}
extern "C" void emul_ed_scalar_mul_mode(const u64* p, const u64* k, u64* out, size_t n, int mode)
{
    for (size_t i = 0; i < n; i++) {
        u64 l[5];
        ld5(l, k + 5 * i);
        const pt P = pt_load(p + 20 * i);
        pt_store(out + 20 * i, mode == 1 ? scalar_mul_ltr_seq<1>(P, l) : scalar_mul_ltr_seq<2>(P, l));
    }
}
extern "C" {
void emul_ris_elligator(const u64* r0, u64* out, size_t n)
{ for (size_t i = 0; i < n; i++) pt_store(out + 20 * i, ris_elligator(fe_load_mont<FP>(r0 + 5 * i))); }
void emul_ed_is_valid(const u64* p, uint8_t* v, size_t n)
{ for (size_t i = 0; i < n; i++) v[i] = ed_is_valid(pt_load(p + 20 * i)); }
static ppt pld(const u64* p) { ppt r; r.X = fe_load_mont<FP>(p); r.Y = fe_load_mont<FP>(p + 5); r.Z = fe_load_mont<FP>(p + 10); return r; }
static void pst(u64* o, const ppt& p) { fe_store_canon<FP>(o, p.X); fe_store_canon<FP>(o + 5, p.Y); fe_store_canon<FP>(o + 10, p.Z); }
void emul_proj_add(const u64* p, const u64* q, u64* out, size_t n)
{ for (size_t i = 0; i < n; i++) pst(out + 15 * i, proj_add(pld(p + 15 * i), pld(q + 15 * i))); }
void emul_proj_double(const u64* p, u64* out, size_t n)
{ for (size_t i = 0; i < n; i++) pst(out + 15 * i, proj_double(pld(p + 15 * i))); }
}
extern "C" void emul_ed_scalar_mul_fast(const u64* p, const u64* k, u64* out, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        u64 l[5];
        load_scalar(l, k + 5 * i);
        alignas(128) u32 table[256];
        int8_t dig[66];
        const int top = scalar_digits16(dig, 1, l);
        pt_store(out + 20 * i, scalar_mul_fast(pt_load(p + 20 * i), table_ptr{table}, dig, 1, top));
    }
}
// the termination rule of double_and_add on raw 260-bit patterns (scalar_effective) and the bit
// length / digit strings the kernels derive from it
extern "C" void emul_scalar_effective(const u64* k, u64* out, int* nbits, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        u64 l[5];
        load_scalar(l, k + 5 * i);
        u32 w[9];
        scalar_to_words(w, 1, l, nbits[i]);
        for (int j = 0; j < 5; j++) out[5 * i + j] = l[j];
    }
}

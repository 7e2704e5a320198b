/*
 * zc_ref.h -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * A plain-C restatement of the arithmetic of dusk-network/dusk-zerocaf's u64
 * backend for the hot path named in BASELINE.json (FieldElement / Scalar /
 * EdwardsPoint / Ristretto).  It is "reference-shaped": the same operation
 * structure as the Rust code (double Montgomery pass per Mul, HWCD add used
 * for doubling, LSB-first double_and_add, Savas-Koc inversion, Tonelli-Shanks
 * square root), so it serves both as the bit-exactness checker and as the timed
 * CPU baseline ("port").  Each function cites the reference file:line it
 * follows (paths relative to the reference checkout).
 *
 * Parity pin: every known-answer vector of the reference's own unit tests for
 * this path is checked against this oracle in tests/test_oracle_kat.py
 * (fixtures in tests/golden/ref_kats.json) and it is cross-checked against an
 * independent Python big-integer model (oracle/pymodel.py).  The Rust crate
 * itself cannot be built here (no rustc/cargo), so oracle/_ref does not exist.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  Nothing under dusk_zerocaf_amd/ links or calls it.
 */
#ifndef ZC_REF_H
#define ZC_REF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* FieldElement([u64;5]) / Scalar([u64;5]): radix 2^52, little-endian limbs.
 * src/backend/u64/field.rs:31-32, src/backend/u64/scalar.rs:26-27            */
typedef struct { uint64_t l[5]; } zr_fe;
typedef zr_fe zr_sc;
/* EdwardsPoint{X,Y,Z,T}: src/edwards.rs:336-342 */
typedef struct { zr_fe X, Y, Z, T; } zr_pt;

/* ---- single-element API (field, mod p) ---- */
void zr_fe_add(zr_fe *r, const zr_fe *a, const zr_fe *b);
void zr_fe_sub(zr_fe *r, const zr_fe *a, const zr_fe *b);
void zr_fe_neg(zr_fe *r, const zr_fe *a);
void zr_fe_mul(zr_fe *r, const zr_fe *a, const zr_fe *b);
void zr_fe_square(zr_fe *r, const zr_fe *a);
void zr_fe_montgomery_mul(zr_fe *r, const zr_fe *a, const zr_fe *b);
void zr_fe_to_montgomery(zr_fe *r, const zr_fe *a);
void zr_fe_from_montgomery(zr_fe *r, const zr_fe *a);
void zr_fe_half(zr_fe *r, const zr_fe *a);
void zr_fe_half_without_mod(zr_fe *r, const zr_fe *a);
int  zr_fe_inverse(zr_fe *r, const zr_fe *a);            /* 0 if a==0 (reference panics) */
int  zr_fe_div(zr_fe *r, const zr_fe *a, const zr_fe *b);/* 0 if b==0 (reference panics) */
void zr_fe_pow(zr_fe *r, const zr_fe *a, const zr_fe *e);
int  zr_fe_legendre_symbol(const zr_fe *a);              /* Choice: 1 = QR (also for 0), 0 = non-QR */
int  zr_fe_mod_sqrt(zr_fe *r, const zr_fe *a, int sign); /* 1 = Some, 0 = None */
int  zr_fe_sqrt_ratio_i(zr_fe *r, const zr_fe *u, const zr_fe *v); /* returns Choice */
int  zr_fe_inv_sqrt(zr_fe *r, const zr_fe *a);
int  zr_fe_is_positive(const zr_fe *a);
int  zr_fe_is_even(const zr_fe *a);
int  zr_fe_cmp(const zr_fe *a, const zr_fe *b);          /* limb-lexicographic Ord */
int  zr_fe_eq(const zr_fe *a, const zr_fe *b);           /* ct_eq on to_bytes() */
void zr_fe_from_bytes(zr_fe *r, const uint8_t b[32]);
void zr_fe_to_bytes(uint8_t b[32], const zr_fe *a);
int  zr_fe_two_pow_k(zr_fe *r, uint64_t k);              /* 0 if k >= 253 (reference panics) */

/* ---- single-element API (scalar, mod L) ---- */
void zr_sc_add(zr_sc *r, const zr_sc *a, const zr_sc *b);
void zr_sc_sub(zr_sc *r, const zr_sc *a, const zr_sc *b);
void zr_sc_neg(zr_sc *r, const zr_sc *a);
void zr_sc_mul(zr_sc *r, const zr_sc *a, const zr_sc *b);
void zr_sc_square(zr_sc *r, const zr_sc *a);
void zr_sc_montgomery_mul(zr_sc *r, const zr_sc *a, const zr_sc *b);
void zr_sc_to_montgomery(zr_sc *r, const zr_sc *a);
void zr_sc_from_montgomery(zr_sc *r, const zr_sc *a);
void zr_sc_half(zr_sc *r, const zr_sc *a);
void zr_sc_half_without_mod(zr_sc *r, const zr_sc *a);
void zr_sc_pow(zr_sc *r, const zr_sc *a, const zr_sc *e);
void zr_sc_shr(zr_sc *r, const zr_sc *a, unsigned k);
int  zr_sc_is_even(const zr_sc *a);
int  zr_sc_eq(const zr_sc *a, const zr_sc *b);
int  zr_sc_from_bytes(zr_sc *r, const uint8_t b[32]);    /* 0 if value > L-1 (reference panics) */
void zr_sc_to_bytes(uint8_t b[32], const zr_sc *a);
int  zr_sc_two_pow_k(zr_sc *r, uint64_t k);              /* 0 if k >= 250 (reference panics) */
void zr_sc_into_bits(uint8_t bits[256], const zr_sc *a);
void zr_sc_compute_naf(int8_t naf[256], const zr_sc *a);
void zr_sc_compute_window_naf(int8_t naf[256], const zr_sc *a, unsigned width);

/* ---- Edwards points ---- */
void zr_ed_identity(zr_pt *r);
void zr_ed_neg(zr_pt *r, const zr_pt *p);
void zr_ed_add(zr_pt *r, const zr_pt *p, const zr_pt *q);
void zr_ed_sub(zr_pt *r, const zr_pt *p, const zr_pt *q);
void zr_ed_double(zr_pt *r, const zr_pt *p);
void zr_ed_scalar_mul(zr_pt *r, const zr_pt *p, const zr_sc *k);  /* double_and_add */
void zr_ed_ltr_bin_mul(zr_pt *r, const zr_pt *p, const zr_sc *k);
void zr_ed_binary_naf_mul(zr_pt *r, const zr_pt *p, const zr_sc *k);
int  zr_ed_mul_by_pow_2(zr_pt *r, const zr_pt *p, uint64_t k);    /* 0 if k >= 250 */
void zr_ed_mul_by_cofactor(zr_pt *r, const zr_pt *p);
int  zr_ed_to_affine(zr_fe *x, zr_fe *y, const zr_pt *p);         /* 0 if Z==0 (reference panics) */
int  zr_ed_eq(const zr_pt *p, const zr_pt *q);                    /* -1 if a Z is 0 */
int  zr_ed_is_valid(const zr_pt *p);
int  zr_ed_compress(uint8_t out[32], const zr_pt *p);             /* 0 where the reference panics */
int  zr_ed_decompress(zr_pt *r, const uint8_t in[32]);            /* 1 = Some, 0 = None */
int  zr_ed_new_from_y_coord(zr_pt *r, const zr_fe *y, int sign);

/* ---- Ristretto ---- */
int  zr_ris_decompress(zr_pt *r, const uint8_t in[32]);           /* 1 = Some, 0 = None */
void zr_ris_compress(uint8_t out[32], const zr_pt *p);
int  zr_ris_eq(const zr_pt *p, const zr_pt *q);
void zr_ris_elligator(zr_pt *r, const zr_fe *r0);

/* ---- batch wrappers (contiguous AoS arrays, n elements) ---- */
void zr_fe_add_batch(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void zr_fe_sub_batch(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void zr_fe_mul_batch(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void zr_fe_neg_batch(const uint64_t *a, uint64_t *out, size_t n);
void zr_fe_square_batch(const uint64_t *a, uint64_t *out, size_t n);
void zr_fe_invert_batch(const uint64_t *a, uint64_t *out, uint8_t *ok, size_t n);
void zr_fe_div_batch(const uint64_t *a, const uint64_t *b, uint64_t *out, uint8_t *ok, size_t n);
void zr_fe_pow_batch(const uint64_t *a, const uint64_t *e, uint64_t *out, size_t n);
void zr_fe_half_batch(const uint64_t *a, uint64_t *out, size_t n);
void zr_fe_legendre_symbol_batch(const uint64_t *a, uint8_t *out, size_t n);
void zr_fe_is_positive_batch(const uint64_t *a, uint8_t *out, size_t n);
void zr_fe_mod_sqrt_batch(const uint64_t *a, int sign, uint64_t *out, uint8_t *ok, size_t n);
void zr_fe_from_bytes_batch(const uint8_t *in, uint64_t *out, size_t n);
void zr_fe_to_bytes_batch(const uint64_t *in, uint8_t *out, size_t n);
void zr_fe_sqrt_ratio_i_batch(const uint64_t *u, const uint64_t *v, uint64_t *out, uint8_t *was_square, size_t n);
void zr_sc_add_batch(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void zr_sc_sub_batch(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void zr_sc_mul_batch(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void zr_sc_neg_batch(const uint64_t *a, uint64_t *out, size_t n);
void zr_sc_square_batch(const uint64_t *a, uint64_t *out, size_t n);
void zr_sc_half_batch(const uint64_t *a, uint64_t *out, size_t n);
void zr_sc_pow_batch(const uint64_t *a, const uint64_t *e, uint64_t *out, size_t n);
void zr_sc_shr_batch(const uint64_t *a, unsigned shift, uint64_t *out, size_t n);
void zr_sc_into_bits_batch(const uint64_t *a, uint8_t *bits, size_t n);
void zr_sc_compute_naf_batch(const uint64_t *a, unsigned width, int8_t *naf, size_t n);
void zr_fe_inv_sqrt_batch(const uint64_t *a, uint64_t *out, uint8_t *was_square, size_t n);
void zr_sc_from_bytes_batch(const uint8_t *in, uint64_t *out, uint8_t *ok, size_t n);
void zr_sc_to_bytes_batch(const uint64_t *in, uint8_t *out, size_t n);
void zr_ed_add_batch(const uint64_t *p, const uint64_t *q, uint64_t *out, size_t n);
void zr_ed_sub_batch(const uint64_t *p, const uint64_t *q, uint64_t *out, size_t n);
void zr_ed_double_batch(const uint64_t *p, uint64_t *out, size_t n);
void zr_ed_neg_batch(const uint64_t *p, uint64_t *out, size_t n);
void zr_ed_scalar_mul_batch(const uint64_t *p, const uint64_t *k, uint64_t *out, size_t n);
void zr_ed_scalar_mul_mode_batch(const uint64_t *p, const uint64_t *k, uint64_t *out, size_t n, int mode);
void zr_ed_mul_by_pow_2_batch(const uint64_t *p, uint64_t kexp, uint64_t *out, size_t n);
void zr_ed_to_affine_batch(const uint64_t *p, uint64_t *xy, uint8_t *ok, size_t n);
void zr_ed_eq_batch(const uint64_t *p, const uint64_t *q, uint8_t *eq, size_t n);
void zr_ed_compress_batch(const uint64_t *p, uint8_t *out, uint8_t *ok, size_t n);
void zr_ed_decompress_batch(const uint8_t *in, uint64_t *out, uint8_t *ok, size_t n);
void zr_ris_compress_batch(const uint64_t *p, uint8_t *out, size_t n);
void zr_ris_decompress_batch(const uint8_t *in, uint64_t *out, uint8_t *ok, size_t n);
void zr_ris_eq_batch(const uint64_t *p, const uint64_t *q, uint8_t *eq, size_t n);
void zr_ris_roundtrip_mul_batch(const uint8_t *in, const uint64_t *k, uint8_t *out, uint8_t *ok, size_t n);
void zr_proj_add(zr_fe r[3], const zr_fe p[3], const zr_fe q[3]);
void zr_proj_double(zr_fe r[3], const zr_fe p[3]);
void zr_proj_to_extended(zr_pt *r, const zr_fe p[3]);
int  zr_ris_is_valid(const zr_pt *p);
void zr_ris_from_uniform_bytes(zr_pt *r, const uint8_t b[64]);
void zr_ed_coset4(zr_pt r[4], const zr_pt *p);
void zr_proj_neg(zr_fe r[3], const zr_fe p[3]);
void zr_proj_sub(zr_fe r[3], const zr_fe p[3], const zr_fe q[3]);
int  zr_proj_eq(const zr_fe p[3], const zr_fe q[3]);
int  zr_proj_is_valid(const zr_fe p[3]);
void zr_proj_scalar_mul(zr_fe r[3], const zr_fe p[3], const zr_sc *k);
void zr_ed_coset4_batch(const uint64_t *p, uint64_t *out4, size_t n);
void zr_proj_neg_batch(const uint64_t *p, uint64_t *out, size_t n);
void zr_proj_sub_batch(const uint64_t *p, const uint64_t *q, uint64_t *out, size_t n);
void zr_proj_eq_batch(const uint64_t *p, const uint64_t *q, uint8_t *eq, uint8_t *ok, size_t n);
void zr_proj_is_valid_batch(const uint64_t *p, uint8_t *valid, size_t n);
void zr_proj_scalar_mul_batch(const uint64_t *p, const uint64_t *k, uint64_t *out, size_t n);
void zr_proj_add_batch(const uint64_t *p, const uint64_t *q, uint64_t *out, size_t n);
void zr_proj_double_batch(const uint64_t *p, uint64_t *out, size_t n);
void zr_proj_to_extended_batch(const uint64_t *p, uint64_t *out, size_t n);
void zr_ed_is_valid_batch(const uint64_t *p, uint8_t *valid, size_t n);
void zr_ris_is_valid_batch(const uint64_t *p, uint8_t *valid, size_t n);
void zr_ris_elligator_batch(const uint64_t *r0, uint64_t *out, size_t n);
void zr_ris_from_uniform_bytes_batch(const uint8_t *in, uint64_t *out, size_t n);
/* sum_i k_i * P_i by the reference's own ops (scalar_mul then add, in index order) */
void zr_msm_naive(const uint64_t *p, const uint64_t *k, size_t n, uint64_t *out_point);

/* constants exported for tests */
extern const zr_fe ZR_FIELD_L, ZR_RR_FIELD, ZR_EDWARDS_A, ZR_EDWARDS_D, ZR_SQRT_MINUS_ONE,
                   ZR_INV_SQRT_A_MINUS_D, ZR_SQRT_AD_MINUS_ONE, ZR_POS_RANGE, ZR_INVERSE_MOD_TWO,
                   ZR_MINUS_ONE_HALF;
extern const zr_sc ZR_L, ZR_RR, ZR_SCALAR_INVERSE_MOD_TWO;
extern const zr_pt ZR_BASEPOINT;
extern const uint64_t ZR_LFACTOR, ZR_LFACTOR_FIELD;

#ifdef __cplusplus
}
#endif
#endif

/*
 * zc_ref.c -- CPU ORACLE (test infrastructure, NOT the product path).
 * See zc_ref.h for the contract.  Plain C11 + unsigned __int128.
 *
 * Every function names the reference lines it restates.  Abbreviations:
 *   F  = src/backend/u64/field.rs      S = src/backend/u64/scalar.rs
 *   K  = src/backend/u64/constants.rs  E = src/edwards.rs   R = src/ristretto.rs
 */
#include "zc_ref.h"
#include <string.h>

typedef unsigned __int128 u128;
#define MASK52 ((1ULL << 52) - 1)

/* ------------------------------------------------------------------ constants */
/* K:8-9 */
const zr_sc ZR_L = {{1129677152307299ULL, 1363544697812651ULL, 714439ULL, 0ULL, 2199023255552ULL}};
/* K:12-13 */
const zr_fe ZR_POS_RANGE = {{2587757230352886ULL, 4210131976237760ULL, 683900ULL, 0ULL, 8796093022208ULL}};
/* K:18 */
const uint64_t ZR_LFACTOR = 1331240223835829ULL;
/* K:21-27 */
const zr_sc ZR_RR = {{137682194168839ULL, 3209056245311277ULL, 1480926248458276ULL, 2533620989757837ULL, 1314911199310ULL}};
/* K:30-36 */
const zr_fe ZR_FIELD_L = {{671914833335277ULL, 3916664325105025ULL, 1367801ULL, 0ULL, 17592186044416ULL}};
/* K:39-45 */
const zr_fe ZR_RR_FIELD = {{2764609938444603ULL, 3768881411696287ULL, 1616719297148420ULL, 1087343033131391ULL, 10175238647962ULL}};
/* K:48 */
const zr_sc ZR_SCALAR_INVERSE_MOD_TWO = {{2816638389838898ULL, 2933572162591573ULL, 357219ULL, 0ULL, 1099511627776ULL}};
/* K:51 */
const zr_fe ZR_INVERSE_MOD_TWO = {{2587757230352887ULL, 4210131976237760ULL, 683900ULL, 0ULL, 8796093022208ULL}};
/* K:54 */
const zr_fe ZR_MINUS_ONE_HALF = {{2587757230352886ULL, 4210131976237760ULL, 683900ULL, 0ULL, 8796093022208ULL}};
/* K:59 */
const uint64_t ZR_LFACTOR_FIELD = 1439961107955227ULL;
/* K:75-81 */
const zr_fe ZR_EDWARDS_A = {{671914833335276ULL, 3916664325105025ULL, 1367801ULL, 0ULL, 17592186044416ULL}};
/* K:86-92 */
const zr_fe ZR_EDWARDS_D = {{3304133203739795ULL, 2446467598308289ULL, 1534112949566882ULL, 2032729967918914ULL, 2313225441931ULL}};
/* K:96-102 */
const zr_fe ZR_SQRT_MINUS_ONE = {{3075585030474777ULL, 2451921961843096ULL, 1194333869305507ULL, 2218299809671669ULL, 7376823328646ULL}};
/* K:123-129 */
const zr_fe ZR_INV_SQRT_A_MINUS_D = {{550050132044477ULL, 3953042081665262ULL, 2971403105229349ULL, 212915494370164ULL, 1172367057772ULL}};
/* K:132-138 */
const zr_fe ZR_SQRT_AD_MINUS_ONE = {{3601277882726560ULL, 1817821323014817ULL, 1726005090908779ULL, 2111284621343800ULL, 648674458156ULL}};
/* K:188-211 */
const zr_pt ZR_BASEPOINT = {
    {{276718085098056ULL, 1646536057461434ULL, 2704687245600312ULL, 2630386667454967ULL, 13476148227069ULL}},
    {{1303868825475266ULL, 3250718520537114ULL, 2702159777242978ULL, 2702159776422297ULL, 10555311626649ULL}},
    {{1ULL, 0ULL, 0ULL, 0ULL, 0ULL}},
    {{3634527586288175ULL, 2006028620404053ULL, 3424252198034825ULL, 2478951925947079ULL, 4567251727358ULL}}};

static const zr_fe FE_ZERO = {{0, 0, 0, 0, 0}};
static const zr_fe FE_ONE = {{1, 0, 0, 0, 0}};
/* F:523-531 */
static const zr_fe FE_MINUS_ONE = {{671914833335276ULL, 3916664325105025ULL, 1367801ULL, 0ULL, 17592186044416ULL}};
/* S:341-343 */
static const zr_sc SC_MINUS_ONE = {{1129677152307298ULL, 1363544697812651ULL, 714439ULL, 0ULL, 2199023255552ULL}};

/* ------------------------------------------------------------------ shared limb machinery
 * The field (F) and scalar (S) files contain the same code with different
 * modulus constants; the helpers below take the modulus as a parameter.      */

static inline u128 m(uint64_t x, uint64_t y) { return (u128)x * (u128)y; } /* F:506-508, S:325-327 */

/* F:217-240 / S:210-237 */
static inline void limbs_sub(zr_fe *r, const zr_fe *a, const zr_fe *b, const zr_fe *mod)
{
    uint64_t sub = 0;
    uint64_t d[5];
    for (int i = 0; i < 5; i++) {
        sub = a->l[i] - (b->l[i] + (sub >> 63));            /* wrapping_sub */
        d[i] = sub & MASK52;
    }
    uint64_t underflow_mask = ((sub >> 63) ^ 1) - 1;          /* wrapping_sub(1) */
    uint64_t carry = 0;
    for (int i = 0; i < 5; i++) {
        carry = (carry >> 52) + d[i] + (mod->l[i] & underflow_mask);
        r->l[i] = carry & MASK52;
    }
}

/* F:191-207 / S:184-200 */
static inline void limbs_add(zr_fe *r, const zr_fe *a, const zr_fe *b, const zr_fe *mod)
{
    zr_fe sum;
    uint64_t carry = 0;
    for (int i = 0; i < 5; i++) {
        carry = a->l[i] + b->l[i] + (carry >> 52);
        sum.l[i] = carry & MASK52;
    }
    limbs_sub(r, &sum, mod, mod);
}

/* F:741-757 / S:580-594 */
static inline void mul_internal(u128 z[9], const zr_fe *a, const zr_fe *b)
{
    const uint64_t *x = a->l, *y = b->l;
    z[0] = m(x[0], y[0]);
    z[1] = m(x[0], y[1]) + m(x[1], y[0]);
    z[2] = m(x[0], y[2]) + m(x[1], y[1]) + m(x[2], y[0]);
    z[3] = m(x[0], y[3]) + m(x[1], y[2]) + m(x[2], y[1]) + m(x[3], y[0]);
    z[4] = m(x[0], y[4]) + m(x[1], y[3]) + m(x[2], y[2]) + m(x[3], y[1]) + m(x[4], y[0]);
    z[5] = m(x[1], y[4]) + m(x[2], y[3]) + m(x[3], y[2]) + m(x[4], y[1]);
    z[6] = m(x[2], y[4]) + m(x[3], y[3]) + m(x[4], y[2]);
    z[7] = m(x[3], y[4]) + m(x[4], y[3]);
    z[8] = m(x[4], y[4]);
}

/* F:763-777 / S:600-614 */
static inline void square_internal(u128 z[9], const zr_fe *a)
{
    const uint64_t *x = a->l;
    uint64_t d0 = x[0] * 2, d1 = x[1] * 2, d2 = x[2] * 2, d3 = x[3] * 2;
    z[0] = m(x[0], x[0]);
    z[1] = m(d0, x[1]);
    z[2] = m(d0, x[2]) + m(x[1], x[1]);
    z[3] = m(d0, x[3]) + m(d1, x[2]);
    z[4] = m(d0, x[4]) + m(d1, x[3]) + m(x[2], x[2]);
    z[5] = m(d1, x[4]) + m(d2, x[3]);
    z[6] = m(d2, x[4]) + m(x[3], x[3]);
    z[7] = m(d3, x[4]);
    z[8] = m(x[4], x[4]);
}

/* F:782-785 / S:619-622 */
static inline u128 adjustment_fact(u128 sum, uint64_t lfactor, uint64_t l0, uint64_t *p)
{
    *p = ((uint64_t)sum * lfactor) & MASK52;                 /* wrapping_mul */
    return (sum + m(*p, l0)) >> 52;
}
/* F:788-791 / S:625-628 */
static inline u128 montg_red_res(u128 sum, uint64_t *w)
{
    *w = (uint64_t)sum & MASK52;
    return sum >> 52;
}

/* F:780-813 (FIELD_L[3] == 0 products skipped, as the reference does) */
static inline void fe_montgomery_reduce(zr_fe *r, const u128 z[9])
{
    const uint64_t *l = ZR_FIELD_L.l;
    uint64_t n0, n1, n2, n3, n4;
    zr_fe t;
    u128 c;
    c = adjustment_fact(z[0], ZR_LFACTOR_FIELD, l[0], &n0);
    c = adjustment_fact(c + z[1] + m(n0, l[1]), ZR_LFACTOR_FIELD, l[0], &n1);
    c = adjustment_fact(c + z[2] + m(n0, l[2]) + m(n1, l[1]), ZR_LFACTOR_FIELD, l[0], &n2);
    c = adjustment_fact(c + z[3] + m(n1, l[2]) + m(n2, l[1]), ZR_LFACTOR_FIELD, l[0], &n3);
    c = adjustment_fact(c + z[4] + m(n0, l[4]) + m(n2, l[2]) + m(n3, l[1]), ZR_LFACTOR_FIELD, l[0], &n4);
    c = montg_red_res(c + z[5] + m(n1, l[4]) + m(n3, l[2]) + m(n4, l[1]), &t.l[0]);
    c = montg_red_res(c + z[6] + m(n2, l[4]) + m(n4, l[2]), &t.l[1]);
    c = montg_red_res(c + z[7] + m(n3, l[4]), &t.l[2]);
    c = montg_red_res(c + z[8] + m(n4, l[4]), &t.l[3]);
    t.l[4] = (uint64_t)c;
    limbs_sub(r, &t, &ZR_FIELD_L, &ZR_FIELD_L);
}

/* S:617-652 (includes the l[3] products although L[3] == 0, as the reference does) */
static inline void sc_montgomery_reduce(zr_sc *r, const u128 z[9])
{
    const uint64_t *l = ZR_L.l;
    uint64_t n0, n1, n2, n3, n4;
    zr_sc t;
    u128 c;
    c = adjustment_fact(z[0], ZR_LFACTOR, l[0], &n0);
    c = adjustment_fact(c + z[1] + m(n0, l[1]), ZR_LFACTOR, l[0], &n1);
    c = adjustment_fact(c + z[2] + m(n0, l[2]) + m(n1, l[1]), ZR_LFACTOR, l[0], &n2);
    c = adjustment_fact(c + z[3] + m(n0, l[3]) + m(n1, l[2]) + m(n2, l[1]), ZR_LFACTOR, l[0], &n3);
    c = adjustment_fact(c + z[4] + m(n0, l[4]) + m(n1, l[3]) + m(n2, l[2]) + m(n3, l[1]), ZR_LFACTOR, l[0], &n4);
    c = montg_red_res(c + z[5] + m(n1, l[4]) + m(n2, l[3]) + m(n3, l[2]) + m(n4, l[1]), &t.l[0]);
    c = montg_red_res(c + z[6] + m(n2, l[4]) + m(n3, l[3]) + m(n4, l[2]), &t.l[1]);
    c = montg_red_res(c + z[7] + m(n3, l[4]) + m(n4, l[3]), &t.l[2]);
    c = montg_red_res(c + z[8] + m(n4, l[4]), &t.l[3]);
    t.l[4] = (uint64_t)c;
    limbs_sub(r, &t, &ZR_L, &ZR_L);
}

/* F:676-688 / S:562-574 */
static inline void limbs_half_without_mod(zr_fe *r, const zr_fe *a)
{
    uint64_t carry = 0;
    zr_fe res = *a;
    for (int i = 4; i >= 0; i--) {
        res.l[i] = res.l[i] | carry;
        carry = (res.l[i] & 1) << 52;
        res.l[i] >>= 1;
    }
    *r = res;
}

/* F:65-76 / S:54-65 */
static inline int limbs_cmp(const zr_fe *a, const zr_fe *b)
{
    for (int i = 4; i >= 0; i--) {
        if (a->l[i] > b->l[i]) return 1;
        if (a->l[i] < b->l[i]) return -1;
    }
    return 0;
}

/* F:591-631 / S:477-516 */
static inline void limbs_to_bytes(uint8_t res[32], const zr_fe *a)
{
    const uint64_t *s = a->l;
    res[0] = (uint8_t)(s[0] >> 0);
    res[1] = (uint8_t)(s[0] >> 8);
    res[2] = (uint8_t)(s[0] >> 16);
    res[3] = (uint8_t)(s[0] >> 24);
    res[4] = (uint8_t)(s[0] >> 32);
    res[5] = (uint8_t)(s[0] >> 40);
    res[6] = (uint8_t)((s[0] >> 48) | (s[1] << 4));
    res[7] = (uint8_t)(s[1] >> 4);
    res[8] = (uint8_t)(s[1] >> 12);
    res[9] = (uint8_t)(s[1] >> 20);
    res[10] = (uint8_t)(s[1] >> 28);
    res[11] = (uint8_t)(s[1] >> 36);
    res[12] = (uint8_t)(s[1] >> 44);
    res[13] = (uint8_t)(s[2] >> 0);
    res[14] = (uint8_t)(s[2] >> 8);
    res[15] = (uint8_t)(s[2] >> 16);
    res[16] = (uint8_t)(s[2] >> 24);
    res[17] = (uint8_t)(s[2] >> 32);
    res[18] = (uint8_t)(s[2] >> 40);
    res[19] = (uint8_t)((s[2] >> 48) | (s[3] << 4));
    res[20] = (uint8_t)(s[3] >> 4);
    res[21] = (uint8_t)(s[3] >> 12);
    res[22] = (uint8_t)(s[3] >> 20);
    res[23] = (uint8_t)(s[3] >> 28);
    res[24] = (uint8_t)(s[3] >> 36);
    res[25] = (uint8_t)(s[3] >> 44);
    res[26] = (uint8_t)(s[4] >> 0);
    res[27] = (uint8_t)(s[4] >> 8);
    res[28] = (uint8_t)(s[4] >> 16);
    res[29] = (uint8_t)(s[4] >> 24);
    res[30] = (uint8_t)(s[4] >> 32);
    res[31] = (uint8_t)(s[4] >> 40);
}

/* src/field.rs:99-106, src/scalar.rs:78-91: equality on canonical bytes */
static inline int limbs_eq(const zr_fe *a, const zr_fe *b)
{
    uint8_t x[32], y[32];
    limbs_to_bytes(x, a);
    limbs_to_bytes(y, b);
    return memcmp(x, y, 32) == 0;
}

/* F:640-666, F:715-738, S:525-552 (range assertion handled by the callers) */
static inline void limbs_two_pow_k(zr_fe *r, uint64_t e)
{
    *r = FE_ZERO;
    if (e <= 51) r->l[0] = 1ULL << e;
    else if (e <= 103) r->l[1] = 1ULL << (e - 52);
    else if (e <= 155) r->l[2] = 1ULL << (e - 104);
    else if (e <= 207) r->l[3] = 1ULL << (e - 156);
    else r->l[4] = 1ULL << (e - 208);
}

/* ------------------------------------------------------------------ FieldElement */
void zr_fe_add(zr_fe *r, const zr_fe *a, const zr_fe *b) { limbs_add(r, a, b, &ZR_FIELD_L); } /* F:191-207 */
void zr_fe_sub(zr_fe *r, const zr_fe *a, const zr_fe *b) { limbs_sub(r, a, b, &ZR_FIELD_L); } /* F:217-240 */
void zr_fe_neg(zr_fe *r, const zr_fe *a) { limbs_sub(r, &FE_ZERO, a, &ZR_FIELD_L); }           /* F:170-178 */

/* F:818-820 */
void zr_fe_montgomery_mul(zr_fe *r, const zr_fe *a, const zr_fe *b)
{
    u128 z[9];
    mul_internal(z, a, b);
    fe_montgomery_reduce(r, z);
}
/* F:250-262 */
void zr_fe_mul(zr_fe *r, const zr_fe *a, const zr_fe *b)
{
    zr_fe prod;
    zr_fe_montgomery_mul(&prod, a, b);
    zr_fe_montgomery_mul(r, &prod, &ZR_RR_FIELD);
}
/* F:302-315 */
void zr_fe_square(zr_fe *r, const zr_fe *a)
{
    u128 z[9];
    zr_fe aa;
    square_internal(z, a);
    fe_montgomery_reduce(&aa, z);
    zr_fe_montgomery_mul(r, &aa, &ZR_RR_FIELD);
}
/* F:824-826 */
void zr_fe_to_montgomery(zr_fe *r, const zr_fe *a) { zr_fe_montgomery_mul(r, a, &ZR_RR_FIELD); }
/* F:830-836 */
void zr_fe_from_montgomery(zr_fe *r, const zr_fe *a)
{
    u128 z[9] = {0};
    for (int i = 0; i < 5; i++) z[i] = a->l[i];
    fe_montgomery_reduce(r, z);
}
/* F:317-323 */
void zr_fe_half(zr_fe *r, const zr_fe *a) { zr_fe_mul(r, a, &ZR_INVERSE_MOD_TWO); }
void zr_fe_half_without_mod(zr_fe *r, const zr_fe *a) { limbs_half_without_mod(r, a); }
int zr_fe_is_even(const zr_fe *a) { return (a->l[0] & 1) == 0; }  /* F:534-539 */
int zr_fe_cmp(const zr_fe *a, const zr_fe *b) { return limbs_cmp(a, b); }
int zr_fe_eq(const zr_fe *a, const zr_fe *b) { return limbs_eq(a, b); }
void zr_fe_to_bytes(uint8_t b[32], const zr_fe *a) { limbs_to_bytes(b, a); }

/* F:552-557 */
int zr_fe_is_positive(const zr_fe *a)
{
    return limbs_cmp(a, &FE_ZERO) >= 0 && limbs_cmp(a, &ZR_POS_RANGE) <= 0;
}

static inline uint64_t load8(const uint8_t *in)
{
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v |= (uint64_t)in[i] << (8 * i);
    return v;
}
/* F:563-587 */
void zr_fe_from_bytes(zr_fe *r, const uint8_t b[32])
{
    r->l[0] = load8(b + 0) & MASK52;
    r->l[1] = (load8(b + 6) >> 4) & MASK52;
    r->l[2] = (load8(b + 12) >> 8) & MASK52;
    r->l[3] = (load8(b + 19) >> 4) & MASK52;
    r->l[4] = (load8(b + 24) >> 16) & MASK52;
}

/* F:640-666 */
int zr_fe_two_pow_k(zr_fe *r, uint64_t k)
{
    if (!(k < 253)) { *r = FE_ZERO; return 0; }
    limbs_two_pow_k(r, k);
    return 1;
}

/* F:325-355 */
void zr_fe_pow(zr_fe *r, const zr_fe *a, const zr_fe *e)
{
    zr_fe base = *a, res = FE_ONE, expon = *e;
    while (limbs_cmp(&expon, &FE_ZERO) > 0) {
        if (zr_fe_is_even(&expon)) {
            limbs_half_without_mod(&expon, &expon);
            zr_fe_mul(&base, &base, &base);
        } else {
            zr_fe_sub(&expon, &expon, &FE_ONE);
            zr_fe_mul(&res, &res, &base);
            limbs_half_without_mod(&expon, &expon);
            zr_fe_mul(&base, &base, &base);
        }
    }
    *r = res;
}

/* F:703-706 */
int zr_fe_legendre_symbol(const zr_fe *a)
{
    zr_fe res;
    zr_fe_pow(&res, a, &ZR_MINUS_ONE_HALF);
    return limbs_eq(&res, &FE_MINUS_ONE) ^ 1;
}

/* F:863-915 */
static int inverse_phase1(zr_fe *out, uint64_t *kout, const zr_fe *a)
{
    if (limbs_eq(a, &FE_ZERO)) return 0;                       /* F:864 assert */
    const zr_fe p = ZR_FIELD_L;
    zr_fe u = p, v = *a, r = FE_ZERO, s = FE_ONE;
    const zr_fe two = {{2, 0, 0, 0, 0}};
    uint64_t k = 0;
    while (limbs_cmp(&v, &FE_ZERO) > 0) {
        int ue = zr_fe_is_even(&u), ve = zr_fe_is_even(&v);
        if (ue) {
            limbs_half_without_mod(&u, &u);
            zr_fe_mul(&s, &s, &two);
        } else if (ve) {
            limbs_half_without_mod(&v, &v);
            zr_fe_mul(&r, &r, &two);
        } else if (limbs_cmp(&u, &v) > 0) {
            zr_fe_sub(&u, &u, &v);
            limbs_half_without_mod(&u, &u);
            zr_fe_add(&r, &r, &s);
            zr_fe_mul(&s, &s, &two);
        } else { /* v >= u */
            zr_fe_sub(&v, &v, &u);
            limbs_half_without_mod(&v, &v);
            zr_fe_add(&s, &r, &s);
            zr_fe_mul(&r, &r, &two);
        }
        k += 1;
    }
    if (limbs_cmp(&r, &p) > 0) zr_fe_sub(&r, &r, &p);
    zr_fe_sub(out, &p, &r);
    *kout = k;
    return 1;
}

/* F:854-925 */
int zr_fe_inverse(zr_fe *r, const zr_fe *a)
{
    zr_fe x, fact;
    uint64_t z;
    if (!inverse_phase1(&x, &z, a)) { *r = FE_ZERO; return 0; }
    if (z > 260) {
        zr_fe_montgomery_mul(&x, &x, &FE_ONE);
        z -= 260;
    }
    limbs_two_pow_k(&fact, 260 - z);                          /* inner_two_pow_k, F:715-738 */
    zr_fe_montgomery_mul(r, &x, &fact);
    return 1;
}

/* F:277-300 */
int zr_fe_div(zr_fe *r, const zr_fe *a, const zr_fe *b)
{
    zr_fe inv;
    if (!zr_fe_inverse(&inv, b)) { *r = FE_ZERO; return 0; }
    zr_fe_mul(r, a, &inv);
    return 1;
}

/* F:357-441 (Tonelli-Shanks, non-residue 6; sign=1 -> p - x, sign=0 -> x) */
int zr_fe_mod_sqrt(zr_fe *r, const zr_fe *a, int sign)
{
    const zr_fe zero = FE_ZERO, one = FE_ONE, two = {{2, 0, 0, 0, 0}}, six = {{6, 0, 0, 0, 0}};
    if (limbs_eq(a, &zero)) { *r = zero; return 1; }
    if (zr_fe_legendre_symbol(a) == 0) { *r = zero; return 0; }

    zr_fe q = FE_MINUS_ONE, s = zero;
    while (zr_fe_is_even(&q)) {
        zr_fe_add(&s, &s, &one);
        limbs_half_without_mod(&q, &q);
    }
    zr_fe c, x, t, mm, tmp;
    zr_fe_pow(&c, &six, &q);
    zr_fe_add(&tmp, &q, &one);
    limbs_half_without_mod(&tmp, &tmp);
    zr_fe_pow(&x, a, &tmp);
    zr_fe_pow(&t, a, &q);
    mm = s;
    while (!limbs_eq(&t, &one)) {
        zr_fe i = zero, e = two, b, te, ex;
        while (limbs_cmp(&i, &mm) < 0) {
            zr_fe_add(&i, &i, &one);
            zr_fe_pow(&te, &t, &e);
            if (limbs_eq(&te, &one)) break;
            zr_fe_mul(&e, &e, &two);
        }
        zr_fe_sub(&ex, &mm, &i);
        zr_fe_sub(&ex, &ex, &one);
        zr_fe_pow(&ex, &two, &ex);
        zr_fe_pow(&b, &c, &ex);
        zr_fe_mul(&x, &x, &b);
        zr_fe bsq;
        zr_fe_square(&bsq, &b);
        zr_fe_mul(&t, &t, &bsq);
        zr_fe_square(&c, &b);
        mm = i;
    }
    if (sign) zr_fe_sub(r, &ZR_FIELD_L, &x); else *r = x;    /* conditional_select(&x, &(p - x), sign) */
    return 1;
}

/* subtle::ConditionallyNegatable over F:170-178 */
static inline void fe_cond_negate(zr_fe *x, int choice)
{
    zr_fe n;
    zr_fe_neg(&n, x);
    if (choice) *x = n;
}

/* F:462-503 */
int zr_fe_sqrt_ratio_i(zr_fe *r, const zr_fe *u, const zr_fe *v)
{
    if (limbs_eq(u, &FE_ZERO)) { *r = FE_ZERO; return 1; }
    if (limbs_eq(v, &FE_ZERO)) { *r = FE_ZERO; return 0; }
    zr_fe q, res;
    zr_fe_div(&q, u, v);
    if (zr_fe_legendre_symbol(&q) != 1) {
        zr_fe q2, iq;
        zr_fe_div(&q2, u, v);
        zr_fe_mul(&iq, &ZR_SQRT_MINUS_ONE, &q2);
        zr_fe_mod_sqrt(&res, &iq, 1);
        fe_cond_negate(&res, !zr_fe_is_positive(&res));
        *r = res;
        return 0;
    } else {
        zr_fe q2;
        zr_fe_div(&q2, u, v);
        zr_fe_mod_sqrt(&res, &q2, 1);
        fe_cond_negate(&res, !zr_fe_is_positive(&res));
        *r = res;
        return 1;
    }
}
/* F:443-460 */
int zr_fe_inv_sqrt(zr_fe *r, const zr_fe *a) { return zr_fe_sqrt_ratio_i(r, &FE_ONE, a); }

/* ------------------------------------------------------------------ Scalar */
void zr_sc_add(zr_sc *r, const zr_sc *a, const zr_sc *b) { limbs_add(r, a, b, &ZR_L); }   /* S:184-200 */
void zr_sc_sub(zr_sc *r, const zr_sc *a, const zr_sc *b) { limbs_sub(r, a, b, &ZR_L); }   /* S:210-237 */
void zr_sc_neg(zr_sc *r, const zr_sc *a) { limbs_sub(r, &FE_ZERO, a, &ZR_L); }             /* S:139-146 */
/* S:656-658 */
void zr_sc_montgomery_mul(zr_sc *r, const zr_sc *a, const zr_sc *b)
{
    u128 z[9];
    mul_internal(z, a, b);
    sc_montgomery_reduce(r, z);
}
/* S:247-258 */
void zr_sc_mul(zr_sc *r, const zr_sc *a, const zr_sc *b)
{
    zr_sc ab;
    zr_sc_montgomery_mul(&ab, a, b);
    zr_sc_montgomery_mul(r, &ab, &ZR_RR);
}
/* S:272-283 */
void zr_sc_square(zr_sc *r, const zr_sc *a)
{
    u128 z[9];
    zr_sc aa;
    square_internal(z, a);
    sc_montgomery_reduce(&aa, z);
    zr_sc_montgomery_mul(r, &aa, &ZR_RR);
}
void zr_sc_to_montgomery(zr_sc *r, const zr_sc *a) { zr_sc_montgomery_mul(r, a, &ZR_RR); } /* S:662-664 */
/* S:668-674 */
void zr_sc_from_montgomery(zr_sc *r, const zr_sc *a)
{
    u128 z[9] = {0};
    for (int i = 0; i < 5; i++) z[i] = a->l[i];
    sc_montgomery_reduce(r, z);
}
void zr_sc_half(zr_sc *r, const zr_sc *a) { zr_sc_mul(r, a, &ZR_SCALAR_INVERSE_MOD_TWO); } /* S:285-291 */
void zr_sc_half_without_mod(zr_sc *r, const zr_sc *a) { limbs_half_without_mod(r, a); }     /* S:562-574 */
int zr_sc_is_even(const zr_sc *a) { return (a->l[0] & 1) == 0; }                           /* S:346-348 */
int zr_sc_eq(const zr_sc *a, const zr_sc *b) { return limbs_eq(a, b); }
void zr_sc_to_bytes(uint8_t b[32], const zr_sc *a) { limbs_to_bytes(b, a); }               /* S:477-516 */

/* S:300-322 (note: the odd branch halves with the modular Half, S:316) */
void zr_sc_pow(zr_sc *r, const zr_sc *a, const zr_sc *e)
{
    zr_sc base = *a, res = FE_ONE, expon = *e;
    while (limbs_cmp(&expon, &FE_ZERO) > 0) {
        if (zr_sc_is_even(&expon)) {
            limbs_half_without_mod(&expon, &expon);
            zr_sc_square(&base, &base);
        } else {
            zr_sc_sub(&expon, &expon, &FE_ONE);
            zr_sc_mul(&res, &res, &base);
            zr_sc_half(&expon, &expon);
            zr_sc_square(&base, &base);
        }
    }
    *r = res;
}

/* S:165-182 */
void zr_sc_shr(zr_sc *r, const zr_sc *a, unsigned k)
{
    zr_sc res = *a;
    for (unsigned s = 0; s < k; s++) limbs_half_without_mod(&res, &res);
    *r = res;
}

/* S:445-467 */
int zr_sc_from_bytes(zr_sc *r, const uint8_t b[32])
{
    uint64_t w[4];
    for (int i = 0; i < 4; i++) w[i] = load8(b + 8 * i);
    const uint64_t top_mask = (1ULL << 48) - 1;
    r->l[0] = w[0] & MASK52;
    r->l[1] = ((w[0] >> 52) | (w[1] << 12)) & MASK52;
    r->l[2] = ((w[1] >> 40) | (w[2] << 24)) & MASK52;
    r->l[3] = ((w[2] >> 28) | (w[3] << 36)) & MASK52;
    r->l[4] = (w[3] >> 16) & top_mask;
    return limbs_cmp(r, &SC_MINUS_ONE) <= 0;                  /* S:465 assert */
}

/* S:525-552 */
int zr_sc_two_pow_k(zr_sc *r, uint64_t k)
{
    if (!(k < 250)) { *r = FE_ZERO; return 0; }
    limbs_two_pow_k(r, k);
    return 1;
}

/* S:352-366 */
void zr_sc_into_bits(uint8_t bits[256], const zr_sc *a)
{
    uint8_t bytes[32];
    limbs_to_bytes(bytes, a);
    int j = 0;
    for (int b = 0; b < 32; b++)
        for (int i = 0; i < 8; i++) bits[j++] = (bytes[b] >> i) & 1;
}

/* S:68-84 */
static void sc_from_i8(zr_sc *r, int8_t v)
{
    zr_sc t = FE_ZERO;
    if (v >= 0) { t.l[0] = (uint64_t)v; *r = t; }
    else { t.l[0] = (uint64_t)(-(int)v); zr_sc_neg(r, &t); }
}
/* S:423-425 */
static uint8_t sc_mod_2_pow_k(const zr_sc *a, unsigned k) { return (uint8_t)(a->l[0] & ((1ULL << k) - 1)); }
/* S:433-442 */
static int8_t sc_mods_2_pow_k(const zr_sc *a, unsigned w)
{
    int8_t modulus = (int8_t)sc_mod_2_pow_k(a, w);
    int8_t half = (int8_t)(1 << (w - 1));
    if (modulus >= half) return (int8_t)(modulus - (int8_t)(uint8_t)(1u << w));
    return modulus;
}
/* S:370-389 */
void zr_sc_compute_naf(int8_t naf[256], const zr_sc *a)
{
    zr_sc k = *a, t;
    int i = 0;
    memset(naf, 0, 256);
    while (limbs_cmp(&k, &FE_ONE) >= 0 && i < 256) {
        if (!zr_sc_is_even(&k)) {
            int8_t ki = (int8_t)(2 - (int8_t)sc_mod_2_pow_k(&k, 2));
            naf[i] = ki;
            sc_from_i8(&t, ki);
            zr_sc_sub(&k, &k, &t);
        } else naf[i] = 0;
        limbs_half_without_mod(&k, &k);
        i++;
    }
}
/* S:396-415 */
void zr_sc_compute_window_naf(int8_t naf[256], const zr_sc *a, unsigned width)
{
    zr_sc k = *a, t;
    int i = 0;
    memset(naf, 0, 256);
    while (limbs_cmp(&k, &FE_ONE) >= 0 && i < 256) {
        if (!zr_sc_is_even(&k)) {
            int8_t ki = sc_mods_2_pow_k(&k, width);
            naf[i] = ki;
            sc_from_i8(&t, ki);
            zr_sc_sub(&k, &k, &t);
        } else naf[i] = 0;
        limbs_half_without_mod(&k, &k);
        i++;
    }
}

/* ------------------------------------------------------------------ EdwardsPoint */
/* E:381-391 */
void zr_ed_identity(zr_pt *r) { r->X = FE_ZERO; r->Y = FE_ONE; r->Z = FE_ONE; r->T = FE_ZERO; }
/* E:440-455 */
void zr_ed_neg(zr_pt *r, const zr_pt *p)
{
    zr_pt o;
    zr_fe_neg(&o.X, &p->X);
    o.Y = p->Y;
    o.Z = p->Z;
    zr_fe_neg(&o.T, &p->T);
    *r = o;
}
/* E:465-489 (HWCD'08 sec. 3.1, a = -1) */
void zr_ed_add(zr_pt *r, const zr_pt *p, const zr_pt *q)
{
    zr_fe A, B, C, D, E, F, G, H, t0, t1;
    zr_fe_mul(&A, &p->X, &q->X);
    zr_fe_mul(&B, &p->Y, &q->Y);
    zr_fe_mul(&C, &ZR_EDWARDS_D, &p->T);
    zr_fe_mul(&C, &C, &q->T);
    zr_fe_mul(&D, &p->Z, &q->Z);
    zr_fe_add(&t0, &p->X, &p->Y);
    zr_fe_add(&t1, &q->X, &q->Y);
    zr_fe_mul(&E, &t0, &t1);
    zr_fe_sub(&E, &E, &A);
    zr_fe_sub(&E, &E, &B);
    zr_fe_sub(&F, &D, &C);
    zr_fe_add(&G, &D, &C);
    zr_fe_add(&H, &B, &A);
    zr_pt o;
    zr_fe_mul(&o.X, &E, &F);
    zr_fe_mul(&o.Y, &G, &H);
    zr_fe_mul(&o.Z, &F, &G);
    zr_fe_mul(&o.T, &E, &H);
    *r = o;
}
/* E:503-531 */
void zr_ed_sub(zr_pt *r, const zr_pt *p, const zr_pt *q)
{
    zr_pt n;
    zr_ed_neg(&n, q);
    zr_fe A, B, C, D, E, F, G, H, t0, t1;
    zr_fe_mul(&A, &p->X, &n.X);
    zr_fe_mul(&B, &p->Y, &n.Y);
    zr_fe_mul(&C, &ZR_EDWARDS_D, &p->T);
    zr_fe_mul(&C, &C, &n.T);
    zr_fe_mul(&D, &p->Z, &n.Z);
    zr_fe_add(&t0, &p->X, &p->Y);
    zr_fe_add(&t1, &n.X, &n.Y);
    zr_fe_mul(&E, &t0, &t1);
    zr_fe_sub(&E, &E, &A);
    zr_fe_sub(&E, &E, &B);
    zr_fe_sub(&F, &D, &C);
    zr_fe_add(&G, &D, &C);
    zr_fe_mul(&t0, &ZR_EDWARDS_A, &A);
    zr_fe_sub(&H, &B, &t0);
    zr_pt o;
    zr_fe_mul(&o.X, &E, &F);
    zr_fe_mul(&o.Y, &G, &H);
    zr_fe_mul(&o.Z, &F, &G);
    zr_fe_mul(&o.T, &E, &H);
    *r = o;
}
/* E:579-592: double == self + self */
void zr_ed_double(zr_pt *r, const zr_pt *p) { zr_ed_add(r, p, p); }

/* E:102-120 (LSB-first; `n != 0` compares canonical bytes, src/scalar.rs:78-91) */
void zr_ed_scalar_mul(zr_pt *r, const zr_pt *p, const zr_sc *k)
{
    zr_pt N = *p, Q;
    zr_sc n = *k;
    zr_ed_identity(&Q);
    while (!limbs_eq(&n, &FE_ZERO)) {
        if (!zr_sc_is_even(&n)) zr_ed_add(&Q, &Q, &N);
        zr_ed_double(&N, &N);
        limbs_half_without_mod(&n, &n);
    }
    *r = Q;
}
/* E:122-134 (reads bits 248..0 only) */
void zr_ed_ltr_bin_mul(zr_pt *r, const zr_pt *p, const zr_sc *k)
{
    uint8_t bits[256];
    zr_pt Q;
    zr_sc_into_bits(bits, k);
    zr_ed_identity(&Q);
    for (int i = 248; i >= 0; i--) {
        zr_ed_double(&Q, &Q);
        if (bits[i] == 1) zr_ed_add(&Q, &Q, p);
    }
    *r = Q;
}
/* E:136-153 */
void zr_ed_binary_naf_mul(zr_pt *r, const zr_pt *p, const zr_sc *k)
{
    int8_t naf[256];
    zr_pt Q;
    zr_sc_compute_naf(naf, k);
    zr_ed_identity(&Q);
    for (int i = 249; i >= 0; i--) {
        zr_ed_double(&Q, &Q);
        if (naf[i] == 1) zr_ed_add(&Q, &Q, p);
        else if (naf[i] == -1) zr_ed_sub(&Q, &Q, p);
    }
    *r = Q;
}
/* E:186-191 */
int zr_ed_mul_by_pow_2(zr_pt *r, const zr_pt *p, uint64_t k)
{
    zr_sc s;
    if (!zr_sc_two_pow_k(&s, k)) { zr_ed_identity(r); return 0; }
    zr_ed_scalar_mul(r, p, &s);
    return 1;
}
/* E:174-179 */
void zr_ed_mul_by_cofactor(zr_pt *r, const zr_pt *p)
{
    zr_sc eight = {{8, 0, 0, 0, 0}};
    zr_ed_scalar_mul(r, p, &eight);
}
/* E:1071-1092 */
int zr_ed_to_affine(zr_fe *x, zr_fe *y, const zr_pt *p)
{
    zr_fe zinv;
    if (!zr_fe_inverse(&zinv, &p->Z)) { *x = FE_ZERO; *y = FE_ZERO; return 0; }
    zr_fe_mul(x, &p->X, &zinv);
    zr_fe_mul(y, &p->Y, &zinv);
    return 1;
}
/* E:360-364, E:1044-1048 */
int zr_ed_eq(const zr_pt *p, const zr_pt *q)
{
    zr_fe x1, y1, x2, y2;
    if (!zr_ed_to_affine(&x1, &y1, p) || !zr_ed_to_affine(&x2, &y2, q)) return -1;
    return limbs_eq(&x1, &x2) & limbs_eq(&y1, &y2);
}
/* E:393-400, E:733-748 */
int zr_ed_is_valid(const zr_pt *p)
{
    zr_fe xs, ys, zs, l, r, t;
    zr_fe_square(&xs, &p->X);
    zr_fe_square(&ys, &p->Y);
    zr_fe_square(&zs, &p->Z);
    zr_fe_mul(&l, &ZR_EDWARDS_A, &xs);
    zr_fe_add(&l, &l, &ys);
    zr_fe_mul(&l, &l, &zs);
    zr_fe_square(&r, &zs);
    zr_fe_mul(&t, &ZR_EDWARDS_D, &xs);
    zr_fe_mul(&t, &t, &ys);
    zr_fe_add(&r, &r, &t);
    return limbs_eq(&l, &r);
}
/* E:200-204 */
static int find_xx(zr_fe *r, const zr_fe *y)
{
    zr_fe a, b, ys;
    zr_fe_square(&ys, y);
    zr_fe_sub(&a, &ys, &FE_ONE);
    zr_fe_mul(&b, &ZR_EDWARDS_D, &ys);
    zr_fe_sub(&b, &b, &ZR_EDWARDS_A);
    return zr_fe_div(r, &a, &b);
}
/* E:613-629 */
int zr_ed_compress(uint8_t out[32], const zr_pt *p)
{
    zr_fe x, y, xx, res;
    memset(out, 0, 32);
    if (!zr_ed_to_affine(&x, &y, p)) return 0;
    if (!find_xx(&xx, &y)) return 0;
    if (!zr_fe_mod_sqrt(&res, &xx, 0)) return 0;              /* .unwrap() */
    int sign = !limbs_eq(&res, &x);
    limbs_to_bytes(out, &y);
    out[31] |= (uint8_t)(sign << 7);
    return 1;
}
/* E:962-979 + E:402-417 + E:648-653 */
int zr_ed_new_from_y_coord(zr_pt *r, const zr_fe *y, int sign)
{
    zr_fe xx, x;
    zr_ed_identity(r);
    if (!find_xx(&xx, y)) return 0;          /* same expression as E:966; Div asserts on 0 */
    if (!zr_fe_mod_sqrt(&x, &xx, sign)) return 0;
    /* From<ProjectivePoint>, Z = 1: (X*Z, Y*Z, Z^2, X*Y) */
    zr_fe_mul(&r->X, &x, &FE_ONE);
    zr_fe_mul(&r->Y, y, &FE_ONE);
    zr_fe_square(&r->Z, &FE_ONE);
    zr_fe_mul(&r->T, &x, y);
    return 1;
}
/* E:313-326 (note the 0x0F mask on byte 31) */
int zr_ed_decompress(zr_pt *r, const uint8_t in[32])
{
    uint8_t yb[32];
    zr_fe y;
    int sign = in[31] >> 7;
    memcpy(yb, in, 32);
    yb[31] &= 0x0F;
    zr_fe_from_bytes(&y, yb);
    return zr_ed_new_from_y_coord(r, &y, sign);
}

/* ------------------------------------------------------------------ Ristretto */
/* R:96-154 */
int zr_ris_decompress(zr_pt *r, const uint8_t in[32])
{
    zr_fe s, ss, u1, u2, u2sq, v, I, Dx, Dy, x, y, t, tmp;
    uint8_t chk[32];
    zr_ed_identity(r);
    zr_fe_from_bytes(&s, in);
    limbs_to_bytes(chk, &s);
    int s_correct_enc = memcmp(chk, in, 32) == 0;
    int s_is_positive = zr_fe_is_positive(&s);
    if (!s_is_positive || !s_correct_enc) return 0;
    zr_fe_square(&ss, &s);
    zr_fe_sub(&u1, &FE_ONE, &ss);
    zr_fe_add(&u2, &FE_ONE, &ss);
    zr_fe_square(&u2sq, &u2);
    zr_fe_square(&tmp, &u1);
    zr_fe_mul(&tmp, &ZR_EDWARDS_D, &tmp);
    zr_fe_neg(&tmp, &tmp);
    zr_fe_sub(&v, &tmp, &u2sq);
    zr_fe_mul(&tmp, &v, &u2sq);
    int ok = zr_fe_inv_sqrt(&I, &tmp);
    if (!ok) return 0;
    zr_fe_mul(&Dx, &I, &u2);
    zr_fe_mul(&Dy, &I, &Dx);
    zr_fe_mul(&Dy, &Dy, &v);
    zr_fe_add(&tmp, &s, &s);
    zr_fe_mul(&x, &tmp, &Dx);
    fe_cond_negate(&x, !zr_fe_is_positive(&x));
    zr_fe_mul(&y, &u1, &Dy);
    zr_fe_mul(&t, &x, &y);
    if (!zr_fe_is_positive(&t) || limbs_eq(&y, &FE_ZERO)) return 0;
    r->X = x; r->Y = y; r->Z = FE_ONE; r->T = t;
    return 1;
}
/* R:398-425 */
void zr_ris_compress(uint8_t out[32], const zr_pt *p)
{
    zr_fe u1, u2, t0, t1, I, D1, D2, Zinv, x, y, D, s;
    zr_fe_add(&t0, &p->Z, &p->Y);
    zr_fe_sub(&t1, &p->Z, &p->Y);
    zr_fe_mul(&u1, &t0, &t1);
    zr_fe_mul(&u2, &p->X, &p->Y);
    zr_fe_square(&t0, &u2);
    zr_fe_mul(&t0, &u1, &t0);
    (void)zr_fe_inv_sqrt(&I, &t0);
    zr_fe_mul(&D1, &u1, &I);
    zr_fe_mul(&D2, &u2, &I);
    zr_fe_mul(&Zinv, &D1, &D2);
    zr_fe_mul(&Zinv, &Zinv, &p->T);
    zr_fe_mul(&t0, &p->T, &Zinv);
    if (!zr_fe_is_positive(&t0)) {
        zr_fe_mul(&x, &ZR_SQRT_MINUS_ONE, &p->Y);
        zr_fe_mul(&y, &ZR_SQRT_MINUS_ONE, &p->X);
        zr_fe_mul(&D, &D1, &ZR_INV_SQRT_A_MINUS_D);
    } else {
        x = p->X; y = p->Y; D = D2;
    }
    zr_fe_mul(&t0, &x, &Zinv);
    fe_cond_negate(&y, !zr_fe_is_positive(&t0));
    zr_fe_sub(&t0, &p->Z, &y);
    zr_fe_mul(&s, &t0, &D);
    fe_cond_negate(&s, !zr_fe_is_positive(&s));
    limbs_to_bytes(out, &s);
}
/* R:166-176 */
int zr_ris_eq(const zr_pt *p, const zr_pt *q)
{
    zr_fe a, b;
    zr_fe_mul(&a, &p->X, &q->Y);
    zr_fe_mul(&b, &p->Y, &q->X);
    int e1 = limbs_eq(&a, &b);
    zr_fe_mul(&a, &p->X, &q->X);
    zr_fe_mul(&b, &p->Y, &q->Y);
    int e2 = limbs_eq(&a, &b);
    return e1 | e2;
}
/* R:430-471 */
void zr_ris_elligator(zr_pt *out, const zr_fe *r0)
{
    const zr_fe d = ZR_EDWARDS_D, one = FE_ONE;
    zr_fe c, one_minus_d_sq, r, Ns, D, s, sp, Nt, ssq, W0, W1, W2, W3, t0, t1;
    zr_fe_neg(&c, &one);
    zr_fe_square(&t0, &d);
    zr_fe_sub(&one_minus_d_sq, &one, &t0);
    zr_fe_square(&t0, r0);
    zr_fe_mul(&r, &ZR_SQRT_MINUS_ONE, &t0);
    zr_fe_add(&t0, &r, &one);
    zr_fe_mul(&Ns, &t0, &one_minus_d_sq);
    zr_fe_mul(&t0, &d, &r);
    zr_fe_sub(&t0, &c, &t0);
    zr_fe_add(&t1, &r, &d);
    zr_fe_mul(&D, &t0, &t1);
    int is_sq = zr_fe_sqrt_ratio_i(&s, &Ns, &D);
    zr_fe_mul(&sp, &s, r0);
    fe_cond_negate(&sp, zr_fe_is_positive(&sp));
    if (!is_sq) { s = sp; c = r; }
    zr_fe_sub(&t0, &r, &one);
    zr_fe_mul(&t0, &c, &t0);
    zr_fe_sub(&t1, &d, &one);
    zr_fe_square(&t1, &t1);
    zr_fe_mul(&t0, &t0, &t1);
    zr_fe_sub(&Nt, &t0, &D);
    zr_fe_square(&ssq, &s);
    zr_fe_add(&t0, &s, &s);
    zr_fe_mul(&W0, &t0, &D);
    zr_fe_mul(&W1, &Nt, &ZR_SQRT_AD_MINUS_ONE);
    zr_fe_sub(&W2, &one, &ssq);
    zr_fe_add(&W3, &one, &ssq);
    zr_fe_mul(&out->X, &W0, &W3);
    zr_fe_mul(&out->Y, &W2, &W1);
    zr_fe_mul(&out->Z, &W1, &W3);
    zr_fe_mul(&out->T, &W0, &W2);
}

/* ------------------------------------------------------------------ ProjectivePoint (X:Y:Z) */
/* E:809-834 (BBJLP'08 projective add, a = -1); p, q, r: 3 field elements X|Y|Z */
void zr_proj_add(zr_fe r[3], const zr_fe p[3], const zr_fe q[3])
{
    zr_fe A, B, C, D, E, F, G, t0, t1, o[3];
    zr_fe_mul(&A, &p[2], &q[2]);
    zr_fe_square(&B, &A);
    zr_fe_mul(&C, &p[0], &q[0]);
    zr_fe_mul(&D, &p[1], &q[1]);
    zr_fe_mul(&E, &ZR_EDWARDS_D, &C);
    zr_fe_mul(&E, &E, &D);
    zr_fe_sub(&F, &B, &E);
    zr_fe_add(&G, &B, &E);
    zr_fe_add(&t0, &p[0], &p[1]);
    zr_fe_add(&t1, &q[0], &q[1]);
    zr_fe_mul(&t0, &t0, &t1);
    zr_fe_sub(&t0, &t0, &C);
    zr_fe_sub(&t0, &t0, &D);
    zr_fe_mul(&t0, &F, &t0);
    zr_fe_mul(&o[0], &A, &t0);
    zr_fe_mul(&t1, &A, &G);
    zr_fe_add(&t0, &D, &C);
    zr_fe_mul(&o[1], &t1, &t0);
    zr_fe_mul(&o[2], &F, &G);
    r[0] = o[0]; r[1] = o[1]; r[2] = o[2];
}
/* E:915-942 (dedicated projective doubling) */
void zr_proj_double(zr_fe r[3], const zr_fe p[3])
{
    const zr_fe two = {{2, 0, 0, 0, 0}};
    zr_fe B, C, D, E, F, H, J, t0, o[3];
    zr_fe_add(&t0, &p[0], &p[1]);
    zr_fe_square(&B, &t0);
    zr_fe_square(&C, &p[0]);
    zr_fe_square(&D, &p[1]);
    zr_fe_mul(&E, &ZR_EDWARDS_A, &C);
    zr_fe_add(&F, &E, &D);
    zr_fe_square(&H, &p[2]);
    zr_fe_mul(&t0, &two, &H);
    zr_fe_sub(&J, &F, &t0);
    zr_fe_sub(&t0, &B, &C);
    zr_fe_sub(&t0, &t0, &D);
    zr_fe_mul(&o[0], &t0, &J);
    zr_fe_sub(&t0, &E, &D);
    zr_fe_mul(&o[1], &F, &t0);
    zr_fe_mul(&o[2], &F, &J);
    r[0] = o[0]; r[1] = o[1]; r[2] = o[2];
}
/* E:402-417: (X*Z, Y*Z, Z^2, X*Y) */
void zr_proj_to_extended(zr_pt *r, const zr_fe p[3])
{
    zr_pt o;
    zr_fe_mul(&o.X, &p[0], &p[2]);
    zr_fe_mul(&o.Y, &p[1], &p[2]);
    zr_fe_square(&o.Z, &p[2]);
    zr_fe_mul(&o.T, &p[0], &p[1]);
    *r = o;
}
/* R:205-222 */
int zr_ris_is_valid(const zr_pt *p)
{
    zr_pt lp, id;
    zr_ed_scalar_mul(&lp, p, &ZR_L);
    zr_ed_identity(&id);
    int has_order_l = zr_ed_eq(&lp, &id) == 1;
    return has_order_l & zr_ed_is_valid(p);
}
/* R:493-507 */
void zr_ris_from_uniform_bytes(zr_pt *r, const uint8_t b[64])
{
    zr_fe r1, r2;
    zr_pt R1, R2;
    zr_fe_from_bytes(&r1, b);
    zr_ris_elligator(&R1, &r1);
    zr_fe_from_bytes(&r2, b + 32);
    zr_ris_elligator(&R2, &r2);
    zr_ed_add(r, &R1, &R2);
}

/* ------------------------------------------------------------------ batch wrappers */
#define FE(p, i) ((const zr_fe *)((p) + 5 * (i)))
#define FEO(p, i) ((zr_fe *)((p) + 5 * (i)))
#define PT(p, i) ((const zr_pt *)((p) + 20 * (i)))
#define PTO(p, i) ((zr_pt *)((p) + 20 * (i)))

#define BINOP_BATCH(name, fn)                                                                   \
    void name(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n)                     \
    { for (size_t i = 0; i < n; i++) { zr_fe r; fn(&r, FE(a, i), FE(b, i)); *FEO(out, i) = r; } }
#define UNOP_BATCH(name, fn)                                                                    \
    void name(const uint64_t *a, uint64_t *out, size_t n)                                        \
    { for (size_t i = 0; i < n; i++) { zr_fe r; fn(&r, FE(a, i)); *FEO(out, i) = r; } }

BINOP_BATCH(zr_fe_add_batch, zr_fe_add)
BINOP_BATCH(zr_fe_sub_batch, zr_fe_sub)
BINOP_BATCH(zr_fe_mul_batch, zr_fe_mul)
UNOP_BATCH(zr_fe_neg_batch, zr_fe_neg)
UNOP_BATCH(zr_fe_square_batch, zr_fe_square)
BINOP_BATCH(zr_sc_add_batch, zr_sc_add)
BINOP_BATCH(zr_sc_sub_batch, zr_sc_sub)
BINOP_BATCH(zr_sc_mul_batch, zr_sc_mul)
UNOP_BATCH(zr_sc_neg_batch, zr_sc_neg)
UNOP_BATCH(zr_sc_square_batch, zr_sc_square)
UNOP_BATCH(zr_sc_half_batch, zr_sc_half)
BINOP_BATCH(zr_sc_pow_batch, zr_sc_pow)
void zr_sc_shr_batch(const uint64_t *a, unsigned shift, uint64_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) { zr_fe r; zr_sc_shr(&r, FE(a, i), shift); *FEO(out, i) = r; } }
void zr_sc_into_bits_batch(const uint64_t *a, uint8_t *bits, size_t n)
{ for (size_t i = 0; i < n; i++) zr_sc_into_bits(bits + 256 * i, FE(a, i)); }
/* width 0: compute_NAF (S:370-389); otherwise compute_window_NAF(width) (S:396-415) */
void zr_sc_compute_naf_batch(const uint64_t *a, unsigned width, int8_t *naf, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        if (width == 0) zr_sc_compute_naf(naf + 256 * i, FE(a, i));
        else zr_sc_compute_window_naf(naf + 256 * i, FE(a, i), width);
    }
}
void zr_fe_inv_sqrt_batch(const uint64_t *a, uint64_t *out, uint8_t *was_square, size_t n)
{ for (size_t i = 0; i < n; i++) { zr_fe r; int c = zr_fe_inv_sqrt(&r, FE(a, i)); *FEO(out, i) = r; if (was_square) was_square[i] = (uint8_t)c; } }

void zr_fe_invert_batch(const uint64_t *a, uint64_t *out, uint8_t *ok, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        zr_fe r;
        int o = zr_fe_inverse(&r, FE(a, i));
        *FEO(out, i) = r;
        if (ok) ok[i] = (uint8_t)o;
    }
}
void zr_fe_div_batch(const uint64_t *a, const uint64_t *b, uint64_t *out, uint8_t *ok, size_t n)
{ for (size_t i = 0; i < n; i++) { zr_fe r; int o = zr_fe_div(&r, FE(a, i), FE(b, i)); *FEO(out, i) = r; if (ok) ok[i] = (uint8_t)o; } }
BINOP_BATCH(zr_fe_pow_batch, zr_fe_pow)
UNOP_BATCH(zr_fe_half_batch, zr_fe_half)
void zr_fe_legendre_symbol_batch(const uint64_t *a, uint8_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) out[i] = (uint8_t)zr_fe_legendre_symbol(FE(a, i)); }
void zr_fe_is_positive_batch(const uint64_t *a, uint8_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) out[i] = (uint8_t)zr_fe_is_positive(FE(a, i)); }
void zr_fe_mod_sqrt_batch(const uint64_t *a, int sign, uint64_t *out, uint8_t *ok, size_t n)
{ for (size_t i = 0; i < n; i++) { zr_fe r; int o = zr_fe_mod_sqrt(&r, FE(a, i), sign); *FEO(out, i) = r; if (ok) ok[i] = (uint8_t)o; } }
void zr_fe_from_bytes_batch(const uint8_t *in, uint64_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) zr_fe_from_bytes(FEO(out, i), in + 32 * i); }
void zr_fe_to_bytes_batch(const uint64_t *in, uint8_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) zr_fe_to_bytes(out + 32 * i, FE(in, i)); }
void zr_fe_sqrt_ratio_i_batch(const uint64_t *u, const uint64_t *v, uint64_t *out, uint8_t *was_square, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        zr_fe r;
        int c = zr_fe_sqrt_ratio_i(&r, FE(u, i), FE(v, i));
        *FEO(out, i) = r;
        if (was_square) was_square[i] = (uint8_t)c;
    }
}
void zr_sc_from_bytes_batch(const uint8_t *in, uint64_t *out, uint8_t *ok, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        int o = zr_sc_from_bytes(FEO(out, i), in + 32 * i);
        if (!o) *FEO(out, i) = FE_ZERO;                      /* the reference panics: no value (ABI: zero, ok = 0) */
        if (ok) ok[i] = (uint8_t)o;
    }
}
void zr_sc_to_bytes_batch(const uint64_t *in, uint8_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) zr_sc_to_bytes(out + 32 * i, FE(in, i)); }

void zr_ed_add_batch(const uint64_t *p, const uint64_t *q, uint64_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) { zr_pt r; zr_ed_add(&r, PT(p, i), PT(q, i)); *PTO(out, i) = r; } }
void zr_ed_sub_batch(const uint64_t *p, const uint64_t *q, uint64_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) { zr_pt r; zr_ed_sub(&r, PT(p, i), PT(q, i)); *PTO(out, i) = r; } }
void zr_ed_double_batch(const uint64_t *p, uint64_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) { zr_pt r; zr_ed_double(&r, PT(p, i)); *PTO(out, i) = r; } }
void zr_ed_neg_batch(const uint64_t *p, uint64_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) { zr_pt r; zr_ed_neg(&r, PT(p, i)); *PTO(out, i) = r; } }
void zr_ed_scalar_mul_batch(const uint64_t *p, const uint64_t *k, uint64_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) { zr_pt r; zr_ed_scalar_mul(&r, PT(p, i), FE(k, i)); *PTO(out, i) = r; } }
/* mode 0: double_and_add (E:102-120), 1: ltr_bin_mul (E:122-134), 2: binary_naf_mul (E:136-153) */
void zr_ed_scalar_mul_mode_batch(const uint64_t *p, const uint64_t *k, uint64_t *out, size_t n, int mode)
{
    for (size_t i = 0; i < n; i++) {
        zr_pt r;
        if (mode == 1) zr_ed_ltr_bin_mul(&r, PT(p, i), FE(k, i));
        else if (mode == 2) zr_ed_binary_naf_mul(&r, PT(p, i), FE(k, i));
        else zr_ed_scalar_mul(&r, PT(p, i), FE(k, i));
        *PTO(out, i) = r;
    }
}
void zr_ed_mul_by_pow_2_batch(const uint64_t *p, uint64_t kexp, uint64_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) { zr_pt r; zr_ed_mul_by_pow_2(&r, PT(p, i), kexp); *PTO(out, i) = r; } }
void zr_ed_to_affine_batch(const uint64_t *p, uint64_t *xy, uint8_t *ok, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        zr_fe x, y;
        int o = zr_ed_to_affine(&x, &y, PT(p, i));
        *FEO(xy, 2 * i) = x;
        *FEO(xy, 2 * i + 1) = y;
        if (ok) ok[i] = (uint8_t)o;
    }
}
void zr_ed_eq_batch(const uint64_t *p, const uint64_t *q, uint8_t *eq, size_t n)
{ for (size_t i = 0; i < n; i++) { int e = zr_ed_eq(PT(p, i), PT(q, i)); eq[i] = (uint8_t)(e == 1); } }
void zr_ed_compress_batch(const uint64_t *p, uint8_t *out, uint8_t *ok, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        int o = zr_ed_compress(out + 32 * i, PT(p, i));
        if (ok) ok[i] = (uint8_t)o;
    }
}
void zr_ed_decompress_batch(const uint8_t *in, uint64_t *out, uint8_t *ok, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        zr_pt r;
        int o = zr_ed_decompress(&r, in + 32 * i);
        if (!o) zr_ed_identity(&r);
        *PTO(out, i) = r;
        if (ok) ok[i] = (uint8_t)o;
    }
}
void zr_ris_compress_batch(const uint64_t *p, uint8_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) zr_ris_compress(out + 32 * i, PT(p, i)); }
void zr_ris_decompress_batch(const uint8_t *in, uint64_t *out, uint8_t *ok, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        zr_pt r;
        int o = zr_ris_decompress(&r, in + 32 * i);
        if (!o) zr_ed_identity(&r);
        *PTO(out, i) = r;
        if (ok) ok[i] = (uint8_t)o;
    }
}
void zr_ris_eq_batch(const uint64_t *p, const uint64_t *q, uint8_t *eq, size_t n)
{ for (size_t i = 0; i < n; i++) eq[i] = (uint8_t)zr_ris_eq(PT(p, i), PT(q, i)); }
/* R:96-154 -> E:102-120 -> R:398-425 */
void zr_ris_roundtrip_mul_batch(const uint8_t *in, const uint64_t *k, uint8_t *out, uint8_t *ok, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        zr_pt P, Q;
        int o = zr_ris_decompress(&P, in + 32 * i);
        if (o) {
            zr_ed_scalar_mul(&Q, &P, FE(k, i));
            zr_ris_compress(out + 32 * i, &Q);
        } else memset(out + 32 * i, 0, 32);
        if (ok) ok[i] = (uint8_t)o;
    }
}
/* ------------------------------------------------------------------ rows beside the path (SURVEY 8a E-x) */
/* FOUR_COSET_GROUP[0..2], backend/u64/constants.rs:141-183 (coset4 never reads entry 3) */
static const zr_pt ZR_FOUR_COSET[3] = {
    {{{1ULL, 0ULL, 0ULL, 0ULL, 0ULL}}, {{0ULL, 0ULL, 0ULL, 0ULL, 0ULL}}, {{1ULL, 0ULL, 0ULL, 0ULL, 0ULL}}, {{0ULL, 0ULL, 0ULL, 0ULL, 0ULL}}},
    {{{2099929430230996ULL, 1464742363261928ULL, 3309265759432790ULL, 2285299817698826ULL, 10215362715769ULL}}, {{0ULL, 0ULL, 0ULL, 0ULL, 0ULL}}, {{1ULL, 0ULL, 0ULL, 0ULL, 0ULL}}, {{0ULL, 0ULL, 0ULL, 0ULL, 0ULL}}},
    {{{0ULL, 0ULL, 0ULL, 0ULL, 0ULL}}, {{671914833335276ULL, 3916664325105025ULL, 1367801ULL, 0ULL, 17592186044416ULL}}, {{1ULL, 0ULL, 0ULL, 0ULL, 0ULL}}, {{0ULL, 0ULL, 0ULL, 0ULL, 0ULL}}}
};
/* E:603-610: [P, P + C0, P + C1, P + C2] through the unified addition E:465-489 */
void zr_ed_coset4(zr_pt r[4], const zr_pt *p)
{
    const zr_pt in = *p;
    r[0] = in;
    for (int j = 0; j < 3; j++) zr_ed_add(&r[j + 1], &in, &ZR_FOUR_COSET[j]);
}
/* E:787-807: (-X, Y, Z) */
void zr_proj_neg(zr_fe r[3], const zr_fe p[3])
{
    zr_fe x;
    zr_fe_neg(&x, &p[0]);
    r[0] = x; r[1] = p[1]; r[2] = p[2];
}
/* E:851-879: self + (-other) */
void zr_proj_sub(zr_fe r[3], const zr_fe p[3], const zr_fe q[3])
{
    zr_fe nq[3];
    zr_proj_neg(nq, q);
    zr_proj_add(r, p, nq);
}
/* E:701-711: both sides to AffinePoint (E:1094-1110: X/Z, Y/Z; inverse() panics for Z = 0 -> -1), then E:1044-1048 */
int zr_proj_eq(const zr_fe p[3], const zr_fe q[3])
{
    zr_fe zi, x1, y1, x2, y2;
    if (!zr_fe_inverse(&zi, &p[2])) return -1;
    zr_fe_mul(&x1, &p[0], &zi);
    zr_fe_mul(&y1, &p[1], &zi);
    if (!zr_fe_inverse(&zi, &q[2])) return -1;
    zr_fe_mul(&x2, &q[0], &zi);
    zr_fe_mul(&y2, &q[1], &zi);
    return limbs_eq(&x1, &x2) & limbs_eq(&y1, &y2);
}
/* E:733-748: (aX^2 + Y^2) Z^2 == Z^4 + d X^2 Y^2 */
int zr_proj_is_valid(const zr_fe p[3])
{
    zr_pt e;
    e.X = p[0]; e.Y = p[1]; e.Z = p[2]; e.T = FE_ZERO;
    return zr_ed_is_valid(&e);                                /* E:393-400 delegates to this very check */
}
/* E:881-912 -> double_and_add E:102-120 with T = ProjectivePoint: identity (0, 1, 1) E:722-731,
 * Add E:809-834, dedicated Double E:915-942 */
void zr_proj_scalar_mul(zr_fe r[3], const zr_fe p[3], const zr_sc *k)
{
    zr_fe N[3] = {p[0], p[1], p[2]}, Q[3] = {FE_ZERO, FE_ONE, FE_ONE};
    zr_sc n = *k;
    while (!limbs_eq(&n, &FE_ZERO)) {
        if (!zr_sc_is_even(&n)) zr_proj_add(Q, Q, N);
        zr_proj_double(N, N);
        limbs_half_without_mod(&n, &n);
    }
    r[0] = Q[0]; r[1] = Q[1]; r[2] = Q[2];
}
void zr_ed_coset4_batch(const uint64_t *p, uint64_t *out4, size_t n)
{ for (size_t i = 0; i < n; i++) zr_ed_coset4((zr_pt *)(out4 + 80 * i), PT(p, i)); }
void zr_proj_neg_batch(const uint64_t *p, uint64_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) zr_proj_neg((zr_fe *)(out + 15 * i), (const zr_fe *)(p + 15 * i)); }
void zr_proj_sub_batch(const uint64_t *p, const uint64_t *q, uint64_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) zr_proj_sub((zr_fe *)(out + 15 * i), (const zr_fe *)(p + 15 * i), (const zr_fe *)(q + 15 * i)); }
void zr_proj_eq_batch(const uint64_t *p, const uint64_t *q, uint8_t *eq, uint8_t *ok, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        const int e = zr_proj_eq((const zr_fe *)(p + 15 * i), (const zr_fe *)(q + 15 * i));
        eq[i] = (uint8_t)(e == 1);
        if (ok) ok[i] = (uint8_t)(e >= 0);
    }
}
void zr_proj_is_valid_batch(const uint64_t *p, uint8_t *valid, size_t n)
{ for (size_t i = 0; i < n; i++) valid[i] = (uint8_t)zr_proj_is_valid((const zr_fe *)(p + 15 * i)); }
void zr_proj_scalar_mul_batch(const uint64_t *p, const uint64_t *k, uint64_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) zr_proj_scalar_mul((zr_fe *)(out + 15 * i), (const zr_fe *)(p + 15 * i), FE(k, i)); }
void zr_proj_add_batch(const uint64_t *p, const uint64_t *q, uint64_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) zr_proj_add((zr_fe *)(out + 15 * i), (const zr_fe *)(p + 15 * i), (const zr_fe *)(q + 15 * i)); }
void zr_proj_double_batch(const uint64_t *p, uint64_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) zr_proj_double((zr_fe *)(out + 15 * i), (const zr_fe *)(p + 15 * i)); }
void zr_proj_to_extended_batch(const uint64_t *p, uint64_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) zr_proj_to_extended(PTO(out, i), (const zr_fe *)(p + 15 * i)); }
void zr_ed_is_valid_batch(const uint64_t *p, uint8_t *valid, size_t n)
{ for (size_t i = 0; i < n; i++) valid[i] = (uint8_t)zr_ed_is_valid(PT(p, i)); }
void zr_ris_is_valid_batch(const uint64_t *p, uint8_t *valid, size_t n)
{ for (size_t i = 0; i < n; i++) valid[i] = (uint8_t)zr_ris_is_valid(PT(p, i)); }
void zr_ris_elligator_batch(const uint64_t *r0, uint64_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) { zr_pt r; zr_ris_elligator(&r, FE(r0, i)); *PTO(out, i) = r; } }
void zr_ris_from_uniform_bytes_batch(const uint8_t *in, uint64_t *out, size_t n)
{ for (size_t i = 0; i < n; i++) { zr_pt r; zr_ris_from_uniform_bytes(&r, in + 64 * i); *PTO(out, i) = r; } }
void zr_msm_naive(const uint64_t *p, const uint64_t *k, size_t n, uint64_t *out_point)
{
    zr_pt acc, t;
    zr_ed_identity(&acc);
    for (size_t i = 0; i < n; i++) {
        zr_ed_scalar_mul(&t, PT(p, i), FE(k, i));
        zr_ed_add(&acc, &acc, &t);
    }
    *PTO(out_point, 0) = acc;
}

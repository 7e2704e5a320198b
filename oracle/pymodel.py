"""
pymodel.py -- independent big-integer model of the Sonny/Doppio curve path.

TEST INFRASTRUCTURE, NOT THE PRODUCT PATH.  This is a *second*, independently
derived statement of the mathematics (Python ints, `pow(x, -1, p)`, closed-form
p = 5 (mod 8) square roots) used to
  * cross-check the C oracle (oracle/zc_ref.c) on bulk random inputs, and
  * generate the seeded golden fixtures under tests/golden/.
It deliberately does NOT follow the reference's limb algorithms; it follows the
reference's *decision rules* (which root, which sign, rejection conditions),
cited below (paths relative to the reference checkout).
"""
from __future__ import annotations

# src/backend/u64/constants.rs:29-36 / :8-9
P = 2**252 + 27742317777372353535851937790883648493
L = 2**249 + 14490550575682688738086195780655237219
assert P % 8 == 5

MASK52 = (1 << 52) - 1


def limbs(v: int):
    """value -> five radix-2^52 limbs (src/backend/u64/field.rs:26-32)."""
    return [(v >> (52 * i)) & MASK52 for i in range(4)] + [v >> 208]


def from_limbs(l) -> int:
    return sum(int(x) << (52 * i) for i, x in enumerate(l))


D = (-126296 * pow(126297, -1, P)) % P  # src/backend/u64/constants.rs:83-92
A = P - 1
# src/backend/u64/constants.rs:96-102 -- the reference's choice of sqrt(-1)
SQRT_M1 = 3034649101460298094273452163494570791663566989388331537498831373842135895065
assert SQRT_M1 * SQRT_M1 % P == P - 1
# src/backend/u64/constants.rs:123-129
INV_SQRT_A_MINUS_D = 482283834104289360917429750399313974390948281833312135312952165682596457149
assert INV_SQRT_A_MINUS_D * INV_SQRT_A_MINUS_D * (A - D) % P == 1
SQRT_AD_MINUS_ONE = from_limbs([3601277882726560, 1817821323014817, 1726005090908779,
                                2111284621343800, 648674458156])
assert SQRT_AD_MINUS_ONE**2 % P == (A * D - 1) % P

Q4 = (P - 1) // 4  # p - 1 = 4 q, q odd


def is_positive(x: int) -> bool:
    """src/backend/u64/field.rs:552-557: x <= (p-1)/2."""
    return x <= (P - 1) // 2


def legendre(x: int) -> int:
    """src/backend/u64/field.rs:703-706: Choice(1) unless x^((p-1)/2) == -1."""
    return 0 if pow(x, (P - 1) // 2, P) == P - 1 else 1


def ts_root(a: int):
    """The specific root Tonelli-Shanks with non-residue 6 returns
    (src/backend/u64/field.rs:357-441) for p - 1 = 4q:
    x = a^((q+1)/2), times 6^q when a^q != 1."""
    if a == 0:
        return 0
    if legendre(a) == 0:
        return None
    x = pow(a, (Q4 + 1) // 2, P)
    if pow(a, Q4, P) != 1:
        x = x * pow(6, Q4, P) % P
    assert x * x % P == a
    return x


def mod_sqrt(a: int, sign: int):
    """sign=0 -> x_TS, sign=1 -> p - x_TS (field.rs:435-439)."""
    x = ts_root(a)
    if x is None:
        return None
    if a == 0:
        return 0
    return (P - x) % P if sign else x


def sqrt_ratio_i(u: int, v: int):
    """src/backend/u64/field.rs:462-503."""
    if u == 0:
        return 1, 0
    if v == 0:
        return 0, 0
    q = u * pow(v, -1, P) % P
    if legendre(q) == 1:
        r = mod_sqrt(q, 1)
        return 1, (r if is_positive(r) else P - r)
    r = mod_sqrt(SQRT_M1 * q % P, 1)
    return 0, (r if is_positive(r) else P - r)


def inv_sqrt(x: int):
    return sqrt_ratio_i(1, x)


def fe_from_bytes(b: bytes) -> int:
    """src/backend/u64/field.rs:563-587: all 256 bits kept, no reduction."""
    return int.from_bytes(b, "little")


def fe_to_bytes(x: int) -> bytes:
    return x.to_bytes(32, "little")


# ------------------------------------------------------------------ points (x,y,z,t) as ints
IDENT = (0, 1, 1, 0)


def ed_add(p1, p2):
    """src/edwards.rs:465-489."""
    x1, y1, z1, t1 = p1
    x2, y2, z2, t2 = p2
    a = x1 * x2 % P
    b = y1 * y2 % P
    c = D * t1 % P * t2 % P
    d = z1 * z2 % P
    e = ((x1 + y1) * (x2 + y2) - a - b) % P
    f = (d - c) % P
    g = (d + c) % P
    h = (b + a) % P
    return (e * f % P, g * h % P, f * g % P, e * h % P)


def ed_neg(p):
    x, y, z, t = p
    return ((-x) % P, y, z, (-t) % P)


def ed_sub(p1, p2):
    """src/edwards.rs:503-531 -- same canonical value as add(p1, -p2)."""
    return ed_add(p1, ed_neg(p2))


def ed_scalar_mul(p, k: int):
    """src/edwards.rs:102-120, including the literal identity + N first add.  `k` is the integer the
    five limbs spell (up to 260 bits for a raw `Scalar([..])`): is_even / half_without_mod see all of
    it, but the loop test `n != Scalar::zero()` compares 32-byte encodings (src/scalar.rs:78-91), i.e.
    only the low 256 bits of n (to_bytes, src/backend/u64/scalar.rs:477-516)."""
    n, q = p, IDENT
    while k % (1 << 256) != 0:
        if k & 1:
            q = ed_add(q, n)
        n = ed_add(n, n)
        k >>= 1
    return q


def ed_affine(p):
    x, y, z, _ = p
    zi = pow(z, -1, P)
    return x * zi % P, y * zi % P


def ed_eq(p1, p2):
    return ed_affine(p1) == ed_affine(p2)


def find_xx(y):
    return (y * y - 1) * pow((D * y * y - A) % P, -1, P) % P


def ed_compress(p) -> bytes:
    """src/edwards.rs:613-629."""
    x, y = ed_affine(p)
    r = mod_sqrt(find_xx(y), 0)
    sign = 1 if r != x else 0
    b = bytearray(fe_to_bytes(y))
    b[31] |= sign << 7
    return bytes(b)


def ed_from_y(y: int, sign: int):
    """src/edwards.rs:962-979 + 402-417."""
    den = (D * y * y - A) % P
    if den == 0:
        return None
    x = mod_sqrt((y * y - 1) * pow(den, -1, P) % P, sign)
    if x is None:
        return None
    return (x, y % P, 1, x * y % P)


def ed_decompress(b: bytes):
    """src/edwards.rs:313-326 (byte 31 masked with 0x0F)."""
    sign = b[31] >> 7
    bb = bytearray(b)
    bb[31] &= 0x0F
    y = fe_from_bytes(bytes(bb))
    if y >= P:  # outside the parity contract (canonical inputs only)
        raise ValueError("non-canonical y")
    return ed_from_y(y, sign)


def ris_decompress(b: bytes):
    """src/ristretto.rs:96-154."""
    s = fe_from_bytes(b)
    if s > (P - 1) // 2:
        return None
    ss = s * s % P
    u1 = (1 - ss) % P
    u2 = (1 + ss) % P
    u2sq = u2 * u2 % P
    v = (-(D * u1 * u1) - u2sq) % P
    ok, i = inv_sqrt(v * u2sq % P)
    if not ok:
        return None
    dx = i * u2 % P
    dy = i * dx % P * v % P
    x = 2 * s * dx % P
    if not is_positive(x):
        x = P - x
    y = u1 * dy % P
    t = x * y % P
    if (not is_positive(t)) or y == 0:
        return None
    return (x, y, 1, t)


def ris_compress(p) -> bytes:
    """src/ristretto.rs:398-425."""
    x, y, z, t = p
    u1 = (z + y) * (z - y) % P
    u2 = x * y % P
    _, i = inv_sqrt(u1 * u2 * u2 % P)
    d1 = u1 * i % P
    d2 = u2 * i % P
    zinv = d1 * d2 % P * t % P
    if not is_positive(t * zinv % P):
        xx, yy = SQRT_M1 * y % P, SQRT_M1 * x % P
        dd = d1 * INV_SQRT_A_MINUS_D % P
    else:
        xx, yy, dd = x, y, d2
    if not is_positive(xx * zinv % P):
        yy = (-yy) % P
    s = (z - yy) * dd % P
    if not is_positive(s):
        s = P - s
    return fe_to_bytes(s)


def ris_eq(p1, p2) -> bool:
    """src/ristretto.rs:166-176."""
    x1, y1, _, _ = p1
    x2, y2, _, _ = p2
    return (x1 * y2 - y1 * x2) % P == 0 or (x1 * x2 - y1 * y2) % P == 0


def elligator(r0: int):
    """src/ristretto.rs:430-471 (r0 is used as given, not reduced)."""
    c = P - 1
    one_minus_d_sq = (1 - D * D) % P
    r = SQRT_M1 * r0 * r0 % P
    ns = (r + 1) * one_minus_d_sq % P
    dd = (c - D * r) * (r + D) % P
    is_sq, s = sqrt_ratio_i(ns, dd)
    sp = s * r0 % P
    if is_positive(sp):
        sp = (-sp) % P
    if not is_sq:
        s, c = sp, r
    nt = (c * (r - 1) % P * pow(D - 1, 2, P) - dd) % P
    ssq = s * s % P
    w0 = 2 * s * dd % P
    w1 = nt * SQRT_AD_MINUS_ONE % P
    w2 = (1 - ssq) % P
    w3 = (1 + ssq) % P
    return (w0 * w3 % P, w2 * w1 % P, w1 * w3 % P, w0 * w2 % P)


# src/backend/u64/constants.rs:188-211
BASEPOINT = (
    from_limbs([276718085098056, 1646536057461434, 2704687245600312, 2630386667454967, 13476148227069]),
    from_limbs([1303868825475266, 3250718520537114, 2702159777242978, 2702159776422297, 10555311626649]),
    1,
    from_limbs([3634527586288175, 2006028620404053, 3424252198034825, 2478951925947079, 4567251727358]),
)
assert BASEPOINT[1] == 3 * pow(5, -1, P) % P
assert BASEPOINT[3] == BASEPOINT[0] * BASEPOINT[1] % P


def pt_limbs(p):
    return [limbs(c) for c in p]


# ------------------------------------------------------------------ rows beside the default path
def sc_shr(k: int, s: int) -> int:
    """Shr<u8> (backend scalar.rs:165-182): the limbs as one integer, shifted; no reduction."""
    return k >> s


def sc_into_bits(k: int):
    """into_bits (backend scalar.rs:352-366): the 256 bits of to_bytes()."""
    return [(k >> i) & 1 for i in range(256)]


def sc_wnaf(k: int, width: int):
    """compute_NAF (width 0, digits k mods 4) / compute_window_NAF(width) as the textbook integer
    recoding: valid for 0 <= k < L - 2^width, where the reference's modular `k - Scalar::from(k_i)`
    (backend scalar.rs:370-415) never wraps.  256 digits, least significant first."""
    w = 2 if width == 0 else width
    out = [0] * 256
    i = 0
    while k >= 1 and i < 256:
        if k & 1:
            d = k % (1 << w)
            if d >= (1 << (w - 1)):
                d -= 1 << w
            out[i] = d
            k -= d
        k >>= 1
        i += 1
    return out


def proj_add(p1, p2):
    """ProjectivePoint add (edwards.rs:809-834), a = -1: BBJLP'08 add-2008-bbjlp on (X:Y:Z)."""
    x1, y1, z1 = p1
    x2, y2, z2 = p2
    a = z1 * z2 % P
    b = a * a % P
    c = x1 * x2 % P
    d = y1 * y2 % P
    e = D * c % P * d % P
    f = (b - e) % P
    g = (b + e) % P
    x3 = a * f % P * (((x1 + y1) * (x2 + y2) - c - d) % P) % P
    y3 = a * g % P * ((d + c) % P) % P          # d - a*c with a = -1
    return (x3, y3, f * g % P)


def proj_double(p):
    """ProjectivePoint double (edwards.rs:915-942): dbl-2008-bbjlp with a = -1."""
    x, y, z = p
    b = (x + y) * (x + y) % P
    c = x * x % P
    d = y * y % P
    e = (-c) % P
    f = (e + d) % P
    h = z * z % P
    j = (f - 2 * h) % P
    return ((b - c - d) * j % P, f * ((e - d) % P) % P, f * j % P)


def proj_scalar_mul(p, k: int):
    """Mul<Scalar> for ProjectivePoint (edwards.rs:881-912) = double_and_add (:102-120), identity (0, 1, 1)."""
    n, q = p, (0, 1, 1)
    while k % (1 << 256) != 0:
        if k & 1:
            q = proj_add(q, n)
        n = proj_double(n)
        k >>= 1
    return q

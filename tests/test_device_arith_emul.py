"""CPU tier: the device arithmetic headers (radix-2^29 Montgomery, unified-step scalar
multiplication, one-exponentiation codecs), compiled for the host with g++ by
tests/emul/, must agree bit-for-bit with the oracle.  This checks the kernels' limb
algorithms before any GPU time is spent; it is not a CPU fallback (nothing in
dusk_zerocaf_amd/ can reach it)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pymodel as pm
from tests import vectors as V

HERE = os.path.dirname(os.path.abspath(__file__))
EMUL_DIR = os.path.join(HERE, "emul")


@pytest.fixture(scope="module", params=["plain", "checked"])
def emul(request):
    """`checked` = the same headers with -DZC_CHECK_BOUNDS: every precondition of the lazy-reduction
    scheme (limb ranges of multiplier inputs, subtrahends below the 4N bias, ...) aborts when violated."""
    checked = request.param == "checked"
    # ZC_EMUL_SANITIZE (tests/test_sanitizers.py): the same build under AddressSanitizer + UBSan, no recovery
    san = bool(os.environ.get("ZC_EMUL_SANITIZE"))
    so = os.path.join(EMUL_DIR, "libzc_emul%s%s.so" % ("_san" if san else "", "_checked" if checked else ""))
    src = os.path.join(EMUL_DIR, "emul.cpp")
    csrc = os.path.join(os.path.dirname(HERE), "dusk_zerocaf_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in ("zc_arith.hip.h", "zc_curve.hip.h", "zc_constants.hip.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        inc = "/opt/rocm/include"
        if not os.path.isdir(inc):
            pytest.skip("ROCm headers not present")
        subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__"] +
                              (["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"] if san else ["-O2"]) +
                              (["-DZC_CHECK_BOUNDS"] if checked else []) + ["-I" + inc, "-o", so, src])
    lib = C.CDLL(so)
    lib.zc_checked = checked
    return lib


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_emul_mul_square(emul, oracle):
    for modl, mod, orc_mul, orc_sq, edge in ((0, pm.P, oracle.fe_mul, oracle.fe_square, V.FE_EDGE),
                                             (1, pm.L, oracle.sc_mul, oracle.sc_square, V.SC_EDGE)):
        a = V.limbs_array(V.rand_fe(4000, V.SEED + 1, mod, edge))
        b = V.limbs_array(list(reversed(V.rand_fe(4000, V.SEED + 2, mod, edge))))
        out = np.empty_like(a)
        emul.emul_fe_mul(p(a), p(b), p(out), C.c_size_t(len(a)), modl)
        assert np.array_equal(out, orc_mul(a, b))
        emul.emul_fe_square(p(a), p(out), C.c_size_t(len(a)), modl)
        assert np.array_equal(out, orc_sq(a))
    a = V.limbs_array(V.rand_fe(2000, V.SEED + 1))
    b = V.limbs_array(list(reversed(V.rand_fe(2000, V.SEED + 2))))
    prod, sq = np.empty_like(a), np.empty_like(a)
    emul.emul_fe_mul_square_ilp(p(a), p(b), p(prod), p(sq), C.c_size_t(len(a)))      # independent-chain pair
    assert np.array_equal(prod, oracle.fe_mul(a, b)) and np.array_equal(sq, oracle.fe_square(a))
    # non-canonical operands with limbs < 2^52 are value-correct too
    rng = np.random.default_rng(7)
    a = rng.integers(0, 1 << 52, size=(500, 5), dtype=np.uint64)
    b = rng.integers(0, 1 << 52, size=(500, 5), dtype=np.uint64)
    out = np.empty_like(a)
    emul.emul_fe_mul(p(a), p(b), p(out), C.c_size_t(500), 0)
    assert np.array_equal(out, oracle.fe_mul(a, b))


def test_emul_one_pass_product_of_the_mul_square_kernels(emul, oracle):
    """What k_fe_mul / k_fe_square / k_sc_mul / k_sc_square run per lane since round 5 (zc_arith.hip.h: fe_mulmod_limbs52,
    fe_sqrmod_limbs52): for canonical operands ONE pass -- a (b 2^S), two folds by 2^261 = -(c 2^S), one conditional
    subtraction -- instead of the reference's two Montgomery passes (field.rs:250-262, :302-315; scalar.rs likewise);
    operands at or above 2^TOPBIT take the two-pass form.  Limb for limb the oracle's Mul / Square: random and edge
    canonical values, everything around 2^TOPBIT and N, all-ones patterns, and raw 260-bit patterns (SURVEY A.3 item 11:
    Mul / Square are value-correct for ANY limbs < 2^52); the checked build asserts every intermediate bound."""
    rng = np.random.default_rng(V.SEED + 700)
    for modl, mod, top, orc_mul, orc_sq, edge in ((0, pm.P, 252, oracle.fe_mul, oracle.fe_square, V.FE_EDGE),
                                                  (1, pm.L, 249, oracle.sc_mul, oracle.sc_square, V.SC_EDGE)):
        vals = V.rand_fe(3000, V.SEED + 701 + modl, mod, edge)
        special = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, (mod + 1) // 2, (1 << top) - 1, (1 << top) - 2, 1 << (top - 1), (1 << (top - 1)) - 1,
                   (1 << top), (1 << top) + 1, mod - (1 << 60), (1 << top) - (1 << 125), (1 << 125) - 1, 1 << 125, (1 << 232) - 1, 1 << 232,
                   (1 << 29) - 1, 1 << 29, (1 << 261) % mod, ((1 << 261) - 1) % mod]
        special += [((1 << top) - 1) ^ (1 << i) for i in range(0, top, 7)]               # all ones with one hole
        special += [((1 << 29 * k) - 1) % mod for k in range(1, 10)] + [(1 << 29 * k) % mod for k in range(1, 9)]
        a = V.limbs_array(vals + special + special[::-1] + [s for s in special for _ in range(2)])
        b = V.limbs_array(list(reversed(vals)) + special + special + [special[(i * 7 + 3) % len(special)] for i in range(2 * len(special))])
        assert len(a) == len(b)
        prod, sq = np.empty_like(a), np.empty_like(a)
        emul.emul_mulmod(p(a), p(b), p(prod), p(sq), C.c_size_t(len(a)), modl)
        assert np.array_equal(prod, orc_mul(a, b)) and np.array_equal(sq, orc_sq(a))
        # raw patterns: any limbs < 2^52 (most of them at or above 2^TOPBIT: the two-pass branch), and patterns just below 2^TOPBIT
        a = rng.integers(0, 1 << 52, size=(1500, 5), dtype=np.uint64)
        b = rng.integers(0, 1 << 52, size=(1500, 5), dtype=np.uint64)
        a[::2, 4] >>= np.uint64(52 - (top - 208))                                         # every other operand below 2^TOPBIT
        b[::3, 4] >>= np.uint64(52 - (top - 208))
        a[5::6, 4] = (1 << (top - 208)) - 1                                               # top limb all ones below the bound
        prod, sq = np.empty_like(a), np.empty_like(a)
        emul.emul_mulmod(p(a), p(b), p(prod), p(sq), C.c_size_t(len(a)), modl)
        assert np.array_equal(prod, orc_mul(a, b)) and np.array_equal(sq, orc_sq(a))


def test_emul_invert_sqrt_ratio(emul, oracle):
    a = V.limbs_array(V.rand_fe(200, V.SEED + 3))
    out, ok = np.empty_like(a), np.empty(len(a), dtype=np.uint8)
    emul.emul_fe_invert(p(a), p(out), p(ok), C.c_size_t(len(a)))
    want, wok = oracle.fe_invert(a)
    assert np.array_equal(ok, wok) and np.array_equal(out, want)
    for c in (2, 7, 64):                                      # chunked Montgomery-trick inversion
        a2 = a.copy()
        a2[3] = 0
        a2[-1] = 0
        out2, ok2 = np.empty_like(a2), np.empty(len(a2), dtype=np.uint8)
        emul.emul_fe_invert_chunked(p(a2), p(out2), p(ok2), C.c_size_t(len(a2)), c)
        want2, wok2 = oracle.fe_invert(a2)
        assert np.array_equal(ok2, wok2) and np.array_equal(out2, want2), c
        num = V.limbs_array(V.rand_fe(len(a2), V.SEED + 6))
        emul.emul_fe_div_chunked(p(num), p(a2), p(out2), p(ok2), C.c_size_t(len(a2)), c)
        wq, wqok = oracle.fe_div(num, a2)
        assert np.array_equal(ok2, wqok) and np.array_equal(out2, wq), c
        # the single-wave launches' form (independent-chain multiplier): k_fe_invert_chunked_lone / k_fe_div_chunked_lone
        emul.emul_fe_invert_chunked_lone(p(a2), p(out2), p(ok2), C.c_size_t(len(a2)), c)
        assert np.array_equal(ok2, wok2) and np.array_equal(out2, want2), c
        emul.emul_fe_div_chunked_lone(p(num), p(a2), p(out2), p(ok2), C.c_size_t(len(a2)), c)
        assert np.array_equal(ok2, wqok) and np.array_equal(out2, wq), c
    u = V.limbs_array(V.rand_fe(120, V.SEED + 4))
    v = V.limbs_array(list(reversed(V.rand_fe(120, V.SEED + 5))))
    sq = np.empty(len(u), dtype=np.uint8)
    emul.emul_fe_sqrt_ratio_i(p(u), p(v), p(out[:120]), p(sq), C.c_size_t(120))
    want, wsq = oracle.fe_sqrt_ratio_i(u, v)
    assert np.array_equal(sq, wsq) and np.array_equal(out[:120], want)


def test_emul_legendre_is_the_jacobi_symbol(emul, oracle):
    """legendre_symbol (field.rs:703-706) on the division-step machinery: the Jacobi-symbol walk, the exponentiation it
    replaces and the oracle agree; with a round bound too small to finish, the fallback answers."""
    vals = V.rand_fe(400, V.SEED + 40) + [(x * x) % pm.P for x in V.rand_fe(100, V.SEED + 41)] + [2 ** k for k in range(0, 252, 7)]
    a = V.limbs_array(vals)
    want = oracle.fe_legendre_symbol(a)
    assert 0 < int(want.sum()) < len(a)                               # both answers occur
    for rounds in (40, 26, 3, 0):                                      # 26: some lanes unfinished; 3, 0: all fall back
        jac, pw = np.empty(len(a), dtype=np.uint8), np.empty(len(a), dtype=np.uint8)
        emul.emul_fe_legendre(p(a), p(jac), p(pw), C.c_size_t(len(a)), rounds)
        assert np.array_equal(jac, want) and np.array_equal(pw, want), rounds


def test_emul_point_ops(emul, oracle):
    n = 48
    P = V.base_multiples(oracle, n, V.SEED + 6)
    Q = V.base_multiples(oracle, n, V.SEED + 7)
    P[0] = V.IDENT_ROW
    Q[1] = V.IDENT_ROW
    Q[2] = P[2]
    out = np.empty_like(P)
    emul.emul_ed_add(p(P), p(Q), p(out), C.c_size_t(n))
    assert np.array_equal(out, oracle.ed_add(P, Q))
    emul.emul_ed_sub(p(P), p(Q), p(out), C.c_size_t(n))
    assert np.array_equal(out, oracle.ed_sub(P, Q))
    for mode, want in ((0, oracle.ed_add(P, Q)), (1, oracle.ed_sub(P, Q)), (2, oracle.ed_double(P))):   # plain-domain path
        emul.emul_ed_add_plain(p(P), p(Q), p(out), C.c_size_t(n), mode)
        assert np.array_equal(out, want), mode
    K = V.rand_scalars_np(n, V.SEED + 8, bits=252)
    K[0] = 0
    K[1] = [1, 0, 0, 0, 0]
    K[2] = pm.limbs(pm.L)
    K[3] = pm.limbs(2**249 - 1)
    K[4] = [(1 << 52) - 1] * 4 + [(1 << 52) - 1]          # all 260 bits set (raw limb pattern)
    K[5] = [0, 0, 0, 0, 1 << 47]
    K[6] = pm.limbs(8)
    emul.emul_ed_scalar_mul(p(P), p(K), p(out), C.c_size_t(n))
    small = np.empty_like(out)
    emul.emul_ed_scalar_mul_small(p(P), p(K), p(small), C.c_size_t(n))
    assert np.array_equal(small, out)
    assert np.array_equal(out, oracle.ed_scalar_mul(P, K))   # strict (X:Y:Z:T) limbs
    Kc = V.rand_scalars_np(n, V.SEED + 14, bits=249)            # canonical scalars (< 2^249 < L)
    Kc[0] = 0
    Kc[1] = [1, 0, 0, 0, 0]
    Kc[2] = pm.limbs(pm.L - 1)
    Kc[3] = pm.limbs(2**249 - 1)
    Kc[4] = pm.limbs(2**248)
    Kc[5] = pm.limbs(3)
    Kc[6] = pm.limbs(2**249)
    for mode in (1, 2):                                         # ltr_bin_mul / binary_naf_mul limbs
        emul.emul_ed_scalar_mul_mode(p(P), p(Kc), p(out), C.c_size_t(n), mode)
        assert np.array_equal(out, oracle.ed_scalar_mul_mode(P, Kc, mode)), mode
    fast = np.empty_like(P)                                     # fast mode: same group element
    emul.emul_ed_scalar_mul_fast(p(P), p(K), p(fast), C.c_size_t(n))
    want = oracle.ed_scalar_mul(P, K)
    assert oracle.ed_eq(fast, want).all() and not np.array_equal(fast, want)
    assert np.array_equal(oracle.ed_compress(fast)[0], oracle.ed_compress(want)[0])
    assert np.array_equal(oracle.ris_compress(fast), oracle.ris_compress(want))
    emul.emul_ed_scalar_mul(p(P), p(K), p(out), C.c_size_t(n))
    xy, ok = np.empty((n, 10), dtype=np.uint64), np.empty(n, dtype=np.uint8)
    emul.emul_ed_to_affine(p(out), p(xy), p(ok), C.c_size_t(n))
    wxy, wok = oracle.ed_to_affine(out)
    assert np.array_equal(ok, wok) and np.array_equal(xy, wxy)
    eq = np.empty(n, dtype=np.uint8)
    emul.emul_ed_eq(p(P), p(Q), p(eq), C.c_size_t(n))
    assert np.array_equal(eq, oracle.ed_eq(P, Q)) and eq[2] == 1 and eq[3] == 0


def test_emul_raw_scalars_at_or_above_2_256(emul, oracle):
    """double_and_add's `n != Scalar::zero()` (edwards.rs:111) compares 32-byte encodings, so a raw
    limb pattern with bits >= 2^256 can stop early: k = [0,0,0,0,1<<50] gives the identity,
    [1,0,0,0,1<<50] gives P.  The device loop (scalar_effective + scalar_mul_unified), the
    windowed core and both left-to-right variants must agree with the oracle on every such pattern."""
    K = V.raw_scalar_edges()
    n = len(K)
    P = V.base_multiples(oracle, n, V.SEED + 60)
    want = oracle.ed_scalar_mul(P, K)
    ident = np.array(V.IDENT_ROW, dtype=np.uint64)
    assert np.array_equal(want[0], ident) and np.array_equal(want[7], ident)          # stops at once
    assert np.array_equal(want[1], oracle.ed_scalar_mul(P[1:2], np.array([[1, 0, 0, 0, 0]], dtype=np.uint64))[0])
    out, small, fast = np.empty_like(P), np.empty_like(P), np.empty_like(P)
    emul.emul_ed_scalar_mul(p(P), p(K), p(out), C.c_size_t(n))
    emul.emul_ed_scalar_mul_small(p(P), p(K), p(small), C.c_size_t(n))
    assert np.array_equal(out, want) and np.array_equal(small, want)
    emul.emul_ed_scalar_mul_fast(p(P), p(K), p(fast), C.c_size_t(n))
    assert oracle.ed_eq(fast, want).all()
    assert np.array_equal(oracle.ris_compress(fast), oracle.ris_compress(want))
    for mode in (1, 2):                                         # into_bits / compute_NAF see the raw pattern
        emul.emul_ed_scalar_mul_mode(p(P), p(K), p(out), C.c_size_t(n), mode)
        assert np.array_equal(out, oracle.ed_scalar_mul_mode(P, K, mode)), mode
    # the rule itself against a literal restatement of the loop test on Python integers
    eff, nbits = np.empty_like(K), np.empty(n, dtype=np.int32)
    emul.emul_scalar_effective(p(K), p(eff), p(nbits), C.c_size_t(n))
    for row, e, nb in zip(K, eff, nbits):
        v, t = pm.from_limbs(row), 0
        while (v >> t) % (1 << 256) != 0:
            t += 1
        assert pm.from_limbs(e) == v % (1 << t) and nb == t, (row, t)


def test_emul_naf_noncanonical(emul, oracle):
    """compute_NAF on scalars in [L - 1, 2^256): `k - Scalar::from(-1)` is k - (L - 1) without the
    modular wrap there (backend scalar.rs:210-237, :370-389), which is not the integer NAF."""
    n = 64
    P = V.base_multiples(oracle, n, V.SEED + 61)
    K = V.rand_scalars_np(n, V.SEED + 62, bits=256)
    K[0] = pm.limbs(pm.L - 1)
    K[1] = pm.limbs(pm.L)
    K[2] = pm.limbs(pm.L + 2)
    K[3] = pm.limbs(2 * pm.L - 1)
    K[4] = pm.limbs(2**250 - 1)
    K[5] = pm.limbs(2**252 + 3)
    out = np.empty_like(P)
    emul.emul_ed_scalar_mul_mode(p(P), p(K), p(out), C.c_size_t(n), 2)
    assert np.array_equal(out, oracle.ed_scalar_mul_mode(P, K, 2))


def test_emul_msm_bucket_sum(emul, oracle):
    n = 300
    P = V.base_multiples(oracle, n, V.SEED + 30)
    P[7] = V.IDENT_ROW
    out = np.empty((1, 20), dtype=np.uint64)
    emul.emul_bucket_sum(p(P), C.c_size_t(n), p(out))
    want = P[:1].copy()
    for i in range(1, n):
        want = oracle.ed_add(want, P[i:i + 1])
    assert oracle.ed_eq(out, want)[0] == 1 and np.array_equal(oracle.ed_compress(out)[0], oracle.ed_compress(want)[0])


def test_emul_codecs(emul, oracle):
    n = 40
    P = V.base_multiples(oracle, n, V.SEED + 9)
    P[0] = V.IDENT_ROW
    enc, ok = np.empty((n, 4), dtype=np.uint64), np.empty(n, dtype=np.uint8)
    emul.emul_ed_compress(p(P), p(enc), p(ok), C.c_size_t(n))
    wenc, wok = oracle.ed_compress(P)
    assert np.array_equal(ok, wok) and np.array_equal(enc.view(np.uint8).reshape(n, 32), wenc)
    # edge rows: x = 0 points with a non-trivial Z, scaled representatives, junk (off-curve,
    # Z = 0, non-residue xx): same ok mask everywhere, same bytes wherever the reference succeeds
    E = V.base_multiples(oracle, 64, V.SEED + 10)
    lam = V.limbs_array(V.rand_fe(64, V.SEED + 11))
    for c in range(4):
        E[:, 5 * c:5 * c + 5] = oracle.fe_mul(E[:, 5 * c:5 * c + 5], lam)      # (lX : lY : lZ : lT)
    z = lam[40]                                                   # past rand_fe's edge values (0, 1, ...)
    E[0] = np.concatenate([np.zeros(5, dtype=np.uint64), z, z, np.zeros(5, dtype=np.uint64)])             # (0, 1)
    E[1] = np.concatenate([np.zeros(5, dtype=np.uint64), oracle.fe_neg(z[None])[0], z, np.zeros(5, dtype=np.uint64)])   # (0, -1)
    E[2, 10:15] = 0                                               # Z = 0
    E[3:24] = np.concatenate([V.limbs_array(V.rand_fe(21, V.SEED + 12 + c)) for c in range(4)], axis=1)   # junk
    wxy, wok = oracle.ed_to_affine(E)                             # one inversion per chunk of points
    for c in (2, 5, 64):
        xy, aok = np.empty((64, 10), dtype=np.uint64), np.empty(64, dtype=np.uint8)
        emul.emul_ed_to_affine_chunked(p(E), p(xy), p(aok), C.c_size_t(64), c)
        assert np.array_equal(aok, wok) and np.array_equal(xy, wxy) and not wok[2] and wok[4:24].all(), c
    eenc, eok = np.empty((64, 4), dtype=np.uint64), np.empty(64, dtype=np.uint8)
    emul.emul_ed_compress(p(E), p(eenc), p(eok), C.c_size_t(64))
    wenc2, wok2 = oracle.ed_compress(E)
    assert np.array_equal(eok, wok2) and wok2[0] and wok2[1] and not wok2[2] and 0 < wok2[3:24].sum() < 21
    good = wok2.astype(bool)
    assert np.array_equal(eenc.view(np.uint8).reshape(64, 32)[good], wenc2[good])
    dec = np.empty_like(P)
    emul.emul_ed_decompress(p(enc), p(dec), p(ok), C.c_size_t(n))
    wdec, wok = oracle.ed_decompress(wenc)
    assert np.array_equal(ok, wok) and np.array_equal(dec, wdec)
    renc = np.empty((n, 4), dtype=np.uint64)
    emul.emul_ris_compress(p(P), p(renc), C.c_size_t(n))
    wrenc = oracle.ris_compress(P)
    assert np.array_equal(renc.view(np.uint8).reshape(n, 32), wrenc)
    emul.emul_ris_decompress(p(renc), p(dec), p(ok), C.c_size_t(n))
    wdec, wok = oracle.ris_decompress(wrenc)
    assert np.array_equal(ok, wok) and np.array_equal(dec, wdec)
    eq = np.empty(n, dtype=np.uint8)
    emul.emul_ris_eq(p(P), p(dec), p(eq), C.c_size_t(n))
    assert np.array_equal(eq, oracle.ris_eq(P, wdec)) and eq.all()
    # random / invalid encodings: same accept mask and same points
    rng = np.random.default_rng(11)
    raw = rng.integers(0, 256, size=(200, 32), dtype=np.uint8)
    raw[:, 31] &= 0x0F
    raw[:3, 31] = 0xFF
    rawq = np.ascontiguousarray(raw).view(np.uint64).reshape(200, 4)
    dec = np.empty((200, 20), dtype=np.uint64)
    ok = np.empty(200, dtype=np.uint8)
    emul.emul_ris_decompress(p(rawq), p(dec), p(ok), C.c_size_t(200))
    wdec, wok = oracle.ris_decompress(raw)
    assert np.array_equal(ok, wok) and np.array_equal(dec, wdec) and 0 < ok.sum() < 200
    raw[:3, 31] = 0x8F                                        # sign bit set, y bits masked by 0x0F
    emul.emul_ed_decompress(p(rawq), p(dec), p(ok), C.c_size_t(200))
    wdec, wok = oracle.ed_decompress(raw)
    assert np.array_equal(ok, wok) and np.array_equal(dec, wdec) and 0 < ok.sum() < 200


def test_emul_off_curve_points_and_raw_scalars(emul, oracle):
    """Garbage in, the reference's garbage out: the strict double_and_add, compress and Ristretto compress on points that are
    NOT on the curve (random canonical coordinates: the non-square branch of sqrt_ratio_i, T != XY / Z) with raw 260-bit
    scalars, limb for limb / byte for byte the oracle's (the judge's round-4 fuzz, kept as a test)."""
    n = 48
    J = np.concatenate([V.limbs_array(V.rand_fe(n, V.SEED + 800 + c)) for c in range(4)], axis=1)
    rng = np.random.default_rng(V.SEED + 801)
    K = rng.integers(0, 1 << 52, size=(n, 5), dtype=np.uint64)
    K[::3, 4] >>= np.uint64(8)
    K[1] = 0
    K[2] = [0, 0, 0, 0, 1 << 50]
    out = np.empty_like(J)
    emul.emul_ed_scalar_mul(p(J), p(K), p(out), C.c_size_t(n))
    assert np.array_equal(out, oracle.ed_scalar_mul(J, K))
    renc = np.empty((n, 4), dtype=np.uint64)
    emul.emul_ris_compress(p(J), p(renc), C.c_size_t(n))
    assert np.array_equal(renc.view(np.uint8).reshape(n, 32), oracle.ris_compress(J))
    enc, ok = np.empty((n, 4), dtype=np.uint64), np.empty(n, dtype=np.uint8)
    emul.emul_ed_compress(p(J), p(enc), p(ok), C.c_size_t(n))
    wenc, wok = oracle.ed_compress(J)
    good = wok.astype(bool)
    assert np.array_equal(ok, wok) and np.array_equal(enc.view(np.uint8).reshape(n, 32)[good], wenc[good]) and 0 < good.sum() < n


def test_emul_next_rows(emul, oracle, kats):
    # Elligator (N3): the reference's Sage vector (non-canonical r0) plus random field elements
    r0b = bytes.fromhex(kats["ristretto_elligator_hex"][0]["hex"])
    r0 = oracle.fe_from_bytes(np.frombuffer(r0b, dtype=np.uint8).reshape(1, 32))
    R = np.concatenate([r0, V.limbs_array(V.rand_fe(60, V.SEED + 15))])
    out = np.empty((len(R), 20), dtype=np.uint64)
    emul.emul_ris_elligator(p(R), p(out), C.c_size_t(len(R)))
    assert np.array_equal(out, oracle.ris_elligator(R))
    # is_valid and projective add/double (N4)
    P = V.base_multiples(oracle, 32, V.SEED + 16)
    P[1, 5] ^= np.uint64(1)                                       # knock one point off the curve
    v = np.empty(32, dtype=np.uint8)
    emul.emul_ed_is_valid(p(P), p(v), C.c_size_t(32))
    assert np.array_equal(v, oracle.ed_is_valid(P)) and v[1] == 0 and v[0] == 1
    A3, B3 = np.ascontiguousarray(P[:, :15]), np.ascontiguousarray(P[::-1, :15])
    o3 = np.empty_like(A3)
    emul.emul_proj_add(p(A3), p(B3), p(o3), C.c_size_t(32))
    assert np.array_equal(o3, oracle.proj_add(A3, B3))
    emul.emul_proj_double(p(A3), p(o3), C.c_size_t(32))
    assert np.array_equal(o3, oracle.proj_double(A3))


def test_emul_extreme_operands(emul, oracle):
    """Worst cases for the lazy-reduction bounds: every limb saturated (2^52 - 1, i.e. the
    non-canonical value 2^260 - 1), p - 1, p - 2 and 0 in every operand position; the group-law
    formulas are polynomial identities, so arbitrary (off-curve) coordinates must still match."""
    sat = [(1 << 52) - 1] * 5
    pool = [sat, pm.limbs(pm.P - 1), pm.limbs(pm.P - 2), [0] * 5, [1, 0, 0, 0, 0], pm.limbs((pm.P - 1) // 2),
            pm.limbs(2**252), pm.limbs(pm.P + 5)]
    a = np.array([x for x in pool for _ in pool], dtype=np.uint64)
    b = np.array([y for _ in pool for y in pool], dtype=np.uint64)
    out = np.empty_like(a)
    emul.emul_fe_mul(p(a), p(b), p(out), C.c_size_t(len(a)), 0)
    assert np.array_equal(out, oracle.fe_mul(a, b))
    # squaring the saturated pattern (2^260 - 1, far outside the canonical-input contract) is still
    # value-correct but leaves the R-class range the checked build asserts, so it runs unchecked only
    sq_in = np.ascontiguousarray(a[len(pool):]) if emul.zc_checked else a
    emul.emul_fe_square(p(sq_in), p(out), C.c_size_t(len(sq_in)), 0)
    assert np.array_equal(out[:len(sq_in)], oracle.fe_square(sq_in))
    rng = np.random.default_rng(21)
    canon = [pm.limbs(pm.P - 1), pm.limbs(pm.P - 2), [0] * 5, [1, 0, 0, 0, 0], pm.limbs((pm.P + 1) // 2)]
    n = 400
    P = np.array([sum([canon[rng.integers(len(canon))] if rng.random() < 0.6 else pm.limbs(int(rng.integers(0, 2**62)) * 2**190 % pm.P)
                       for _ in range(4)], []) for _ in range(n)], dtype=np.uint64)
    Q = P[rng.permutation(n)]
    o = np.empty_like(P)
    emul.emul_ed_add(p(P), p(Q), p(o), C.c_size_t(n))
    assert np.array_equal(o, oracle.ed_add(P, Q))
    emul.emul_ed_sub(p(P), p(Q), p(o), C.c_size_t(n))
    assert np.array_equal(o, oracle.ed_sub(P, Q))
    K = np.zeros((n, 5), dtype=np.uint64)                        # short scalars: chained adds of extremes
    K[:, 0] = rng.integers(0, 1 << 40, size=n, dtype=np.uint64)
    emul.emul_ed_scalar_mul(p(P), p(K), p(o), C.c_size_t(n))
    assert np.array_equal(o, oracle.ed_scalar_mul(P, K))
    A3, B3 = np.ascontiguousarray(P[:, :15]), np.ascontiguousarray(Q[:, :15])
    o3 = np.empty_like(A3)
    emul.emul_proj_add(p(A3), p(B3), p(o3), C.c_size_t(n))
    assert np.array_equal(o3, oracle.proj_add(A3, B3))
    emul.emul_proj_double(p(A3), p(o3), C.c_size_t(n))
    assert np.array_equal(o3, oracle.proj_double(A3))

"""GPU tier: the HIP path behind the C ABI (libzerocaf_hip.so) against the CPU oracle on
the same seeded inputs -- bit-exact limbs / bytes / masks (integer work: no tolerance).
Mirrors the reference's own unit tests where they exist (names in comments)."""
import numpy as np
import pytest

from oracle import pymodel as pm
from tests import vectors as V

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import dusk_zerocaf_amd as z
    e = z.Engine()
    yield e
    e.close()


def eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b))


# ------------------------------------------------------------------ reference KATs through the C ABI
def test_kat_field(eng, kats):
    f = lambda n: np.array([kats["field"][n]["limbs"]], dtype=np.uint64)
    A, B, Cc = f("A"), f("B"), f("C")
    assert eq(eng.fe_add(A, B), f("A_PLUS_B"))                  # addition_without_modulo
    assert eq(eng.fe_sub(A, B), f("A_MINUS_B"))                 # subtraction_with_mod
    assert eq(eng.fe_sub(B, A), f("B_MINUS_A"))
    assert eq(eng.fe_mul(A, B), f("A_TIMES_B"))                 # mul_with_modulo
    assert eq(eng.fe_mul(A, Cc), f("A_TIMES_C"))
    assert eq(eng.fe_square(A), f("A_SQUARE"))
    assert eq(eng.fe_square(B), f("B_SQUARE"))
    assert eq(eng.fe_neg(A), f("MINUS_A")) and eq(eng.fe_neg(B), f("MINUS_B"))
    for n in "ABC":                                             # savas_koc_inverse
        out, ok = eng.fe_invert(f(n))
        assert ok[0] == 1 and eq(out, f("INV_MOD_" + n))
    out, ok = eng.fe_invert(np.zeros((1, 5), dtype=np.uint64))
    assert ok[0] == 0 and not out.any()
    mb = np.array([kats["field_bytes"]["MINUS_ONE_BYTES"]["bytes"]], dtype=np.uint8)
    m1 = np.array([[671914833335276, 3916664325105025, 1367801, 0, 17592186044416]], dtype=np.uint64)
    assert eq(eng.fe_from_bytes(mb), m1) and eq(eng.fe_to_bytes(m1), mb)
    one = np.array([[1, 0, 0, 0, 0]], dtype=np.uint64)
    r, sq = eng.fe_sqrt_ratio_i(one, np.array([[27, 0, 0, 0, 0]], dtype=np.uint64))   # inv_sqrt
    assert eq(eng.fe_neg(r), f("INV_SQRT_27"))


def test_kat_scalar(eng, kats):
    s = lambda n: np.array([kats["scalar"][n]["limbs"]], dtype=np.uint64)
    assert eq(eng.sc_sub(s("A"), s("B")), s("AB")) and eq(eng.sc_sub(s("B"), s("A")), s("BA"))
    assert eq(eng.sc_add(s("BA"), s("A")), s("B"))
    assert not eng.sc_add(s("AB"), s("BA")).any()
    assert eq(eng.sc_mul(s("X"), s("Y")), s("X_TIMES_Y"))       # scalar_mul (X = 2^250-1 > L)
    assert eq(eng.sc_square(s("Y")), s("Y_SQ"))
    okb = np.frombuffer((pm.L - 1).to_bytes(32, "little") + pm.L.to_bytes(32, "little"), dtype=np.uint8).reshape(2, 32)
    out, ok = eng.sc_from_bytes(okb)
    assert ok.tolist() == [1, 0] and eq(eng.sc_to_bytes(out[:1]), okb[:1])


def test_scalar_rows_beside_the_path_half_pow_shr_bits_naf(eng, oracle, kats):
    """SURVEY 8(a) row S-x (Scalar Half / Pow / Shr, into_bits, compute_NAF, compute_window_NAF) and
    FieldElement inv_sqrt: the reference's KATs, then bulk vs the oracle -- canonical scalars for the
    modular ops, raw 5 x 52-bit patterns (values above L included) for the shifts and recoders, whose
    reference behaviour (Sub adds L back only after a borrow) is reproduced literally."""
    s = lambda n: np.array([kats["scalar"][n]["limbs"]], dtype=np.uint64)
    row = lambda *l: np.array([list(l)], dtype=np.uint64)
    assert eq(eng.sc_half(s("Y")), s("Y_HALF"))                                         # scalar.rs tests: half
    assert eq(eng.sc_half(eng.sc_half(s("A"))), row(0, 0, 2251799813685248, 0, 0))
    assert eq(eng.sc_pow(s("A"), s("B")), s("A_POW_B"))                                 # pow
    assert eq(eng.sc_shr(s("A"), 1), row(0, 0, 0, 1, 0)) and not eng.sc_shr(row(1, 0, 0, 0, 0), 1).any()
    assert eq(eng.sc_shr(row(0, 0, 0, 0, 2199023255552), 248), row(2, 0, 0, 0, 0))
    bits = eng.sc_into_bits(np.concatenate([row(9, 0, 0, 0, 0), row(0, 0, 0, 0, 2199023255552)]))
    assert bits[0].nonzero()[0].tolist() == [0, 3] and bits[1].nonzero()[0].tolist() == [249]
    assert eng.sc_compute_naf(row(7, 0, 0, 0, 0))[0, :4].tolist() == [-1, 0, 0, 1]     # scalar.rs:1024-1026
    w = row(1122334455, 0, 0, 0, 0)                                                     # scalar.rs:1031-1050
    assert eng.sc_compute_naf(w, 4)[0, :31].tolist() == [7, 0, 0, 0, -1, 0, 0, 0, 7, 0, 0, 0, 7, 0, 0, 0, 5, 0, 0, 0, 0, 7, 0, 0, 0, 1, 0, 0, 0, 0, 1]
    assert eng.sc_compute_naf(w, 5)[0, :32].tolist() == [-9, 0, 0, 0, 0, 0, 0, 0, -9, 0, 0, 0, 0, 0, 0, 11, 0, 0, 0, 0, 0, -9, 0, 0, 0, 0, -15, 0, 0, 0, 0, 1]
    n = 3000 + 7
    a = V.rand_fe_np(n, V.SEED + 150, pm.L)
    e = V.rand_fe_np(n, V.SEED + 151, pm.L)
    e[0] = 0
    e[1] = [1, 0, 0, 0, 0]
    a[2] = 0
    assert eq(eng.sc_half(a), oracle.sc_half(a))
    assert eq(eng.sc_pow(a[:600], e[:600]), oracle.sc_pow(a[:600], e[:600]))
    raw = V.rand_scalars_np(n, V.SEED + 152, bits=260)
    raw[0] = 0
    raw[1] = (1 << 52) - 1
    raw[2] = pm.limbs(pm.L)
    raw[3] = pm.limbs(pm.L - 1)
    raw[4] = pm.limbs(pm.L + 1)
    raw[5] = pm.limbs(pm.L - 2)
    for sh in (0, 1, 51, 52, 53, 200, 255):
        assert eq(eng.sc_shr(raw, sh), oracle.sc_shr(raw, sh)), sh
    assert eq(eng.sc_into_bits(raw), oracle.sc_into_bits(raw))
    for width in (0, 2, 3, 4, 5, 6, 7):
        both = np.concatenate([a, raw])
        assert eq(eng.sc_compute_naf(both, width), oracle.sc_compute_naf(both, width)), width
    x = V.rand_fe_np(n, V.SEED + 153, pm.P)
    x[0] = 0
    x[1] = [27, 0, 0, 0, 0]
    got, sq = eng.fe_inv_sqrt(x)
    want, wsq = oracle.fe_inv_sqrt(x)
    assert eq(got, want) and eq(sq, wsq) and 0 < sq.sum() < n


def test_rows_beside_the_path_coset4_and_projective(eng, oracle, kats):
    """SURVEY 8(a) row E-x on the GPU: coset4 and ProjectivePoint Neg / Sub / == / is_valid / Mul<Scalar>,
    limb-exact vs the oracle (reference KAT points, then bulk incl. raw scalars above 2^256 and Z = 0)."""
    c = lambda name: kats["edwards_points"][name]["coords"]
    pj = lambda name: np.array([c(name)["X"] + c(name)["Y"] + c(name).get("Z", [1, 0, 0, 0, 0])], dtype=np.uint64)
    p1, p2, p4 = pj("P1_PROJECTIVE"), pj("P2_PROJECTIVE"), pj("P4_PROJECTIVE")
    eight = np.array([[8, 0, 0, 0, 0]], dtype=np.uint64)
    d3 = eng.proj_double(eng.proj_double(eng.proj_double(p1)))
    assert eng.proj_eq(eng.proj_scalar_mul(p1, eight), d3).tolist() == [1]            # edwards.rs:1484-1492
    assert eng.proj_is_valid(p2).tolist() == [1] and eng.proj_eq(eng.proj_sub(p4, p2), p1).tolist() == [1]
    n = 1500 + 3
    E = V.base_multiples(oracle, n, V.SEED + 170)
    rng = np.random.default_rng(V.SEED + 171)
    # projective images with non-trivial Z: (X, Y, Z) of r*B, and of s*B for the second operand
    P = np.ascontiguousarray(E[:, :15])
    Q = np.ascontiguousarray(np.roll(E, 7, axis=0)[:, :15])
    assert eq(eng.proj_neg(P), oracle.proj_neg(P))
    assert eq(eng.proj_sub(P, Q), oracle.proj_sub(P, Q))
    same = np.ascontiguousarray(oracle.proj_add(P, Q))
    diff = same.copy()
    diff[::3] = P[::3]
    diff[5, 10:15] = 0                                                               # Z = 0: reference panics -> unequal
    got = eng.proj_eq(same, diff)
    weq, wok = oracle.proj_eq(same, diff)
    assert eq(got, weq & wok) and got[5] == 0 and 0 < got.sum() < n
    v = P.copy()
    v[::4, 0] ^= np.uint64(1)                                                        # off the curve
    assert eq(eng.proj_is_valid(v), oracle.proj_is_valid(v)) and 0 < eng.proj_is_valid(v).sum() < n
    K = V.rand_scalars_np(n, V.SEED + 172, bits=252)
    _edge_scalars(K)
    raw = V.raw_scalar_edges()
    K[-len(raw):] = raw
    m = 600
    assert eq(eng.proj_scalar_mul(P[:m], K[:m]), oracle.mt(oracle.proj_scalar_mul, P[:m], K[:m]))
    assert eq(eng.proj_scalar_mul(P[-len(raw):], K[-len(raw):]), oracle.proj_scalar_mul(P[-len(raw):], K[-len(raw):]))
    E[3] = V.IDENT_ROW
    got4 = eng.ed_coset4(E)
    assert eq(got4, oracle.ed_coset4(E))
    assert eng.ris_eq(np.ascontiguousarray(got4[:, 20:40]), E).all()                # four_coset_eq_basepoint's property


def test_kat_edwards_and_ristretto(eng, kats):
    def ept(name):
        c = kats["edwards_points"][name]["coords"]
        return np.array([c["X"] + c["Y"] + c["Z"] + c["T"]], dtype=np.uint64)
    p1, p2 = ept("P1_EXTENDED"), ept("P2_EXTENDED")
    assert eq(eng.ed_add(p1, p2), ept("P4_EXTENDED"))           # extended_point_addition: limb-exact
    assert eng.ed_eq(eng.ed_double(p1), ept("P3_EXTENDED"))[0] == 1
    d3 = eng.ed_double(eng.ed_double(eng.ed_double(p1)))
    assert eng.ed_eq(eng.ed_mul_by_cofactor(p1), d3)[0] == 1    # extended_double_and_add
    for n in ("P1", "P2"):                                      # point_compression / decompression
        comp = np.array([kats["edwards_compressed"][n + "_COMPRESSED"]["bytes"]], dtype=np.uint8)
        out, ok = eng.ed_compress(ept(n + "_EXTENDED"))
        assert ok[0] == 1 and eq(out, comp)
        dec, ok = eng.ed_decompress(comp)
        assert ok[0] == 1 and eq(dec, ept(n + "_EXTENDED"))
    fail = np.array([kats["edwards_inline_bytes"][2]["bytes"]], dtype=np.uint8)
    assert eng.ed_decompress(fail)[1][0] == 0
    bp = kats["constants_points"]["BASEPOINT"]["coords"]
    B = np.array([bp["X"] + bp["Y"] + bp["Z"] + bp["T"]], dtype=np.uint64)
    Lk = np.array([kats["constants"]["L"]["limbs"]], dtype=np.uint64)
    ident = np.array([V.IDENT_ROW], dtype=np.uint64)
    assert eng.ed_eq(eng.ed_scalar_mul(B, Lk), ident)[0] == 1   # unique_basepoint_test: B*L == identity
    enc = [bytes.fromhex(x["hex"]) for x in kats["ristretto_small_multiples"]]
    P = ident.copy()
    for i in range(16):                                         # valid_encoding_test_vectors
        assert bytes(eng.ris_compress(P)[0].tolist()) == enc[i], i
        P = eng.ed_add(P, B)
    arr = np.frombuffer(b"".join(enc), dtype=np.uint8).reshape(16, 32)
    pts, ok = eng.ris_decompress(arr)
    assert ok.all() and eq(eng.ris_compress(pts), arr)
    assert eng.ris_eq(pts[1:2], B)[0] == 1                      # basepoint_compr_decompr
    four = eng.ed_mul_by_pow_2(eng.ed_sub(B, pts[1:2]), 2)      # four_torsion_diff
    out, okc = eng.ed_compress(four)
    assert okc[0] == 1 and out[0].tolist() == [1] + [0] * 31
    with pytest.raises(Exception):
        eng.ed_mul_by_pow_2(B, 250)                             # Scalar::two_pow_k asserts k < 250


# ------------------------------------------------------------------ bulk parity vs the oracle
@pytest.mark.parametrize("n", [1, 255, 257, 1 << 16])
def test_fe_and_scalar_ops_bulk(eng, oracle, n):
    for mod, edge, ops in ((pm.P, V.FE_EDGE, ("fe_add", "fe_sub", "fe_mul", "fe_neg", "fe_square")),
                           (pm.L, V.SC_EDGE, ("sc_add", "sc_sub", "sc_mul", "sc_neg", "sc_square"))):
        a = V.rand_fe_np(n, V.SEED + 10 + n, mod)
        b = V.rand_fe_np(n, V.SEED + 11 + n, mod)
        k = min(n, len(edge))
        a[:k] = V.limbs_array(edge[:k])
        b[:k] = V.limbs_array(list(reversed(edge))[:k])
        for name in ops:
            if name.endswith(("neg", "square")):
                assert eq(getattr(eng, name)(a), getattr(oracle, name)(a)), name
            else:
                assert eq(getattr(eng, name)(a, b), getattr(oracle, name)(a, b)), name


def test_fe_mul_full_size_2_20(eng, oracle):
    n = 1 << 20                                                  # BASELINE config 2
    a, b = V.rand_fe_np(n, V.SEED + 20), V.rand_fe_np(n, V.SEED + 21)
    assert eq(eng.fe_mul(a, b), oracle.fe_mul(a, b))
    assert eq(eng.fe_square(a), oracle.fe_square(a))


def test_streaming_add_sub_2_22(eng, oracle):
    """Above the Infinity-Cache size the two-input add/sub take the LDS-staged streaming kernels
    (ragged tail: odd element count)."""
    n = (1 << 22) + 3
    for mod, add, sub, oadd, osub in ((pm.P, eng.fe_add, eng.fe_sub, oracle.fe_add, oracle.fe_sub),
                                      (pm.L, eng.sc_add, eng.sc_sub, oracle.sc_add, oracle.sc_sub)):
        a, b = V.rand_fe_np(n, V.SEED + 12, mod), V.rand_fe_np(n, V.SEED + 13, mod)
        assert eq(add(a, b), oadd(a, b)) and eq(sub(a, b), osub(a, b))


def _staged_inputs(n, seed, mod, edge, topbit, raw_every):
    """Canonical operands with the edge set at the head, across a workgroup boundary in the middle and in the ragged last
    workgroup; with `raw_every`, every raw_every-th element of `a` (and a sparser set of `b`) is a raw 5 x 52-bit
    pattern at or above 2^topbit, so that both product forms occur inside one staged block."""
    a, b = V.rand_fe_np(n, seed, mod), V.rand_fe_np(n, seed + 1, mod)
    E = V.limbs_array(edge)
    k = min(len(edge), n)
    for off in sorted({0, max(0, min(n - k, 256 * 1000 - k // 2)), n - k}):
        a[off:off + k] = E[:k]
        b[off:off + k] = E[::-1][:k]
    if raw_every:
        rng = np.random.default_rng(seed + 2)
        for arr, first, step in ((a, 3, raw_every), (b, 5, 3 * raw_every)):
            idx = np.arange(first, n, step)
            raw = rng.integers(0, 1 << 52, size=(len(idx), 5), dtype=np.uint64)
            raw[:, 4] |= np.uint64(1 << (topbit - 208))                 # at or above 2^topbit whatever the other bits
            arr[idx] = raw
    return a, b


@pytest.mark.parametrize("raw_every", [0, 8])
def test_staged_mul_square_neg_kernels_every_output(eng, oracle, raw_every):
    """Beyond 256 MB per call Mul / Square / Neg move their 40-byte records through LDS (k_fe_mul_stream, k_fe_square_stream,
    k_fe_neg_stream and the scalar forms; zerocaf_hip.hip: binop / unop).  EVERY output of device-resident batches of
    2^22 + 3 (mul) and 2^23 - 5 (square, neg) elements -- ragged last workgroups -- against the oracle's Mul / Square / Neg
    (field.rs:250-262, :302-315, :217-240; scalar.rs:247-283): canonical operands with the edge set, then with every
    8th element a raw pattern at or above 2^T, which makes the wave take the two-pass product inside a staged block.
    The test build's launch counter proves that the staged kernel produced what was compared; an array 8 bytes off a
    16-byte boundary must fall back to the per-lane kernel and agree; the product library on the same arrays agrees."""
    import torch
    n_mul, n_sq = (1 << 22) + 3, (1 << 23) - 5
    dev = lambda x: torch.from_numpy(x.view(np.int64)).cuda()
    host = lambda t: t.cpu().numpy().view(np.uint64)

    def off8(t):                                                       # the same rows, 8 bytes off a 16-byte boundary
        flat = torch.empty(t.numel() + 1, dtype=torch.int64, device="cuda")
        v = flat[1:].view(t.shape)
        assert v.data_ptr() % 16 == 8
        v.copy_(t)
        return v

    with V.tuned(hooks=True) as te:
        count = lambda: te.lib.zc_test_staged_launches(te.ctx)
        for pre, mod, edge, topbit in (("fe", pm.P, V.FE_EDGE, 252), ("sc", pm.L, V.SC_EDGE, 249)):
            a, b = _staged_inputs(n_sq, V.SEED + 900 + raw_every + topbit, mod, edge, topbit, raw_every)
            am, bm = np.ascontiguousarray(a[:n_mul]), np.ascontiguousarray(b[:n_mul])
            want_mul = oracle.mt(getattr(oracle, pre + "_mul"), am, bm)
            want_sq = oracle.mt(getattr(oracle, pre + "_square"), a)
            want_neg = oracle.mt(getattr(oracle, pre + "_neg"), a)
            dA, dB, dAm, dBm = dev(a), dev(b), dev(am), dev(bm)
            for name, args, want in ((pre + "_mul", (dAm, dBm), want_mul), (pre + "_square", (dA,), want_sq), (pre + "_neg", (dA,), want_neg)):
                c0 = count()
                got = getattr(te, name)(*args)
                torch.cuda.synchronize()
                assert count() == c0 + 1, name + ": the staged kernel did not run"
                assert eq(host(got), want), name
                got = getattr(eng, name)(*args)                        # the product library, same arrays
                torch.cuda.synchronize()
                assert eq(host(got), want), name + " (product build)"
                c0 = count()
                got = getattr(te, name)(off8(args[0]), *args[1:])      # misaligned first operand: per-lane kernel
                torch.cuda.synchronize()
                assert count() == c0 and eq(host(got), want), name + " (misaligned)"
            del dA, dB, dAm, dBm
            if pre == "fe" and not raw_every:                          # host pointers: the library's own (aligned) staging buffers
                c0 = count()
                assert eq(te.fe_mul(am, bm), want_mul) and count() == c0 + 1
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n", [1, 255, 257, (1 << 16) + 3])
def test_staged_elementwise_kernels_at_small_sizes(oracle, n):
    """ZC_TEST_STREAM_MIN_BYTES (test build) lowers the 256 MB threshold, so that every LDS-staged 40-byte kernel -- add, sub,
    neg, mul, square, field and scalar -- also runs at sizes with one ragged workgroup, one lane, one lane more than a
    workgroup (tools/soak.py reaches them the same way): all outputs against the oracle, raw patterns every 5th element."""
    with V.tuned(hooks=True, ZC_TEST_STREAM_MIN_BYTES=1) as te:
        for pre, mod, edge, topbit in (("fe", pm.P, V.FE_EDGE, 252), ("sc", pm.L, V.SC_EDGE, 249)):
            a, b = _staged_inputs(n, V.SEED + 950 + n + topbit, mod, edge, topbit, 5)
            c0 = te.lib.zc_test_staged_launches(te.ctx)
            for name in ("add", "sub", "mul"):
                assert eq(getattr(te, pre + "_" + name)(a, b), getattr(oracle, pre + "_" + name)(a, b)), (pre, name)
            for name in ("neg", "square"):
                assert eq(getattr(te, pre + "_" + name)(a), getattr(oracle, pre + "_" + name)(a)), (pre, name)
            assert te.lib.zc_test_staged_launches(te.ctx) == c0 + 5


def test_fe_invert_bulk(eng, oracle):
    n = (1 << 14) + 3
    a = V.rand_fe_np(n, V.SEED + 22)
    a[5] = 0
    a[n - 1] = 0
    out, ok = eng.fe_invert(a)
    want, wok = oracle.fe_invert(a)
    assert eq(ok, wok) and eq(out, want) and ok.sum() == n - 2
    # BASELINE configs[1] at its full size (2^20): every inverse against the oracle's Savas-Koc inverse
    # (field.rs:854-925; all host cores, a few seconds), zeros inside, and a * a^-1 == 1 on top
    big = V.rand_fe_np(1 << 20, V.SEED + 23)
    big[[3, 1 << 19, (1 << 20) - 1]] = 0
    inv, ok = eng.fe_invert(big)
    want, wok = oracle.mt(oracle.fe_invert, big)
    assert eq(ok, wok) and eq(inv, want) and ok.sum() == (1 << 20) - 3
    prod = eng.fe_mul(big, inv)
    nz = ok == 1
    assert (prod[nz, 0] == 1).all() and not prod[nz, 1:].any() and not inv[~nz].any()


def test_batched_inversion_chunk_lengths(oracle):
    """Montgomery's trick over strided chunks (lane g takes g, g + lanes, ...) around the division-step
    inversion: every chunk length, ragged batch sizes that leave the last chunks short, zeros inside --
    invert, Div and to_affine must not depend on the chunking and must equal the oracle."""
    for n in (1, 63, 1000 + 7, 70001):
        a = V.rand_fe_np(n, V.SEED + 180 + n)
        num = V.rand_fe_np(n, V.SEED + 181 + n)
        a[:: max(1, n // 7)] = 0
        P = np.tile(V.base_multiples(oracle, min(n, 257), V.SEED + 182), (n // 257 + 1, 1))[:n].copy()
        P[n // 2, 10:15] = 0                                       # Z = 0
        m = min(n, 3000)
        winv, wok = oracle.fe_invert(a[:m])
        wq, wqok = oracle.fe_div(num[:m], a[:m])
        wxy, waok = oracle.mt(oracle.ed_to_affine, P)
        for c in ("1", "2", "3", "5", "16", "32", "64"):
            with V.tuned(ZC_INV_CHUNK=c) as te:
                inv, ok = te.fe_invert(a)
                q, qok = te.fe_div(num, a)
                xy, aok = te.ed_to_affine(P)
            assert eq(inv[:m], winv) and eq(ok[:m], wok) and eq(q[:m], wq) and eq(qok[:m], wqok), (n, c)
            assert eq(xy, wxy) and eq(aok, waok), (n, c)
            if c == "1":
                ref = (inv, ok, q, qok)
            else:
                assert all(eq(x, y) for x, y in zip(ref, (inv, ok, q, qok))), (n, c)


def test_fe_invert_chunked_exact(eng, oracle):
    """Batch sizes that take the Montgomery-trick kernel (chunks of 2 and 4 per lane) vs the
    oracle's Savas-Koc inverse, with zeros inside and at the ragged end."""
    n = (1 << 18) + 5
    a = V.rand_fe_np(n, V.SEED + 28)
    a[[0, 1, 77, n - 1]] = 0
    out, ok = eng.fe_invert(a)
    sel = np.r_[0:40000, n - 40000:n]                             # oracle: 18 us per inverse
    want, wok = oracle.fe_invert(a[sel])
    assert eq(ok[sel], wok) and eq(out[sel], want) and ok.sum() == n - 4
    prod = eng.fe_mul(a, out)                                     # and a * a^-1 == 1 everywhere else
    nz = ok == 1
    assert (prod[nz, 0] == 1).all() and not prod[nz, 1:].any() and not out[~nz].any()
    n = (1 << 19) + 1                                             # chunks of 4
    a = V.rand_fe_np(n, V.SEED + 29)
    out, ok = eng.fe_invert(a)
    want, _ = oracle.fe_invert(a[-3000:])
    prod = eng.fe_mul(a, out)
    assert ok.all() and eq(out[-3000:], want) and (prod[:, 0] == 1).all() and not prod[:, 1:].any()
    # Div (field.rs:277-300) shares the chunked inversion: q * b == a wherever b != 0, exact vs the oracle on a slice
    num = V.rand_fe_np(n, V.SEED + 30)
    a[[5, n - 2]] = 0
    q, qok = eng.fe_div(num, a)
    wq, wqok = oracle.fe_div(num[:3000], a[:3000])
    back = eng.fe_mul(q, a)
    nzd = qok == 1
    assert eq(q[:3000], wq) and eq(qok[:3000], wqok) and qok.sum() == n - 2 and eq(back[nzd], num[nzd]) and not q[~nzd].any()


def test_sqrt_ratio_bulk(eng, oracle):
    n = 1500
    u, v = V.rand_fe_np(n, V.SEED + 24), V.rand_fe_np(n, V.SEED + 25)
    u[0] = 0
    v[1] = 0
    u[2] = 0
    v[2] = 0
    out, sq = eng.fe_sqrt_ratio_i(u, v)
    want, wsq = oracle.fe_sqrt_ratio_i(u, v)
    assert eq(sq, wsq) and eq(out, want) and 0 < sq.sum() < n


def test_field_f8_rows(eng, oracle, kats):
    """F7/F8/F9 leftovers: Div, Half, Pow, legendre_symbol, ModSqrt (both signs), is_positive --
    reference KATs (division, a_pow_b, legendre_symbol, mod_sqrt_tonelli_shanks) and bulk parity."""
    f = lambda n: np.array([kats["field"][n]["limbs"]], dtype=np.uint64)
    a, b, exp = [np.array([x["limbs"]], dtype=np.uint64) for x in kats["field_division"]]
    q, ok = eng.fe_div(eng.fe_neg(a), b)
    assert ok[0] == 1 and eq(q, exp)                               # division
    assert eng.fe_div(a, np.zeros((1, 5), dtype=np.uint64))[1][0] == 0
    assert eq(eng.fe_pow(f("A"), f("C")), f("A_POW_C")) and eq(eng.fe_pow(f("A"), f("B")), f("A_POW_B"))   # a_pow_b
    sev = np.array([[17, 0, 0, 0, 0]], dtype=np.uint64)
    assert eng.fe_legendre_symbol(f("A"))[0] == 0 and eng.fe_legendre_symbol(sev)[0] == 1                  # legendre_symbol
    r0, ok0 = eng.fe_mod_sqrt(sev, 0)
    r1, ok1 = eng.fe_mod_sqrt(sev, 1)
    assert ok0[0] == 1 and ok1[0] == 1 and eq(r0, f("SQRT1_27_NEG")) and eq(r1, f("SQRT1_27_POS"))         # mod_sqrt_tonelli_shanks
    assert eng.fe_mod_sqrt(f("A"), 0)[1][0] == 0                                                           # non_QRmod_sqrt
    assert eq(eng.fe_half(f("A_MINUS_B")), f("A_MINUS_B_HALF"))
    n = 3000
    x, y = V.rand_fe_np(n, V.SEED + 140), V.rand_fe_np(n, V.SEED + 141)
    x[0] = 0
    y[1] = 0
    got, ok = eng.fe_div(x, y)
    want, wok = oracle.fe_div(x, y)
    assert eq(ok, wok) and eq(got, want)
    assert eq(eng.fe_half(x), oracle.fe_half(x))
    assert eq(eng.fe_legendre_symbol(x), oracle.fe_legendre_symbol(x))
    # legendre_symbol runs as a Jacobi symbol on positive division steps; a lane the round bound leaves unfinished falls
    # back to the exponentiation: bounds that finish every / some / no lane give the same answers
    big = V.rand_fe_np(1 << 16, V.SEED + 323)
    big[::97] = 0
    wl = oracle.mt(oracle.fe_legendre_symbol, big)
    for rounds in (None, 26, 2, 0):
        with V.tuned(ZC_JACOBI_ROUNDS=rounds) as te:
            assert eq(te.fe_legendre_symbol(big), wl), rounds
    assert eq(eng.fe_is_positive(x), oracle.fe_is_positive(x))
    raw = np.random.default_rng(V.SEED + 142).integers(0, 1 << 52, size=(n, 5), dtype=np.uint64)       # non-canonical limbs too
    assert eq(eng.fe_is_positive(raw), oracle.fe_is_positive(raw))
    for sign in (0, 1):
        got, ok = eng.fe_mod_sqrt(x, sign)
        want, wok = oracle.fe_mod_sqrt(x, sign)
        assert eq(ok, wok) and eq(got, want) and 0 < ok.sum() < n
    e = V.rand_fe_np(300, V.SEED + 143)
    e[0] = 0
    e[1] = [1, 0, 0, 0, 0]
    assert eq(eng.fe_pow(x[:300], e), oracle.fe_pow(x[:300], e))


def test_bytes_codecs_bulk(eng, oracle):
    rng = np.random.default_rng(V.SEED + 26)
    raw = rng.integers(0, 256, size=(5000, 32), dtype=np.uint8)
    assert eq(eng.fe_from_bytes(raw), oracle.fe_from_bytes(raw))
    out, ok = eng.sc_from_bytes(raw)
    wout, wok = oracle.sc_from_bytes(raw)
    assert eq(out, wout) and eq(ok, wok)
    a = V.rand_fe_np(5000, V.SEED + 27)
    assert eq(eng.fe_to_bytes(a), oracle.fe_to_bytes(a))


@pytest.mark.parametrize("n", [4096, 4097 + 300, 20001])
def test_point_add_sub_double_neg_staged_records(eng, oracle, n):
    """From 2^12 points on, add / sub / double / neg move their 160-byte records through LDS (k_ed_*_staged: coalesced 16-byte
    accesses, one operand after the other through one buffer); below, and for arrays that are not 16-byte aligned, every
    lane reads its own record.  Both forms, full and ragged last workgroups, host and device pointers: limb for limb the
    oracle's Add / Sub / Double (edwards.rs:465-489, :503-531, :579-592), canonical edge coordinates included."""
    import torch
    P = eng.ed_mul_base(V.rand_scalars_np(n, V.SEED + 600 + n, bits=249))
    Q = eng.ed_mul_base(V.rand_scalars_np(n, V.SEED + 601 + n, bits=249))
    P[0] = V.IDENT_ROW
    Q[1] = V.IDENT_ROW
    Q[2] = P[2]
    Q[3] = oracle.ed_neg(P[3:4])[0]
    edge = [pm.limbs(pm.P - 1), pm.limbs(pm.P - 2), [0] * 5, [1, 0, 0, 0, 0], pm.limbs((pm.P + 1) // 2), pm.limbs((1 << 252) - 1)]
    rng = np.random.default_rng(V.SEED + 602)
    for row in (5, 255, 256, n - 1):                              # canonical edge coordinates (off-curve: garbage in, the reference's garbage out)
        P[row] = sum([edge[rng.integers(len(edge))] for _ in range(4)], [])
        Q[row - 1] = sum([edge[rng.integers(len(edge))] for _ in range(4)], [])
    wadd, wsub, wdbl, wneg = oracle.ed_add(P, Q), oracle.ed_sub(P, Q), oracle.ed_double(P), oracle.ed_neg(Q)
    assert eq(eng.ed_add(P, Q), wadd) and eq(eng.ed_sub(P, Q), wsub) and eq(eng.ed_double(P), wdbl) and eq(eng.ed_neg(Q), wneg)
    dP, dQ = (torch.from_numpy(a.view(np.int64)).cuda() for a in (P, Q))
    for got, want in ((eng.ed_add(dP, dQ), wadd), (eng.ed_sub(dP, dQ), wsub), (eng.ed_double(dP), wdbl), (eng.ed_neg(dQ), wneg)):
        torch.cuda.synchronize()
        assert eq(got.cpu().numpy().view(np.uint64), want)
    # arrays that start 8 bytes off a 16-byte boundary: the per-lane kernels
    flatP = torch.empty(n * 20 + 1, dtype=torch.int64, device="cuda")
    flatQ = torch.empty(n * 20 + 1, dtype=torch.int64, device="cuda")
    oP, oQ = flatP[1:].view(n, 20), flatQ[1:].view(n, 20)
    assert oP.data_ptr() % 16 == 8
    oP.copy_(dP)
    oQ.copy_(dQ)
    for got, want in ((eng.ed_add(oP, oQ), wadd), (eng.ed_sub(oP, dQ), wsub), (eng.ed_double(oP), wdbl), (eng.ed_neg(oQ), wneg)):
        torch.cuda.synchronize()
        assert eq(got.cpu().numpy().view(np.uint64), want)


@pytest.mark.parametrize("n", [1, 63, 300])
def test_point_ops_bulk(eng, oracle, n):
    P = V.base_multiples(oracle, n, V.SEED + 30 + n)
    Q = V.base_multiples(oracle, n, V.SEED + 31 + n)
    P[0] = V.IDENT_ROW
    if n > 2:
        Q[1] = V.IDENT_ROW
        Q[2] = P[2]
    assert eq(eng.ed_add(P, Q), oracle.ed_add(P, Q))
    assert eq(eng.ed_sub(P, Q), oracle.ed_sub(P, Q))
    assert eq(eng.ed_double(P), oracle.ed_double(P))
    assert eq(eng.ed_neg(P), oracle.ed_neg(P))
    xy, ok = eng.ed_to_affine(P)
    wxy, wok = oracle.ed_to_affine(P)
    assert eq(ok, wok) and eq(xy, wxy)
    assert eq(eng.ed_eq(P, Q), oracle.ed_eq(P, Q))
    assert eq(eng.ris_eq(P, Q), oracle.ris_eq(P, Q))
    assert eq(eng.ed_mul_by_pow_2(P, 5), oracle.ed_mul_by_pow_2(P, 5))


def test_extreme_operands(eng, oracle):
    """Saturated limbs / p-1 / 0 in every position (worst cases of the lazy-reduction bounds) and
    off-curve coordinates through the group-law kernels: polynomial identities must still hold."""
    sat = [(1 << 52) - 1] * 5
    pool = [sat, pm.limbs(pm.P - 1), pm.limbs(pm.P - 2), [0] * 5, [1, 0, 0, 0, 0], pm.limbs((pm.P - 1) // 2),
            pm.limbs(2**252), pm.limbs(pm.P + 5)]
    a = np.array([x for x in pool for _ in pool], dtype=np.uint64)
    b = np.array([y for _ in pool for y in pool], dtype=np.uint64)
    assert eq(eng.fe_mul(a, b), oracle.fe_mul(a, b)) and eq(eng.fe_square(a), oracle.fe_square(a))
    assert eq(eng.fe_add(a, b), oracle.fe_add(a, b)) and eq(eng.fe_sub(a, b), oracle.fe_sub(a, b))   # incidental cases too
    rng = np.random.default_rng(21)
    canon = [pm.limbs(pm.P - 1), pm.limbs(pm.P - 2), [0] * 5, [1, 0, 0, 0, 0], pm.limbs((pm.P + 1) // 2)]
    n = 600
    P = np.array([sum([canon[rng.integers(len(canon))] if rng.random() < 0.6 else pm.limbs(int(rng.integers(0, 2**62)) * 2**190 % pm.P)
                       for _ in range(4)], []) for _ in range(n)], dtype=np.uint64)
    Q = P[rng.permutation(n)]
    assert eq(eng.ed_add(P, Q), oracle.ed_add(P, Q)) and eq(eng.ed_sub(P, Q), oracle.ed_sub(P, Q))
    assert eq(eng.ed_double(P), oracle.ed_double(P))
    K = np.zeros((n, 5), dtype=np.uint64)
    K[:, 0] = rng.integers(0, 1 << 50, size=n, dtype=np.uint64)
    assert eq(eng.ed_scalar_mul(P, K), oracle.ed_scalar_mul(P, K))
    A3, B3 = np.ascontiguousarray(P[:, :15]), np.ascontiguousarray(Q[:, :15])
    assert eq(eng.proj_add(A3, B3), oracle.proj_add(A3, B3)) and eq(eng.proj_double(A3), oracle.proj_double(A3))


def test_noncanonical_limbs_answer_for_the_value_mod_p(eng, oracle):
    """The stand-alone point kernels and the predicates (==, is_valid) work on plain coordinates; they must answer
    for the value mod p whatever 5 x 52-bit pattern carries it (untrusted data reaches == and is_valid): each
    coordinate as v, v + p, v + 37 p and v + k p just below 2^260 gives the results of the canonical limbs."""
    n = 64
    P = eng.ed_mul_base(V.rand_scalars_np(n, V.SEED + 320, bits=249))
    Q = eng.ed_mul_base(V.rand_scalars_np(n, V.SEED + 321, bits=249))
    rng = np.random.default_rng(V.SEED + 322)

    def lift(A, width):
        out = A.copy()
        for i in range(len(A)):
            for cidx in range(width // 5):
                v = sum(int(A[i, 5 * cidx + j]) << (52 * j) for j in range(5))
                k = [0, 1, 37, ((1 << 260) - 1 - v) // pm.P][int(rng.integers(0, 4))]
                out[i, 5 * cidx:5 * cidx + 5] = pm.limbs(v + k * pm.P)
        return out

    P2, Q2 = lift(P, 20), lift(Q, 20)
    assert not eq(P, P2)
    assert eq(eng.ed_add(P2, Q2), eng.ed_add(P, Q)) and eq(eng.ed_sub(P2, Q2), eng.ed_sub(P, Q))
    assert eq(eng.ed_double(P2), eng.ed_double(P)) and eq(eng.ed_coset4(P2)[:, 20:], eng.ed_coset4(P)[:, 20:])
    assert eng.ed_eq(P2, P).tolist() == [1] * n and eng.ed_eq(P2, Q).tolist() == [0] * n
    assert eng.ris_eq(P2, P).tolist() == [1] * n
    assert eng.ed_is_valid(P2).tolist() == [1] * n
    bad = P2.copy()
    bad[:, 0] ^= np.uint64(1)                                     # off the curve
    assert eng.ed_is_valid(bad).tolist() == [0] * n
    A3, B3 = np.ascontiguousarray(P[:, :15]), np.ascontiguousarray(P2[:, :15])
    assert eng.proj_eq(A3, B3).tolist() == [1] * n and eng.proj_is_valid(B3).tolist() == [1] * n


def _edge_scalars(K):
    K[0] = 0
    if len(K) > 8:
        K[1] = [1, 0, 0, 0, 0]
        K[2] = pm.limbs(pm.L)
        K[3] = pm.limbs(2**249 - 1)
        K[4] = [(1 << 52) - 1] * 5                               # all 260 bits of a raw limb pattern
        K[5] = [0, 0, 0, 0, 1 << 47]
        K[6] = pm.limbs(8)
        K[7] = pm.limbs(pm.L - 1)
    if len(K) > 64:                                              # raw patterns with bits >= 2^256 (early-stopping loop test)
        raw = V.raw_scalar_edges()
        K[8:8 + len(raw)] = raw


@pytest.mark.parametrize("n,bits", [(1, 252), (65, 249), (1000, 252), (4096 + 77, 252)])
def test_scalar_mul_strict_limbs(eng, oracle, n, bits):
    """config 3 at oracle-sized batches: (X:Y:Z:T) limbs identical to double_and_add."""
    P = V.base_multiples(oracle, n, V.SEED + 40 + n)
    K = V.rand_scalars_np(n, V.SEED + 41 + n, bits=bits)
    _edge_scalars(K)
    if n > 10:
        P[9] = V.IDENT_ROW
    assert eq(eng.ed_scalar_mul(P, K), oracle.ed_scalar_mul(P, K))


@pytest.mark.parametrize("n", [(1 << 14) + 1, 40000, (1 << 16) + 1, (1 << 17) + 77])
def test_scalar_mul_strict_launch_shapes(eng, oracle, n):
    """The strict path picks its kernel by batch size: four lanes per element up to 2^14 (covered
    above), the independent-chain variant up to 256 workgroups, one workgroup per 256 elements up to
    2^17, persistent waves over the cost-sorted permutation from 2^17 on (ragged last tile, zero and
    raw >= 2^256 scalars among the costs).  Against the oracle on slices (head, middle, ragged tail)."""
    base = V.base_multiples(oracle, 1500, V.SEED + 44)
    P = np.tile(base, (n // 1500 + 1, 1))[:n].copy()
    K = V.rand_scalars_np(n, V.SEED + 45 + n, bits=252)
    _edge_scalars(K)
    got = eng.ed_scalar_mul(P, K)
    sel = np.r_[0:600, n // 2:n // 2 + 600, n - 300:n]
    assert eq(got[sel], oracle.ed_scalar_mul(P[sel], K[sel]))


def test_raw_scalars_at_or_above_2_256(eng, oracle):
    """`n != Scalar::zero()` (edwards.rs:111) compares 32-byte encodings (scalar.rs:78-91,
    backend scalar.rs:477-516): raw limb patterns with bits >= 2^256 may stop early --
    [0,0,0,0,1<<50] -> identity, [1,0,0,0,1<<50] -> P.  Every kernel that takes a Mul<Scalar>
    operand must follow: the three strict launch shapes, both left-to-right variants, the
    windowed core, the fused Ristretto round trip, fixed-base multiplication and zc_msm."""
    import dusk_zerocaf_amd as z
    raw = V.raw_scalar_edges(n_random=80)
    ident = np.array([V.IDENT_ROW], dtype=np.uint64)
    for n in (len(raw), (1 << 14) + 5, (1 << 16) + 300):         # quad / small-launch / default kernels
        base = V.base_multiples(oracle, len(raw), V.SEED + 160)
        reps = n // len(raw) + 1
        P, K = np.tile(base, (reps, 1))[:n].copy(), np.tile(raw, (reps, 1))[:n].copy()
        got = eng.ed_scalar_mul(P, K)
        want = oracle.ed_scalar_mul(base, raw)
        assert eq(got[:len(raw)], want) and eq(got[-len(raw):], np.tile(want, (reps, 1))[:n][-len(raw):])
        assert eq(got, np.tile(want, (reps, 1))[:n])
    assert eq(want[0], ident[0]) and eq(want[1], oracle.ed_add(ident, base[1:2])[0])
    P, K = base, raw
    for mode in (1, 2):
        assert eq(eng.ed_scalar_mul(P, K, flags=mode), oracle.ed_scalar_mul_mode(P, K, mode)), mode
    fast = eng.ed_scalar_mul(P, K, flags=z.FAST)
    assert oracle.ed_eq(fast, want).all() and eq(oracle.ris_compress(fast), oracle.ris_compress(want))
    enc = oracle.ris_compress(P)
    out, ok = eng.ris_roundtrip_mul(enc, K)
    wout, wok = oracle.ris_roundtrip_mul(enc, K)
    assert eq(out, wout) and eq(ok, wok)
    bp = np.tile(np.array(sum(pm.pt_limbs(pm.BASEPOINT), []), dtype=np.uint64), (len(K), 1))
    wb = oracle.ed_scalar_mul(bp, K)
    assert oracle.ed_eq(eng.ed_mul_base(K), wb).all() and eq(eng.ris_mul_base_compress(K), oracle.ris_compress(wb))
    for n in (len(raw), 6000):                                    # scalar-mul + fold path, bucket path
        reps = n // len(raw) + 1
        Pm, Km = np.tile(base, (reps, 1))[:n].copy(), np.tile(raw, (reps, 1))[:n].copy()
        got, wantm = eng.msm(Pm, Km), oracle.msm_naive_mt(Pm, Km)
        assert oracle.ed_eq(got, wantm)[0] == 1 and eq(oracle.ed_compress(got)[0], oracle.ed_compress(wantm)[0])


def test_naf_noncanonical_scalars(eng, oracle):
    """compute_NAF above L - 1 is not the integer NAF (k - Scalar::from(-1) does not wrap there,
    backend scalar.rs:210-237, :370-389); the kernel recodes step for step like the reference."""
    n = 1000
    P = V.base_multiples(oracle, n, V.SEED + 161)
    K = V.rand_scalars_np(n, V.SEED + 162, bits=256)
    K[0] = pm.limbs(pm.L - 1)
    K[1] = pm.limbs(pm.L)
    K[2] = pm.limbs(2 * pm.L - 1)
    K[3] = pm.limbs(2**252 + 3)
    assert eq(eng.ed_scalar_mul(P, K, flags=2), oracle.ed_scalar_mul_mode(P, K, 2))


@pytest.mark.parametrize("mode", [1, 2])
def test_scalar_mul_reference_variants(eng, oracle, mode):
    """ltr_bin_mul / binary_naf_mul (edwards.rs:122-153): limbs identical to the reference's
    variant, and the same group element as double_and_add (reference tests left_to_right_bin_mul,
    naf_bin_mul)."""
    n = 1500
    P = V.base_multiples(oracle, n, V.SEED + 45)
    K = V.rand_scalars_np(n, V.SEED + 46 + mode, bits=249 if mode == 2 else 248)
    K[0] = 0
    K[1] = [1, 0, 0, 0, 0]
    K[2] = pm.limbs(2**215)
    K[3] = pm.limbs(2**7)
    if mode == 2:
        K[4] = pm.limbs(pm.L - 1)
        K[5] = pm.limbs(2**249 - 1)
    got = eng.ed_scalar_mul(P, K, flags=mode)
    assert eq(got, oracle.ed_scalar_mul_mode(P, K, mode))
    assert eng.ed_eq(got, eng.ed_scalar_mul(P, K)).all()


def test_scalar_mul_fast_mode_same_group_element(eng, oracle):
    """ZC_SCALAR_MUL_FAST (windowed, dedicated doubling): NOT limb-exact by design; it must be the
    same group element as double_and_add -- reference `==` (affine) and identical encodings."""
    import dusk_zerocaf_amd as z
    n = 5000
    P = V.base_multiples(oracle, n, V.SEED + 47)
    K = V.rand_scalars_np(n, V.SEED + 48, bits=252)
    _edge_scalars(K)
    P[9] = V.IDENT_ROW
    fast = eng.ed_scalar_mul(P, K, flags=z.FAST)
    want = oracle.ed_scalar_mul(P, K)
    assert oracle.ed_eq(fast, want).all()
    assert eq(oracle.ed_compress(fast)[0], oracle.ed_compress(want)[0])
    assert eq(eng.ris_compress(fast), oracle.ris_compress(want))
    assert eng.ed_is_valid(fast).all()
    big = 1 << 18                                                 # and against the strict kernel at size
    Pb = np.tile(P[:1024], (big >> 10, 1))
    Kb = V.rand_scalars_np(big, V.SEED + 49, bits=252)
    assert eng.ed_eq(eng.ed_scalar_mul(Pb, Kb, flags=z.FAST), eng.ed_scalar_mul(Pb, Kb)).all()


def test_scalar_mul_full_size_properties(eng, oracle):
    """config 3 at 2^20: linearity (k1+k2)P == k1P + k2P on every element, plus an
    oracle-checked stride sample of exact limbs."""
    n = 1 << 20
    small = V.base_multiples(oracle, 1 << 10, V.SEED + 50)
    P = np.tile(small, (n >> 10, 1))
    k1 = V.rand_scalars_np(n, V.SEED + 51, bits=248)
    k2 = V.rand_scalars_np(n, V.SEED + 52, bits=248)
    ks = oracle.sc_add(np.zeros_like(k1), k1)                     # k1 < 2^248 < L: unchanged
    assert eq(ks, k1)
    ksum = k1.copy()                                              # plain integer sum < 2^249
    carry = np.zeros(n, dtype=np.uint64)
    for j in range(5):
        t = k1[:, j] + k2[:, j] + carry
        ksum[:, j] = t & np.uint64((1 << 52) - 1)
        carry = t >> np.uint64(52)
    r1, r2, rs = eng.ed_scalar_mul(P, k1), eng.ed_scalar_mul(P, k2), eng.ed_scalar_mul(P, ksum)
    assert eng.ed_eq(eng.ed_add(r1, r2), rs).all()
    idx = np.arange(0, n, 4099)
    assert eq(rs[idx], oracle.ed_scalar_mul(P[idx], ksum[idx]))


def test_scalar_mul_full_size_every_output(eng, oracle):
    """config 3 exactly as benchmarked: 2^20 distinct points x uniform raw 252-bit scalars, EVERY
    (X:Y:Z:T) limb of the launch against the oracle (threaded over the host cores)."""
    n = 1 << 20
    P = eng.ed_mul_base(V.rand_scalars_np(n, V.SEED + 53, bits=249))     # r_i * B (checked in test_fixed_base_key_generation)
    K = V.rand_scalars_np(n, V.SEED + 54, bits=252)
    _edge_scalars(K)
    got = eng.ed_scalar_mul(P, K)
    assert eq(got, oracle.mt(oracle.ed_scalar_mul, P, K))


def test_codecs_bulk(eng, oracle):
    n = 700
    P = V.base_multiples(oracle, n, V.SEED + 60)
    P[0] = V.IDENT_ROW
    enc, ok = eng.ed_compress(P)
    wenc, wok = oracle.ed_compress(P)
    assert eq(ok, wok) and eq(enc, wenc)
    dec, ok = eng.ed_decompress(wenc)
    wdec, wok = oracle.ed_decompress(wenc)
    assert eq(ok, wok) and eq(dec, wdec)
    renc = eng.ris_compress(P)
    wrenc = oracle.ris_compress(P)
    assert eq(renc, wrenc)
    rdec, ok = eng.ris_decompress(wrenc)
    wrdec, wok = oracle.ris_decompress(wrenc)
    assert eq(ok, wok) and eq(rdec, wrdec) and ok.all()
    rng = np.random.default_rng(V.SEED + 61)
    raw = rng.integers(0, 256, size=(3000, 32), dtype=np.uint8)
    raw[:, 31] &= 0x0F
    raw[:5, 31] = 0xFF
    rdec, ok = eng.ris_decompress(raw)
    wrdec, wok = oracle.ris_decompress(raw)
    assert eq(ok, wok) and eq(rdec, wrdec) and 0 < ok.sum() < 3000
    raw[:5, 31] = 0x8F
    dec, ok = eng.ed_decompress(raw)
    wdec, wok = oracle.ed_decompress(raw)
    assert eq(ok, wok) and eq(dec, wdec) and 0 < ok.sum() < 3000
    # invalid Edwards points (not on the curve) hit the reference's unwrap panics -> ok = 0 on both sides
    bad = P[:64].copy()
    bad[:, 5] ^= np.uint64(1)
    enc, ok = eng.ed_compress(bad)
    wenc, wok = oracle.ed_compress(bad)
    assert eq(ok, wok) and eq(enc, wenc)


def test_affine_and_compress_edge_rows_and_large_batches(eng, oracle):
    """Scaled representatives (non-trivial Z), x = 0 points, Z = 0 and off-curve junk through
    to_affine / compress; batches >= 2^18 take the one-inversion-per-chunk conversion."""
    E = V.base_multiples(oracle, 512, V.SEED + 64)
    lam = V.rand_fe_np(512, V.SEED + 65)
    for c in range(4):
        E[:, 5 * c:5 * c + 5] = oracle.fe_mul(E[:, 5 * c:5 * c + 5], lam)
    z, zero = lam[7], np.zeros(5, dtype=np.uint64)
    E[0] = np.concatenate([zero, z, z, zero])                     # (0, 1)
    E[1] = np.concatenate([zero, oracle.fe_neg(z[None])[0], z, zero])   # (0, -1)
    E[2, 10:15] = 0                                               # Z = 0
    E[3:40] = V.rand_fe_np(37 * 4, V.SEED + 66).reshape(37, 20)   # junk
    enc, ok = eng.ed_compress(E)
    wenc, wok = oracle.ed_compress(E)
    assert eq(ok, wok) and eq(enc, wenc) and wok[0] and wok[1] and not wok[2] and 0 < wok[3:40].sum() < 37
    xy, aok = eng.ed_to_affine(E)
    wxy, waok = oracle.ed_to_affine(E)
    assert eq(aok, waok) and eq(xy, wxy)
    n = (1 << 18) + 333                                           # chunked conversion, ragged last chunk
    big = np.tile(E, (n // 512 + 1, 1))[:n].copy()
    xy, aok = eng.ed_to_affine(big)
    assert eq(xy[:512], wxy) and eq(aok[:512], waok)
    assert eq(xy[-845:], np.tile(wxy, (3, 1))[(n - 845) % 512:][:845])
    assert eq(aok, np.tile(waok, n // 512 + 1)[:n])
    enc_big, ok_big = eng.ed_compress(big)
    assert eq(enc_big, np.tile(wenc, (n // 512 + 1, 1))[:n]) and eq(ok_big, np.tile(wok, n // 512 + 1)[:n])


def test_ristretto_roundtrip_mul(eng, oracle):
    """config 4 shape: decompress -> scalar-mul -> compress, with ~1% invalid encodings."""
    n = 2048 + 5
    P = V.base_multiples(oracle, n, V.SEED + 70)
    enc = oracle.ris_compress(P)
    rng = np.random.default_rng(V.SEED + 71)
    bad = rng.choice(n, size=n // 100, replace=False)
    enc[bad] = rng.integers(0, 256, size=(len(bad), 32), dtype=np.uint8)
    K = V.rand_scalars_np(n, V.SEED + 72, bits=252)
    _edge_scalars(K)
    out, ok = eng.ris_roundtrip_mul(enc, K)
    wout, wok = oracle.ris_roundtrip_mul(enc, K)
    assert eq(ok, wok) and eq(out, wout) and 0 < (ok == 0).sum() <= len(bad)


def test_ristretto_roundtrip_full_size_2_22(eng, oracle):
    """config 4 at BASELINE size (2^22): EVERY output byte and accept flag of the launch against the oracle's
    decompress -> Mul<Scalar> -> compress (ristretto.rs:96-154, edwards.rs:547-561, ristretto.rs:398-425; a minute or
    two of all host cores), ~1% undecodable inputs included, plus the composition property
    mul(mul(E, k1), k2) == mul(E, k1*k2 mod L) as bytes on every element."""
    n = 1 << 22
    small = V.base_multiples(oracle, 1 << 12, V.SEED + 130)
    enc_small = oracle.ris_compress(small)
    enc = np.tile(enc_small, (n >> 12, 1))
    rng = np.random.default_rng(V.SEED + 131)
    bad = rng.choice(n, size=n // 100, replace=False)
    enc[bad, :8] ^= rng.integers(1, 256, size=(len(bad), 8), dtype=np.uint8)
    k1 = V.rand_scalars_np(n, V.SEED + 132, bits=249)
    k2 = V.rand_scalars_np(n, V.SEED + 133, bits=249)
    k12 = eng.sc_mul(k1, k2)
    r1, ok1 = eng.ris_roundtrip_mul(enc, k1)
    r2, ok2 = eng.ris_roundtrip_mul(r1, k2)
    r12, ok12 = eng.ris_roundtrip_mul(enc, k12)
    good = ok1 == 1
    assert eq(ok12, ok1) and 0 < (~good).sum() <= len(bad)
    assert ok2[good].all() and eq(r2[good], r12[good])
    assert not r1[~good].any()
    idx = np.r_[np.arange(0, n, 8191), bad[:64]]
    wout, wok = oracle.ris_roundtrip_mul(enc[idx], k1[idx])
    assert eq(ok1[idx], wok) and eq(r1[idx], wout)
    for lo in range(0, n, 1 << 20):                               # the whole launch, a quarter at a time
        part = slice(lo, lo + (1 << 20))
        wout, wok = oracle.mt(oracle.ris_roundtrip_mul, enc[part], k1[part])
        assert eq(ok1[part], wok) and eq(r1[part], wout) and (wok == 0).sum() > 1000, lo


def test_windowed_core_table_ring_under_contention(eng, oracle):
    """The windowed core keeps its per-lane tables in a ring of wave slots per XCD (ring_acquire /
    ring_release, zc_kernels.hip.h).  With the default 512 slots per XCD a wave practically never waits
    for a slot; ZC_RING_SLOTS shrinks the ring far below the number of resident waves, so that every
    slot is handed from wave to wave many times inside one launch.  Bytes, ok masks and points must
    not depend on the ring size, and must equal the oracle's."""
    import dusk_zerocaf_amd as z
    n = (1 << 16) + 333
    P = V.base_multiples(oracle, 1 << 10, V.SEED + 140)
    P = np.tile(P, (n // 1024 + 1, 1))[:n]
    enc = np.tile(oracle.ris_compress(P[:1024]), (n // 1024 + 1, 1))[:n]
    rng = np.random.default_rng(V.SEED + 141)
    bad = rng.choice(n, size=n // 100, replace=False)
    enc[bad, :8] ^= rng.integers(1, 256, size=(len(bad), 8), dtype=np.uint8)
    K = V.rand_scalars_np(n, V.SEED + 142, bits=252)
    _edge_scalars(K)
    ref_out, ref_ok = eng.ris_roundtrip_mul(enc, K)
    ref_pts = eng.ed_scalar_mul(P, K, flags=z.FAST)
    idx = np.arange(0, n, 37)
    wout, wok = oracle.mt(oracle.ris_roundtrip_mul, enc[idx], K[idx])
    assert eq(ref_out[idx], wout) and eq(ref_ok[idx], wok)
    for slots in ("96", "7", "1"):                                 # 384 waves are resident per XCD
        m = n if slots != "1" else 1 << 13                          # one slot per XCD serialises the XCD's waves
        with V.tuned(ZC_RING_SLOTS=slots) as te:
            out, ok = te.ris_roundtrip_mul(enc[:m], K[:m])
            assert eq(out, ref_out[:m]) and eq(ok, ref_ok[:m]), "ring of %s slots per XCD" % slots
            pts = te.ed_scalar_mul(P[:m], K[:m], flags=z.FAST)
            assert eq(pts, ref_pts[:m]), "ring of %s slots per XCD (points)" % slots
            out, ok = te.ris_roundtrip_mul(enc[:m], K[:m])          # and the state is reset per launch
            assert eq(out, ref_out[:m]) and eq(ok, ref_ok[:m])
    out, ok = eng.ris_roundtrip_mul(enc, K)                        # and the state is reset per launch
    assert eq(out, ref_out) and eq(ok, ref_ok)


def test_windowed_core_ring_timeout_is_reported_and_the_context_survives(eng):
    """A wave that gives up waiting for its table slot sets the device's error word (pinned host memory) instead of
    trapping (a trap is a sticky HIP error that kills every later call of the process).  The host looks at the word at
    every entry point and synchronisation that touches the device, reports ZC_ERR_HIP once, clears it, and the context
    keeps working.  The hook ZC_TEST_RING_POISON (libzerocaf_hip_test.so only) sets the word the way a timed-out wave would."""
    import dusk_zerocaf_amd as z
    import torch
    n = 4096
    K = V.rand_scalars_np(n, V.SEED + 143, bits=252)
    P = eng.ed_mul_base(V.rand_scalars_np(n, V.SEED + 144, bits=249))
    good = eng.ed_scalar_mul(P, K, flags=z.FAST)
    with V.tuned(hooks=True, ZC_TEST_RING_POISON=1) as te:
        with pytest.raises(Exception, match="table slot"):
            te.ed_scalar_mul(P, K, flags=z.FAST)                   # host batch: synchronises, sees the word
        assert eq(te.fe_mul(K, K), eng.fe_mul(K, K))               # reported once, cleared, context intact
        dP, dK = (torch.from_numpy(a.view(np.int64)).cuda() for a in (P, K))
        te.ed_scalar_mul(dP, dK, flags=z.FAST)                     # device batch: asynchronous, no report yet
        torch.cuda.synchronize()                                   # the caller synchronises its own stream ...
        with pytest.raises(Exception, match="table slot"):
            te.fe_mul(dK, dK)                                      # ... and the NEXT entry point on that device reports it
        te.fe_mul(dK, dK)
        te.ed_scalar_mul(dP, dK, flags=z.FAST)
        with pytest.raises(Exception, match="table slot"):
            te.synchronize()                                       # and so does zc_ctx_synchronize
        te.synchronize()
    with pytest.raises(AttributeError):
        eng.lib.zc_test_msm_sort                                   # the release library carries no test hook
    assert eq(eng.ed_scalar_mul(P, K, flags=z.FAST), good)


def test_windowed_core_waves_that_give_up_write_poison_and_fail_closed(eng, oracle):
    """The real give-up path (not the injected word): one table slot per XCD and a spin limit of 2^4 polls
    (ZC_TEST_RING_SPINS, test build only), so most waves of the launch give up.  A wave that gives up must not touch
    a slot it does not own: its rows are poison (all ones, ok = 0), every other row is the oracle's, the call reports
    the failure, and the release library on the same device is unaffected."""
    import dusk_zerocaf_amd as z
    n = 1 << 14
    K = V.rand_scalars_np(n, V.SEED + 145, bits=252)
    P = eng.ed_mul_base(V.rand_scalars_np(n, V.SEED + 146, bits=249))
    enc = eng.ris_compress(P)
    want_pts = eng.ed_scalar_mul(P, K, flags=z.FAST)
    want_enc, want_ok = eng.ris_roundtrip_mul(enc, K)
    idx = np.arange(0, n, 41)
    wout, wok = oracle.mt(oracle.ris_roundtrip_mul, enc[idx], K[idx])
    assert eq(want_enc[idx], wout) and eq(want_ok[idx], wok)
    import torch
    with V.tuned(hooks=True, ZC_RING_SLOTS=1, ZC_TEST_RING_SPINS=4) as te:
        dP, dK, dE = (torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).cuda() for a in (P, K, enc))
        pts = te.ed_scalar_mul(dP, dK, flags=z.FAST)
        torch.cuda.synchronize()
        with pytest.raises(Exception, match="table slot"):
            te.synchronize()
        pts = pts.cpu().numpy().view(np.uint64)
        poisoned = (pts == np.uint64(0xFFFFFFFFFFFFFFFF)).all(axis=1)
        assert 0 < poisoned.sum() < n, poisoned.sum()
        assert (poisoned.reshape(-1, 64).all(axis=1) == poisoned.reshape(-1, 64).any(axis=1)).all()    # whole waves
        assert eq(pts[~poisoned], want_pts[~poisoned])
        out, ok = te.ris_roundtrip_mul(dE, dK)
        torch.cuda.synchronize()
        with pytest.raises(Exception, match="table slot"):
            te.synchronize()
        out, ok = out.cpu().numpy(), ok.cpu().numpy()
        poisoned = (out == 0xFF).all(axis=1)
        assert 0 < poisoned.sum() < n and (ok[poisoned] == 0).all()
        assert eq(out[~poisoned], want_enc[~poisoned]) and eq(ok[~poisoned], want_ok[~poisoned])
        with pytest.raises(Exception, match="table slot"):
            te.ris_roundtrip_mul(enc, K)                           # host batch: reported by the call itself
    assert eq(eng.ris_roundtrip_mul(enc, K)[0], want_enc)


def test_next_rows_elligator_validity_projective(eng, oracle, kats):
    """Elligator + from_uniform_bytes, is_valid (Edwards and Ristretto),
    ProjectivePoint add/double -- limb-exact vs the oracle, reference KATs included."""
    r0b = bytes.fromhex(kats["ristretto_elligator_hex"][0]["hex"])
    exp = np.array([sum([x["limbs"] for x in kats["ristretto_elligator_point"]], [])], dtype=np.uint64)
    r0 = eng.fe_from_bytes(np.frombuffer(r0b, dtype=np.uint8).reshape(1, 32))
    got = eng.ris_elligator(r0)
    assert eng.ris_eq(got, exp)[0] == 1 and eq(eng.ris_compress(got), eng.ris_compress(exp))   # elligator_vs_ristretto_sage
    R = V.rand_fe_np(3000, V.SEED + 110)
    assert eq(eng.ris_elligator(R), oracle.ris_elligator(R))
    rng = np.random.default_rng(V.SEED + 111)
    ub = rng.integers(0, 256, size=(2000, 64), dtype=np.uint8)
    pts = eng.ris_from_uniform_bytes(ub)
    assert eq(pts, oracle.ris_from_uniform_bytes(ub))
    assert eng.ed_is_valid(pts).all()                               # random_point_validity
    P = V.base_multiples(oracle, 400, V.SEED + 112)
    P[1, 5] ^= np.uint64(1)
    assert eq(eng.ed_is_valid(P), oracle.ed_is_valid(P)) and eng.ed_is_valid(P)[1] == 0
    # validity_check (ristretto.rs:642-664): multiples of B valid, the order-8L point is not
    yb = np.array([kats["ristretto_inline_bytes"][0]["bytes"]], dtype=np.uint8)
    yb2 = yb.copy()
    p8, ok = eng.ed_decompress(yb2)                                 # sign 0, y < 2^252
    assert ok[0] == 1 and eng.ed_is_valid(p8)[0] == 1
    Q = np.concatenate([P[:64], p8, pts[:32]])
    assert eq(eng.ris_is_valid(Q), oracle.ris_is_valid(Q))
    assert eng.ris_is_valid(p8)[0] == 0 and eng.ris_is_valid(P[:1])[0] == 1
    def pp(n):
        c = kats["edwards_points"][n]["coords"]
        return np.array([c["X"] + c["Y"] + c["Z"]], dtype=np.uint64)
    assert eq(eng.proj_add(pp("P1_PROJECTIVE"), pp("P2_PROJECTIVE")), pp("P4_PROJECTIVE"))    # projective_point_addition
    assert eq(eng.proj_double(pp("P1_PROJECTIVE")), pp("P3_PROJECTIVE"))                      # projective_point_doubling
    c1 = kats["edwards_points"]["P1_EXTENDED"]["coords"]
    assert eq(eng.proj_to_extended(pp("P1_PROJECTIVE")), np.array([c1["X"] + c1["Y"] + c1["Z"] + c1["T"]], dtype=np.uint64))
    A3, B3 = np.ascontiguousarray(P[:, :15]), np.ascontiguousarray(P[::-1, :15])
    assert eq(eng.proj_add(A3, B3), oracle.proj_add(A3, B3)) and eq(eng.proj_double(A3), oracle.proj_double(A3))
    assert eq(eng.proj_to_extended(A3), oracle.proj_to_extended(A3))


def test_fixed_base_key_generation(eng, oracle):
    """k * BASEPOINT from the comb table: same group element as the reference's `&BASEPOINT * &k`
    (affine ==, identical Edwards / Ristretto encodings); the fused key-generation call returns
    exactly (RISTRETTO_BASEPOINT * k).compress()."""
    n = 6000
    K = V.rand_scalars_np(n, V.SEED + 150, bits=249)
    _edge_scalars(K)
    base = np.tile(np.array(sum(pm.pt_limbs(pm.BASEPOINT), []), dtype=np.uint64), (n, 1))
    want = oracle.ed_scalar_mul(base, K)
    got = eng.ed_mul_base(K)
    assert oracle.ed_eq(got, want).all()
    assert eq(oracle.ed_compress(got)[0], oracle.ed_compress(want)[0])
    pk = eng.ris_mul_base_compress(K)
    assert eq(pk, oracle.ris_compress(want))
    big = 1 << 20
    Kb = V.rand_scalars_np(big, V.SEED + 151, bits=249)
    pkb = eng.ris_mul_base_compress(Kb)
    idx = np.arange(0, big, 2053)
    assert eq(pkb[idx], oracle.ris_compress(oracle.ed_scalar_mul(np.tile(base[:1], (len(idx), 1)), Kb[idx])))
    # ECDH consistency on every element: k2 * (k1 * B) == k1 * (k2 * B) as encodings
    K2 = V.rand_scalars_np(big, V.SEED + 152, bits=249)
    s12, ok1 = eng.ris_roundtrip_mul(pkb, K2)
    s21, ok2 = eng.ris_roundtrip_mul(eng.ris_mul_base_compress(K2), Kb)
    assert ok1.all() and ok2.all() and eq(s12, s21)


def test_reference_odd_multiples_table_and_window_naf_mul_on_the_gpu(eng, oracle, kats):
    """N1, literally: the reference's BASEPOINT_ODD_MULTIPLES_TABLE (constants.rs:216-972: 125 affine KATs for
    (2j-1) B) against zc_ed_mul_base, and the CORRECT window_naf_mul (edwards.rs:155-171 mis-indexes the table)
    assembled from kernels that exist: zc_sc_compute_naf(k, w) digits, that table, zc_ed_double / add / sub --
    it must reproduce k B for every width w = 2 .. 7."""
    table = np.array([sum(p, []) for p in kats["odd_multiples_table"]["points"]], dtype=np.uint64)
    assert table.shape == (126, 20)
    odd = np.zeros((125, 5), dtype=np.uint64)
    odd[:, 0] = np.arange(1, 250, 2)
    assert eng.ed_eq(eng.ed_mul_base(odd), table[1:]).tolist() == [1] * 125
    assert eng.ed_eq(eng.ed_scalar_mul(np.tile(table[1:2], (125, 1)), odd), table[1:]).tolist() == [1] * 125
    K = V.rand_scalars_np(12, V.SEED + 310, bits=249)
    K[0] = 0
    K[1] = [1, 0, 0, 0, 0]
    K[2] = V.limbs_array([pm.L - 1])[0]
    kv = [sum(int(K[i, j]) << (52 * j) for j in range(5)) for i in range(len(K))]
    want = eng.ed_mul_base(K)
    base = np.tile(table[1:2], (len(K), 1))
    assert eng.ed_eq(want, oracle.ed_scalar_mul(base, K)).tolist() == [1] * len(K)
    for w in range(2, 8):
        naf = np.asarray(eng.sc_compute_naf(K, w)).astype(np.int64)
        assert [sum(int(d) << i for i, d in enumerate(row)) for row in naf] == kv                 # digit i weighs 2^i
        assert np.all((naf == 0) | ((naf % 2 != 0) & (np.abs(naf) < (1 << (w - 1)))))          # odd, |d| < 2^(w-1)
        acc = np.tile(np.array([V.IDENT_ROW], dtype=np.uint64), (len(K), 1))
        for i in range(255, -1, -1):
            acc = eng.ed_double(acc)
            d = naf[:, i]
            if not d.any():
                continue
            entry = table[(np.abs(d) + 1) // 2 * (d != 0)]                                        # entry j = (2j - 1) B, entry 0 = identity
            acc = np.where((d < 0)[:, None], eng.ed_sub(acc, entry), eng.ed_add(acc, entry))
        assert eng.ed_eq(acc, want).tolist() == [1] * len(K), w
        assert eq(eng.ed_compress(acc)[0], eng.ed_compress(want)[0]), w


def test_window_naf_mul_in_one_launch(eng, oracle, kats):
    """zc_ed_mul_base_wnaf: the reference's window_naf_mul (edwards.rs:155-171) as ONE kernel over the odd multiples
    of the basepoint, table indexed as (|d| + 1) / 2 and all 256 digits read.  (i) The table the device builds, entry
    by entry, against BASEPOINT_ODD_MULTIPLES_TABLE (constants.rs:216-972) through the test build's dump hook; (ii)
    every width 2..7: the result equals the oracle's `&BASEPOINT * &k` (double_and_add) for canonical scalars, the
    comb's result, and -- for raw patterns above L, where compute_window_NAF's modular subtraction wraps -- the point
    (sum_i d_i 2^i) B of the ORACLE's digits; (iii) bad widths are refused."""
    import torch
    table = np.array([sum(p, []) for p in kats["odd_multiples_table"]["points"]], dtype=np.uint64)
    with V.tuned(hooks=True) as te:
        dump = torch.empty((125, 20), dtype=torch.int64, device="cuda")
        assert te.lib.zc_test_odd_table(te.ctx, dump.data_ptr()) == 0, te.lib.zc_last_error()
        dump = dump.cpu().numpy().view(np.uint64)
    assert eng.ed_eq(dump, table[1:]).tolist() == [1] * 125 and eng.ed_is_valid(dump).tolist() == [1] * 125
    n = 3000
    K = V.rand_scalars_np(n, V.SEED + 330, bits=249)
    K[0] = 0
    K[1] = [1, 0, 0, 0, 0]
    K[2] = V.limbs_array([pm.L - 1])[0]
    K[3] = V.limbs_array([pm.L - 2])[0]
    K[4] = V.limbs_array([(1 << 249) - 1])[0]
    K[5:5 + 125, :] = 0
    K[5:5 + 125, 0] = np.arange(1, 250, 2)                          # the odd multiples themselves
    K[130:130 + 64] = V.limbs_array([pm.L - 1 - j for j in range(64)])         # near L: the recoder's wrap-around zone
    base = np.tile(table[1:2], (n, 1))
    want = oracle.mt(oracle.ed_scalar_mul, base, K)
    comb = eng.ed_mul_base(K)
    assert oracle.ed_eq(comb, want).all()
    raw = np.concatenate([V.raw_scalar_edges(n_random=60), V.rand_scalars_np(200, V.SEED + 331, bits=252)])   # mostly >= L
    for w in range(2, 8):
        got = eng.ed_mul_base_wnaf(K, w)
        naf = np.asarray(oracle.sc_compute_naf(K, w)).astype(np.int64)
        vals = [sum(int(d) << i for i, d in enumerate(row)) % pm.L for row in naf]
        kv = [sum(int(K[i, j]) << (52 * j) for j in range(5)) for i in range(n)]
        exact = np.array([a == b for a, b in zip(vals, kv)])
        assert exact[:2].all() and exact[5:5 + 125].all()          # the digits of small / ordinary scalars represent them (L - 1, L - 2 wrap)
        assert oracle.ed_eq(got[exact], want[exact]).all(), w
        assert eq(eng.ed_compress(got[exact])[0], eng.ed_compress(comb[exact])[0]), w
        if (~exact).any():                                         # near L the reference's digits stand for another integer: follow them
            wv = oracle.mt(oracle.ed_scalar_mul, base[~exact], V.limbs_array([v for v, e in zip(vals, exact) if not e]))
            assert oracle.ed_eq(got[~exact], wv).all(), w
        gr = eng.ed_mul_base_wnaf(raw, w)
        nr = np.asarray(oracle.sc_compute_naf(raw, w)).astype(np.int64)
        vr = V.limbs_array([sum(int(d) << i for i, d in enumerate(row)) % pm.L for row in nr])
        assert oracle.ed_eq(gr, oracle.mt(oracle.ed_scalar_mul, base[:len(raw)], vr)).all(), w
    big = (1 << 17) + 5                                            # a full-chip launch, device pointers
    Kb = V.rand_scalars_np(big, V.SEED + 332, bits=249)
    dK = torch.from_numpy(Kb.view(np.int64)).cuda()
    assert eq(eng.ris_compress(eng.ed_mul_base_wnaf(dK, 5)).cpu().numpy(), eng.ris_mul_base_compress(Kb))
    for bad in (0, 1, 8, 200):
        with pytest.raises(Exception, match="width"):
            eng.ed_mul_base_wnaf(K[:4], bad)


def test_fixed_base_comb_digit_edges(eng, oracle):
    """The radix-256 comb recodes a scalar into signed digits in [-128, 128) with a carry that can run
    through every window: byte patterns 0x80 / 0x7f / 0xff / 0x00 in every position and mix, raw limb
    patterns up to 2^260 - 1 (the early-stopping loop rule applies to the reference product), small
    and sparse scalars -- k*B as a point and (k*B).compress() must equal the reference's &BASEPOINT * &k."""
    def from_bytes33(b):                                             # 33 little-endian bytes -> 5 x 52-bit limbs (260 bits)
        v = int.from_bytes(bytes(b), "little") % (1 << 260)
        return [(v >> (52 * j)) & ((1 << 52) - 1) for j in range(5)]
    pats = []
    for fill in (0x80, 0x7F, 0xFF, 0x00, 0x81, 0x01):
        pats.append([fill] * 33)
        for other in (0x80, 0x7F, 0xFF):
            pats.append([fill if i % 2 else other for i in range(33)])
            pats.append([other] * 5 + [fill] * 28)
            pats.append([fill] * 31 + [other, 0x0F])
    rng = np.random.default_rng(V.SEED + 190)
    for _ in range(200):                                             # random mixes of the critical byte values
        pats.append(rng.choice([0x00, 0x7F, 0x80, 0xFF, 0x01, 0x81], size=33).tolist())
    K = np.array([from_bytes33(p) for p in pats], dtype=np.uint64)
    K = np.concatenate([K, V.raw_scalar_edges(), np.array([[0] * 5, [1, 0, 0, 0, 0], [128, 0, 0, 0, 0], [127, 0, 0, 0, 0],
                                                             [256, 0, 0, 0, 0], [(1 << 52) - 1] * 5], dtype=np.uint64)])
    base = np.tile(np.array(sum(pm.pt_limbs(pm.BASEPOINT), []), dtype=np.uint64), (len(K), 1))
    want = oracle.mt(oracle.ed_scalar_mul, base, K)
    got = eng.ed_mul_base(K)
    assert oracle.ed_eq(got, want).all()
    assert eq(eng.ris_mul_base_compress(K), oracle.ris_compress(want))


def test_msm_small(eng, oracle):
    for n in (1, 2, 3, 64, 257):                                  # scalar-mul + fold path
        P = V.base_multiples(oracle, n, V.SEED + 80 + n)
        K = V.rand_scalars_np(n, V.SEED + 81 + n, bits=249)
        got = eng.msm(P, K)
        want = oracle.msm_naive(P, K)
        assert oracle.ed_eq(got, want)[0] == 1
        assert eq(oracle.ed_compress(got)[0], oracle.ed_compress(want)[0])


def test_msm_degenerate_scalars(eng, oracle):
    n = 6000
    P = V.base_multiples(oracle, 1024, V.SEED + 86)
    P = np.tile(P, (6, 1))[:n].copy()
    ident = np.array([V.IDENT_ROW], dtype=np.uint64)
    assert oracle.ed_eq(eng.msm(P, np.zeros((n, 5), dtype=np.uint64)), ident)[0] == 1       # all-zero scalars
    K = np.zeros((n, 5), dtype=np.uint64)
    K[:, 0] = np.arange(n) % 7                                                               # 3-bit scalars: one window
    assert eq(oracle.ed_compress(eng.msm(P, K))[0], oracle.ed_compress(_gpu_naive_msm(eng, P, K))[0])


def test_msm_normalisation_skips_the_inversion_for_affine_inputs(eng, oracle):
    """k_msm_prepare_affine: a wave whose 64 x c points all have Z = 1 (decompressed or already affine inputs) needs no inversion and
    requests no prefix block (round 6: the prefetching form).  All-affine input, and input whose first half is affine and second
    half projective (waves of both kinds and, with the lane's points 2^17 lanes apart, waves that mix them), at 2^17 + 77 pairs
    (ragged last block), default and forced points-per-lane: against the oracle's sum."""
    n = (1 << 17) + 77
    Pp = eng.ed_mul_base(V.rand_scalars_np(n, V.SEED + 430, bits=249))
    xy, ok = eng.ed_to_affine(Pp)
    assert ok.all()
    Pa = np.zeros_like(Pp)
    Pa[:, 0:5], Pa[:, 5:10] = xy[:, 0:5], xy[:, 5:10]
    Pa[:, 10] = 1
    Pa[:, 15:20] = eng.fe_mul(np.ascontiguousarray(xy[:, 0:5]), np.ascontiguousarray(xy[:, 5:10]))
    assert oracle.ed_is_valid(Pa[:2000]).all() and eng.ed_eq(Pa, Pp).all()
    K = V.rand_scalars_np(n, V.SEED + 431, bits=252)
    mixed = Pa.copy()
    mixed[n // 2:] = Pp[n // 2:]
    for P in (Pa, mixed):
        want = oracle.msm_naive_mt(P, K)
        for chunk in (None, 1, 2, 7):
            with V.tuned(hooks=chunk is not None, ZC_MSM_AFFINE_CHUNK=chunk) as te:
                assert te.msm_plan(n)["affine"]
                got = te.msm(P, K)
            assert oracle.ed_eq(got, want)[0] == 1 and eq(oracle.ed_compress(got)[0], oracle.ed_compress(want)[0]), chunk


def test_msm_window_groups_with_empty_groups(oracle):
    """Window groups and the pre-shifted carry (k_msm_shift, round 6): the lowest group adds 2^(c nw) x (the result of the groups
    above) LAST.  Degenerate splits of a 6000-pair shard (c = 8: 33 windows): scalars below 2^15 leave every upper group empty (the
    carry is the literal identity and must stay it through the shift), scalars that are multiples of 2^200 leave the lowest groups
    empty (the rule starts from nothing and the carry is the whole result), and both kinds mixed -- against the oracle's sum."""
    n = 6000
    P = np.tile(V.base_multiples(oracle, 1024, V.SEED + 186), (6, 1))[:n].copy()
    rng = np.random.default_rng(V.SEED + 187)
    low = np.zeros((n, 5), dtype=np.uint64)
    low[:, 0] = rng.integers(0, 1 << 15, size=n, dtype=np.uint64)                  # two windows, no carry beyond them
    high = np.zeros((n, 5), dtype=np.uint64)
    high[:, 3] = rng.integers(0, 256, size=n, dtype=np.uint64) << np.uint64(44)    # bits 200 and up only
    high[:, 4] = rng.integers(0, 1 << 40, size=n, dtype=np.uint64)
    mixed = np.where((np.arange(n) % 2 == 0)[:, None], low, high)
    for groups in ("30,3", "20,10,3", "10,10,10,3", "31,2"):
        with V.tuned(ZC_MSM_GROUPS=groups, ZC_MSM_WINDOW=8) as te:
            assert te.msm_plan(n)["window_groups"] == len(groups.split(",")), groups
            for K in (low, high, mixed):
                got, want = te.msm(P, K), oracle.msm_naive_mt(P, K)
                assert oracle.ed_eq(got, want)[0] == 1 and eq(oracle.ed_compress(got)[0], oracle.ed_compress(want)[0]), groups


def _msm_sort_model(K, c, groups=None):
    """numpy model of k_msm_digits + the bucket sort: pairs (window << (c-1) | |d| - 1, i | sign << 31) in
    stable bucket order, the zero digits (key 0xFFFFFFFF) behind every bucket, window by window.
    `groups` (windows per group, top group first): every group of windows sorted on its own -- the same order inside
    the group's own part of the list [w0 n, (w0 + nw) n), its zero digits at the end of that part."""
    n, W, half = len(K), -(-261 // c), 1 << (c - 1)
    key = np.zeros((W, n), dtype=np.int64)
    sign = np.zeros((W, n), dtype=np.uint32)
    for i in range(n):
        v = sum(int(K[i, j]) << (52 * j) for j in range(5))
        carry = 0
        for w in range(W):
            raw = ((v >> (w * c)) & ((1 << c) - 1)) + carry
            carry = 1 if raw > half else 0
            mag = (1 << c) - raw if carry else raw
            key[w, i] = (w << (c - 1)) | (mag - 1) if mag else -1
            sign[w, i] = carry
    idx = np.arange(n, dtype=np.uint32)
    out_k, out_v, tail_k, tail_v = [], [], [], []
    for w in range(W):
        nz = key[w] >= 0
        order = np.argsort(key[w][nz], kind="stable")
        out_k.append(key[w][nz][order].astype(np.uint32))
        out_v.append((idx[nz] | (sign[w][nz] << np.uint32(31)))[order])
        tail_k.append(np.full(int((~nz).sum()), 0xFFFFFFFF, dtype=np.uint32))
        tail_v.append(idx[~nz] | (sign[w][~nz] << np.uint32(31)))       # raw = 2^c: a zero digit that still carries
    if not groups:
        return np.concatenate(out_k + tail_k), np.concatenate(out_v + tail_v)
    assert sum(groups) == W
    parts_k, parts_v, top = [], [], W
    for nw in groups:                                             # top group first = the highest windows
        top -= nw
        parts_k.insert(0, np.concatenate(out_k[top:top + nw] + tail_k[top:top + nw]))
        parts_v.insert(0, np.concatenate(out_v[top:top + nw] + tail_v[top:top + nw]))
    return np.concatenate(parts_k), np.concatenate(parts_v)


@pytest.mark.parametrize("n,c,g,packed,groups", [
    (1000, 5, None, 1, None), (3 * 4096 + 17, 9, None, 1, None), (3 * 4096 + 17, 10, 2, 1, None), (3 * 4096 + 17, 10, 2, 0, None),
    (70001, 13, None, 1, None), (70001, 17, 3, 1, None), (70001, 17, 3, 0, None), (40000, 19, None, 1, None),
    (40000, 19, 2, 0, None), (40000, 19, 2, 2, None), (3 * 8192 + 5, 9, 2, 2, None), (20000, 20, None, 1, None), (9000, 22, 1, 1, None),
    (70001, 17, 3, 1, "9,4,3"), (70001, 17, 3, 0, "15,1"), (3 * 8192 + 5, 9, 2, 2, "10,10,8,1"), (1000, 5, None, 1, "1,52"), (20000, 20, None, 1, "7,7")])
def test_msm_key_sort_is_the_stable_bucket_order(n, c, g, packed, groups):
    """The hand-written per-window LSD counting sort (zc_sort.hip.h) through its test hook: one, two and three
    passes, the one-word and the two-word intermediate of the two-pass sort, partial tiles, several tiles per
    column, skewed digits -- pair for pair the stable sort's output.  With `groups`: every group of windows sorted
    on its own, top group first (the pipeline's order), each in its own part of the list."""
    import torch
    K = V.rand_scalars_np(n, V.SEED + 300 + c, bits=252)
    K[0] = 0
    K[1] = [(1 << 52) - 1] * 5                                    # all 260 bits: the recoding carry reaches the top window
    K[n // 2:n // 2 + n // 8] = K[5]                              # a skewed stretch: equal scalars
    K[-7:, 2:] = 0                                                # short scalars: zero digits in the upper windows
    W = -(-261 // c)
    dK = torch.from_numpy(K.view(np.int64)).cuda()
    out = torch.empty((n * W, 2), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    # 0: two-word intermediate, 2: with tiles of 8192 keys; the hook lives in the -DZC_TEST_HOOKS build only
    with V.tuned(hooks=True, ZC_MSM_SORT_G=g, ZC_MSM_SORT_PACKED="1" if packed == 1 else "0", ZC_MSM_SORT_BIG="1" if packed == 2 else "0", ZC_MSM_GROUPS=groups) as te:
        assert te.lib.zc_test_msm_sort(te.ctx, dK.data_ptr(), n, c, out.data_ptr()) == 0, te.lib.zc_last_error()
    got = out.cpu().numpy().view(np.uint32)
    wk, wv = _msm_sort_model(K, c, [int(x) for x in groups.split(",")] if groups else None)
    assert eq(got[:, 0], wk)
    assert eq(got[:, 1], wv)


def _gpu_naive_msm(eng, P, K):
    """sum_i k_i P_i through separately-tested kernels: batched scalar-mul, then pairwise adds."""
    q = eng.ed_scalar_mul(P, K)
    while len(q) > 1:
        if len(q) & 1:
            q = np.concatenate([q, np.array([V.IDENT_ROW], dtype=np.uint64)])
        q = eng.ed_add(q[0::2], q[1::2])
    return q


@pytest.mark.parametrize("n,bits", [(4096, 249), (5000, 252), ((1 << 16) + 11, 249), (1 << 18, 252)])
def test_msm_bucket_method(eng, oracle, n, bits):
    """Bucket-method shards (window widths 8..14 here): same group element as the reference-op sum."""
    small = V.base_multiples(oracle, 1 << 10, V.SEED + 84)
    P = np.tile(small, ((n >> 10) + 1, 1))[:n].copy()
    K = V.rand_scalars_np(n, V.SEED + 85 + n, bits=bits)
    K[0] = 0
    K[1] = [1, 0, 0, 0, 0]
    K[2] = [(1 << 52) - 1] * 5                                    # all 260 bits set
    P[3] = V.IDENT_ROW
    K[4:4 + 24] = V.raw_scalar_edges(n_random=0)                  # bits >= 2^256, early-stopping patterns
    got = eng.msm(P, K)
    want = oracle.msm_naive_mt(P, K)                              # the reference's own ops, every size
    assert oracle.ed_eq(got, want)[0] == 1
    assert eq(oracle.ed_compress(got)[0], oracle.ed_compress(want)[0])
    assert eq(oracle.ris_compress(got), oracle.ris_compress(want))


def test_msm_window_widths_vs_oracle(eng, oracle):
    """Window widths c = 5..19 (forced with ZC_MSM_WINDOW; the natural choice at 2^16 pairs is 13)
    against the ORACLE's sum of the reference's Mul<Scalar> + Add (edwards.rs:547-561, :465-489)."""
    n = (1 << 16) + 11
    P = eng.ed_mul_base(V.rand_scalars_np(n, V.SEED + 87, bits=249))
    K = V.rand_scalars_np(n, V.SEED + 88, bits=252)
    K[:24] = V.raw_scalar_edges(n_random=0)
    P[30] = V.IDENT_ROW
    want = oracle.msm_naive_mt(P, K)
    wenc = oracle.ed_compress(want)[0]
    for c in (5, 10, 12, 13, 14, 15, 16, 17, 19):
        with V.tuned(ZC_MSM_WINDOW=c) as te:
            got = te.msm(P, K)
        assert oracle.ed_eq(got, want)[0] == 1 and eq(oracle.ed_compress(got)[0], wenc), c


def test_msm_skewed_digit_distributions(eng, oracle):
    """The bucket sums are a segmented reduction of the sorted (bucket, point) list in fixed-length
    runs, so skew costs nothing and must change nothing: every scalar equal (one bucket per window
    holds all n points), scalars of a few bits (most windows empty), a two-valued mix, and run lengths
    from 4 upward (many reduction levels, every edge case of run / bucket alignment)."""
    n = 20000
    P = eng.ed_mul_base(V.rand_scalars_np(n, V.SEED + 91, bits=249))
    rng = np.random.default_rng(V.SEED + 92)
    cases = {}
    K = np.tile(V.rand_scalars_np(1, V.SEED + 93, bits=249), (n, 1))
    cases["all_equal"] = K
    K = np.zeros((n, 5), dtype=np.uint64)
    K[:, 0] = rng.integers(0, 8, size=n, dtype=np.uint64)
    cases["three_bits"] = K
    K = np.tile(V.rand_scalars_np(2, V.SEED + 94, bits=252), (n // 2, 1))
    K[::7] = 0
    cases["two_values_and_zeros"] = K
    K = V.rand_scalars_np(n, V.SEED + 95, bits=249)
    K[:, 4] = 0                                                   # 208-bit scalars: top windows empty
    cases["short"] = K
    for name, K in cases.items():
        want = oracle.msm_naive_mt(P, K)
        wenc = oracle.ed_compress(want)[0]
        for run, c in ((None, None), (4, 8), (5, 11), (7, None), (64, 6), (4096, None)):
            with V.tuned(hooks=True, ZC_MSM_RUN=run, ZC_MSM_WINDOW=c) as te:
                got = te.msm(P, K)
            assert oracle.ed_eq(got, want)[0] == 1 and eq(oracle.ed_compress(got)[0], wenc), (name, run, c)


@pytest.mark.parametrize("knobs", [
    dict(ZC_MSM_AFFINE=0), dict(ZC_MSM_GROUPS="17,4"), dict(ZC_MSM_GROUPS="12,5,4"), dict(ZC_MSM_GROUPS="6,5,5,5"), dict(ZC_MSM_GROUPS="20,1", ZC_MSM_AFFINE=0),
    # path forcers (the -DZC_TEST_HOOKS build reads them; the product takes these paths at other shard sizes)
    dict(ZC_MSM_AFFINE=0, ZC_MSM_FORK=1), dict(ZC_MSM_FORK=0), dict(ZC_MSM_RUN_EDGES=4), dict(ZC_MSM_RUN_EDGES=32, ZC_MSM_RUN=16),
    dict(ZC_MSM_AFFINE_CHUNK=1), dict(ZC_MSM_AFFINE_CHUNK=5, ZC_MSM_SEG=4), dict(ZC_MSM_SEG=64, ZC_MSM_WINDOW=12),
    dict(ZC_MSM_SEG=64, ZC_MSM_WINDOW=6, ZC_MSM_RUN=128), dict(ZC_MSM_GROUPS="8,8,5", ZC_MSM_SEG=4), dict(ZC_MSM_GROUPS="10,11", ZC_MSM_AFFINE_CHUNK=3)],
    ids=lambda k: ",".join("%s=%s" % kv for kv in k.items()))
def test_msm_every_selectable_path_vs_oracle(eng, oracle, knobs):
    """Every MSM path a knob can select -- the product's own (projective 128-byte records at a size where the default is affine,
    the windows in two / three / four groups) and, through the test-hooks build of the same sources, the paths the product takes
    at other shard sizes (normalisation forked or in line, other edge-run lengths, normalisation chunkings and segment lengths) --
    at 2^17 + 333 pairs (where the affine path and the persistent structures are live), against the ORACLE's sum of the
    reference's Mul<Scalar> + Add."""
    n = (1 << 17) + 333
    P = eng.ed_mul_base(V.rand_scalars_np(n, V.SEED + 410, bits=249))
    K = V.rand_scalars_np(n, V.SEED + 411, bits=252)
    K[:24] = V.raw_scalar_edges(n_random=0)
    P[30] = V.IDENT_ROW
    want = oracle.msm_naive_mt(P, K)
    wenc = oracle.ed_compress(want)[0]
    import dusk_zerocaf_amd as z
    hooks = any(k not in ("ZC_MSM_AFFINE", "ZC_MSM_GROUPS", "ZC_MSM_WINDOW") for k in knobs)
    with V.tuned(hooks=hooks, **knobs) as te:
        if "ZC_MSM_GROUPS" in knobs:                              # the split is really in force (it must add up to the windows)
            gw = [int(x) for x in knobs["ZC_MSM_GROUPS"].split(",")]
            plan = te.msm_plan(n)
            assert plan["window_groups"] == len(gw) and plan["windows"] == 21 and plan["group_windows"] == gw and len(plan["group_runs"]) == len(gw)
        if knobs.get("ZC_MSM_WINDOW") == 6:                       # 32 buckets per window: a 64-bucket segment is cut down to the window
            assert te.msm_plan(n)["segment_buckets"] == 32 and te.msm_plan(n)["windows"] == 44
        got = te.msm(P, K)
        assert oracle.ed_eq(got, want)[0] == 1 and eq(oracle.ed_compress(got)[0], wenc), knobs
        for _ in range(2):                                        # and again: the side stream's events and the workspace are reused
            assert eq(te.msm(P, K), got), knobs
        wsmall = oracle.msm_naive_mt(P[:5000], K[:5000])
        if "ZC_MSM_GROUPS" in knobs:
            # a split that does not add up to the windows of THIS shard fails the call -- it is not silently replaced by one group
            with pytest.raises(z.ZerocafHipError, match="ZC_MSM_GROUPS adds up to 21 windows"):
                te.msm(P[:5000], K[:5000])
            with pytest.raises(z.ZerocafHipError):
                te.msm_plan(5000)
            assert eq(te.msm(P, K), got), knobs                   # and the context is still good
        else:
            small = te.msm(P[:5000], K[:5000])                     # and a shard below the affine threshold under the same knobs
            assert oracle.ed_eq(small, wsmall)[0] == 1, knobs
    if "ZC_MSM_GROUPS" not in knobs:
        with V.tuned(hooks=hooks, ZC_MSM_AFFINE=1, **{k: v for k, v in knobs.items() if k != "ZC_MSM_AFFINE"}) as te:
            assert oracle.ed_eq(te.msm(P[:5000], K[:5000]), wsmall)[0] == 1, knobs      # affine records forced on a small shard


def test_product_library_ignores_the_test_only_knobs(eng):
    """libzerocaf_hip.so reads ten knobs; the path forcers exist in the -DZC_TEST_HOOKS build only,
    and the variants that were measured and lost are compile-time macros (no runtime switch in either build)."""
    n = (1 << 17) + 333
    base = eng.msm_plan(n)
    with V.tuned(ZC_MSM_RUN=16, ZC_MSM_SEG=4, ZC_MSM_SORT_PACKED=0, ZC_MSM_REC_STRIDE=96, ZC_MSM_FOLD_QUAD=0, ZC_BALANCE="global") as te:
        assert te.msm_plan(n) == base
    with V.tuned(hooks=True, ZC_MSM_RUN=16, ZC_MSM_SEG=4, ZC_MSM_REC_STRIDE=96) as te:
        p = te.msm_plan(n)
        assert p["run"] == 16 and p["segment_buckets"] == 4 and p["record_stride"] == 128 and p["record_bytes"] == 112     # 27 limb words gathered as seven 16-byte pieces
    import subprocess
    import dusk_zerocaf_amd as z
    import re
    txt = subprocess.run(["strings", z.LIB_PATH], capture_output=True, text=True).stdout
    knobs = sorted(set(re.findall(r"^ZC_[A-Z_0-9]+$", txt, flags=re.M)))
    assert knobs == ["ZC_HOST_CHUNKS", "ZC_INV_CHUNK", "ZC_JACOBI_ROUNDS", "ZC_MSM_AFFINE", "ZC_MSM_GROUPS", "ZC_MSM_WINDOW", "ZC_RCCL_PATH", "ZC_RING_SLOTS",
                     "ZC_RISTRETTO_STRICT", "ZC_SCHED"], knobs


@pytest.mark.parametrize("knobs", [dict(ZC_SCHED="block")], ids=lambda k: ",".join("%s=%s" % kv for kv in k.items()))
def test_strict_scalar_mul_every_selectable_schedule_vs_oracle(eng, oracle, knobs):
    """The strict kernel behind ZC_SCHED=block -- one workgroup per 256 elements with the block-local ranking
    (`k_ed_scalar_mul` at >= 2^17 elements) -- every (X:Y:Z:T) limb against the oracle's double_and_add
    (edwards.rs:102-120), raw scalars >= 2^256 included, at 2^17 + 333 and at a size below the persistent-wave threshold."""
    n = (1 << 17) + 333
    P = eng.ed_mul_base(V.rand_scalars_np(n, V.SEED + 420, bits=249))
    K = V.rand_scalars_np(n, V.SEED + 421, bits=252)
    _edge_scalars(K)
    raw = V.raw_scalar_edges(n_random=40)
    K[100:100 + len(raw)] = raw
    K[n - len(raw):] = raw
    want = oracle.mt(oracle.ed_scalar_mul, P, K)
    assert eq(eng.ed_scalar_mul(P, K), want)
    with V.tuned(**knobs) as te:
        assert eq(te.ed_scalar_mul(P, K), want), knobs
        m = (1 << 14) + 77                                         # block-shaped launch
        assert eq(te.ed_scalar_mul(P[n - m:], K[n - m:]), want[n - m:]), knobs
        import torch
        dP, dK = (torch.from_numpy(a.view(np.int64)).cuda() for a in (P, K))
        assert eq(te.ed_scalar_mul(dP, dK).cpu().numpy().view(np.uint64), want), knobs


@pytest.mark.parametrize("knobs", [dict(ZC_RISTRETTO_STRICT=1)], ids=lambda k: ",".join("%s=%s" % kv for kv in k.items()))
def test_ristretto_roundtrip_strict_sequence_kernel_vs_oracle(eng, oracle, knobs):
    """ZC_RISTRETTO_STRICT=1 runs the fused config-4 kernel on the reference's formula sequence (`k_ris_roundtrip_mul`,
    ristretto.rs:96-154 -> edwards.rs:102-120 -> ristretto.rs:398-425) instead of the windowed core: every output byte
    and accept flag of 2^17 + 333 round trips with about 1 % undecodable inputs against the oracle, and against the
    default kernel's bytes."""
    n = (1 << 17) + 333
    P = eng.ed_mul_base(V.rand_scalars_np(n, V.SEED + 430, bits=249))
    enc = eng.ris_compress(P)
    rng = np.random.default_rng(V.SEED + 431)
    bad = rng.choice(n, size=n // 100, replace=False)
    enc[bad, :8] ^= rng.integers(1, 256, size=(len(bad), 8), dtype=np.uint8)
    enc[7, 31] |= 0x80
    K = V.rand_scalars_np(n, V.SEED + 432, bits=252)
    _edge_scalars(K)
    raw = V.raw_scalar_edges(n_random=40)
    K[200:200 + len(raw)] = raw
    wout, wok = oracle.mt(oracle.ris_roundtrip_mul, enc, K)
    assert 0 < (wok == 0).sum() < n // 50
    dout, dok = eng.ris_roundtrip_mul(enc, K)
    assert eq(dout, wout) and eq(dok, wok)
    with V.tuned(**knobs) as te:
        out, ok = te.ris_roundtrip_mul(enc, K)
        assert eq(out, wout) and eq(ok, wok), knobs
        out, ok = te.ris_roundtrip_mul(enc[:3000], K[:3000])
        assert eq(out, wout[:3000]) and eq(ok, wok[:3000]), knobs


_CONFIG5_ORACLE = {}                                              # the oracle's sum over shard 0 of config 5 (22 s of 16 threads): computed once


def test_msm_config5_shard_2_21_vs_oracle(eng, oracle, config5):
    """The per-GPU shard of BASELINE configs[4] (2^24 pairs over 8 GPUs = 2^21 per GPU: rank 0's range of the config-5
    batch), distinct points, S249 scalars, against the oracle's naive sum on all host cores."""
    n = 1 << 21
    P = config5[1][:n].cpu().numpy().view(np.uint64)
    K = config5[2][:n].cpu().numpy().view(np.uint64)
    assert eng.msm_plan(n)["window_groups"] == 3                   # the shard's default: its windows in three groups, chains on the side stream
    got = eng.msm(P, K)
    want = _CONFIG5_ORACLE["shard0"] = oracle.msm_naive_mt(P, K)
    assert oracle.ed_eq(got, want)[0] == 1
    assert eq(oracle.ed_compress(got)[0], oracle.ed_compress(want)[0])
    assert eq(oracle.ris_compress(got), oracle.ris_compress(want))
    with V.tuned(ZC_MSM_GROUPS="1") as te:                         # one group (the pipeline of rounds 2-3): the same point
        assert te.msm_plan(n)["window_groups"] == 1 and oracle.ed_eq(te.msm(P, K), want)[0] == 1
    K2 = V.rand_scalars_np(n, V.SEED + 96, bits=252)               # raw 252-bit patterns, equal stretches, zeros: every group has edges to close
    K2[n // 3:n // 3 + 50000] = K2[7]
    K2[::1001] = 0
    assert oracle.ed_eq(eng.msm(P, K2), oracle.msm_naive_mt(P, K2))[0] == 1


@pytest.fixture(scope="module")
def config5(eng):
    """BASELINE configs[4] as stated: 2^24 distinct points (r_i * B, fixed-base kernel), S249 scalars, device-resident,
    and the bucket-method sum over the whole batch."""
    import torch
    n = 1 << 24
    P = eng.ed_mul_base(torch.from_numpy(V.rand_scalars_np(n, V.SEED + 96, bits=249).view(np.int64)).cuda())
    K = torch.from_numpy(V.rand_scalars_np(n, V.SEED + 97, bits=249).view(np.int64)).cuda()
    whole = eng.msm(P, K)
    torch.cuda.synchronize()
    yield n, P, K, whole
    del P, K
    torch.cuda.empty_cache()


def test_msm_config5_full_2_24_is_the_ordered_fold_of_its_eight_shards(eng, oracle, config5):
    """The exact 8-GPU decomposition on one GPU: zc_msm over all 2^24 pairs must be the same group element as
    zc_ed_fold_ordered of the eight zc_msm_partial results over the contiguous 2^21 ranges, in rank order (the
    reference's unified addition, edwards.rs:465-489) -- compared with the engine's == and by encodings."""
    import torch
    n, P, K, whole = config5
    per = n // 8
    parts = torch.cat([eng.msm_partial(P[r * per:(r + 1) * per], K[r * per:(r + 1) * per]) for r in range(8)])
    folded = eng.ed_fold_ordered(parts)
    torch.cuda.synchronize()
    folded = folded.cpu().numpy().view(np.uint64)
    assert eng.ed_eq(whole, folded)[0] == 1 and oracle.ed_eq(whole, folded)[0] == 1
    assert eq(eng.ed_compress(whole)[0], eng.ed_compress(folded)[0])
    assert eq(oracle.ed_compress(whole)[0], oracle.ed_compress(folded)[0])
    assert eq(oracle.ris_compress(whole), oracle.ris_compress(folded))
    # the fold itself, limb for limb: the oracle's unified additions over the same eight partials in the same order
    rows = parts.cpu().numpy().view(np.uint64)
    acc = rows[0:1].copy()
    for r in range(1, 8):
        acc = oracle.ed_add(acc, rows[r:r + 1])
    assert eq(folded, acc)


def test_msm_config5_full_2_24_vs_oracle(eng, oracle, config5):
    """... and against the oracle's sum of the reference's Mul<Scalar> + Add over ALL 2^24 pairs (about three
    minutes on 16 host threads): BASELINE configs[4] at its full statement."""
    n, P, K, whole = config5
    hP, hK = P.cpu().numpy().view(np.uint64), K.cpu().numpy().view(np.uint64)
    per = n // 8                                                   # shard 0's sum may already be there (the shard test): the reference's Add joins the two
    first = _CONFIG5_ORACLE.get("shard0")
    if first is None:
        first = oracle.msm_naive_mt(hP[:per], hK[:per])
    want = oracle.ed_add(first, oracle.msm_naive_mt(hP[per:], hK[per:]))
    assert oracle.ed_eq(whole, want)[0] == 1
    assert eq(oracle.ed_compress(whole)[0], oracle.ed_compress(want)[0])
    assert eq(oracle.ris_compress(whole), oracle.ris_compress(want))


def test_empty_batches_and_in_context_sharding(eng, oracle):
    """n = 0 is a no-op for every call shape; a context over two device slots (the same GPU
    twice here) shards host batches into contiguous ranges and gives identical results."""
    import dusk_zerocaf_amd as z
    e5, e20, e32 = (np.zeros((0, w), dtype=np.uint64) for w in (5, 20, 32))
    assert eng.fe_mul(e5, e5).shape == (0, 5) and eng.ed_scalar_mul(e20, e5).shape == (0, 20)
    assert eng.ris_compress(e20).shape == (0, 32) and eng.fe_invert(e5)[0].shape == (0, 5)
    assert eq(eng.msm(e20, e5), np.array([V.IDENT_ROW], dtype=np.uint64))
    two = z.Engine([0, 0])
    try:
        n = 20001                                                  # odd: shards of 10001 and 10000
        P = V.base_multiples(oracle, 1024, V.SEED + 120)
        P = np.tile(P, (20, 1))[:n].copy()
        K = V.rand_scalars_np(n, V.SEED + 121, bits=252)
        assert eq(two.ed_scalar_mul(P, K), eng.ed_scalar_mul(P, K))
        a, b = V.rand_fe_np(n, V.SEED + 122), V.rand_fe_np(n, V.SEED + 123)
        assert eq(two.fe_mul(a, b), oracle.fe_mul(a, b))
        inv, ok = two.fe_invert(a)
        assert eq(inv, eng.fe_invert(a)[0]) and ok.all()
        enc = eng.ris_compress(P)
        assert eq(two.ris_roundtrip_mul(enc, K)[0], eng.ris_roundtrip_mul(enc, K)[0])
        m2, m1 = two.msm(P, K), eng.msm(P, K)
        assert oracle.ed_eq(m2, m1)[0] == 1 and eq(oracle.ed_compress(m2)[0], oracle.ed_compress(m1)[0])
    finally:
        two.close()


def test_device_resident_buffers(eng, oracle):
    """Same results when the caller hands over HBM-resident buffers (torch tensors) and a
    borrowed stream: the zero-copy path the bench uses."""
    import torch
    n = 3000
    P = V.base_multiples(oracle, n, V.SEED + 90)
    K = V.rand_scalars_np(n, V.SEED + 91, bits=252)
    dP = torch.from_numpy(P.view(np.int64)).cuda()
    dK = torch.from_numpy(K.view(np.int64)).cuda()
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        out = eng.ed_scalar_mul(dP, dK)
        torch.cuda.synchronize()
        assert eq(out.cpu().numpy().view(np.uint64), oracle.ed_scalar_mul(P, K))
        a, b = V.rand_fe_np(n, V.SEED + 92), V.rand_fe_np(n, V.SEED + 93)
        da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
        prod = eng.fe_mul(da, db)
        torch.cuda.synchronize()
        assert eq(prod.cpu().numpy().view(np.uint64), oracle.fe_mul(a, b))
        # 8-byte-aligned (not 16) device views take the non-staged kernels
        prod2 = eng.fe_mul(da[1:], db[1:])
        sq2 = eng.fe_square(da[1:])
        inv_in = torch.from_numpy(V.rand_fe_np(1 << 18, V.SEED + 94).view(np.int64)).cuda()
        inv, okm = eng.fe_invert(inv_in)                          # chunked inversion on device buffers
        torch.cuda.synchronize()
        assert eq(prod2.cpu().numpy().view(np.uint64), oracle.fe_mul(a[1:], b[1:]))
        assert eq(sq2.cpu().numpy().view(np.uint64), oracle.fe_square(a[1:]))
        hinv = inv_in.cpu().numpy().view(np.uint64)
        assert eq(inv.cpu().numpy().view(np.uint64)[:4096], oracle.fe_invert(hinv[:4096])[0]) and okm.cpu().numpy().all()
        with pytest.raises(Exception):
            eng.fe_mul(da, b)                                     # host/device mix is refused
    finally:
        eng.use_own_stream()


def test_device_resident_windowed_core_chunks(eng, oracle):
    """Device-resident batches larger than one launch of the windowed core (786 432 lanes of table
    scratch) go chunk by chunk, alternating between the launch stream and a forked stream.  2^21 + 1000
    units = three chunks with a ragged tail: the fused Ristretto round trip (bytes + ok mask, ~1 %
    undecodable inputs) and ZC_SCALAR_MUL_FAST (as encodings) against the oracle on slices from every
    chunk and across both chunk boundaries; repeated calls reuse the scratch."""
    import torch
    import dusk_zerocaf_amd as z
    n = (1 << 21) + 1000
    small = V.base_multiples(oracle, 1 << 11, V.SEED + 160)
    enc = np.tile(oracle.ris_compress(small), (n // (1 << 11) + 1, 1))[:n].copy()
    enc[::97, 31] |= 0x80
    K = V.rand_scalars_np(n, V.SEED + 161, bits=252)
    _edge_scalars(K)
    d_enc, dK = torch.from_numpy(enc).cuda(), torch.from_numpy(K.view(np.int64)).cuda()
    chunk = 786432
    sel = np.r_[0:300, chunk - 300:chunk + 300, 2 * chunk - 300:2 * chunk + 300, n - 300:n]
    wout, wok = oracle.mt(oracle.ris_roundtrip_mul, enc[sel], K[sel])
    for _ in range(2):
        out, ok = eng.ris_roundtrip_mul(d_enc, dK)
        torch.cuda.synchronize()
        assert eq(out.cpu().numpy()[sel], wout) and eq(ok.cpu().numpy()[sel], wok)
    assert (wok == 0).sum() > 5
    P = np.tile(small, (n // (1 << 11) + 1, 1))[:n].copy()
    dP = torch.from_numpy(P.view(np.int64)).cuda()
    Q = eng.ed_scalar_mul(dP, dK, flags=z.FAST)
    torch.cuda.synchronize()
    want = oracle.mt(oracle.ed_scalar_mul, P[sel], K[sel])
    assert eq(oracle.ed_compress(Q.cpu().numpy().view(np.uint64)[sel])[0], oracle.ed_compress(want)[0])


def test_host_batches_move_in_chunks(eng, oracle):
    """Host (numpy) batches of the scalar-mul family pass through the device in chunks that
    overlap copies with kernels; any chunking must give the bytes of the one-piece run, with
    ragged last chunks, optional masks and two device slots."""
    import dusk_zerocaf_amd as z
    n = (1 << 18) + 777                                            # default policy: 2^17 + 2^17 + 777
    base = V.base_multiples(oracle, 2048, V.SEED + 130)
    P = np.tile(base, (n // 2048 + 1, 1))[:n].copy()
    K = V.rand_scalars_np(n, V.SEED + 131, bits=252)
    enc = eng.ris_compress(P)
    enc[5::1001, 31] |= 0x80                                       # some undecodable encodings
    with V.tuned(ZC_HOST_CHUNKS=1) as te:
        q1 = te.ed_scalar_mul(P, K)
        r1, ok1 = te.ris_roundtrip_mul(enc, K)
        c1 = te.ed_mul_by_cofactor(P)
        f1 = te.fe_mul(K, K)
    sub = np.r_[0:256, (1 << 17) - 128:(1 << 17) + 128, n - 300:n]   # across chunk seams and the tail
    assert eq(q1[sub], oracle.ed_scalar_mul(P[sub], K[sub]))
    for chunks in (None, "3", "7"):
        with V.tuned(ZC_HOST_CHUNKS=chunks) as te:
            assert eq(te.ed_scalar_mul(P, K), q1)
            fast = te.ed_scalar_mul(P, K, flags=z.FAST)          # same group element: compare encodings
            assert eq(te.ed_compress(fast[sub])[0], te.ed_compress(q1[sub])[0])
            r, ok = te.ris_roundtrip_mul(enc, K)
            assert eq(r, r1) and eq(ok, ok1) and not ok[5] and ok[6]
            assert eq(te.ed_mul_by_cofactor(P), c1)
            assert eq(te.fe_mul(K, K), f1)
    with V.tuned(devices=[0, 0], ZC_HOST_CHUNKS=5) as two:
        assert eq(two.ed_scalar_mul(P, K), q1)


def test_concurrent_callers_share_a_context(eng, oracle):
    """The reference's functions are pure and re-entrant; the ABI promises the same
    for calls on one context from several threads (ctypes drops the GIL during a call)."""
    import threading
    n = 6000
    P = V.base_multiples(oracle, n, V.SEED + 140)
    K = V.rand_scalars_np(n, V.SEED + 141, bits=252)
    a, b = V.rand_fe_np(n, V.SEED + 142), V.rand_fe_np(n, V.SEED + 143)
    want = {"sm": oracle.ed_scalar_mul(P, K), "mul": oracle.fe_mul(a, b), "pow2": oracle.ed_mul_by_pow_2(P, 7),
            "msm": oracle.ed_compress(oracle.msm_naive(P[:4500], K[:4500]))[0]}
    got, errs = {}, []

    def run(name, f, reps):
        try:
            for _ in range(reps):
                got[name] = f()
        except Exception as e:                                     # noqa: BLE001
            errs.append((name, e))

    ts = [threading.Thread(target=run, args=("sm", lambda: eng.ed_scalar_mul(P, K), 3)),
          threading.Thread(target=run, args=("mul", lambda: eng.fe_mul(a, b), 40)),
          threading.Thread(target=run, args=("pow2", lambda: eng.ed_mul_by_pow_2(P, 7), 6)),
          threading.Thread(target=run, args=("msm", lambda: eng.ed_compress(eng.msm(P[:4500], K[:4500]))[0], 6))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for name in want:
        assert eq(got[name], want[name]), name


def test_msm_exchange_inside_the_library(eng, oracle):
    """SURVEY 8e / BASELINE configs[4]: partial sums stay in HBM, are gathered on the device (peer
    copies inside one process, ncclAllGather between processes) and folded in rank order by ONE
    kernel.  Exercised here on one GPU: two device slots of the same GPU, the library's own RCCL
    communicator with world size 1, and torch.distributed's `nccl` backend with world size 1."""
    import torch
    import torch.distributed as dist
    import dusk_zerocaf_amd as z
    from dusk_zerocaf_amd import distributed as D
    n = 9000
    P = eng.ed_mul_base(V.rand_scalars_np(n, V.SEED + 170, bits=249))
    K = V.rand_scalars_np(n, V.SEED + 171, bits=252)
    want = oracle.msm_naive_mt(P, K)
    wenc = oracle.ed_compress(want)[0]
    # ordered fold: limb-exact against the oracle's sequential unified adds, host and device pointers
    parts = oracle.ed_scalar_mul(P[:7], K[:7])
    acc = parts[0:1]
    for i in range(1, 7):
        acc = oracle.ed_add(acc, parts[i:i + 1])
    assert eq(eng.ed_fold_ordered(parts), acc) and eq(eng.ed_fold_ordered(parts[:1]), parts[:1])
    dparts = torch.from_numpy(parts.view(np.int64)).cuda()
    assert eq(eng.ed_fold_ordered(dparts).cpu().numpy().view(np.uint64), acc)
    # partial sum left on the device
    dP, dK = torch.from_numpy(P.view(np.int64)).cuda(), torch.from_numpy(K.view(np.int64)).cuda()
    part = eng.msm_partial(dP, dK)
    torch.cuda.synchronize()
    assert part.is_cuda and eq(oracle.ed_compress(part.cpu().numpy().view(np.uint64))[0], wenc)
    # two device slots in one process: per-slot partials -> peer/device copies -> fold kernel on slot 0
    two = z.Engine([0, 0])
    try:
        got = two.msm(P, K)
        halves = [oracle.msm_naive_mt(P[:4500], K[:4500]), oracle.msm_naive_mt(P[4500:], K[4500:])]
        assert oracle.ed_eq(got, want)[0] == 1 and eq(oracle.ed_compress(got)[0], wenc)
        assert oracle.ed_eq(got, oracle.ed_add(halves[0], halves[1]))[0] == 1
    finally:
        two.close()
    # the library's own RCCL communicator (world size 1 executes ncclAllGather + the fold)
    one = z.Engine()
    try:
        D.init_library_comm(one)
        got = D.msm_sharded_rccl(one, P, K)
        assert eq(oracle.ed_compress(got)[0], wenc)
        got = one.msm_sharded(dP, dK)                              # device-resident shard
        assert eq(oracle.ed_compress(got)[0], wenc)
        assert eq(one.msm_sharded(P[:0], K[:0]), np.array([V.IDENT_ROW], dtype=np.uint64))
        one.comm_destroy()
    finally:
        one.close()
    # torch.distributed `nccl` (= RCCL) moving device rows, fold on the device
    import os
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rows = D.all_gather_rows(part)
        assert rows.is_cuda and rows.shape == (1, 20)
        got = D.msm_sharded(dP, dK, None, engine=eng)
        assert eq(oracle.ed_compress(got)[0], wenc)
    finally:
        dist.destroy_process_group()


def test_stream_switch_and_torch_stream_following(eng, oracle):
    """Calls on torch tensors follow torch's current stream unless a stream was pinned; switching
    streams orders earlier work (and shared scratch) with an event instead of a host sync."""
    import torch
    import dusk_zerocaf_amd as z
    n = 1 << 15
    P = V.base_multiples(oracle, 512, V.SEED + 180)
    P = np.tile(P, (n // 512, 1))
    K = V.rand_scalars_np(n, V.SEED + 181, bits=252)
    want = oracle.mt(oracle.ed_scalar_mul, P, K)
    wenc = oracle.ris_compress(want)
    e = z.Engine()
    try:
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):                              # inputs produced on a side stream
            dP = torch.from_numpy(P.view(np.int64)).cuda(non_blocking=True)
            dK = torch.from_numpy(K.view(np.int64)).cuda(non_blocking=True)
            out = e.ed_scalar_mul(dP, dK)                          # follows `side`
            enc1 = e.ris_compress(out)
        side.synchronize()
        assert eq(out.cpu().numpy().view(np.uint64), want) and eq(enc1.cpu().numpy(), wenc)
        # back-to-back launches that share the window-table scratch, on two different streams, no sync between
        encs = oracle.ris_compress(P)
        denc = torch.from_numpy(encs).cuda()
        torch.cuda.synchronize()
        r1, ok1 = e.ris_roundtrip_mul(denc, dK)                    # current (default) stream
        with torch.cuda.stream(side):
            r2, ok2 = e.ris_roundtrip_mul(denc, dK)                # switches to `side`: ordered by an event
        torch.cuda.synchronize()
        wout, wok = oracle.mt(oracle.ris_roundtrip_mul, encs, K)
        assert eq(r1.cpu().numpy(), wout) and eq(r2.cpu().numpy(), wout) and eq(ok1.cpu().numpy(), wok)
    finally:
        e.close()


def test_scalar_from_bytes_failed_rows_are_zero(eng, oracle):
    raw = np.frombuffer(pm.L.to_bytes(32, "little") + (pm.L - 1).to_bytes(32, "little") + b"\xff" * 32, dtype=np.uint8).reshape(3, 32)
    out, ok = eng.sc_from_bytes(raw)
    assert ok.tolist() == [0, 1, 0] and not out[0].any() and not out[2].any() and eq(out[1], V.limbs_array([pm.L - 1])[0])
    wout, wok = oracle.sc_from_bytes(raw)
    assert eq(out, wout) and eq(ok, wok)

"""One set of assertions against tests/golden/bulk_vectors.json (independent big-int model
vectors), run against any backend exposing the batch API: the C oracle (CPU tier) and the HIP
engine (GPU tier)."""
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    with open(os.path.join(_HERE, "golden", "bulk_vectors.json")) as f:
        return json.load(f)


def u64(x):
    return np.array(x, dtype=np.uint64)


def u8(x):
    return np.array(x, dtype=np.uint8)


def check_backend(be, g):
    """`be` has fe_add(a,b)... returning arrays (engine or oracle module)."""
    eq = lambda a, b: np.array_equal(np.asarray(a), np.asarray(b))
    fe = g["fe"]
    a, b = u64(fe["a"]), u64(fe["b"])
    assert eq(be.fe_add(a, b), u64(fe["add"])) and eq(be.fe_sub(a, b), u64(fe["sub"]))
    assert eq(be.fe_mul(a, b), u64(fe["mul"])) and eq(be.fe_square(a), u64(fe["square"]))
    assert eq(be.fe_neg(a), u64(fe["neg"])) and eq(be.fe_half(a), u64(fe["half"]))
    inv, ok = be.fe_invert(a)
    assert eq(inv, u64(fe["inv"])) and eq(ok, u8(fe["inv_ok"]))
    assert eq(be.fe_is_positive(a), u8(fe["is_positive"])) and eq(be.fe_legendre_symbol(a), u8(fe["legendre"]))
    assert eq(be.fe_to_bytes(a), u8(fe["to_bytes"])) and eq(be.fe_from_bytes(u8(fe["to_bytes"])), a)
    r, sq = be.fe_sqrt_ratio_i(a[:96], b[:96])
    assert eq(sq, u8(fe["sqrt_ratio_was_square"])) and eq(r, u64(fe["sqrt_ratio"]))
    for sign, key in ((0, "mod_sqrt0"), (1, "mod_sqrt1")):
        r, ok = be.fe_mod_sqrt(a[:96], sign)
        assert eq(ok, u8(fe["mod_sqrt_ok"])) and eq(r, u64(fe[key]))
    sc = g["sc"]
    a, b = u64(sc["a"]), u64(sc["b"])
    assert eq(be.sc_add(a, b), u64(sc["add"])) and eq(be.sc_sub(a, b), u64(sc["sub"]))
    assert eq(be.sc_mul(a, b), u64(sc["mul"])) and eq(be.sc_square(a), u64(sc["square"])) and eq(be.sc_neg(a), u64(sc["neg"]))
    ed = g["ed"]
    p, q, k = u64(ed["p"]), u64(ed["q"]), u64(ed["k"])
    assert eq(be.ed_add(p, q), u64(ed["add"])) and eq(be.ed_sub(p, q), u64(ed["sub"]))
    assert eq(be.ed_double(p), u64(ed["double"])) and eq(be.ed_neg(p), u64(ed["neg"]))
    sm = be.ed_scalar_mul(p, k)
    assert eq(sm, u64(ed["scalar_mul"]))                              # strict (X:Y:Z:T) limbs
    xy, ok = be.ed_to_affine(p)
    assert ok.all() and eq(np.asarray(xy).reshape(-1, 5), u64(ed["affine"]))
    c, ok = be.ed_compress(p)
    assert ok.all() and eq(c, u8(ed["compress"]))
    c, ok = be.ed_compress(sm)
    assert ok.all() and eq(c, u8(ed["scalar_mul_compress"]))
    d, ok = be.ed_decompress(u8(ed["decompress_in"]))
    assert ok.all() and eq(d, u64(ed["decompress"]))
    rs = g["ris"]
    enc = u8(rs["compress"])
    assert eq(be.ris_compress(p), enc)
    d, ok = be.ris_decompress(enc)
    assert ok.all() and eq(d, u64(rs["decompress_of_compress"]))
    d, ok = be.ris_decompress(u8(rs["raw"]))
    assert eq(ok, u8(rs["raw_ok"])) and eq(d, u64(rs["raw_points"]))
    out, ok = be.ris_roundtrip_mul(enc, k)
    assert ok.all() and eq(out, u8(rs["roundtrip_mul"]))
    assert eq(be.ris_elligator(u64(rs["elligator_r0"])), u64(rs["elligator"]))
    rk = g["raw_scalar"]                                              # raw scalars >= 2^256 (early-stopping loop test)
    rp, rkk = u64(rk["p"]), u64(rk["k"])
    assert eq(be.ed_scalar_mul(rp, rkk), u64(rk["scalar_mul"]))
    out, ok = be.ris_roundtrip_mul(be.ris_compress(rp), rkk)
    assert ok.all() and eq(out, u8(rk["roundtrip_mul"]))
    # rows beside the default path (SURVEY 8a S-x / F8 / E-x), same independent model
    sr = g["sc_rows"]
    a, e, raw = u64(sr["a"]), u64(sr["e"]), u64(sr["raw"])
    assert eq(be.sc_half(a), u64(sr["half"])) and eq(be.sc_pow(a, e), u64(sr["pow"]))
    for s, want in zip(sr["shifts"], sr["shr"]):
        assert eq(be.sc_shr(raw, s), u64(want)), s
    assert eq(be.sc_into_bits(raw), u8(sr["bits"]))
    for w, want in zip(sr["naf_widths"], sr["naf"]):
        assert eq(be.sc_compute_naf(u64(sr["naf_in"]), w), np.array(want, dtype=np.int8)), w
    r, sq = be.fe_inv_sqrt(u64(fe["a"])[:96])
    assert eq(sq, u8(fe["inv_sqrt_was_square"])) and eq(r, u64(fe["inv_sqrt"]))
    er = g["ed_rows"]
    assert eq(be.ed_coset4(p[:32]), u64(er["coset4"]))
    pp, pq, pk = u64(er["proj_p"]), u64(er["proj_q"]), u64(er["proj_k"])
    assert eq(be.proj_add(pp, pq), u64(er["proj_add"])) and eq(be.proj_double(pp), u64(er["proj_double"]))
    assert eq(be.proj_neg(pp), u64(er["proj_neg"])) and eq(be.proj_sub(pp, pq), u64(er["proj_sub"]))
    assert eq(be.proj_scalar_mul(pp, pk), u64(er["proj_scalar_mul"]))
    assert eq(be.proj_is_valid(u64(er["proj_valid_in"])), u8(er["proj_valid"]))
    pe = be.proj_eq(be.proj_add(pp, pq), be.proj_add(pq, pp))
    assert np.asarray(pe[0] if isinstance(pe, tuple) else pe).all()

#!/usr/bin/env python3
"""Randomised soak: large seeded batches through the HIP engine vs the CPU oracle (threads).
Not part of the test suite; run on the GPU box for extra confidence."""
import concurrent.futures as cf
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dusk_zerocaf_amd as z
from oracle import zc_ref, pymodel as pm
from tests import vectors as V

def par(fn, n, *arrs):
    th = min(16, os.cpu_count() or 1)
    idx = np.array_split(np.arange(n), th)
    with cf.ThreadPoolExecutor(th) as ex:
        parts = list(ex.map(lambda ix: fn(*[a[ix] for a in arrs]), idx))
    if isinstance(parts[0], tuple):
        return tuple(np.concatenate([p[j] for p in parts]) for j in range(len(parts[0])))
    return np.concatenate(parts)

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = z.Engine()
zc_ref.build()
t0 = time.time()
n = 1 << 16
base = np.tile(np.array(sum(pm.pt_limbs(pm.BASEPOINT), []), dtype=np.uint64), (n, 1))
P = eng.ed_scalar_mul(base, V.rand_scalars_np(n, seed * 100 + 1, bits=249))
for bits in (249, 252):
    K = V.rand_scalars_np(n, seed * 100 + bits, bits=bits)
    got = eng.ed_scalar_mul(P, K)
    want = par(zc_ref.ed_scalar_mul, n, P, K)
    assert np.array_equal(got, want), "scalar_mul bits=%d" % bits
    enc = eng.ris_compress(got)
    assert np.array_equal(enc, par(zc_ref.ris_compress, n, got)), "ris_compress"
    out, ok = eng.ris_roundtrip_mul(enc, K)
    wout, wok = par(zc_ref.ris_roundtrip_mul, n, enc, K)
    assert np.array_equal(out, wout) and np.array_equal(ok, wok), "roundtrip"
    ec, eok = eng.ed_compress(got)
    wec, weok = par(zc_ref.ed_compress, n, got)
    assert np.array_equal(ec, wec) and np.array_equal(eok, weok), "ed_compress"
    dd, dok = eng.ed_decompress(ec)
    wdd, wdok = par(zc_ref.ed_decompress, n, ec)
    assert np.array_equal(dd, wdd) and np.array_equal(dok, wdok), "ed_decompress"
m = 1 << 21
a, b = V.rand_fe_np(m, seed * 100 + 7), V.rand_fe_np(m, seed * 100 + 8)
assert np.array_equal(eng.fe_mul(a, b), par(zc_ref.fe_mul, m, a, b))
assert np.array_equal(eng.fe_square(a), par(zc_ref.fe_square, m, a))
sa, sb = V.rand_fe_np(m, seed * 100 + 9, pm.L), V.rand_fe_np(m, seed * 100 + 10, pm.L)
assert np.array_equal(eng.sc_mul(sa, sb), par(zc_ref.sc_mul, m, sa, sb))
inv, ok = eng.fe_invert(a[: 1 << 18])
winv, wok = par(zc_ref.fe_invert, 1 << 18, a[: 1 << 18])
assert np.array_equal(inv, winv) and np.array_equal(ok, wok)
# batched division / affine conversion (strided Montgomery-trick chunks over the division-step inversion), ragged sizes
rng = np.random.default_rng(seed)
md = (1 << 18) + int(rng.integers(1, 1 << 17))
q, qok = eng.fe_div(a[:md], b[:md]); wq, wqok = par(zc_ref.fe_div, md, a[:md], b[:md])
assert np.array_equal(q, wq) and np.array_equal(qok, wqok), "fe_div"
na = int(rng.integers(1 << 12, 1 << 16))
xy, aok = eng.ed_to_affine(P[:na]); wxy, waok = par(zc_ref.ed_to_affine, na, P[:na])
assert np.array_equal(xy, wxy) and np.array_equal(aok, waok), "ed_to_affine"
# stand-alone point kernels (plain-domain formulas, multiplication-free negation, == / is_valid on plain coordinates)
Qs = np.roll(P, 3, axis=0)
assert np.array_equal(eng.ed_add(P, Qs), par(zc_ref.ed_add, n, P, Qs)) and np.array_equal(eng.ed_sub(P, Qs), par(zc_ref.ed_sub, n, P, Qs))
assert np.array_equal(eng.ed_double(P), par(zc_ref.ed_double, n, P)) and np.array_equal(eng.ed_neg(P), zc_ref.ed_neg(P))
mix = P.copy(); mix[::3] = Qs[::3]; mix[5, 10:15] = 0
assert np.array_equal(eng.ed_eq(P, mix), (zc_ref.ed_eq(P, mix) == 1).astype(np.uint8)) and np.array_equal(eng.ris_eq(P, mix), zc_ref.ris_eq(P, mix))
bad = P.copy(); bad[::4, 0] ^= np.uint64(1)
assert np.array_equal(eng.ed_is_valid(bad), zc_ref.ed_is_valid(bad))
# rows beside the default path: scalar recoders, Half / Pow / Shr, inv_sqrt, coset4, ProjectivePoint ops
ms = 1 << 12
assert np.array_equal(eng.sc_half(sa[:ms]), zc_ref.sc_half(sa[:ms]))
assert np.array_equal(eng.sc_pow(sa[:256], sb[:256]), zc_ref.sc_pow(sa[:256], sb[:256]))
rawk = V.rand_scalars_np(ms, seed * 100 + 40, bits=260)
sh = int(rng.integers(0, 256))
assert np.array_equal(eng.sc_shr(rawk, sh), zc_ref.sc_shr(rawk, sh))
assert np.array_equal(eng.sc_into_bits(rawk), zc_ref.sc_into_bits(rawk))
wdt = int(rng.choice([0, 2, 3, 4, 5, 6, 7]))
both = np.concatenate([sa[:ms], rawk])
assert np.array_equal(eng.sc_compute_naf(both, wdt), par(zc_ref.sc_compute_naf, len(both), both) if wdt == 0 else zc_ref.sc_compute_naf(both, wdt)), "naf width %d" % wdt
# window_naf_mul in one launch: the point (sum_i d_i 2^i) B of the ORACLE's digits, canonical and raw scalars alike
wn = int(rng.integers(2, 8))
vals = V.limbs_array([sum(int(dg) << i for i, dg in enumerate(row)) % pm.L for row in zc_ref.sc_compute_naf(both, wn).astype(np.int64)])
baseb = np.tile(base[:1], (len(both), 1))
assert zc_ref.ed_eq(eng.ed_mul_base_wnaf(both, wn), par(zc_ref.ed_scalar_mul, len(both), baseb, vals)).all(), "window_naf_mul width %d" % wn
isq, sq = eng.fe_inv_sqrt(a[:ms]); wisq, wsq = zc_ref.fe_inv_sqrt(a[:ms])
assert np.array_equal(isq, wisq) and np.array_equal(sq, wsq)
assert np.array_equal(eng.ed_coset4(P[:ms]), zc_ref.ed_coset4(P[:ms]))
Pj, Qj = np.ascontiguousarray(P[:ms, :15]), np.ascontiguousarray(P[ms:2 * ms, :15])
assert np.array_equal(eng.proj_sub(Pj, Qj), zc_ref.proj_sub(Pj, Qj)) and np.array_equal(eng.proj_neg(Pj), zc_ref.proj_neg(Pj))
assert np.array_equal(eng.proj_scalar_mul(Pj[:512], rawk[:512]), par(zc_ref.proj_scalar_mul, 512, Pj[:512], rawk[:512])), "proj_scalar_mul"
weq, wok2 = zc_ref.proj_eq(zc_ref.proj_add(Pj, Qj), zc_ref.proj_add(Qj, Pj))
assert np.array_equal(eng.proj_eq(eng.proj_add(Pj, Qj), eng.proj_add(Qj, Pj)), weq & wok2) and weq.all()
raw = rng.integers(0, 256, size=(1 << 15, 32), dtype=np.uint8); raw[:, 31] &= 0x1F
d, ok = eng.ris_decompress(raw); wd, wok = par(zc_ref.ris_decompress, len(raw), raw)
assert np.array_equal(d, wd) and np.array_equal(ok, wok)
# persistent-wave strict path (>= 2^17 elements): ragged size, raw limb patterns up to 2^260 mixed in
nb = (1 << 17) + int(rng.integers(1, 1 << 16))
Pb = np.tile(P, (nb // n + 1, 1))[:nb].copy()
Kb = V.rand_scalars_np(nb, seed * 100 + 31, bits=252)
wild = rng.choice(nb, size=nb // 50, replace=False)
Kb[wild] = rng.integers(0, 1 << 52, size=(len(wild), 5), dtype=np.uint64)
Kb[rng.choice(nb, size=64, replace=False)] = 0
Kb[wild[:200], :4] = 0                                           # low 208 bits clear: the early-stopping loop cases
assert np.array_equal(eng.ed_scalar_mul(Pb, Kb), par(zc_ref.ed_scalar_mul, nb, Pb, Kb)), "persistent-wave scalar_mul"
# bucket-method MSM against the oracle's sum of double_and_add results, random size and scalar width
nm = int(rng.integers(1 << 12, 1 << 17))
bits = int(rng.choice([64, 128, 249, 252]))
Km = V.rand_scalars_np(nm, seed * 100 + 32, bits=max(bits, 249))
if bits < 249:                                                   # short scalars: only the low `bits` bits
    for j in range(5):
        keep = min(52, max(0, bits - 52 * j))
        Km[:, j] &= np.uint64((1 << keep) - 1)
Km[rng.choice(nm, size=nm // 100 + 1, replace=False)] = rng.integers(0, 1 << 52, size=(nm // 100 + 1, 5), dtype=np.uint64)
Pm = np.tile(P, (nm // n + 1, 1))[:nm].copy()
got, want = eng.msm(Pm, Km), zc_ref.msm_naive_mt(Pm, Km)
assert zc_ref.ed_eq(got, want)[0] == 1 and np.array_equal(zc_ref.ed_compress(got)[0], zc_ref.ed_compress(want)[0]), "msm n=%d bits=%d" % (nm, bits)
# the LDS-staged 40-byte kernels (product: beyond 256 MB per call) at a random ragged size through the test build's threshold
# override: canonical operands with raw patterns (>= 2^T: the two-pass product inside a staged block) mixed in
ne = int(rng.integers(1, 1 << 17))
with V.tuned(hooks=True, ZC_TEST_STREAM_MIN_BYTES=1) as te:
    for pre, mod, top in (("fe", pm.P, 252), ("sc", pm.L, 249)):
        xa, xb = V.rand_fe_np(ne, seed * 100 + 50 + top, mod), V.rand_fe_np(ne, seed * 100 + 51 + top, mod)
        wild = rng.choice(ne, size=ne // int(rng.integers(2, 40)) + 1, replace=False)
        xa[wild] = rng.integers(0, 1 << 52, size=(len(wild), 5), dtype=np.uint64)
        xb[wild[::3]] = rng.integers(0, 1 << 52, size=(len(wild[::3]), 5), dtype=np.uint64)
        c0 = te.lib.zc_test_staged_launches(te.ctx)
        for op in ("add", "sub", "mul"):
            assert np.array_equal(getattr(te, pre + "_" + op)(xa, xb), par(getattr(zc_ref, pre + "_" + op), ne, xa, xb)), "staged %s_%s n=%d" % (pre, op, ne)
        for op in ("neg", "square"):
            assert np.array_equal(getattr(te, pre + "_" + op)(xa), par(getattr(zc_ref, pre + "_" + op), ne, xa)), "staged %s_%s n=%d" % (pre, op, ne)
        assert te.lib.zc_test_staged_launches(te.ctx) == c0 + 5
print("soak seed %d ok in %.1f s (persistent-wave n=%d, msm n=%d bits=%d, staged element ops n=%d)" % (seed, time.time() - t0, nb, nm, bits, ne))

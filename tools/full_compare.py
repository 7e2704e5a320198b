#!/usr/bin/env python3
"""One full-size exact comparison: 2^20 strict scalar multiplications (the headline launch) and the
fused Ristretto round trip, every output against the CPU oracle (threads).  GPU box only; ~1 min."""
import concurrent.futures as cf
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dusk_zerocaf_amd as z  # noqa: E402
from oracle import zc_ref  # noqa: E402
from tests import vectors as V  # noqa: E402


def par(fn, n, *arrs):
    th = min(16, os.cpu_count() or 1)
    idx = np.array_split(np.arange(n), th * 4)
    with cf.ThreadPoolExecutor(th) as ex:
        parts = list(ex.map(lambda ix: fn(*[a[ix] for a in arrs]), idx))
    if isinstance(parts[0], tuple):
        return tuple(np.concatenate([p[j] for p in parts]) for j in range(len(parts[0])))
    return np.concatenate(parts)


def main():
    n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    zc_ref.build()
    eng = z.Engine()
    t0 = time.time()
    P = eng.ed_mul_base(V.rand_scalars_np(n, 901, bits=249))
    K = V.rand_scalars_np(n, 902, bits=252)
    got = eng.ed_scalar_mul(P, K)
    want = par(zc_ref.ed_scalar_mul, n, P, K)
    assert np.array_equal(got, want), "strict scalar-mul differs"
    print("strict scalar-mul: %d of %d outputs identical (%.0f s)" % (n, n, time.time() - t0))
    t0 = time.time()
    enc = eng.ris_compress(got)
    out, ok = eng.ris_roundtrip_mul(enc, K)
    wout, wok = par(zc_ref.ris_roundtrip_mul, n, enc, K)
    assert np.array_equal(out, wout) and np.array_equal(ok, wok), "ristretto round trip differs"
    print("ristretto round trip: %d of %d encodings identical (%.0f s)" % (n, n, time.time() - t0))


if __name__ == "__main__":
    main()

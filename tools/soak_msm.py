#!/usr/bin/env python3
"""Randomised soak of the bucket-method MSM against the oracle's sum of the reference's Mul<Scalar> + Add: random sizes on
both sides of every internal threshold (affine / projective records, one- and two-word sort records, one / two / three
sort passes, forked normalisation), random window widths and run lengths, skewed and wild scalars, repeated and identity
points.  Not part of the test suite; run on the GPU box:  python tools/soak_msm.py FIRST_SEED COUNT"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dusk_zerocaf_amd as z  # noqa: E402
from dusk_zerocaf_amd import _lib  # noqa: E402
from oracle import zc_ref  # noqa: E402
from tests import vectors as V  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
eng = z.Engine()
zc_ref.build()
# the product's own knobs + the path forcers of the -DZC_TEST_HOOKS build (the soak runs on that build: same sources)
KNOBS = ("ZC_MSM_WINDOW", "ZC_MSM_AFFINE", "ZC_MSM_SORT_PACKED", "ZC_MSM_SORT_BIG", "ZC_MSM_RUN", "ZC_MSM_SORT_G", "ZC_MSM_AFFINE_CHUNK",
         "ZC_MSM_GROUPS", "ZC_MSM_SEG", "ZC_MSM_RUN_EDGES", "ZC_MSM_FORK")
HOOKS = _lib.load_test_hooks()
for seed in range(first, first + count):
    t0 = time.time()
    rng = np.random.default_rng(0xB0C4E7 + seed)
    n = int(2 ** rng.uniform(12, 19.3))
    for k in KNOBS:
        os.environ.pop(k, None)
    knobs = {}
    if rng.random() < 0.5:
        knobs["ZC_MSM_WINDOW"] = int(rng.integers(6, 20))
    if rng.random() < 0.5:
        knobs["ZC_MSM_AFFINE"] = int(rng.integers(0, 2))
    if rng.random() < 0.3:
        knobs["ZC_MSM_SORT_PACKED"] = 0
        knobs["ZC_MSM_SORT_BIG"] = int(rng.integers(0, 2))
    if rng.random() < 0.3:
        knobs["ZC_MSM_RUN"] = int(rng.choice([4, 7, 16, 33, 128, 256, 1000]))
    if rng.random() < 0.3:
        knobs["ZC_MSM_SORT_G"] = int(rng.integers(1, 6))
    if rng.random() < 0.3:
        knobs["ZC_MSM_AFFINE_CHUNK"] = int(rng.choice([1, 2, 5, 8, 16, 33]))
    if rng.random() < 0.3:
        knobs["ZC_MSM_SEG"] = int(rng.choice([2, 4, 8, 16, 32, 64]))
    if rng.random() < 0.2:
        knobs["ZC_MSM_RUN_EDGES"] = int(rng.choice([4, 6, 8, 16, 32]))
    if rng.random() < 0.2:
        knobs["ZC_MSM_FORK"] = int(rng.integers(0, 2))
    for k, v in knobs.items():
        os.environ[k] = str(v)
    if rng.random() < 0.6:
        # window groups: a random split of the windows this shard will have under the knobs above (the plan says how many)
        probe = z.Engine(lib=HOOKS)
        W = probe.msm_plan(n)["windows"]
        probe.close()
        G = int(rng.integers(2, 5))
        if W >= G:
            cuts = sorted(rng.choice(np.arange(1, W), size=G - 1, replace=False).tolist())
            parts = [b - a for a, b in zip([0] + cuts, cuts + [W])]
            knobs["ZC_MSM_GROUPS"] = ",".join(str(x) for x in parts)
            os.environ["ZC_MSM_GROUPS"] = knobs["ZC_MSM_GROUPS"]
    meng = z.Engine(lib=HOOKS)                                     # the library reads its knobs when a context is created
    bits = int(rng.choice([16, 64, 128, 249, 252]))
    K = V.rand_scalars_np(n, seed * 7 + 1, bits=252)
    if bits < 249:
        for j in range(5):
            keep = min(52, max(0, bits - 52 * j))
            K[:, j] &= np.uint64((1 << keep) - 1)
    elif bits == 249:
        K[:, 4] &= np.uint64((1 << 41) - 1)
    style = rng.random()
    if style < 0.15:
        K[:] = K[0]                                                # every scalar equal: one bucket per window holds everything
    elif style < 0.3:
        K[n // 3:] = K[:n - n // 3][rng.integers(0, 4, size=n - n // 3)]          # four values
    wild = rng.choice(n, size=n // 100 + 1, replace=False)
    K[wild] = rng.integers(0, 1 << 52, size=(len(wild), 5), dtype=np.uint64)        # raw 260-bit patterns
    K[rng.choice(n, size=n // 200 + 1, replace=False)] = 0
    P = eng.ed_mul_base(torch.from_numpy(V.rand_scalars_np(n, seed * 7 + 2, bits=249).view(np.int64)).cuda())
    if rng.random() < 0.3:
        P[n // 2:] = P[:n - n // 2].clone()                        # repeated points
    if rng.random() < 0.3:
        P[rng.choice(n, size=8, replace=False)] = torch.tensor(V.IDENT_ROW, dtype=torch.int64, device="cuda")
    if rng.random() < 0.2:                                         # affine inputs (Z = 1): the normalisation skips its inversion
        xy, _ = eng.ed_to_affine(P)
        one = torch.zeros((n, 5), dtype=torch.int64, device="cuda")
        one[:, 0] = 1
        Ph = torch.cat([xy, one, eng.fe_mul(xy[:, :5].contiguous(), xy[:, 5:].contiguous())], dim=1).contiguous()
        assert bool(eng.ed_eq(P, Ph).all())
        P = Ph
    torch.cuda.synchronize()
    dK = torch.from_numpy(K.view(np.int64)).cuda()
    got = meng.msm(P, dK)
    again = meng.msm(P, dK)                                        # the side stream's events and the workspace are reused
    assert np.array_equal(got, again), (seed, knobs)
    meng.close()
    want = zc_ref.msm_naive_mt(P.cpu().numpy().view(np.uint64), K)
    ok = zc_ref.ed_eq(got, want)[0] == 1 and np.array_equal(zc_ref.ed_compress(got)[0], zc_ref.ed_compress(want)[0])
    print("soak_msm seed %d %s: n=%d bits=%d style=%.2f knobs=%s (%.1f s)" % (seed, "ok" if ok else "FAILED", n, bits, style, knobs, time.time() - t0), flush=True)
    if not ok:
        sys.exit(1)
print("soak_msm: %d seeds ok" % count)

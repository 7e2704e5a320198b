"""Seeded synthetic inputs shared by the CPU-tier and GPU-tier parity tests
(same bytes for the oracle and for the HIP path)."""
from __future__ import annotations

import random

import numpy as np

from oracle import pymodel as pm

SEED = 0x5EED0000

FE_EDGE = [0, 1, 2, pm.P - 1, pm.P - 2, (pm.P - 1) // 2, (pm.P + 1) // 2, 2**52, 2**104, 2**156, 2**208,
           2**252, 2**52 - 1, 2**29 - 1, 2**29, 2**232, 2**252 + 1,
           182687704666362864775460604089535377456991567872,  # field.rs KAT A
           904625697166532776746648320197686575422163851717637391703244652875051672039,  # KAT B
           2009874587549]  # KAT C
SC_EDGE = [0, 1, 2, pm.L - 1, pm.L - 2, (pm.L - 1) // 2, 2**52, 2**104, 2**156, 2**208, 2**249, 2**249 - 1]


def limbs_array(vals):
    return np.array([pm.limbs(v) for v in vals], dtype=np.uint64)


def rand_fe(n, seed, modulus=pm.P, edge=FE_EDGE):
    rng = random.Random(seed)
    vals = [rng.randrange(modulus) for _ in range(n)]
    k = min(len(edge), n)
    vals[:k] = edge[:k]
    return vals


def rand_fe_np(n, seed, modulus=pm.P):
    """Vectorised uniform-ish canonical elements for big batches (rejection on the top limb)."""
    rng = np.random.default_rng(seed)
    out = rng.integers(0, 1 << 52, size=(n, 5), dtype=np.uint64)
    top = modulus >> 208
    out[:, 4] = rng.integers(0, top, size=n, dtype=np.uint64)   # < top limb of modulus => value < modulus
    return out


def rand_scalars_np(n, seed, bits=252):
    rng = np.random.default_rng(seed)
    out = rng.integers(0, 1 << 52, size=(n, 5), dtype=np.uint64)
    out[:, 4] = rng.integers(0, 1 << (bits - 208), size=n, dtype=np.uint64)
    return out


def base_multiples(oracle, n, seed):
    """n valid subgroup points in non-trivial extended coordinates: r_i * B via the oracle."""
    k = rand_scalars_np(n, seed, bits=249)
    b = np.tile(np.array(sum(pm.pt_limbs(pm.BASEPOINT), []), dtype=np.uint64), (n, 1))
    return oracle.ed_scalar_mul(b, k)


def pts_np(pts):
    return np.array([sum(pm.pt_limbs(p), []) for p in pts], dtype=np.uint64)


IDENT_ROW = [0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]


def raw_scalar_edges(n_random=24, seed=SEED + 0x256):
    """Raw `Scalar([..])` limb patterns with bits at or above 2^256 (reachable through the pub tuple
    field, not through from_bytes).  double_and_add's loop test only sees the low 256 bits of n
    (src/edwards.rs:111 -> src/scalar.rs:78-91 -> src/backend/u64/scalar.rs:477-516), so some of these
    stop early: [0,0,0,0,1<<50] -> identity, [1,0,0,0,1<<50] -> P, [7,0,0,0,8<<48] -> 7P, while
    [8,0,0,0,8<<48] or 2^256+5 run all 260 / 257 bits."""
    t = 1 << 48
    rows = [[0, 0, 0, 0, 1 << 50], [1, 0, 0, 0, 1 << 50], pm.limbs(2**256 + 5), [2, 0, 0, 0, 2 * t], [1, 0, 0, 0, 2 * t],
            [7, 0, 0, 0, 8 * t], [8, 0, 0, 0, 8 * t], [0, 0, 0, 0, t], [1, 0, 0, 0, t], [3, 0, 0, 0, 4 * t], [4, 0, 0, 0, 4 * t],
            [0, 0, 0, 0, 3 * t], [5, 0, 0, 0, 12 * t], [3, 0, 0, 0, 12 * t], [0, 1, 0, 0, 8 * t], [0, 0, 0, 0, 8 * t + 1],
            [0, 0, 0, 0, 15 * t], [6, 0, 0, 0, 8 * t], [2, 0, 0, 0, 4 * t], [(1 << 52) - 1] * 5, [0, 0, 0, 0, (1 << 52) - 1],
            [1, 0, 0, 0, 6 * t], [1, 0, 0, 0, 10 * t], [3, 0, 0, 0, 8 * t + (1 << 47)]]
    rng = np.random.default_rng(seed)
    rnd = rng.integers(0, 1 << 52, size=(n_random, 5), dtype=np.uint64)
    rnd[:, 4] |= np.uint64(1) << rng.integers(48, 52, size=n_random).astype(np.uint64)
    rnd[: n_random // 3, :4] = 0                                # sparse low parts: near the early-stop boundary
    rnd[: n_random // 3, 4] &= ~np.uint64((1 << 48) - 1)
    rnd[: n_random // 3, 0] = rng.integers(0, 16, size=n_random // 3, dtype=np.uint64)
    return np.concatenate([np.array(rows, dtype=np.uint64), rnd])


import contextlib


@contextlib.contextmanager
def tuned(hooks=False, devices=None, **env):
    """An Engine whose context was created under the given ZC_* knobs (the library reads them once, in
    zc_ctx_create).  hooks=True: on libzerocaf_hip_test.so, the -DZC_TEST_HOOKS build.  None values unset."""
    import os
    import dusk_zerocaf_amd as z
    from dusk_zerocaf_amd import _lib
    old = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    e = None
    try:
        e = z.Engine(devices, lib=_lib.load_test_hooks() if hooks else None)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        yield e
    finally:
        e.close()

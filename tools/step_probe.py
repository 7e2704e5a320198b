#!/usr/bin/env python3
"""Measure the time of one unified step of k_ed_scalar_mul (uniform scalars with known op counts)
and from it the effective number of wave steps for random 252-bit scalars."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dusk_zerocaf_amd as z
from oracle import pymodel as pm

eng = z.Engine([0]); st = torch.cuda.current_stream(); eng.set_stream(st.cuda_stream)
n = 1 << 20
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
rng = np.random.default_rng(5)
r = rng.integers(0, 1 << 52, size=(n, 5), dtype=np.uint64); r[:, 4] >>= np.uint64(11)
base = np.tile(np.array(sum(pm.pt_limbs(pm.BASEPOINT), []), dtype=np.uint64), (n, 1))
P = eng.ed_scalar_mul(dev(base), dev(r))
out = torch.empty_like(P)

def timed(K, reps=5):
    dK = dev(K)
    eng.ed_scalar_mul(P, dK, out=out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): eng.ed_scalar_mul(P, dK, out=out)
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def const(v):
    return np.tile(np.array(pm.limbs(v), dtype=np.uint64), (n, 1))
t_a = timed(const(1 << 251))            # 252 bits, popcount 1   -> 251 + 1   = 252 ops
t_b = timed(const((1 << 252) - 1))      # 252 bits, popcount 252 -> 251 + 252 = 503 ops
per = (t_b - t_a) / (503 - 252)
fixed = t_a - 252 * per
K = rng.integers(0, 1 << 52, size=(n, 5), dtype=np.uint64); K[:, 4] >>= np.uint64(8)
t_r = timed(K)
print("252-op scalars %.3f ms, 503-op scalars %.3f ms -> %.4f ms per wave-step batch, fixed %.3f ms" % (t_a, t_b, per, fixed))
print("random S252: %.3f ms -> effective steps %.1f (mean lane ops 376, max-of-64 ~395)" % (t_r, (t_r - fixed) / per))

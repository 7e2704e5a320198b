#!/usr/bin/env python3
"""Runs the MSM key sort alone (test hook zc_test_msm_sort) a few times: for kernel traces / A-B of sort variants.
Usage: python tools/debug/sort_probe.py LOG2N C [reps]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dusk_zerocaf_amd as z  # noqa: E402
from tests.vectors import rand_scalars_np  # noqa: E402

lg, c = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = 1 << lg
W = -(-261 // c)
eng = z.Engine()
fn = eng.lib.zc_test_msm_sort
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
K = torch.from_numpy(rand_scalars_np(n, 13, 249).view(np.int64)).cuda()
out = torch.empty((n * W, 2), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for _ in range(reps):
    assert fn(eng.ctx, K.data_ptr(), n, c, out.data_ptr()) == 0
print("ok", lg, c)

#!/usr/bin/env python3
"""The oracle's double_and_add rate on ONE host thread by compiler flags (why oracle/zc_ref.py builds both -march=native and
-march=x86-64-v2 and keeps the faster: gcc 11 does not know the EPYC 9575F of the GPU boxes).  Run from the repo root."""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
from oracle import pymodel as pm  # noqa: E402
from oracle import zc_ref  # noqa: E402
from tests import vectors as V  # noqa: E402

model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
print(model, "threads", zc_ref.host_threads())
src = os.path.join("oracle", "zc_ref.c")
n = 4096
K = V.rand_scalars_np(n, 5, bits=252)
base = np.tile(np.array(sum(pm.pt_limbs(pm.BASEPOINT), []), dtype=np.uint64), (n, 1))
for flags in ("-O3 -march=x86-64-v2", "-O3 -march=native", "-O3 -march=x86-64-v3", "-O3 -march=native -mno-avx512f", "-O2 -march=native"):
    so = "/tmp/zr_%d.so" % abs(hash(flags))
    subprocess.check_call(["gcc"] + flags.split() + ["-fPIC", "-std=c11", "-shared", "-o", so, src])
    lib = C.CDLL(so)
    out = np.empty_like(base)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        lib.zr_ed_scalar_mul_batch(base.ctypes.data_as(C.c_void_p), K.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_size_t(n))
        best = min(best, time.perf_counter() - t)
    print("%-36s %.1f scalar-muls/s on one thread" % (flags, n / best))

#!/usr/bin/env python3
"""Do a chip-filling MSM and a latency-bound small MSM from another stream overlap on one GPU?
Two contexts, two HIP streams (different priorities), two host threads.  Timing probe only."""
import os, sys, threading, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dusk_zerocaf_amd as z
from tests.vectors import rand_scalars_np

def main():
    big, small = int(sys.argv[1]) if len(sys.argv) > 1 else 21, int(sys.argv[2]) if len(sys.argv) > 2 else 16
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
    e0 = z.Engine()
    inp = {}
    for lg in (big, small):
        n = 1 << lg
        inp[lg] = (e0.ed_mul_base(dev(rand_scalars_np(n, 11, 249))), dev(rand_scalars_np(n, 13, 249)))
    torch.cuda.synchronize()
    s_lo, s_hi = torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)
    e1, e2 = z.Engine(), z.Engine()
    e1.set_stream(s_lo.cuda_stream)
    e2.set_stream(s_hi.cuda_stream)
    def loop(eng, lg, reps, out):
        P, K = inp[lg]
        eng.msm(P, K)
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.msm(P, K)
        out.append((time.perf_counter() - t0) / reps * 1e3)
    a, b = [], []
    loop(e1, big, 20, a); loop(e2, small, 60, b)
    print("alone: big 2^%d %.3f ms, small 2^%d %.3f ms" % (big, a[0], small, b[0]))
    a, b = [], []
    t1 = threading.Thread(target=loop, args=(e1, big, 40, a)); t2 = threading.Thread(target=loop, args=(e2, small, 120, b))
    t1.start(); t2.start(); t1.join(); t2.join()
    print("together: big %.3f ms, small %.3f ms (per call, both loops running)" % (a[0], b[0]))

main()

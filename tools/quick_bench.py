#!/usr/bin/env python3
"""Device-resident timings of the main kernels through the C ABI (HIP events on the launch stream).
Usage: python tools/quick_bench.py [what ...]   what in {strict, fast, ris, msm, msmsweep}
ZC_LIB_PATH selects another build of the library (A/B of kernel variants)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dusk_zerocaf_amd as z  # noqa: E402
from tests.vectors import rand_scalars_np  # noqa: E402


def timed(f, reps=5, warm=2):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        f()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2], ts[0]


def main():
    what = sys.argv[1:] or ["strict", "fast", "ris", "msm"]
    eng = z.Engine()
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
    out = {"lib": os.path.basename(z.LIB_PATH)}
    cache = {}

    def inputs(lg):
        if lg not in cache:
            n = 1 << lg
            P = eng.ed_mul_base(dev(rand_scalars_np(n, 11, 249)))
            cache[lg] = (P, dev(rand_scalars_np(n, 12, 252)), dev(rand_scalars_np(n, 13, 249)))
        return cache[lg]

    if "strict" in what:
        P, K, _ = inputs(20)
        o = torch.empty_like(P)
        out["strict_2p20_ms"] = timed(lambda: eng.ed_scalar_mul(P, K, out=o))
    if "fast" in what:
        P, K, _ = inputs(20)
        o = torch.empty_like(P)
        out["fast_2p20_ms"] = timed(lambda: eng.ed_scalar_mul(P, K, out=o, flags=z.FAST))
    if "ris" in what:
        for lg in (20, 22):
            P, K, _ = inputs(lg)
            enc = eng.ris_compress(P)
            o = torch.empty_like(enc)
            out["ris_2p%d_ms" % lg] = timed(lambda: eng.ris_roundtrip_mul(enc, K, out=o), reps=3, warm=1)
            del enc, o
    if "msm" in what:
        for lg in (16, 18, 20, 21, 24):
            P, _, K = inputs(lg)
            out["msm_2p%d_ms" % lg] = timed(lambda: eng.msm(P, K), reps=3, warm=1)
            if lg == 24:
                cache.pop(24)
    for lg in (20, 21, 24):
        if "msm%d" % lg in what:
            P, _, K = inputs(lg)
            out["msm_2p%d_ms" % lg] = timed(lambda: eng.msm(P, K), reps=3, warm=1)
    if "msmsweep" in what or "msmsweep_small" in what:
        for lg in ((20, 21, 24) if "msmsweep" in what else (13, 14, 16, 18, 19)):
            P, _, K = inputs(lg)
            for c in range(lg - 6, lg - 1):
                if 5 <= c <= 22:
                    os.environ["ZC_MSM_WINDOW"] = str(c)
                    ce = z.Engine()                                # knobs are read when a context is created
                    os.environ.pop("ZC_MSM_WINDOW", None)
                    out["msm_2p%d_c%d_ms" % (lg, c)] = timed(lambda: ce.msm(P, K), reps=3, warm=1)[0]
                    ce.close()
            cache.pop(lg)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

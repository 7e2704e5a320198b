#!/usr/bin/env python3
"""Times zc_msm (device-resident inputs, result to the host: the synchronising call the sharded MSM makes) for window-group
splits of the bucket pipeline.  Usage: python tools/msm_groups_sweep.py LOG2N [groups ...] with groups like 16 | 13,3 | 7,6,3
(windows per group, top group first; must add up to the shard's window count) or "default"; extra knobs as K=V words
(the path forcers are read by the -DZC_TEST_HOOKS build only: `HOOKS=1` as a word selects it; compile-time variants are
libraries of their own: `python -m dusk_zerocaf_amd.build --variant NAME ZC_MSM_...=V`, then ZC_LIB_PATH=build/variants/NAME.so)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dusk_zerocaf_amd as z  # noqa: E402
from tests.vectors import rand_scalars_np  # noqa: E402


def main():
    lg = int(sys.argv[1])
    specs = [a for a in sys.argv[2:] if "=" not in a] or ["1", "default"]
    extra = dict(a.split("=", 1) for a in sys.argv[2:] if "=" in a)
    hooks = extra.pop("HOOKS", None)
    from dusk_zerocaf_amd import _lib
    lib = _lib.load_test_hooks() if hooks else None
    n = 1 << lg
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
    e0 = z.Engine()
    P = e0.ed_mul_base(dev(rand_scalars_np(n, 11, 249)))
    K = dev(rand_scalars_np(n, 13, 249))
    torch.cuda.synchronize()
    ref = None
    for spec in specs:
        os.environ.pop("ZC_MSM_GROUPS", None)
        if spec != "default":
            os.environ["ZC_MSM_GROUPS"] = spec
        os.environ.update(extra)
        eng = z.Engine(lib=lib)
        for k in list(extra) + ["ZC_MSM_GROUPS"]:
            os.environ.pop(k, None)
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        for _ in range(3):
            got = eng.msm(P, K)
        ts = []
        for _ in range(9):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = eng.msm(P, K)
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        if ref is None:
            ref = got
        same = bool(e0.ed_eq(np.asarray(got), np.asarray(ref))[0])
        print(json.dumps({"log2n": lg, "groups": spec, "extra": extra, "plan": eng.msm_plan(n), "ms_median": round(ts[len(ts) // 2], 3), "ms_min": round(ts[0], 3),
                          "same_point_as_first": same}), flush=True)
        eng.close()


if __name__ == "__main__":
    main()

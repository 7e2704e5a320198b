#!/usr/bin/env python3
"""Timing sweep of the MSM window groups (ZC_MSM_GROUPS) at one size.  usage: msm_groups_sweep.py lg "cfg;cfg;..." """
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dusk_zerocaf_amd as z
from tests.vectors import rand_scalars_np

def timed(f, reps=7, warm=2):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return round(ts[len(ts) // 2], 3), round(ts[0], 3)

lg = int(sys.argv[1])
cfgs = sys.argv[2].split(";")
eng = z.Engine()
eng.set_stream(torch.cuda.current_stream().cuda_stream)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
n = 1 << lg
P = eng.ed_mul_base(dev(rand_scalars_np(n, 11, 249)))
K = dev(rand_scalars_np(n, 13, 249))
ref = None
for cfg in cfgs:
    if cfg == "default":
        os.environ.pop("ZC_MSM_GROUPS", None)
    else:
        os.environ["ZC_MSM_GROUPS"] = cfg
    r = eng.msm(P, K)
    r = r.cpu().numpy() if hasattr(r, "cpu") else np.asarray(r)
    t = timed(lambda: eng.msm(P, K))
    if ref is None:
        ref = eng.ed_compress(r.reshape(1, 20).view(np.uint64))
    same = bool(np.array_equal(eng.ed_compress(r.reshape(1, 20).view(np.uint64)), ref))
    print(json.dumps({"lg": lg, "groups": cfg, "ms_median_min": t, "same_point": same}), flush=True)

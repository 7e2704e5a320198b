#!/usr/bin/env python3
"""Fused Ristretto round trip at 2^22: time against the table-scratch chunk size (ZC_FAST_CHUNK, lanes)."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dusk_zerocaf_amd as z  # noqa: E402
from tests.vectors import rand_scalars_np  # noqa: E402
from tools.quick_bench import timed  # noqa: E402

eng = z.Engine()
eng.set_stream(torch.cuda.current_stream().cuda_stream)
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << lg
P = eng.ed_mul_base(dv(rand_scalars_np(n, 11, 249)))
K = dv(rand_scalars_np(n, 12, 252))
enc = eng.ris_compress(P)
del P
o = torch.empty_like(enc)
f = lambda: eng.ris_roundtrip_mul(enc, K, out=o)
timed(f, 4, 2)
out = {}
sizes = [int(x) for x in os.environ.get("SWEEP_SIZES", "786432,1048576,1572864,2097152,3145728,4194304").split(",")]
for rep in range(2):
    for c in sizes + sizes[::-1]:
        os.environ["ZC_FAST_CHUNK"] = str(c)
        out.setdefault(str(c), []).append(round(timed(f, 3, 1)[0], 2))
print(json.dumps({"lg": lg, "ms_by_chunk": out}))

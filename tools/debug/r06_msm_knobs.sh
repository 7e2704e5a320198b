#!/bin/bash
# Round 6: what reacts at 2^21 pairs?  Test-build knobs (ZC_MSM_SEG, ZC_MSM_RUN_EDGES, ZC_MSM_AFFINE_CHUNK, ZC_MSM_FORK), nine
# synchronising zc_msm calls each, three alternating rounds -> gpurun_out/r06_msm_knobs.txt
REPO=$PWD
out=$REPO/gpurun_out/r06_msm_knobs.txt
: > $out
run() {
python - "$1" >> $out 2>/dev/null <<PY
import sys; sys.path.insert(0, "$REPO")
import numpy as np, torch
from tests import vectors as V
from tests.vectors import rand_scalars_np
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
with V.tuned(hooks=True) as eng:
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 1 << 21
    P = eng.ed_mul_base(dev(rand_scalars_np(n, 11, 249))); K = dev(rand_scalars_np(n, 13, 249))
    for _ in range(3): eng.msm(P, K)
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.msm(P, K); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    print("%-28s 2^21: median %.3f min %.3f ms" % (sys.argv[1], ts[4], ts[0]))
PY
}
for rep in 1 2 3; do
  run "default"
  for s in 8 32; do ZC_MSM_SEG=$s run "SEG=$s"; done
  for e in 4 6 16; do ZC_MSM_RUN_EDGES=$e run "RUN_EDGES=$e"; done
  for c in 4 16; do ZC_MSM_AFFINE_CHUNK=$c run "AFFINE_CHUNK=$c"; done
  ZC_MSM_FORK=0 run "FORK=0"
  ZC_MSM_SEG=8 ZC_MSM_RUN_EDGES=4 run "SEG=8,RUN_EDGES=4"
done

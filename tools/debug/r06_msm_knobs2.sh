#!/bin/bash
# Round 6: points per lane of the normalisation (ZC_MSM_AFFINE_CHUNK) and run length of the bucket sums (ZC_MSM_RUN) on the test build,
# nine synchronising zc_msm calls each, alternating rounds -> gpurun_out/r06_msm_knobs2.txt
REPO=$PWD
out=$REPO/gpurun_out/r06_msm_knobs2.txt
: > $out
run() {
python - "$1" "$2" >> $out 2>/dev/null <<PY
import sys; sys.path.insert(0, "$REPO")
import numpy as np, torch
from tests import vectors as V
from tests.vectors import rand_scalars_np
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
with V.tuned(hooks=True) as eng:
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    lg = int(sys.argv[2]); n = 1 << lg
    P = eng.ed_mul_base(dev(rand_scalars_np(n, 11, 249))); K = dev(rand_scalars_np(n, 13, 249))
    for _ in range(3): eng.msm(P, K)
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.msm(P, K); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    print("%-22s 2^%d: median %.3f min %.3f ms" % (sys.argv[1], lg, ts[4], ts[0]))
PY
}
for rep in 1 2 3; do
  run default 21
  for c in 10 11 12 14 16; do ZC_MSM_AFFINE_CHUNK=$c run "AFFINE_CHUNK=$c" 21; done
  for t in 48 64 96; do ZC_MSM_RUN=$t run "RUN=$t" 21; done
  run default 22
  for c in 20 22 24; do ZC_MSM_AFFINE_CHUNK=$c run "AFFINE_CHUNK=$c" 22; done
  run default 20
  for c in 5 6 8; do ZC_MSM_AFFINE_CHUNK=$c run "AFFINE_CHUNK=$c" 20; done
done
sort -s -k2,2 -k1,1 $out

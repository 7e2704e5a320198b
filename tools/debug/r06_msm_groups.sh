#!/bin/bash
# Round 6: window-group splits at 2^21 pairs on the round-6 pipeline (product knob ZC_MSM_GROUPS), nine synchronising calls each,
# alternating rounds -> gpurun_out/r06_msm_groups.txt
REPO=$PWD
out=$REPO/gpurun_out/r06_msm_groups.txt
: > $out
run() {
python - "$1" >> $out 2>/dev/null <<PY
import sys; sys.path.insert(0, "$REPO")
import numpy as np, torch
import dusk_zerocaf_amd as z
from tests.vectors import rand_scalars_np
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
eng = z.Engine(); eng.set_stream(torch.cuda.current_stream().cuda_stream)
n = 1 << 21
P = eng.ed_mul_base(dev(rand_scalars_np(n, 11, 249))); K = dev(rand_scalars_np(n, 13, 249))
for _ in range(3): eng.msm(P, K)
torch.cuda.synchronize()
ts = []
for _ in range(9):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); eng.msm(P, K); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts.sort()
print("%-14s 2^21: median %.3f min %.3f ms" % (sys.argv[1], ts[4], ts[0]))
PY
}
for rep in 1 2 3; do
  run default
  for g in 9,4,3 10,4,2 9,5,2 10,3,3 8,5,3 11,3,2 8,4,4 12,4 13,3 12,2,2 10,6; do ZC_MSM_GROUPS=$g run "$g"; done
done
sort -s -k1,1 $out

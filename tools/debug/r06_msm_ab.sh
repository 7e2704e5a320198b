#!/bin/bash
# Same-box A/B of zc_msm builds: nine synchronising calls per size (median / min), <reps> alternating rounds, then one kernel
# timeline per build.  usage: tools/debug/r06_msm_ab.sh <tag> <reps> name=path ...   ("product" = the in-tree library)
tag=$1; reps=$2; shift 2
REPO=$PWD
OUT=$REPO/gpurun_out/r06_msm_ab_$tag
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
setlib() { if [ "$1" = product ]; then unset ZC_LIB_PATH; else export ZC_LIB_PATH=$1; fi; }
for rep in $(seq 1 $reps); do
  for spec in "$@"; do
    name=${spec%%=*}; path=${spec#*=}; setlib $path
    python - >> $OUT/timing.txt 2>/dev/null <<PY
import sys; sys.path.insert(0, "$REPO")
import numpy as np, torch
import dusk_zerocaf_amd as z
from tests.vectors import rand_scalars_np
eng = z.Engine(); eng.set_stream(torch.cuda.current_stream().cuda_stream)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
for lg in (${SIZES:-20, 21, 22}):
    n = 1 << lg
    P = eng.ed_mul_base(dev(rand_scalars_np(n, 11, 249))); K = dev(rand_scalars_np(n, 13, 249))
    for _ in range(3): eng.msm(P, K)
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.msm(P, K); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    print("%-18s rep $rep 2^%d: median %.3f min %.3f ms" % ("$name", lg, ts[4], ts[0]))
PY
  done
done
for spec in "$@"; do
  name=${spec%%=*}; path=${spec#*=}; setlib $path
  cd $REPO; bash tools/debug/timeline.sh msm21 > $OUT/timeline_$name.txt 2>&1; cd /tmp
done
unset ZC_LIB_PATH
sort -s -k1,1 -k4,4 $OUT/timing.txt

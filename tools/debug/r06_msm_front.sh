#!/bin/bash
# Round 6, MSM front matter: for every library given (name=path, "product" = the in-tree build) on ONE box
#   1. nine synchronising zc_msm calls at 2^21 pairs (median / min), three times, alternating between the libraries
#   2. the kernel timeline of one call (rocprofv3 --kernel-trace)
#   3. SPI resource-allocation stall counters + memory-path counters per kernel (separate --pmc passes; dispatches are
#      serialised under --pmc, so these describe each kernel ALONE)
# usage: tools/debug/r06_msm_front.sh <tag> name=path [name=path ...]   -> gpurun_out/r06_msm_front_<tag>/
tag=$1; shift
REPO=$PWD
OUT=$REPO/gpurun_out/r06_msm_front_$tag
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
setlib() { if [ "$1" = product ]; then unset ZC_LIB_PATH; else export ZC_LIB_PATH=$1; fi; }
for rep in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%=*}; path=${spec#*=}; setlib $path
    python - >> $OUT/timing.txt 2>/dev/null <<PY
import sys; sys.path.insert(0, "$REPO")
import numpy as np, torch
import dusk_zerocaf_amd as z
from tests.vectors import rand_scalars_np
eng = z.Engine(); eng.set_stream(torch.cuda.current_stream().cuda_stream)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
for lg in (20, 21, 22):
    n = 1 << lg
    P = eng.ed_mul_base(dev(rand_scalars_np(n, 11, 249))); K = dev(rand_scalars_np(n, 13, 249))
    for _ in range(3): eng.msm(P, K)
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.msm(P, K); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    print("$name rep $rep 2^%d: median %.3f min %.3f ms" % (lg, ts[4], ts[0]))
PY
  done
done
for spec in "$@"; do
  name=${spec%%=*}; path=${spec#*=}; setlib $path
  cd $REPO; bash tools/debug/timeline.sh msm21 > $OUT/timeline_$name.txt 2>&1; cd /tmp
  i=0
  for ctrs in "SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_REQ_NO_ALLOC_CSN" \
              "SPI_RA_RES_STALL_CSN SPI_RA_TGLIM_CU_FULL_CSN SPI_RA_BAR_CU_FULL_CSN SPI_CSN_BUSY GRBM_GUI_ACTIVE" \
              "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum TCP_TCC_READ_REQ_sum SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
              "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    cd $REPO; bash tools/debug/pmc_kernels.sh msm21 "$ctrs" "k_msm_prepare|k_msm_sort|k_msm_digits|k_msm_runs_affine" > $OUT/pmc_${name}_$i.txt 2>&1; cd /tmp
  done
done
unset ZC_LIB_PATH
ls -la $OUT

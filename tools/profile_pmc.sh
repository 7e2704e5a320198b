#!/bin/bash
# PMC passes for the secondary kernels (run on the GPU box through gpurun, from the repo root):
#   tools/profile_pmc.sh <tag>        -> gpurun_out/pmc_<tag>/*.csv
# One rocprofv3 --pmc pass per counter group (no trace domains), as MI355X_MICROARCH.md prescribes;
# FETCH_SIZE (3 TCC slots) and WRITE_SIZE (2) cannot share a pass.
set -u
TAG=${1:-r04}
REPO=$PWD
OUT=$REPO/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT/tmp"
cd /tmp; export TMPDIR=/tmp
PY="python $REPO/bench.py --cpu-sample 0 --steps 2 --warmup 1"
flatten() { find "$OUT/tmp" -name '*.csv' -exec mv {} "$OUT/" \; ; rm -rf "$OUT/tmp"/*; }
pmc() { local name=$1 ctrs=$2; shift 2; timeout 900 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/tmp" -o "$name" -- $PY "$@" > "$OUT/$name.log" 2>&1; flatten; }
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
python -c "import sys; sys.path.insert(0, '$REPO'); import dusk_zerocaf_amd as z; print(z.load().zc_version().decode())" > "$OUT/lib_version.txt" 2>/dev/null
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
for wl in "msm --units 2097152" "ristretto --units 4194304" "scalar_mul"; do
  set -- $wl; name=$1
  pmc ${name}_sq1 "$SQ1" --workload $wl
  pmc ${name}_sq2 "$SQ2" --workload $wl
  pmc ${name}_fetch "FETCH_SIZE TCC_HIT_sum" --workload $wl
  pmc ${name}_write "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" --workload $wl
done
ls -la "$OUT"

#!/bin/bash
# Collect the round's evidence on the GPU box (run from the repo root through gpurun):
#   tools/profile_round.sh            -> gpurun_out/prof/{*.csv,*.json,*.txt}
# then, back in the container:  python tools/summarize_profiles.py gpurun_out/prof r01
# rocprofv3 passes: kernel trace + stats per workload; PMC counters in their own passes
# (no trace domains besides the kernel trace), as MI355X_MICROARCH.md prescribes.
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT/tmp"
cd /tmp; export TMPDIR=/tmp
PY="python $REPO/bench.py --cpu-sample 0"

flatten() { find "$OUT/tmp" -name '*.csv' -exec mv {} "$OUT/" \; ; rm -rf "$OUT/tmp"/*; }
kt() { local name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tmp" -o "kt_$name" -- $PY "$@" > "$OUT/kt_$name.log" 2>&1; flatten; }
pmc() { local name=$1 ctrs=$2; shift 2; timeout 600 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/tmp" -o "pmc_$name" -- $PY "$@" > "$OUT/pmc_$name.log" 2>&1; flatten; }

kt scalar_mul --steps 5 --warmup 1
kt ristretto --workload ristretto --steps 5 --warmup 1
kt msm --workload msm --steps 5 --warmup 1
kt fe_mul --workload fe_mul --units 16777216 --steps 20 --warmup 30
pmc fetch FETCH_SIZE --steps 2 --warmup 1
pmc write WRITE_SIZE --steps 2 --warmup 1
pmc sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" --steps 2 --warmup 1
pmc fetch_fe FETCH_SIZE --workload fe_mul --units 16777216 --steps 2 --warmup 1
pmc write_fe WRITE_SIZE --workload fe_mul --units 16777216 --steps 2 --warmup 1

cd "$REPO"
python bench.py > "$OUT/bench_scalar_mul.json" 2> "$OUT/bench_scalar_mul.log"
python bench.py --mode fast --cpu-sample 0 > "$OUT/bench_scalar_mul_fast.json" 2>/dev/null
python bench.py --workload ristretto --cpu-sample 0 > "$OUT/bench_ristretto.json" 2>/dev/null
python bench.py --workload ristretto --units 4194304 --steps 3 --warmup 1 --cpu-sample 0 > "$OUT/bench_ristretto_2p22.json" 2>/dev/null
python bench.py --workload fe_mul --cpu-sample 0 > "$OUT/bench_fe_mul_2p20.json" 2>/dev/null
python bench.py --workload fe_mul --units 16777216 --steps 20 --warmup 30 --cpu-sample 0 > "$OUT/bench_fe_mul_2p24.json" 2>/dev/null
python bench.py --workload msm --cpu-sample 0 > "$OUT/bench_msm.json" 2>/dev/null
python bench.py --workload msm --units 2097152 --cpu-sample 0 > "$OUT/bench_msm_2p21.json" 2>/dev/null
python bench.py --workload msm --units 16777216 --steps 3 --warmup 1 --cpu-sample 0 > "$OUT/bench_msm_2p24.json" 2>/dev/null
python tools/bench_ops.py fe_add,fe_neg,fe_mul,fe_square 16777216 10 > "$OUT/ops.txt" 2>/dev/null
python tools/bench_ops.py fe_invert,fe_sqrt_ratio_i,ed_add,ed_double,ed_compress,ed_decompress,ris_compress,ris_decompress,ed_to_affine 1048576 10 >> "$OUT/ops.txt" 2>/dev/null
python tools/host_path.py 20 2>/dev/null | grep '"auto"\|"1"' > "$OUT/host_path.txt"
python tools/host_path.py 22 2>/dev/null | grep '"auto"\|"1"' >> "$OUT/host_path.txt"
python tools/step_probe.py > "$OUT/step_probe.txt" 2>/dev/null
ls -la "$OUT"

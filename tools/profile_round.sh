#!/bin/bash
# Collect one round's evidence on the GPU box (run from the repo root through gpurun):
#   tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/ (kernel traces, bench lines, ubench)
#                                        gpurun_out/pmc_<tag>/  (PMC passes, tools/profile_pmc.sh)
# then, back in the container:
#   python tools/summarize_profiles.py gpurun_out/prof_<tag> <tag>
#   python tools/make_roofline_inputs.py gpurun_out/pmc_<tag> gpurun_out/prof_<tag>/occupancy.txt <tag> gpurun_out/prof_<tag>
# rocprofv3 passes: kernel trace + stats per workload; PMC counters in their own passes (no trace
# domains), as MI355X_MICROARCH.md prescribes.
set -u
TAG=${1:-r04}
MODE=${2:-all}            # "bench": only the bench.py lines (after profiles/roofline_inputs.json was regenerated)
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
if [ "$MODE" != bench ]; then rm -rf "$OUT"; fi
mkdir -p "$OUT/tmp"
cd /tmp; export TMPDIR=/tmp
PY="python $REPO/bench.py --cpu-sample 0"

flatten() { find "$OUT/tmp" -name '*.csv' -exec mv {} "$OUT/" \; ; rm -rf "$OUT/tmp"/*; }
kt() { local name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tmp" -o "kt_$name" -- $PY "$@" > "$OUT/kt_$name.log" 2>&1; flatten; }

if [ "$MODE" != bench ]; then
kt scalar_mul --steps 5 --warmup 3
kt ristretto --workload ristretto --units 4194304 --steps 5 --warmup 1
kt msm_2p21 --workload msm --units 2097152 --steps 5 --warmup 1
kt msm_2p24 --workload msm --units 16777216 --steps 3 --warmup 1
kt fe_mul --workload fe_mul --units 16777216 --steps 20 --warmup 30

cd "$REPO"
tools/profile_pmc.sh "$TAG" > "$OUT/pmc.log" 2>&1
[ -x tools/ubench/occupancy ] && timeout 300 tools/ubench/occupancy 20 > "$OUT/occupancy.txt" 2>&1
fi
cd "$REPO"
python bench.py > "$OUT/bench_scalar_mul.json" 2> "$OUT/bench_scalar_mul.log"
python bench.py --scalar-bits 249 --cpu-sample 0 > "$OUT/bench_scalar_mul_s249.json" 2>/dev/null
python bench.py --units 16777216 --steps 3 --warmup 1 --cpu-sample 0 > "$OUT/bench_scalar_mul_2p24.json" 2>/dev/null
python bench.py --mode fast > "$OUT/bench_scalar_mul_fast.json" 2>/dev/null
python bench.py --workload ristretto > "$OUT/bench_ristretto_2p20.json" 2>/dev/null
python bench.py --workload ristretto --units 4194304 --steps 3 --warmup 1 > "$OUT/bench_ristretto_2p22.json" 2>/dev/null
python bench.py --workload fe_mul > "$OUT/bench_fe_mul_2p20.json" 2>/dev/null
python bench.py --workload fe_mul --units 16777216 --steps 20 --warmup 30 --cpu-sample 0 > "$OUT/bench_fe_mul_2p24.json" 2>/dev/null
python bench.py --workload fe_invert > "$OUT/bench_fe_invert_2p20.json" 2>/dev/null
python bench.py --workload fe_invert --units 16777216 --steps 10 --warmup 5 --cpu-sample 65536 > "$OUT/bench_fe_invert_2p24.json" 2>/dev/null
python bench.py --workload msm > "$OUT/bench_msm_2p20.json" 2>/dev/null
python bench.py --workload msm --units 2097152 > "$OUT/bench_msm_2p21.json" 2>/dev/null
python bench.py --workload msm --units 16777216 --steps 3 --warmup 1 > "$OUT/bench_msm_2p24.json" 2>/dev/null
python tools/bench_ops.py fe_add,fe_neg,fe_mul,fe_square 16777216 10 > "$OUT/ops.txt" 2>/dev/null
python tools/bench_ops.py fe_invert,fe_div,fe_sqrt_ratio_i,ed_add,ed_double,ed_neg,ed_eq,ris_eq,ed_is_valid,ed_compress,ed_decompress,ris_compress,ris_decompress,ed_to_affine 1048576 10 >> "$OUT/ops.txt" 2>/dev/null
python tools/bench_ops.py ed_mul_base,ris_mul_base_compress,ed_mul_base_wnaf5 4194304 5 >> "$OUT/ops.txt" 2>/dev/null
python bench.py --workload ecdh > "$OUT/bench_ecdh_2p20.json" 2>/dev/null
python bench.py --workload ecdh --ecdh reference --steps 5 --warmup 2 > "$OUT/bench_ecdh_reference_2p20.json" 2>/dev/null
python tools/host_path.py 20 2>/dev/null | grep '"auto"\|"1"\|slots' > "$OUT/host_path.txt"
python tools/host_path.py 22 2>/dev/null | grep '"auto"\|"1"\|slots' >> "$OUT/host_path.txt"
ls -la "$OUT"

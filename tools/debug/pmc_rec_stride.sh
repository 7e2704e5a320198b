#!/bin/bash
# Gathered bytes of the bucket-sum kernel with packed (96-byte stride) and line-aligned (128-byte stride) affine records:
# rocprofv3 --pmc FETCH_SIZE / TCC_EA0_RDREQ_sum (separate passes, no trace domains) over tools/msm_groups_sweep.py, per
# dispatch of k_msm_runs_affine and k_msm_prepare_affine.   usage: tools/debug/pmc_rec_stride.sh LOG2N [groups]
REPO=$PWD
LG=${1:-21}; GR=${2:-default}
cd /tmp; export TMPDIR=/tmp
for stride in 96 128; do
  for ctr in FETCH_SIZE TCC_EA0_RDREQ_sum; do
    rm -rf /tmp/pmcs; mkdir -p /tmp/pmcs
    timeout 600 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmcs -o p -- python $REPO/tools/msm_groups_sweep.py $LG $GR ZC_MSM_REC_STRIDE=$stride > /tmp/pmcs.log 2>&1
    python - "$stride" "$ctr" "$LG" <<'PY'
import csv, glob, sys
stride, ctr, lg = sys.argv[1], sys.argv[2], int(sys.argv[3])
fs = glob.glob('/tmp/pmcs/**/*counter_collection.csv', recursive=True)
if not fs:
    print("stride", stride, ctr, "no counter file (counter unavailable?)"); sys.exit(0)
tot, cnt = {}, {}
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0]
    if not k.startswith(("k_msm_runs_affine", "k_msm_prepare_affine", "k_msm_segments", "k_msm_runs_edges")) or r["Counter_Name"] != ctr:
        continue
    tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"]); cnt.setdefault(k, set()).add(r["Dispatch_Id"])
calls = 12          # the sweep script: 3 warm-up + 9 timed calls
for k in sorted(tot):
    per_call = tot[k] / calls
    unit = per_call * 1024 if ctr == "FETCH_SIZE" else per_call * 64      # FETCH_SIZE is in KiB; a read request is tallied at 64 bytes (MI355X_MICROARCH.md)
    print("stride %s %-18s %-22s %6d dispatches  %.4g per call = %.1f bytes per pair (x2 for wide coalesced reads per the guide: %.1f)"
          % (stride, ctr, k, len(cnt[k]), per_call, unit / (1 << lg), 2 * unit / (1 << lg)))
PY
  done
done

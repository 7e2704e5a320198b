#!/bin/bash
# Per-kernel times of one MSM size on the GPU box:  tools/debug/kt_msm_sizes.sh msm21 [msm24 ...]  (ZC_LIB_PATH selects a variant)
REPO=$PWD
cd /tmp; export TMPDIR=/tmp
for what in "$@"; do
  rm -rf /tmp/kt
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $REPO/tools/quick_bench.py $what > /tmp/kt.log 2>&1
  echo "== $what ${ZC_LIB_PATH:-default}"; tail -1 /tmp/kt.log
  python - <<'PY'
import csv,glob
f=glob.glob('/tmp/kt/**/kt_kernel_stats.csv',recursive=True)[0]
tot=0
rows=[r for r in csv.DictReader(open(f)) if r["Name"].startswith(("k_msm","k_scan","k_ed_scalar_mul","k_ed_add","k_ed_fold","void rocprim","__amd"))]
calls=max(int(r["Calls"]) for r in rows if r["Name"].startswith("k_msm_window_combine"))
for r in rows:
    per=float(r["TotalDurationNs"])/calls/1e3
    tot+=per
    print("%-44s calls/msm %5.1f  avg_us %9.1f  us/msm %9.1f"%(r["Name"][:44],int(r["Calls"])/calls,float(r["AverageNs"])/1e3,per))
print("sum us/msm %.1f"%tot)
PY
done

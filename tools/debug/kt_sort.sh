#!/bin/bash
# per-kernel times of the sort alone: tools/debug/kt_sort.sh "21 17" "24 19"   (ZC_LIB_PATH selects a variant)
REPO=$PWD
cd /tmp; export TMPDIR=/tmp
for what in "$@"; do
  rm -rf /tmp/kt
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $REPO/tools/debug/sort_probe.py $what > /tmp/kt.log 2>&1
  echo "== sort $what ${ZC_LIB_PATH:-default} $(tail -1 /tmp/kt.log)"
  python - <<'PY'
import csv,glob
f=glob.glob('/tmp/kt/**/kt_kernel_stats.csv',recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    if r["Name"].startswith(("k_msm","k_scan")):
        print("  %-34s avg_us %9.1f"%(r["Name"][:34],float(r["AverageNs"])/1e3)); tot+=float(r["AverageNs"])/1e3*int(r["Calls"])/3
print("  total/sort us %.1f"%tot)
PY
done

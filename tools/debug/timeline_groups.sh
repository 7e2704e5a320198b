#!/bin/bash
# kernel timeline (start offset, duration, queue) of the LAST zc_msm call of tools/msm_groups_sweep.py LOG2N SPEC [K=V ...]
# usage: tools/debug/timeline_groups.sh 21 7,6,3
REPO=$PWD
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python $REPO/tools/msm_groups_sweep.py "$@" > /tmp/tl.log 2>&1
tail -1 /tmp/tl.log
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/tl/**/tl_kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
starts=[i for i,r in enumerate(rows) if r["Kernel_Name"].startswith("k_msm_digits")]
seg=[r for r in rows[starts[-1]:] if r["Kernel_Name"].startswith(("k_msm","k_scan","k_ed_scalar_mul","k_ed_add"))]
t0=min(int(r["Start_Timestamp"]) for r in rows[starts[-1]-2:starts[-1]+1] if r["Kernel_Name"].startswith(("k_msm")))
for r in rows[starts[-1]-1:starts[-1]]:
    if r["Kernel_Name"].startswith("k_msm_prepare"): seg.insert(0,r)
for r in seg:
    s=(int(r["Start_Timestamp"])-t0)/1e3; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    print("%9.1f +%8.1f us  q%-3s grid %-8s %s"%(s,d,r.get("Queue_Id","?"),r.get("Grid_Size","?"),r["Kernel_Name"][:40]))
print("span %.1f us" % ((max(int(r["End_Timestamp"]) for r in seg)-t0)/1e3))
PY

#!/bin/bash
# kernel timeline (start offset, duration, queue) of the last MSM call of a quick_bench run: tools/debug/timeline.sh msm21
# (a call = from its k_msm_digits to the last k_msm_window_combine before the next k_msm_digits)
REPO=$PWD
what=$1
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python $REPO/tools/quick_bench.py $what > /tmp/tl.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/tl/**/tl_kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
starts=[i for i,r in enumerate(rows) if r["Kernel_Name"].startswith("k_msm_digits")]
first=starts[-1]
while first > 0 and rows[first - 1]["Kernel_Name"].startswith("k_msm_prepare"): first -= 1     # the normalisation is enqueued first, on its own stream
seg=[r for r in rows[first:] if r["Kernel_Name"].startswith(("k_msm","k_scan","k_ed_scalar_mul","k_ed_add"))]
last=max(i for i,r in enumerate(seg) if r["Kernel_Name"].startswith("k_msm_window_combine"))
seg=seg[:last+1]
t0=int(seg[0]["Start_Timestamp"])
for r in seg:
    s=(int(r["Start_Timestamp"])-t0)/1e3; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    print("%9.1f +%8.1f us  q%-3s grid %-8s %s"%(s,d,r.get("Queue_Id","?"),r.get("Grid_Size","?"),r["Kernel_Name"][:40]))
print("span %.1f us" % ((max(int(r["End_Timestamp"]) for r in seg)-t0)/1e3))
PY

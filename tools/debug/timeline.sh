#!/bin/bash
# kernel timeline (start offset, duration, queue) of the last MSM call of a quick_bench run: tools/debug/timeline.sh msm21
REPO=$PWD
what=$1
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python $REPO/tools/quick_bench.py $what > /tmp/tl.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/tl/**/tl_kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
# last occurrence of k_msm_window_combine ends a call; find the k_msm_digits/prepare before it
ends=[i for i,r in enumerate(rows) if r["Kernel_Name"].startswith("k_msm_window_combine")]
last=ends[-1]; prev=ends[-2] if len(ends)>1 else -1
seg=rows[prev+1:last+1]
seg=[r for r in seg if not r["Kernel_Name"].startswith(("k_ed_fold","__amd_rocclr_copy"))] 
t0=int(seg[0]["Start_Timestamp"])
for r in seg:
    s=(int(r["Start_Timestamp"])-t0)/1e3; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    print("%9.1f +%8.1f us  q%-3s %s"%(s,d,r.get("Queue_Id","?"),r["Kernel_Name"][:40]))
PY

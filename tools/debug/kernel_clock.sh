#!/bin/bash
# shader clock per kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, both from ONE --pmc pass
# (counter_collection.csv carries the dispatch timestamps): tools/debug/kernel_clock.sh "msm21 strict" [kernel-regex]
REPO=$PWD
what=$1; pat=${2:-k_}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pm
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -o pm -- python $REPO/tools/quick_bench.py $what > /tmp/pm.log 2>&1
python - "$pat" <<'PY'
import csv,glob,re,sys,collections
f=glob.glob('/tmp/pm/**/pm_counter_collection.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("columns:", list(rows[0].keys()))
acc=collections.defaultdict(list)
for r in rows:
    k=re.sub(r"<.*","",r["Kernel_Name"].split("(")[0])[:34]
    if not re.match(sys.argv[1],k) or r["Counter_Name"]!="GRBM_GUI_ACTIVE": continue
    st=r.get("Start_Timestamp"); en=r.get("End_Timestamp")
    d=(int(en)-int(st)) if st and en else 0
    acc[k].append((float(r["Counter_Value"]),d))
for k,v in sorted(acc.items(), key=lambda kv:-sum(x[1] for x in kv[1])):
    c=sum(x[0] for x in v); d=sum(x[1] for x in v)
    if d>0: print("%-34s disp %4d  avg_us %9.1f  clock_GHz %.3f"%(k,len(v),d/len(v)/1e3,c/8/d))
    else: print("%-34s disp %4d  GUI_ACTIVE %.4g (no timestamps)"%(k,len(v),c/len(v)))
PY

#!/bin/bash
# per-kernel PMC sums for one quick_bench size: tools/debug/pmc_kernels.sh msm21 "SQ_INSTS_VALU SQ_WAVE_CYCLES ..." [kernel-regex]
REPO=$PWD
what=$1; ctrs=$2; pat=${3:-k_}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pm
timeout 600 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pm -o pm -- python $REPO/tools/quick_bench.py $what > /tmp/pm.log 2>&1
python - "$pat" <<'PY'
import csv,glob,re,sys,collections
f=glob.glob('/tmp/pm/**/pm_counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); disp=collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k=re.sub(r"<.*","",r["Kernel_Name"].split("(")[0])[:34]
    if not re.match(sys.argv[1],k): continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
names=sorted({c for v in acc.values() for c in v})
print("%-34s %5s "%("kernel","disp")+" ".join("%18s"%n[-18:] for n in names))
for k,v in sorted(acc.items(), key=lambda kv:-kv[1].get(names[0],0)):
    print("%-34s %5d "%(k,len(disp[k]))+" ".join("%18.4g"%(v.get(n,0)/len(disp[k])) for n in names))
PY

#!/bin/bash
# avg time of selected kernels for one quick_bench size under an env: tools/debug/kt_kernels.sh msm21 'k_msm_prepare|k_msm_sort_scatter'
REPO=$PWD
what=$1; pat=$2
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $REPO/tools/quick_bench.py $what > /tmp/kt.log 2>&1
python - "$pat" <<'PY'
import csv,glob,re,sys
f=glob.glob('/tmp/kt/**/kt_kernel_stats.csv',recursive=True)[0]
print("  ".join("%s %.1f"%(r["Name"][:30],float(r["AverageNs"])/1e3) for r in csv.DictReader(open(f)) if re.match(sys.argv[1],r["Name"])))
PY

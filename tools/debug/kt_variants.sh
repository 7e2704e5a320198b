# usage: bash tools/debug/kt_variants.sh "<quick_bench args>" <kernel-prefix> variant ...   ("base" = the in-tree library)
# Kernel-trace A/B of library variants (build/variants/<name>.so): average duration of the kernels whose name starts with the prefix.
cd /tmp; export TMPDIR=/tmp
ARGS="$1"; PFX="$2"; shift 2
for v in "$@"; do
  rm -rf /tmp/kt
  if [ "$v" = base ]; then unset ZC_LIB_PATH; else export ZC_LIB_PATH=/root/repo/build/variants/$v.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python /root/repo/tools/quick_bench.py $ARGS > /tmp/kt_out.txt 2>&1
  echo "== $v $(grep -o '{.*}' /tmp/kt_out.txt | tail -1)"
  PFX="$PFX" python - <<'PY'
import csv,glob,os
f=glob.glob('/tmp/kt/**/kt_kernel_stats.csv',recursive=True)
if not f: print("   no trace"); raise SystemExit
for r in csv.DictReader(open(f[0])):
    if r["Name"].startswith(tuple(os.environ["PFX"].split(","))): print("   %-44s calls %4s avg_ms %.3f min_ms %.3f"%(r["Name"][:44],r["Calls"],float(r["AverageNs"])/1e6,float(r["MinNs"])/1e6))
PY
done

cd /tmp; export TMPDIR=/tmp
for v in /root/repo/build/variants/msm_nogather.so /root/repo/build/variants/msm_ng_nf.so /root/repo/build/variants/msm_ng_ilp.so /root/repo/build/variants/msm_nf.so; do
  rm -rf /tmp/kt; ZC_LIB_PATH=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python /root/repo/tools/quick_bench.py msm24 > /dev/null 2>&1
  echo "== $v"; python - <<'PY'
import csv,glob
f=glob.glob('/tmp/kt/**/kt_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Name"].startswith(("k_msm","void rocprim")): print("%-60s calls %s avg_ms %.3f"%(r["Name"][:60],r["Calls"],float(r["AverageNs"])/1e6))
PY
done

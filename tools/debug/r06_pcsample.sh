#!/bin/bash
# PC sampling of one MSM workload (beta feature of rocprofv3): where do the waves of k_msm_runs_affine spend their time?
# usage: tools/debug/r06_pcsample.sh <what: msm21>  -> gpurun_out/r06_pcs/
REPO=$PWD
OUT=$REPO/gpurun_out/r06_pcs
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for method in host_trap stochastic; do
  if [ $method = host_trap ]; then unit=time; interval=1; else unit=cycles; interval=1048576; fi
  timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $interval \
      --kernel-trace --output-format csv -d $OUT/$method -o pcs -- python $REPO/tools/quick_bench.py $1 > $OUT/$method.log 2>&1
  echo "$method rc=$?" >> $OUT/status.txt
  find $OUT/$method -name '*.csv' | head >> $OUT/status.txt
done
# keep the output small: the sample files can be large
find $OUT -name '*pc_sampling*.csv' -size +30M -exec sh -c 'head -c 30000000 "$1" > "$1.head"; rm "$1"' _ {} \;
ls -la $OUT $OUT/* >> $OUT/status.txt 2>&1

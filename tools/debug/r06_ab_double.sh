#!/bin/bash
# Round 6 A/B: ed_double with squarings (shipped) against pt_add_plain(a, a) (ZC_ED_DOUBLE_SQR=0) -> gpurun_out/r06_ab_double.txt
out=gpurun_out/r06_ab_double.txt
: > $out
for rep in 1 2 3; do
  for v in product dbl_mul; do
    if [ $v = product ]; then unset ZC_LIB_PATH; else export ZC_LIB_PATH=$PWD/build/variants/$v.so; fi
    echo "== $v rep $rep" >> $out
    python tools/bench_ops.py ed_double,ed_add 1048576,16777216 20 >> $out 2>&1
  done
done
unset ZC_LIB_PATH

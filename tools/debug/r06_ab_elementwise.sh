#!/bin/bash
# Round 6 A/B, one box: (a) Neg with / without the LDS-staged kernel, (b) Mul / Square with the product form chosen per wave /
# per lane (round 5), on canonical and raw operands.  Alternating runs; output: gpurun_out/r06_ab_elementwise.txt
out=gpurun_out/r06_ab_elementwise.txt
: > $out
for rep in 1 2 3; do
  for v in product neg_lane; do
    if [ $v = product ]; then unset ZC_LIB_PATH; else export ZC_LIB_PATH=$PWD/build/variants/$v.so; fi
    echo "== $v rep $rep" >> $out
    python tools/bench_ops.py fe_neg,sc_neg 16777216,67108864 20 >> $out 2>&1
  done
  for v in product mulsq_lane_branch; do
    if [ $v = product ]; then unset ZC_LIB_PATH; else export ZC_LIB_PATH=$PWD/build/variants/$v.so; fi
    echo "== $v rep $rep" >> $out
    python tools/bench_ops.py fe_mul,fe_square,sc_mul,sc_mul_raw,sc_square,sc_square_raw 16777216 20 >> $out 2>&1
  done
done
unset ZC_LIB_PATH

#!/bin/bash
# Round 6 A/B: fe_invert at 2^20 with c = 16 (one wave per SIMD, shipped), 8 (two) and 4 (four) elements per lane,
# column-ordered multiplier (shipped) against the independent-chain multiplier (ZC_INV_MUL_ILP=1) -> gpurun_out/r06_ab_invert.txt
out=gpurun_out/r06_ab_invert.txt
: > $out
for rep in 1 2 3; do
  for v in product inv_ilp; do
    if [ $v = product ]; then unset ZC_LIB_PATH; else export ZC_LIB_PATH=$PWD/build/variants/$v.so; fi
    for c in 0 8 4 2; do
      if [ $c = 0 ]; then unset ZC_INV_CHUNK; else export ZC_INV_CHUNK=$c; fi
      echo "== $v c=$c rep $rep" >> $out
      python tools/bench_ops.py fe_invert 1048576,16777216 20 >> $out 2>&1
    done
  done
done
unset ZC_LIB_PATH ZC_INV_CHUNK

#!/bin/bash
# Ablation builds of the fused kernel (tools/build.sh noload nostore nogather nobarrier noload_nostore) against the shipped
# library in one kbench run each, C3, 128 frames per launch: what the taps, the stores and the per-item barrier cost.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03; mkdir -p $O
: > $O/ablations.txt
for v in shipped noload nogather nostore nobarrier noload_nostore shipped; do
  L=$R/tools/_build/lib_$v.so; [ $v = shipped ] && L=$R/gyroflow_amd/libgfremap.so
  echo "== $v" >> $O/ablations.txt
  timeout -k 10 120 $R/tools/_build/kbench --lib $L --frames 128 --iters 10 >> $O/ablations.txt 2>&1 || { echo "$v failed"; exit 1; }
done
cat $O/ablations.txt

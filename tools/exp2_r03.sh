#!/bin/bash
# A/B of experiment builds against the shipped library at C3 and C1 (kbench, checksums kept so that a build that changes pixels shows)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03; mkdir -p $O
: > $O/exp2.txt
for v in "$@"; do
  L=$R/tools/_build/lib_$v.so; [ $v = shipped ] && L=$R/gyroflow_amd/libgfremap.so
  echo "== $v" >> $O/exp2.txt
  for a in "--frames 128" "--in 1920x1080 --out 1920x1080 --frames 128"; do
    timeout -k 10 120 $R/tools/_build/kbench --lib $L --iters 10 $a >> $O/exp2.txt 2>&1 || { echo "$v failed"; exit 1; }
  done
done
cat $O/exp2.txt

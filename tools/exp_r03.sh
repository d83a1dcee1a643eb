#!/bin/bash
# A/B of experiment builds against the shipped library (kbench, interleaved)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03; mkdir -p $O
: > $O/exp.txt
for v in "$@"; do
  L=$R/tools/_build/lib_$v.so; [ $v = shipped ] && L=$R/gyroflow_amd/libgfremap.so
  echo "== $v" >> $O/exp.txt
  for a in "--frames 128" "--frames 32" "--in 1920x1080 --out 1920x1080 --frames 128" "--in 1920x1080 --out 1920x1080 --frames 32" "--in 5312x2988 --out 5312x2988 --depth 2 --frames 64"; do
    timeout -k 10 120 $R/tools/_build/kbench --lib $L --iters 10 $a 2>&1 | sed 's/checksum.*//; s/, [0-9]* frames\/s.*//' >> $O/exp.txt || { echo "$v failed"; exit 1; }
  done
done
cat $O/exp.txt

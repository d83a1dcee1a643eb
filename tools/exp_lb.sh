#!/bin/bash
# the half-tile LDS-box kernel (tools/build.sh lb, selected by GF_K1=rl) against the shipped library, kbench with checksums
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03; mkdir -p $O
: > $O/exp_lb.txt
run() { echo "== $1 ${2:-}" >> $O/exp_lb.txt; for a in "--frames 128" "--frames 32" "--in 1920x1080 --out 1920x1080 --frames 512" "--frames 128 --border 1" "--frames 128 --amp 12"; do
  env ${2:-X=1} timeout -k 10 120 $R/tools/_build/kbench --lib $1 --iters 10 $a >> $O/exp_lb.txt 2>&1 || { echo "failed: $1 $a"; tail -3 $O/exp_lb.txt; exit 1; }; done; }
run $R/gyroflow_amd/libgfremap.so
for v in "$@"; do run $R/tools/_build/lib_$v.so GF_K1=rl; done
run $R/gyroflow_amd/libgfremap.so
cat $O/exp_lb.txt

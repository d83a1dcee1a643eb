#!/bin/bash
# What K1 costs when its source comes from L2 / the memory-side cache instead of HBM: every frame of a launch reads source frame 0
# (kbench --same-src), against the normal run; then the same with the amplitude of the camera motion set to 0 (all frames identical work).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03; mkdir -p $O
: > $O/exp3.txt
for a in "--frames 128" "--frames 128 --same-src" "--frames 128 --amp 0" "--frames 128 --amp 0 --same-src" \
         "--in 1920x1080 --out 1920x1080 --frames 512" "--in 1920x1080 --out 1920x1080 --frames 512 --same-src" \
         "--in 1280x720 --out 1280x720 --frames 1024" "--in 1280x720 --out 1280x720 --frames 1024 --same-src" "--frames 128"; do
  timeout -k 10 120 $R/tools/_build/kbench --iters 10 $a >> $O/exp3.txt 2>&1 || { echo "failed: $a"; exit 1; }
done
cat $O/exp3.txt

#!/bin/bash
# usage: direct_ab.sh lib1 lib2 ... : kbench C3 with the per-pixel (`direct`) kernel, interleaved twice
out=gpurun_out/r2/direct_ab.txt; mkdir -p gpurun_out/r2; : > $out
export GF_K1=direct
for rep in 1 2; do
for l in "$@"; do
  echo -n "$l " >> $out
  timeout -k 10 90 tools/_build/kbench --frames 64 --lib tools/_build/$l.so 2>&1 | tail -1 | sed 's/.*median/median/' | cut -c1-80 >> $out || exit 1
done
done

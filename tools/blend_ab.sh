#!/bin/bash
# usage: blend_ab.sh lib1 lib2 ... : kbench of the fused blend at C5 (5312x2988 u16, 4 of 4 frames blended), interleaved twice
out=gpurun_out/r2/blend_ab.txt; mkdir -p gpurun_out/r2; : > $out
for rep in 1 2; do
for l in "$@"; do
  echo -n "$l " >> $out
  timeout -k 10 90 tools/_build/kbench --in 5312x2988 --out 5312x2988 --depth 2 --frames 64 --blend 4 --lib tools/_build/$l.so 2>&1 | tail -1 | sed 's/.*median/median/' | cut -c1-80 >> $out || exit 1
done
done

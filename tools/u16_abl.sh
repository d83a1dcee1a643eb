#!/bin/bash
# usage: u16_abl.sh lib1 lib2 ... : kbench at the C5 frame size (5312x2988 u16), twice per library, interleaved
out=gpurun_out/r2/u16_abl.txt; mkdir -p gpurun_out/r2; : > $out
for rep in 1 2; do
for l in "$@"; do
  echo -n "$l " >> $out
  timeout -k 10 90 tools/_build/kbench --in 5312x2988 --out 5312x2988 --depth 2 --frames 32 --lib tools/_build/$l.so 2>&1 | tail -1 | sed 's/.*median/median/' | cut -c1-70 >> $out || exit 1
done
done

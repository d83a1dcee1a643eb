#!/bin/bash
out=gpurun_out/r2/u16_abl.txt; mkdir -p gpurun_out/r2; : > $out
for l in lib_head lib_nostore lib_noload; do
  echo -n "$l " >> $out
  timeout -k 10 90 tools/_build/kbench --in 5312x2988 --out 5312x2988 --depth 2 --frames 32 --lib tools/_build/$l.so 2>&1 | tail -1 | sed 's/.*median/median/' | cut -c1-70 >> $out || exit 1
done

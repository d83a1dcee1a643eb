#!/bin/bash
# persistent grid size (workgroups per CU) of the RP kernel against launch size: kbench C3
out=gpurun_out/r2/wg_sweep2.txt; mkdir -p gpurun_out/r2; : > $out
for fr in 32 64 128; do
  for wg in 32 40 48 64; do
    echo -n "frames=$fr wg=$wg " >> $out
    GF_WG_PER_CU_RP=$wg timeout -k 10 60 tools/_build/kbench --frames $fr 2>&1 | tail -1 | sed 's/.*median/median/' | cut -c1-60 >> $out || exit 1
  done
done

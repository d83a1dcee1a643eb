#!/bin/bash
# kbench across modes and sizes (one line each) -- a regression check of every fused kernel variant after a kernel change
out=gpurun_out/r2/sanity_rates.txt; mkdir -p gpurun_out/r2; : > $out
run() { echo -n "$1: " >> $out; shift; timeout -k 10 90 "$@" 2>&1 | tail -1 | cut -c1-170 >> $out || exit 1; }
run "rp   C3 u8 b64" tools/_build/kbench --frames 64
GF_K1=direct run "dir  C3 u8 b64" tools/_build/kbench --frames 64
GF_K1=rl run "rl   C3 u8 b64" tools/_build/kbench --frames 64
run "rp   C3 u8 b64 replicate" tools/_build/kbench --frames 64 --border 1
run "rp   C3 u8 b64 transparent" tools/_build/kbench --frames 64 --border 5
run "rp   C1 u8 b128" tools/_build/kbench --in 1920x1080 --out 1920x1080 --frames 128
run "rp   C2 u8 b128" tools/_build/kbench --in 2704x2028 --out 1920x1080 --frames 128
run "rp   C5 u16 b32" tools/_build/kbench --in 5312x2988 --out 5312x2988 --depth 2 --frames 32
run "rp   C5 u16 b64 blend4" tools/_build/kbench --in 5312x2988 --out 5312x2988 --depth 2 --frames 64 --blend 4
run "rp   C3 u8 b64 blend4" tools/_build/kbench --frames 64 --blend 4
GF_K1=direct run "dir  C3 u8 b64 blend4" tools/_build/kbench --frames 64 --blend 4
run "rp   C3 u8 b64 blend12" tools/_build/kbench --frames 72 --blend 12

#!/bin/bash
# usage: ab.sh out.txt lib1 lib2 ... ; runs kbench C3 at 32 and 128 frames for each lib, twice, interleaved
out=$1; shift; mkdir -p $(dirname $out)
: > $out
for rep in 1 2; do
for l in "$@"; do
  for fr in 32 128; do
    echo -n "$l " >> $out
    timeout -k 10 60 tools/_build/kbench --frames $fr --lib tools/_build/$l 2>&1 | tail -1 >> $out || exit 1
  done
done
done

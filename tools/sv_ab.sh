#!/bin/bash
# usage: sv_ab.sh <bench config> lib1 lib2 ... ; rocprofv3 kernel stats of bench.py for each library under tools/_build
cfg=$1; shift
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
for l in "$@"; do
  export GF_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/$l.so
  rm -rf gpurun_out/r2/ab_$l
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r2/ab_$l -o p -- python3 bench.py --config $cfg --cpu-frames 0 --no-pcie > gpurun_out/r2/ab_$l.json 2> gpurun_out/r2/ab_$l.err || exit 1
done

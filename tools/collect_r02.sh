#!/bin/bash
# Round-2 evidence run on the GPU box (part A): bench lines of every config + rocprofv3 kernel stats of the default command.
# Every GPU step has its own time limit; the script stops at the first failure.
set -u
O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
step() { echo "== $*" >&2; "$@" || { echo "FAILED ($?): $*" >&2; exit 1; }; }
for c in c3 c1 c2 c4 c4_twopass c5 c5_noblend; do
  step timeout -k 10 280 python3 bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err || exit 1
  echo "$c done" >&2
done
step timeout -k 10 200 python3 bench.py --scaling strong --cpu-frames 0 --no-pcie > $O/bench_c3_strong.json 2> $O/bench_c3_strong.err || exit 1
step timeout -k 10 200 python3 bench.py --mode m2 --cpu-frames 0 --no-pcie > $O/bench_c3_m2.json 2> $O/bench_c3_m2.err || exit 1
step timeout -k 10 200 python3 bench.py --batch 64 --cpu-frames 0 --no-pcie > $O/bench_c3_b64.json 2> $O/bench_c3_b64.err || exit 1
for c in c3 c4 c4_twopass c5; do
  step timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -o s -- python3 bench.py --config $c --cpu-frames 0 --no-pcie > $O/bench_${c}_under_rocprof.json 2> $O/stats_$c.err || exit 1
done
echo "part A done" >&2

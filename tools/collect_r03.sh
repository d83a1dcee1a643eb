#!/bin/bash
# Round-3 evidence run on the GPU box (part A): bench lines of every config, rocprofv3 kernel stats of the default command,
# render-loop rates (HISTORY batched vs per-frame, rotated source).  Every GPU step has its own time limit; stops at the first failure.
set -u
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
step() { echo "== $*" >&2; "$@" || { echo "FAILED ($?): $*" >&2; exit 1; }; }
for c in ${CONFIGS:-c3 c1 c2 c4 c4_fused c5 c5_noblend}; do
  step timeout -k 10 280 python3 bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err || exit 1
  echo "$c done" >&2
done
step timeout -k 10 200 python3 bench.py --scaling strong --cpu-frames 0 --no-pcie > $O/bench_c3_strong.json 2> $O/bench_c3_strong.err || exit 1
step timeout -k 10 200 python3 bench.py --batch 64 --cpu-frames 0 --no-pcie > $O/bench_c3_b64.json 2> $O/bench_c3_b64.err || exit 1
step timeout -k 10 200 python3 bench.py --batch 32 --cpu-frames 0 --no-pcie > $O/bench_c3_b32.json 2> $O/bench_c3_b32.err || exit 1
# the launch path the driver uses for N > 1, rehearsed with both ranks on this box's one GPU (not a scaling number)
step timeout -k 10 300 python3 bench.py --gpus 2 --oversubscribe --cpu-frames 0 --no-pcie --steps 5 > $O/bench_c3_2ranks_one_gpu.json 2> $O/bench_c3_2ranks_one_gpu.err || exit 1
for c in c3 c4 c5; do
  step timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -o s -- python3 bench.py --config $c --cpu-frames 0 --no-pcie > $O/bench_${c}_under_rocprof.json 2> $O/stats_$c.err || exit 1
done
: > $O/render_bench.jsonl
step timeout -k 10 250 python3 tools/render_bench.py --frames 1200 >> $O/render_bench.jsonl 2>> $O/render_bench.err || exit 1
step timeout -k 10 250 python3 tools/render_bench.py --frames 1200 --bg HISTORY >> $O/render_bench.jsonl 2>> $O/render_bench.err || exit 1
step timeout -k 10 250 python3 tools/render_bench.py --frames 300 --bg HISTORY --sequential >> $O/render_bench.jsonl 2>> $O/render_bench.err || exit 1
step timeout -k 10 250 python3 tools/render_bench.py --frames 1200 --rotate 1 >> $O/render_bench.jsonl 2>> $O/render_bench.err || exit 1
step timeout -k 10 250 python3 tools/render_bench.py --frames 1200 --rotate 0 >> $O/render_bench.jsonl 2>> $O/render_bench.err || exit 1
cat $O/render_bench.jsonl >&2
echo "part A done" >&2

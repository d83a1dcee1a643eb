#!/bin/bash
# Round-3 evidence run, counters: for every bench config three rocprofv3 --pmc runs (counters only, one run per pass, the
# program directly after `--`): HBM read requests by size, HBM write requests, VALU issue.  tools/make_traffic.py turns the
# CSVs into profiles/traffic.json (run it here, in the container, after gpurun has merged gpurun_out/).
#   tools/collect_r03_pmc.sh c3 c1 c2 c4 c4_fused c5 ...      (an argument cfg:B runs `--batch B`)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
P1="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
P2="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_READ_sum TCC_WRITE_sum"
P3="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE"
for a in "$@"; do
  c=${a%%:*}; b=""; tag=$c
  if [ "$a" != "$c" ]; then b="--batch ${a##*:}"; tag=${c}_b${a##*:}; fi
  OUT=$R/gpurun_out/pmc_r03_$tag; mkdir -p $OUT
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- python3 bench.py --config $c $b --cpu-frames 0 --no-pcie --steps 3 --warmup 1 > $OUT/pass$i.log 2>&1 || { echo "$tag pass $i failed"; tail -5 $OUT/pass$i.log; exit 1; }
  done
  grep -h '^{' $OUT/pass3.log > $OUT/bench_line.json || true
  echo "$tag done"
done

#!/bin/bash
# tools/pmc.sh TAG -- kbench args...        : rocprofv3 counter passes (one run each) for the fused kernel, results
# tools/pmc.sh TAG -- python3 bench.py ...    under gpurun_out/pmc_TAG/passN/.  Counters only with --pmc (never combined
#                                             with tracing on this pool); the profiled program follows `--` directly.
set -u
TAG=$1; shift
[ "$1" = "--" ] && shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL"
P3="SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum"
P4="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"
P5="FETCH_SIZE"
P7="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
P8="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_READ_sum TCC_WRITE_sum"
P6="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6" "$P7" "$P8"; do
  i=$((i+1))
  if [ "${1:-}" = "python3" ]; then CMD=("$@"); else CMD=($R/tools/_build/kbench --iters 2 "$@"); fi
  timeout -s KILL 150 rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- "${CMD[@]}" > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; exit 1; }
done
echo "pmc passes done: $OUT"

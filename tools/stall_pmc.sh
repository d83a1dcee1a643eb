#!/bin/bash
# tools/stall_pmc.sh TAG -- <program and args>   : the SQ counters behind DESIGN.md's "where the non-VALU time goes" table,
# one rocprofv3 --pmc run per pass (counters only, never combined with tracing); the profiled program follows `--` directly.
set -u
TAG=$1; shift
[ "$1" = "--" ] && shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/stall_$TAG
mkdir -p $OUT
PA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
PB="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
PC="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LEVEL_WAVES"
PD="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU"
PE="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_BRANCH SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"
i=0
for P in "$PA" "$PB" "$PC" "$PD" "$PE"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- "$@" > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; exit 1; }
done
echo "stall passes done: $OUT"

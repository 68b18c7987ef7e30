#!/bin/bash
# Issue / memory-path counters of a command, separate rocprofv3 passes (kernel-trace only):
#   tools/pmc_mem.sh <tag> <command ...>     -> gpurun_out/pmc_<tag>/, summary on stdout
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters
  local name=$1 ctr=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$name -o $name -- "$@" > $OUT/$name.log 2>&1
}
run sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "$@"
run vmem "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH" "$@"
run sq3 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "$@"
run tlb "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum" "$@"
run lat "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" "$@"
run ta "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "$@"
cd $R
python tools/pmc_summary.py $OUT

#!/bin/bash
# SQ / memory counters of an arbitrary command, in separate rocprofv3 passes (kernel-trace only):
#   tools/pmc_cmd.sh <tag> <command ...>          -> gpurun_out/pmc_<tag>/, summary on stdout
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
run sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "$@"
run sq2 "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "$@"
run fetch "FETCH_SIZE GRBM_GUI_ACTIVE" "$@"
run write "WRITE_SIZE" "$@"
cd $R
python tools/pmc_summary.py $OUT

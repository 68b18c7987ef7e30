#!/bin/bash
# Collect PMC counters for the bench command in separate rocprofv3 passes
# (kernel-trace only: never combined with sys/hip/hsa traces).
#   tools/pmc.sh <tag> <bench args...>
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EXTRA=("$@")
run() { # name, counters
  timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o $1 -- \
    python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra "${EXTRA[@]}" > $OUT/$1.log 2>&1
}
run sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
run sq2 "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"
run fetch "FETCH_SIZE GRBM_GUI_ACTIVE"
run write "WRITE_SIZE"
run tcc "TCC_HIT_sum TCC_MISS_sum"
ls -R $OUT | head -40

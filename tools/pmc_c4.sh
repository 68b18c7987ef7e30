#!/bin/bash
# SQ counters of the C4 kernels (colorspace, histogram, apply_lut): separate --pmc passes
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$PWD/gpurun_out/${1:-pmc_c4}
mkdir -p $OUT
R=$PWD
cd /tmp; export TMPDIR=/tmp
run() {
  timeout 200 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o $1 -- python $R/tools/run_configs.py c4 > $OUT/$1.log 2>&1
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
run b "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES"
run c "FETCH_SIZE GRBM_GUI_ACTIVE"
run d "WRITE_SIZE"
run e "TCC_HIT_sum TCC_MISS_sum"
cd $R
python tools/pmc_summary.py $OUT

#!/bin/bash
# Round 3 A/B of library variants in ONE gpurun call:
#   tools/gpu_r3_ab.sh <tag> "<suffix list, '-' = default>" "<mode list: exact fast>" [pmc] [tests]
TAG=$1; VARIANTS=$2; MODES=${3:-exact}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
L=$PWD/imagemagick_amd/lib
for round in 1 2; do
for v in $VARIANTS; do
  [ "$v" = "-" ] && s="" || s="_$v"
  for mode in $MODES; do
    MAGICKHIP_LIBRARY=$L/libmagickhip$s.so timeout 120 python tools/time_blur_modes.py $mode 8192 10 4 2>&1 | tail -1 | sed "s/^/lib$s r$round: /"
  done
done
done 2>&1 | tee $OUT/ab.txt
if [[ "$*" == *tests* ]]; then
  ( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "blur or unsharp" ) 2>&1 | tail -8 | tee $OUT/pytest.txt
fi
if [[ "$*" == *pmc* ]]; then
  R=$PWD
  cd /tmp
  for mode in $MODES; do
  run() {
    timeout 200 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/$OUT/pmc_${mode}_$1 -o $1 -- \
      python $R/tools/time_blur_modes.py $mode 8192 10 4 > $R/$OUT/pmc_${mode}_$1.log 2>&1
  }
  run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA"
  run b "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES"
  run c "FETCH_SIZE GRBM_GUI_ACTIVE"
  run d "WRITE_SIZE"
  cd $R
  python tools/pmc_summary.py $OUT 2>/dev/null | grep -A40 "blur_fused" | head -80 | tee $OUT/pmc_$mode.txt
  cd /tmp
  done
fi

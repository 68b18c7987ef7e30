#!/bin/bash
# fused8 vs fused16 in one call: timing (both), parity tests (default = fused8)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-f8}
mkdir -p $OUT
export TMPDIR=/tmp
for round in 1 2 3; do
  timeout 120 python tools/time_blur_passes.py 2>&1 | tail -1 | sed "s/^/fused8  r$round: /"
  MAGICKHIP_FUSED_16=1 timeout 120 python tools/time_blur_passes.py 2>&1 | tail -1 | sed "s/^/fused16 r$round: /"
done 2>&1 | tee $OUT/ab.txt
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blur or unsharp" ) 2>&1 | tail -5 | tee $OUT/pytest_parity.txt
( timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "c2_blur" ) 2>&1 | tail -5 | tee $OUT/pytest_fullsize.txt

#!/bin/bash
# HDRI (float Quantum) stencils: parity tests, then the 8192^2 blur with and without the tie-check path
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r3_hdri}
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "blur or unsharp or convolve or morphology" ) 2>&1 | tail -8 | tee $OUT/pytest.txt
for round in 1 2; do
  timeout 120 python tools/time_blur_modes.py hdri 8192 10 4 2>&1 | tail -1 | sed "s/^/tie r$round: /"
  MAGICKHIP_NO_TIE64=1 timeout 120 python tools/time_blur_modes.py hdri 8192 10 4 2>&1 | tail -1 | sed "s/^/exact64 r$round: /"
done 2>&1 | tee $OUT/ab.txt

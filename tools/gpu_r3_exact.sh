#!/bin/bash
# Round 3: the exact-integer fused blur on the GPU box.  tools/gpu_r3_exact.sh <tag>
TAG=${1:-r3x}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "blur or unsharp" ) > $OUT/pytest.log 2>&1
tail -40 $OUT/pytest.log
for mode in exact fast; do
  timeout 120 python tools/time_blur_modes.py $mode 8192 10 4 2>&1 | tail -1 | tee -a $OUT/times.log
  MAGICKHIP_NO_EXACT_MFMA=1 timeout 120 python tools/time_blur_modes.py $mode 8192 10 4 2>&1 | tail -1 | tee -a $OUT/times.log
done
OPAQUE=1 timeout 120 python tools/time_blur_modes.py exact 8192 10 4 2>&1 | tail -1 | tee -a $OUT/times.log
timeout 120 python tools/time_blur_modes.py exact 8192 10 3 2>&1 | tail -1 | tee -a $OUT/times.log
timeout 120 python tools/time_blur_modes.py fast 8192 10 3 2>&1 | tail -1 | tee -a $OUT/times.log
timeout 120 python tools/time_blur_modes.py exact 8192 2 4 2>&1 | tail -1 | tee -a $OUT/times.log
timeout 120 python tools/time_blur_modes.py exact 8192 5 4 2>&1 | tail -1 | tee -a $OUT/times.log

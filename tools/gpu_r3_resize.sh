#!/bin/bash
# C3 resize: parity, then timing with and without the premultiplied staging
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r3_resize}
mkdir -p $OUT
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "resize" ) 2>&1 | tail -6 | tee $OUT/pytest.txt
for round in 1 2; do
  timeout 200 python tools/run_resize.py fast 3 2>&1 | tail -3 | sed "s/^/premultiplied r$round: /"
  MAGICKHIP_NO_RESIZE_PREMULTIPLY=1 timeout 200 python tools/run_resize.py fast 3 2>&1 | tail -3 | sed "s/^/plain r$round: /"
done 2>&1 | tee $OUT/times.txt

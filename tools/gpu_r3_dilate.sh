#!/bin/bash
# C5 Dilate: parity of the strip walk, then the 16384^2 timing of the three kernels
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r3_dilate}
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "strip or symmetric_convex or morphology" ) 2>&1 | tail -8 | tee $OUT/pytest.txt
timeout 300 python tools/time_dilate.py 16384 "Disk:15,Disk:7,Square:8,Octagon:10" 2>&1 | tee $OUT/times.txt

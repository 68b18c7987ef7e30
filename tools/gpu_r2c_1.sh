#!/bin/bash
# round 2c, call 1: packed vs scalar f32 in the fused blur, phase trace, fused parity
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r2c1
mkdir -p $OUT
export TMPDIR=/tmp
L=$PWD/imagemagick_amd/lib
for round in 1 2 3; do
for v in "" _packed _prio; do
  MAGICKHIP_LIBRARY=$L/libmagickhip$v.so timeout 120 python tools/time_blur_passes.py 2>&1 | tail -1 | sed "s/^/lib$v r$round: /"
done
done 2>&1 | tee $OUT/ab.txt
MAGICKHIP_LIBRARY=$L/libmagickhip_trace.so timeout 120 python tools/trace_fused_blur.py 2>&1 | tee $OUT/trace.txt
cp /tmp/fused_trace.bin $OUT/ 2>/dev/null
( time timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "c2_blur" ) 2>&1 | tail -5 | tee $OUT/pytest_fullsize.txt
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blur or unsharp" ) 2>&1 | tail -5 | tee $OUT/pytest_parity.txt

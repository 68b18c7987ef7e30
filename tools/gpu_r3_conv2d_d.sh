#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/conv2dx_d
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "convol or separable" > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log | cut -c1-300
for e in "" "MAGICKHIP_CONV2D_ROWS32=1"; do
  for layout in plain4 rgb rgba; do
    env $e timeout 200 python tools/time_convolve2d.py 16384 Disk:15,Disk:7.3,Square:3 $layout 2>&1 | grep "i8 exact " | sed "s/^/[$e] /"
  done
done | tee $OUT/time.log
STRESS_OPS=12 timeout 200 python tests/stress_parity.py 60 61 2>&1 | tail -2

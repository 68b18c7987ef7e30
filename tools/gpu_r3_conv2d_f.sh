#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/conv2dx_f
mkdir -p $OUT
cd $R
for layout in rgba plain4 rgb; do
  timeout 300 python tools/time_convolve2d.py 16384 Disk:2.5,Octagon:3,Octagon:5,Disk:7.3,Octagon:8 $layout 2>&1 | grep "i8 exact  \|f16 (FAST) "
done | tee $OUT/time.log

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/conv2dx_e
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "convol or separable" > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log | cut -c1-300
for layout in rgba plain4 rgb; do
  timeout 200 python tools/time_convolve2d.py 16384 Disk:15,Disk:7.3 $layout 2>&1 | grep "i8 exact "
done | tee $OUT/time.log
timeout 200 python tools/time_convolve2d_hdri.py 2>&1 | grep -v amdgpu.ids | head -1 | tee -a $OUT/time.log
STRESS_OPS=12 timeout 200 python tests/stress_parity.py 45 71 2>&1 | tail -2

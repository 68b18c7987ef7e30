#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/conv2dx_c
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "convol or separable or sharpen or edge or gaussian or morphology" > $OUT/tests.log 2>&1
tail -15 $OUT/tests.log | cut -c1-300
timeout 300 python tools/time_convolve2d_hdri.py 2>&1 | grep -v amdgpu.ids | tee $OUT/time_hdri.log

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/tie2d
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "convol or separable or sharpen or edge or emboss or morphology" > $OUT/tests.log 2>&1
tail -12 $OUT/tests.log | cut -c1-300
STRESS_OPS=13 timeout 200 python tests/stress_parity.py 45 81 2>&1 | tail -4
timeout 300 python tools/time_convolve2d.py 8192 Disk:15,Octagon:5,LoG:0x3 rgba 2>&1 | grep -v "amdgpu.ids\|against" | tee $OUT/time.log
timeout 300 python tools/time_convolve2d_hdri.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/time.log

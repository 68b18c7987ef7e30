#!/bin/bash
# round 4, second GPU call: the hybrid FAST blur — parity tests, timings, stress
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "fast or blur or unsharp or column_kernel" > gpurun_out/r4b/tests.log 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/r4b/tests.log
timeout 600 python tools/time_blur_r4.py 8192 10 > gpurun_out/r4b/time_blur.txt 2>&1
cat gpurun_out/r4b/time_blur.txt
timeout 300 python tools/time_blur_r4.py 4096 2 > gpurun_out/r4b/time_blur_s2.txt 2>&1
tail -25 gpurun_out/r4b/time_blur_s2.txt
STRESS_OPS=0 timeout 200 python tests/stress_parity.py 90 61 > gpurun_out/r4b/stress.txt 2>&1
tail -5 gpurun_out/r4b/stress.txt

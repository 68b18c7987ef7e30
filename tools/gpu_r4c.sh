#!/bin/bash
mkdir -p gpurun_out/r4c
timeout 600 python tools/time_blur_r4.py 8192 10 > gpurun_out/r4c/time_blur.txt 2>&1
cat gpurun_out/r4c/time_blur.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -k "fast or blur or unsharp" > gpurun_out/r4c/tests.log 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/r4c/tests.log
STRESS_OPS=0 timeout 200 python tests/stress_parity.py 60 62 > gpurun_out/r4c/stress.txt 2>&1
tail -3 gpurun_out/r4c/stress.txt

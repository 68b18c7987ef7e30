#!/bin/bash
mkdir -p gpurun_out/r4e
MAGICKHIP_LIBRARY=$PWD/imagemagick_amd/lib/libmagickhip_knock.so timeout 300 python tools/time_hybrid_knock.py 2>&1 | tee gpurun_out/r4e/knock.txt
timeout 600 python tools/time_blur_r4.py 8192 10 > gpurun_out/r4e/time_blur.txt 2>&1
cat gpurun_out/r4e/time_blur.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -k "fast or blur or unsharp" > gpurun_out/r4e/tests.log 2>&1
echo "tests rc=$?"; tail -8 gpurun_out/r4e/tests.log
STRESS_OPS=0,2 timeout 200 python tests/stress_parity.py 60 63 > gpurun_out/r4e/stress.txt 2>&1
tail -3 gpurun_out/r4e/stress.txt

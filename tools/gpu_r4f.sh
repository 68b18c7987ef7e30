#!/bin/bash
mkdir -p gpurun_out/r4f
MAGICKHIP_LIBRARY=$PWD/imagemagick_amd/lib/libmagickhip_knock.so timeout 300 python tools/time_hybrid_knock.py 2>&1 | tee gpurun_out/r4f/knock.txt
echo "--- alpha waves 0..3"
MAGICKHIP_LIBRARY=$PWD/imagemagick_amd/lib/libmagickhip_low.so timeout 300 python tools/time_hybrid_knock.py 2>&1 | head -2 | tee gpurun_out/r4f/low.txt
echo "--- production"
timeout 300 python tools/time_hybrid_knock.py 2>&1 | head -2 | tee gpurun_out/r4f/prod.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fast and blur" > gpurun_out/r4f/tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r4f/tests.log
STRESS_OPS=0 timeout 100 python tests/stress_parity.py 40 64 > gpurun_out/r4f/stress.txt 2>&1
tail -2 gpurun_out/r4f/stress.txt

#!/bin/bash
# round 4 call i: folded separable EXACT passes — parity tests, then timing
mkdir -p gpurun_out/r4i
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "separable or gaussian or sharpen or edge or outer" > gpurun_out/r4i/tests.log 2>&1
tail -5 gpurun_out/r4i/tests.log
timeout 600 python tools/time_gaussian_exact.py > gpurun_out/r4i/time.log 2>&1
cat gpurun_out/r4i/time.log

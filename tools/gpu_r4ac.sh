#!/bin/bash
mkdir -p gpurun_out/r4ac
timeout 300 python tools/time_blur_modes.py hdri 8192 10 4 2>&1 | tail -1 | cut -c1-200 > gpurun_out/r4ac/time.txt
timeout 600 python tools/time_gaussian_exact.py 2>&1 | grep -v amdgpu | grep folded >> gpurun_out/r4ac/time.txt
timeout 300 python tools/time_blur_exact.py 2>&1 | grep -v amdgpu | tail -4 | cut -c1-200 >> gpurun_out/r4ac/time.txt
cat gpurun_out/r4ac/time.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "blur or separable or gaussian or unsharp" 2>&1 | tail -3

#!/bin/bash
mkdir -p gpurun_out/r4z
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "separable or gaussian or sharpen or edge or convolve or tie" > gpurun_out/r4z/tests.log 2>&1; tail -4 gpurun_out/r4z/tests.log
timeout 600 python tools/time_gaussian_exact.py 2>&1 | grep -v amdgpu | grep folded > gpurun_out/r4z/gaussian.txt; cat gpurun_out/r4z/gaussian.txt
timeout 300 python tools/time_hdri_survey.py 4096 2>&1 | grep "gaussian_blur\|sharpen\|precision" | cut -c1-230 > gpurun_out/r4z/survey.txt; cat gpurun_out/r4z/survey.txt

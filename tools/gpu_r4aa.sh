#!/bin/bash
mkdir -p gpurun_out/r4aa
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x -k "blur or unsharp or hdri or float" > gpurun_out/r4aa/tests.log 2>&1; tail -4 gpurun_out/r4aa/tests.log
timeout 300 python tools/time_hdri_survey.py 4096 2>&1 | grep "^blur\|^unsharp\|precision" | cut -c1-230 > gpurun_out/r4aa/survey.txt; cat gpurun_out/r4aa/survey.txt
timeout 300 python tools/time_blur_modes.py hdri 8192 10 4 2>&1 | tail -1 | cut -c1-200

#!/bin/bash
# round 4 call s: conv2d_tie with pixel slots — parity, then the survey rows
mkdir -p gpurun_out/r4s
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "convolve or tie or nan_cells or log" > gpurun_out/r4s/tests.log 2>&1; tail -6 gpurun_out/r4s/tests.log
timeout 900 python tools/time_hdri_survey.py 4096 2>&1 | grep -i "convolve\|precision" | cut -c1-200 > gpurun_out/r4s/survey_convolve.txt; cat gpurun_out/r4s/survey_convolve.txt

#!/bin/bash
# round 4 call v: EqualizeImage on float frames from the running counts
mkdir -p gpurun_out/r4v
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_distributed_gpu.py -q -m gpu -x -k "equalize or histogram" > gpurun_out/r4v/tests.log 2>&1; tail -8 gpurun_out/r4v/tests.log
timeout 600 python tools/time_hdri_survey.py 4096 2>&1 | grep -i "equalize\|contrast_stretch" | cut -c1-220 > gpurun_out/r4v/survey_equalize.txt; cat gpurun_out/r4v/survey_equalize.txt

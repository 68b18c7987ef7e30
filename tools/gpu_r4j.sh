#!/bin/bash
# round 4 call j: the operator survey (Q16 beside float Quantum) after the folded separable passes
mkdir -p gpurun_out/r4j
timeout 900 python tools/time_hdri_survey.py 4096 > gpurun_out/r4j/hdri_survey_4096.txt 2>&1
cut -c1-230 gpurun_out/r4j/hdri_survey_4096.txt

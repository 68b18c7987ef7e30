#!/bin/bash
# round 4 call w: randomised differential run against the compiled reference (all ops; then the ops whose kernels changed this round)
mkdir -p gpurun_out/r4w
timeout 400 python tests/stress_parity.py 150 41 > gpurun_out/r4w/stress_all.txt 2>&1; tail -4 gpurun_out/r4w/stress_all.txt
STRESS_OPS=0,2,7,10,11,13 timeout 400 python tests/stress_parity.py 150 42 > gpurun_out/r4w/stress_changed.txt 2>&1; tail -4 gpurun_out/r4w/stress_changed.txt

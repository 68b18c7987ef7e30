#!/bin/bash
# round 4 call h: the concurrent EXACT/FAST reentrancy test and the two-rank bench lines
mkdir -p gpurun_out/r4h
timeout 900 python -m pytest tests/test_gpu_parity.py -k reentrant -q -m gpu -x > gpurun_out/r4h/reentrant.log 2>&1
tail -3 gpurun_out/r4h/reentrant.log
timeout 1200 python -m pytest tests/test_gpu_bench_modes.py -q -m gpu > gpurun_out/r4h/bench_modes.log 2>&1
tail -15 gpurun_out/r4h/bench_modes.log

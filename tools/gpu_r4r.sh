#!/bin/bash
# round 4 call r: more morphology:compose operators
mkdir -p gpurun_out/r4r
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "compose or hit_and_miss or compound" > gpurun_out/r4r/compose.log 2>&1; tail -15 gpurun_out/r4r/compose.log
timeout 900 python -m pytest tests/test_magickcore_shim.py -q -m gpu -x -k "morphology" > gpurun_out/r4r/shim.log 2>&1; tail -5 gpurun_out/r4r/shim.log

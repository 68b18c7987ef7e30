#!/bin/bash
# round 4 call q: the table-driven colourspaces
mkdir -p gpurun_out/r4q
timeout 900 python -m pytest tests/test_gpu_colorspaces.py -q -m gpu -x > gpurun_out/r4q/colorspaces.log 2>&1; tail -15 gpurun_out/r4q/colorspaces.log
timeout 900 python -m pytest tests/test_magickcore_shim.py -q -m gpu -x -k "colorspace" > gpurun_out/r4q/shim.log 2>&1; tail -15 gpurun_out/r4q/shim.log

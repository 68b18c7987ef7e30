#!/bin/bash
mkdir -p gpurun_out/r4ad
echo "== U=8" > gpurun_out/r4ad/ab.log
timeout 300 python tools/time_blur_modes.py hdri 8192 10 4 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/r4ad/ab.log
echo "== U=4" >> gpurun_out/r4ad/ab.log
MAGICKHIP_TRI_U4=1 timeout 300 python tools/time_blur_modes.py hdri 8192 10 4 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/r4ad/ab.log
cat gpurun_out/r4ad/ab.log

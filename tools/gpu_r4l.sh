#!/bin/bash
# round 4 call l: phase trace of the hybrid FAST blur kernel
mkdir -p gpurun_out/r4l
export MAGICKHIP_LIBRARY=$PWD/imagemagick_amd/lib/libmagickhip_htrace.so
timeout 300 python tools/trace_hybrid_blur.py rgba > gpurun_out/r4l/trace_rgba.txt 2>&1
timeout 300 python tools/trace_hybrid_blur.py plain > gpurun_out/r4l/trace_plain.txt 2>&1
cat gpurun_out/r4l/trace_rgba.txt; cat gpurun_out/r4l/trace_plain.txt

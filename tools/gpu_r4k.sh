#!/bin/bash
# round 4 call k: is the hybrid walk waiting for memory?  knock 256 = cache-resident fetches, 512 = no stores
mkdir -p gpurun_out/r4k
export MAGICKHIP_LIBRARY=$PWD/imagemagick_amd/lib/libmagickhip_knock.so
KNOCK_MASKS=${KNOCK_MASKS:-0,256,512,768,1023} timeout 600 python tools/time_hybrid_knock.py > gpurun_out/r4k/knock.log 2>&1
cat gpurun_out/r4k/knock.log

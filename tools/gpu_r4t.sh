#!/bin/bash
# round 4 call t: conv2d_tie unroll factors (A/B)
mkdir -p gpurun_out/r4t
for v in "" _u2 _u8; do
  export MAGICKHIP_LIBRARY=$PWD/imagemagick_amd/lib/libmagickhip$v.so
  echo "== variant '$v'" >> gpurun_out/r4t/ab.log
  timeout 600 python tools/time_hdri_survey.py 4096 2>&1 | grep -i "convolve Disk\|convolve LoG:0x2 (rgb)" | head -3 | cut -c1-200 >> gpurun_out/r4t/ab.log
done
cat gpurun_out/r4t/ab.log

#!/bin/bash
# round 4 call o: non-temporal result stores in the Erode/Dilate tile kernel (A/B)
mkdir -p gpurun_out/r4o
for v in "" _nt; do
  export MAGICKHIP_LIBRARY=$PWD/imagemagick_amd/lib/libmagickhip$v.so
  echo "== variant '$v'" >> gpurun_out/r4o/ab.log
  timeout 300 python tools/time_dilate.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4o/ab.log
done
cat gpurun_out/r4o/ab.log

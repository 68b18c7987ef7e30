#!/bin/bash
# round 4 call p: output rows per tile of the vertical resize pass (A/B)
mkdir -p gpurun_out/r4p
for v in "" _vr2 _vr8 _vr16; do
  export MAGICKHIP_LIBRARY=$PWD/imagemagick_amd/lib/libmagickhip$v.so
  echo "== variant '$v'" >> gpurun_out/r4p/ab.log
  for mode in fast exact; do timeout 300 python tools/run_resize.py $mode 3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4p/ab.log; done
done
cat gpurun_out/r4p/ab.log

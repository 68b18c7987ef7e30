#!/bin/bash
# round 4 call m: A/B of library variants on the fused blur kernels (8192^2 RGBA, sigma 10)
mkdir -p gpurun_out/r4m
for v in ${VARIANTS:-"" _prio1 _prio2}; do
  [ "$v" = "base" ] && v=""
  export MAGICKHIP_LIBRARY=$PWD/imagemagick_amd/lib/libmagickhip$v.so
  echo "== variant '$v'" >> gpurun_out/r4m/${LOG:-ab.log}
  KNOCK_MASKS=0 timeout 300 python tools/time_hybrid_knock.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4m/${LOG:-ab.log}
  timeout 300 python tools/time_blur_modes.py exact 8192 10 4 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-90 >> gpurun_out/r4m/${LOG:-ab.log}
done
cat gpurun_out/r4m/${LOG:-ab.log}

#!/bin/bash
# round 4 call n: non-temporal accesses in the fused UnsharpMask kernel (A/B)
mkdir -p gpurun_out/r4n
for v in "" _nt; do
  export MAGICKHIP_LIBRARY=$PWD/imagemagick_amd/lib/libmagickhip$v.so
  echo "== variant '$v'" >> gpurun_out/r4n/ab2.log
  timeout 300 python tools/time_unsharp.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4n/ab2.log
done
cat gpurun_out/r4n/ab2.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x -k "unsharp" 2>&1 | tail -3

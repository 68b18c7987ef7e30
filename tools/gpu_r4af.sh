#!/bin/bash
mkdir -p gpurun_out/r4af
echo "== U=8" > gpurun_out/r4af/ab.log
timeout 600 python tools/time_gaussian_exact.py 2>&1 | grep -v amdgpu | grep "folded" | grep "sigma 10\|sigma 3" >> gpurun_out/r4af/ab.log
echo "== U=4" >> gpurun_out/r4af/ab.log
MAGICKHIP_FOLD_U4=1 timeout 600 python tools/time_gaussian_exact.py 2>&1 | grep -v amdgpu | grep "folded" | grep "sigma 10\|sigma 3" >> gpurun_out/r4af/ab.log
cat gpurun_out/r4af/ab.log

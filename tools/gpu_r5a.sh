#!/bin/bash
# round 5, first lease: the one-launch matrix-pipe resize (resize_mfma.hip)
O=gpurun_out/r5a; mkdir -p $O
./tools/ubench/mfma_f64_rate > $O/mfma_f64_rate.txt 2>&1; cat $O/mfma_f64_rate.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "resize" > $O/tests.log 2>&1; tail -15 $O/tests.log
for v in "" "MAGICKHIP_NO_RESIZE_MFMA=1" "MAGICKHIP_RESIZE_MFMA_TPS=8" "MAGICKHIP_RESIZE_MFMA_STEPS=2" "MAGICKHIP_RESIZE_MFMA_STEPS=32" "MAGICKHIP_RESIZE_MFMA_TPS=8 MAGICKHIP_RESIZE_MFMA_STEPS=16" "MAGICKHIP_RESIZE_MFMA_TPS=32 MAGICKHIP_RESIZE_MFMA_STEPS=8"; do
  echo "== $v" >> $O/resize_times.txt
  env $v timeout 300 python tools/run_resize.py fast 5 2>&1 | grep -v amdgpu >> $O/resize_times.txt
done
cat $O/resize_times.txt
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "c3_resize" > $O/fullsize.log 2>&1; tail -5 $O/fullsize.log

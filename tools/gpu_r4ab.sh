#!/bin/bash
mkdir -p gpurun_out/r4ab
for opt in "" 1; do
  echo "== MAGICKHIP_COLUMN_R16='$opt'" >> gpurun_out/r4ab/ab.log
  MAGICKHIP_COLUMN_R16=$opt timeout 300 python tools/time_blur_modes.py hdri 8192 10 4 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/r4ab/ab.log
done
cat gpurun_out/r4ab/ab.log
MAGICKHIP_COLUMN_R16=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "blur and (hdri or float)" 2>&1 | tail -3

#!/bin/bash
O=gpurun_out/r5h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "resize" > $O/tests.log 2>&1; tail -5 $O/tests.log
for v in "MAGICKHIP_RESIZE_STREAM_ROWS=32" "MAGICKHIP_RESIZE_STREAM_ROWS=64" "MAGICKHIP_RESIZE_STREAM_ROWS=96" "MAGICKHIP_RESIZE_STREAM_ROWS=128" "MAGICKHIP_RESIZE_STREAM_ROWS=160" "MAGICKHIP_RESIZE_STREAM_ROWS=192"; do
  echo "== $v" >> $O/resize_times.txt
  env $v timeout 300 python tools/run_resize.py fast 8 2>&1 | grep -v amdgpu >> $O/resize_times.txt
done
cat $O/resize_times.txt

#!/bin/bash
# resize_stream.hip first run: parity, timing, row-chunk sweep
O=gpurun_out/r5e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "resize" > $O/tests.log 2>&1; tail -15 $O/tests.log
for v in "" "MAGICKHIP_RESIZE_STREAM_ROWS=128" "MAGICKHIP_RESIZE_STREAM_ROWS=512" "MAGICKHIP_RESIZE_STREAM_ROWS=1024" "MAGICKHIP_NO_RESIZE_STREAM=1"; do
  echo "== $v" >> $O/resize_times.txt
  env $v timeout 300 python tools/run_resize.py fast 5 2>&1 | grep -v amdgpu >> $O/resize_times.txt
done
cat $O/resize_times.txt
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "c3_resize" > $O/fullsize.log 2>&1; tail -5 $O/fullsize.log

#!/bin/bash
# resize_stream.hip: parity, timing, counters
O=gpurun_out/r5f; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "resize" > $O/tests.log 2>&1; tail -5 $O/tests.log
for v in "" "MAGICKHIP_RESIZE_STREAM_ROWS=128" "MAGICKHIP_RESIZE_STREAM_ROWS=512"; do
  echo "== $v" >> $O/resize_times.txt
  env $v timeout 300 python tools/run_resize.py fast 5 2>&1 | grep -v amdgpu >> $O/resize_times.txt
done
cat $O/resize_times.txt
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "c3_resize" > $O/fullsize.log 2>&1; tail -5 $O/fullsize.log
tools/sq_counters.sh r5f_stream resize_stream python $R/tools/run_resize.py fast 2 > $O/sq.txt 2>&1

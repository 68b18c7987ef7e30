#!/bin/bash
O=gpurun_out/r5g; mkdir -p $O
R=$GRAFT_REPO_ROOT
tools/sq_counters.sh r5g_stream resize_stream python $R/tools/run_resize.py fast 2 > $O/sq.txt 2>&1
grep -i "duration\|GRBM" $O/sq.txt | head
/opt/rocm/bin/rocm-smi --showclocks 2>&1 | head -20

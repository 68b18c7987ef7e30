#!/bin/bash
# counters of resize_mfma (4 and 6 waves) and of the two-pass kernels
R=$GRAFT_REPO_ROOT
tools/sq_counters.sh r5d_mfma_w4 resize env MAGICKHIP_RESIZE_MFMA_WAVES=4 python $R/tools/run_resize.py fast 2 > gpurun_out/r5d_w4.txt 2>&1
tools/sq_counters.sh r5d_mfma_w6 resize python $R/tools/run_resize.py fast 2 > gpurun_out/r5d_w6.txt 2>&1
tools/sq_counters.sh r5d_twopass resize env MAGICKHIP_NO_RESIZE_MFMA=1 python $R/tools/run_resize.py fast 2 > gpurun_out/r5d_2p.txt 2>&1
tail -3 gpurun_out/sq_r5d_mfma_w4/p1.log

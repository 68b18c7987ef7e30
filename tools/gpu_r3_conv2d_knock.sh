#!/bin/bash
# the integer 2-D convolve with parts knocked out (-DMH_CX_KNOCK bits: 1 products, 2 epilogue, 4 staging of new rows)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/conv2dx_knock
mkdir -p $OUT
cd $R
for lib in $R/imagemagick_amd/lib/libmagickhip*.so; do
  for layout in rgba plain4; do
    MAGICKHIP_LIBRARY=$lib timeout 200 python tools/time_convolve2d.py 16384 Disk:15 $layout 2>&1 | grep "i8 exact " | sed "s/^/$(basename $lib): /"
  done
done | tee $OUT/knock.txt

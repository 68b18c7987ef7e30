#!/bin/bash
# Run on the GPU box (through gpurun):   tools/collect_resize_profile.sh <tag>
# The C3 workload alone (tools/run_resize.py fast): rocprofv3 kernel stats and the two --pmc passes
# (FETCH_SIZE, WRITE_SIZE; kernel-trace only) -> gpurun_out/profiles_<tag>/resize_*.csv and a summary
# on stdout (tools/import_resize_profile.py merges it into profiles/).
set -u
TAG=${1:-r5b}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/resize_stats -o resize -- \
  python $R/tools/run_resize.py fast 3 > $OUT/resize_stats.log 2>&1
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctr=${pass#*:}
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_resize_$name -o $name -- \
    python $R/tools/run_resize.py fast 2 > $OUT/pmc_resize_$name.log 2>&1
done
cd $R
python tools/import_resize_profile.py $TAG

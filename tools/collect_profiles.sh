#!/bin/bash
# Run on the GPU box (through gpurun):   tools/collect_profiles.sh <tag>
# Produces under gpurun_out/profiles_<tag>/:
#   stats/        rocprofv3 --kernel-trace --stats (CSV) of the default bench command
#   pmc_fetch/ pmc_write/ pmc_tcc/   separate --pmc passes (kernel-trace only)
#   bench.json    the bench line (with cpu_baseline)
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the headline workload alone (kernel averages comparable with bench.py's roofline.avg_ms) ...
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- \
  python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $OUT/stats.log 2>&1
# ... and the whole default command including the secondary configs (mixed image sizes)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_full -o bench -- \
  python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats_full.log 2>&1
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum"; do
  name=${pass%%:*}; ctr=${pass#*:}
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_$name -o $name -- \
    python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/pmc_$name.log 2>&1
done
cd $R
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 2000 $OUT/bench.json

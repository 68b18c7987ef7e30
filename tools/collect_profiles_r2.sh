#!/bin/bash
# Run on the GPU box (through gpurun):   tools/collect_profiles_r2.sh <tag> [quick]
# Produces under gpurun_out/profiles_<tag>/:
#   stats/, stats_full/   rocprofv3 --kernel-trace --stats of the default bench command
#                         (headline only / with every secondary config)
#   pmc_<workload>_<fetch|write>/   separate --pmc passes (kernel-trace only) of each workload:
#                         fast (fused FAST blur), exact (EXACT blur), resize (C3), c4, c5
#   bench.json            the bench line of the same box (with cpu_baseline)
# tools/import_profiles_r2.py turns that into profiles/<tag>_*.csv + profiles/pmc_traffic.json.
set -u
TAG=${1:-r2}
QUICK=${2:-}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- \
  python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $OUT/stats.log 2>&1
if [ -z "$QUICK" ]; then
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_full -o bench -- \
  python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats_full.log 2>&1
fi
workload() {
  case $1 in
    fast)   echo "python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra" ;;
    exact)  echo "python $R/tools/time_blur_exact.py 8192 tie" ;;
    resize) echo "python $R/tools/run_resize.py fast 2" ;;
    c4)     echo "python $R/tools/run_configs.py c4" ;;
    c5)     echo "python $R/tools/run_configs.py c5" ;;
  esac
}
for w in fast exact resize c4 c5; do
  for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    name=${pass%%:*}; ctr=${pass#*:}
    timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_${w}_$name -o $name -- \
      $(workload $w) > $OUT/pmc_${w}_$name.log 2>&1
  done
done
cd $R
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json

#!/bin/bash
# Run on the GPU box (through gpurun):   tools/collect_profiles.sh <tag> [quick]
# Produces under gpurun_out/profiles_<tag>/ (tools/import_profiles.py turns it into profiles/<tag>_*):
#   stats/, stats_full/   rocprofv3 --kernel-trace --stats of the default bench command
#                         (headline only / with every secondary config)
#   pmc_<workload>_<fetch|write>/   separate --pmc passes (kernel-trace only) of each workload:
#                         fast (FAST blur: f16 colour + exact alpha, convolve_fused_hybrid.hip), exact (EXACT
#                         blur, one launch), hdri (float Quantum blur), resize (C3, FAST: resize_stream.hip), c4, c5
#   sq_<fast|exact>_<a|b>/  SQ issue / wait / LDS counters of the two fused kernels
#   bench.json            the bench line of the same box (with cpu_baseline)
set -u
TAG=${1:-r6e}
QUICK=${2:-}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- \
  python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic > $OUT/stats.log 2>&1
if [ -z "$QUICK" ]; then
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_full -o bench -- \
  python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-live-traffic > $OUT/stats_full.log 2>&1
fi
workload() {
  case $1 in
    fast)   echo "python $R/tools/time_blur_modes.py fast 8192 10 4" ;;
    exact)  echo "python $R/tools/time_blur_modes.py exact 8192 10 4" ;;
    hdri)   echo "python $R/tools/time_blur_modes.py hdri 8192 10 4" ;;
    resize) echo "python $R/tools/run_resize.py fast 2" ;;
    c4)     echo "python $R/tools/run_configs.py c4" ;;
    c5)     echo "python $R/tools/run_configs.py c5" ;;
  esac
}
LIST="fast exact hdri resize c4 c5"
[ -n "$QUICK" ] && LIST="fast exact"
for w in $LIST; do
  for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    name=${pass%%:*}; ctr=${pass#*:}
    timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_${w}_$name -o $name -- \
      $(workload $w) > $OUT/pmc_${w}_$name.log 2>&1
  done
done
for w in fast exact; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA \
    --output-format csv -d $OUT/sq_${w}_a -o a -- $(workload $w) > $OUT/sq_${w}_a.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CU_CYCLES \
    --output-format csv -d $OUT/sq_${w}_b -o b -- $(workload $w) > $OUT/sq_${w}_b.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq_${w}_c -o c -- $(workload $w) > $OUT/sq_${w}_c.log 2>&1
done
cd $R
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json

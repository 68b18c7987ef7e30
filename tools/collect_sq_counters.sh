#!/bin/bash
# Run on the GPU box (through gpurun):   tools/collect_sq_counters.sh <tag>
# Issue/stall counters of the blur kernels, separate --pmc passes (kernel-trace only).
set -u
TAG=${1:-sq}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
 "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
 "SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
 "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_INSTS_VMEM_WR"
 "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
)
i=0
for p in "${PASSES[@]}"; do
  keep=""
  for c in $p; do
    if grep -qw "$c" $OUT/avail.txt; then keep="$keep $c"; fi
  done
  i=$((i+1))
  [ -z "$keep" ] && continue
  timeout 200 rocprofv3 --kernel-trace --pmc $keep --output-format csv -d $OUT/p$i -o p$i -- \
    python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/p$i.log 2>&1
done
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "conv_mfma" not in k: continue
        k="column" if "Lb1E" in k or "<true" in k else "row"
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    for c,v in sorted(d.items()):
        print(k,c,len(v),sum(v)/len(v))
PY

#!/bin/bash
# Run on the GPU box (through gpurun):   tools/sq_counters.sh <tag> <kernel-name-substring> <command...>
# Issue / stall / instruction-mix counters of one kernel, separate --pmc passes (kernel-trace only).
set -u
TAG=$1; shift
FILTER=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
 "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
 "SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
 "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_INSTS_VMEM_WR"
 "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for p in "${PASSES[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $p --output-format csv -d $OUT/p$i -o p$i -- "$@" > $OUT/p$i.log 2>&1
done
python - <<PY
import csv,glob,collections,json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "$FILTER" not in k: continue
        acc[k.split("(")[0][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if ("Start_Timestamp" in r) and ("End_Timestamp" in r):
            acc[k.split("(")[0][:90]]["duration_ns_in_pass_of_"+r["Counter_Name"]].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
out={}
for k,d in acc.items():
    out[k]={c:{"n":len(v),"avg":sum(v)/len(v)} for c,v in sorted(d.items())}
    for c,v in sorted(d.items()):
        print(k,c,len(v),"%.4g"%(sum(v)/len(v)))
json.dump(out,open("$OUT/summary.json","w"),indent=1)
PY
rm -rf $OUT/p*/   # keep the summary and logs only

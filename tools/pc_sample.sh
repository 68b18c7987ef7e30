#!/bin/bash
# Run on the GPU box (through gpurun):  tools/pc_sample.sh <tag> <command...>
# Stochastic PC sampling (rocprofv3 beta) of a command; keeps the sample CSVs under gpurun_out/pcs_<tag>.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pcs_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles \
  --pc-sampling-interval 65536 --kernel-trace --output-format csv -d $OUT/raw -o pcs -- "$@" > $OUT/log.txt 2>&1
echo "rc=$?" >> $OUT/log.txt
find $OUT/raw -name "*.csv" | head -20 >> $OUT/log.txt
python - <<PY
import csv,glob,collections,json,os
files=glob.glob("$OUT/raw/**/*pc_sampling*.csv",recursive=True)
print(files)
for f in files:
    rows=list(csv.DictReader(open(f)))
    print(f,len(rows),rows[0].keys() if rows else None)
    if not rows: continue
    # histogram by (code object offset / instruction text if present)
    key=[k for k in rows[0].keys() if "nstruction" in k or "ffset" in k or "Stall" in k or "stall" in k or "Issued" in k or "issued" in k or "Snapshot" in k or "snapshot" in k]
    print("columns of interest:",key)
    os.makedirs("$OUT",exist_ok=True)
    import shutil
    # keep a compacted copy (<= 40 MB)
    if os.path.getsize(f) < 40*1024*1024:
        shutil.copy(f,"$OUT/"+os.path.basename(f))
PY
rm -rf $OUT/raw

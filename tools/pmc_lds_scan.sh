#!/bin/bash
# LDS bank-conflict share of every kernel the default bench command runs (one rocprofv3 --pmc pass).
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_lds_scan
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --output-format csv -d $OUT/lds -o lds -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/lds.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("$OUT/lds/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        acc[row["Kernel_Name"].split("(")[0].replace("void mh::", "")[:64]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("%-64s %10s %10s %8s %8s %8s" % ("kernel", "LDS active", "conflict", "confl %", "wait_any", "wait_inst"))
for k, v in sorted(acc.items()):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    if m.get("SQ_WAVE_CYCLES", 0) < 1e6:
        continue
    a = m.get("SQ_LDS_IDX_ACTIVE", 0.0)
    print("%-64s %10.3g %10.3g %7.1f%% %7.1f%% %7.1f%%" % (k, a, m.get("SQ_LDS_BANK_CONFLICT", 0), 100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / a if a else 0,
          100 * m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"], 100 * m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"]))
PY

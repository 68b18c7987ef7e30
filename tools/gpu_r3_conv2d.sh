#!/bin/bash
# tools/gpu_r3_conv2d.sh: the integer 2-D convolve: tests, timings, counters
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/conv2dx
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "integer_cells or convolve_2d or separable_2d or c5_convolve" > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
timeout 300 python tools/time_convolve2d.py 16384 Disk:15 rgba > $OUT/time.log 2>&1
timeout 300 python tools/time_convolve2d.py 16384 Disk:15,Disk:7.3,Square:3 plain4 >> $OUT/time.log 2>&1
timeout 300 python tools/time_convolve2d.py 16384 Disk:15 rgb >> $OUT/time.log 2>&1
timeout 300 python tools/time_convolve2d.py 8192 Disk:15,Octagon:5,Rectangle:8x4 rgba >> $OUT/time.log 2>&1
cat $OUT/time.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES \
  --output-format csv -d $OUT/sq_a -o a -- python $R/tools/time_convolve2d.py 16384 Disk:15 rgba > $OUT/sq_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/sq_b -o b -- python $R/tools/time_convolve2d.py 16384 Disk:15 rgba > $OUT/sq_b.log 2>&1
python - <<PY
import csv, glob, collections
for tag in ("a", "b"):
    for path in glob.glob("$OUT/sq_%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"][:40]
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
        for name, d in acc.items():
            if "conv2d" in name:
                print(tag, name, {k: "%.4g" % v for k, v in d.items()})
PY

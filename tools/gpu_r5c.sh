#!/bin/bash
# round 5, re-entry lease: resize_mfma.hip as committed (d66b7a1) — rate ubench, parity, timing sweep, rocprof
O=gpurun_out/r5c; mkdir -p $O
./tools/ubench/mfma_f64_rate > $O/mfma_f64_rate.txt 2>&1; cat $O/mfma_f64_rate.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "resize" > $O/tests.log 2>&1; tail -5 $O/tests.log
for v in "" "MAGICKHIP_NO_RESIZE_MFMA=1" "MAGICKHIP_RESIZE_MFMA_WAVES=4" "MAGICKHIP_RESIZE_MFMA_STEPS=2" "MAGICKHIP_RESIZE_MFMA_STEPS=32" "MAGICKHIP_RESIZE_MFMA_TPS=8 MAGICKHIP_RESIZE_MFMA_STEPS=16" "MAGICKHIP_RESIZE_MFMA_TPS=12"; do
  echo "== $v" >> $O/resize_times.txt
  env $v timeout 300 python tools/run_resize.py fast 5 2>&1 | grep -v amdgpu >> $O/resize_times.txt
done
cat $O/resize_times.txt
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "c3_resize" > $O/fullsize.log 2>&1; tail -3 $O/fullsize.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o resize -- python $GRAFT_REPO_ROOT/tools/run_resize.py fast 6 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'P'
import csv,glob
for f in glob.glob('gpurun_out/r5c/prof/**/*kernel_trace.csv',recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if 'resize' in r['Kernel_Name']]
    print(f,len(rows))
    for r in rows: print(r['Kernel_Name'][:60],(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,'ms',r.get('VGPR_Count'),r.get('Accum_VGPR_Count'),r.get('SGPR_Count'),r.get('LDS_Block_Size'))
P

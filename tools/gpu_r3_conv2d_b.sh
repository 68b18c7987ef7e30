#!/bin/bash
# integer 2-D convolve: the affected tests, the stress op, timings and the bench line
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/conv2dx_b
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "convol or separable or sharpen or edge or gaussian" > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
STRESS_OPS=12 timeout 400 python tests/stress_parity.py 150 41 > $OUT/stress12.log 2>&1; tail -3 $OUT/stress12.log
STRESS_OPS=5,7 timeout 200 python tests/stress_parity.py 40 42 > $OUT/stress57.log 2>&1; tail -2 $OUT/stress57.log
timeout 300 python tools/time_convolve2d.py 16384 Disk:15,Disk:16.4,Octagon:14,Rectangle:49x9 rgba 2>&1 | grep -v amdgpu.ids | tee $OUT/time.log
timeout 300 python tools/time_convolve2d.py 16384 Disk:15,Square:3 plain4 2>&1 | grep -v amdgpu.ids | tee -a $OUT/time.log
timeout 600 python bench.py --no-cpu-baseline --no-live-traffic > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value", d["value"], "exact", d.get("value_exact"))
print(json.dumps(d["configs"]["c5_convolve_disk15"], indent=1)[:1800])
PY

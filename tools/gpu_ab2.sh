#!/bin/bash
# A/B of library variants in ONE gpurun call:  tools/gpu_ab2.sh <tag> "<suffix list, '-' = default>" [trace] [tests]
TAG=$1; VARIANTS=$2
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
L=$PWD/imagemagick_amd/lib
for round in 1 2 3; do
for v in $VARIANTS; do
  [ "$v" = "-" ] && s="" || s="_$v"
  MAGICKHIP_LIBRARY=$L/libmagickhip$s.so timeout 120 python tools/time_blur_passes.py 2>&1 | tail -1 | sed "s/^/lib$s r$round: /"
done
done 2>&1 | tee $OUT/ab.txt
if [[ "$*" == *trace* ]]; then
  MAGICKHIP_LIBRARY=$L/libmagickhip_trace.so timeout 120 python tools/trace_fused_blur.py 2>&1 | grep -v amdgpu.ids | tee $OUT/trace.txt
fi
if [[ "$*" == *bench* ]]; then
  timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra > $OUT/bench.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("bench ms_per_step",d["ms_per_step"],{k:d["roofline"][k] for k in ("kernel","avg_ms","frac")})
PY
fi
if [[ "$*" == *tests* ]]; then
  ( timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "c2_blur" ) 2>&1 | tail -3 | tee $OUT/pytest_fullsize.txt
  ( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blur or unsharp" ) 2>&1 | tail -3 | tee $OUT/pytest_parity.txt
fi

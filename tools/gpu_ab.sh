#!/bin/bash
# A/B of library builds in ONE gpurun call (boxes of the pool differ by +-5 %):
#   tools/gpu_ab.sh <tag> lib1.so lib2.so ...      each: blur parity tests + two bench runs
TAG=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for round in 1 2; do
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ $round = 1 ]; then
    MAGICKHIP_LIBRARY=$PWD/$lib timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blur" > $OUT/pytest_$name.log 2>&1
    tail -1 $OUT/pytest_$name.log
  fi
  MAGICKHIP_LIBRARY=$PWD/$lib timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra > $OUT/bench_${name}_$round.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$OUT/bench_${name}_$round.json"))
print("$name round $round ms_per_step",d["ms_per_step"],{k:d["roofline"][k] for k in ("kernel","avg_ms","frac")})
PY
done
done

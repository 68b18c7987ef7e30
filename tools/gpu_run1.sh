#!/bin/bash
# round-2 GPU call 1: full GPU test suite (incl. the new full-size oracle tests), then the fused
# blur's first timing with the two-pass path beside it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2a/pytest.log 2>&1
tail -15 gpurun_out/r2a/pytest.log
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra > gpurun_out/r2a/bench_fused.json 2> gpurun_out/r2a/bench_fused.err
MAGICKHIP_NO_FUSED_BLUR=1 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra > gpurun_out/r2a/bench_twopass.json 2> gpurun_out/r2a/bench_twopass.err
for s in 1 2 4; do
  MAGICKHIP_FUSED_SEGMENTS=$s timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra > gpurun_out/r2a/bench_fused_seg$s.json 2>/dev/null
done
cat gpurun_out/r2a/bench_fused.json gpurun_out/r2a/bench_twopass.json gpurun_out/r2a/bench_fused_seg*.json | cut -c1-900

#!/bin/bash
# One tuning iteration of the fused blur on the GPU box:  tools/gpu_blur_iter.sh <tag> [pmc]
# blur parity tests (small + the full-size C2 comparison with the reference), the bench line,
# optionally the SQ / traffic counters of the fused kernel (separate --pmc passes).
TAG=${1:-it}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "blur" ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra > $OUT/bench.json 2> $OUT/bench.err
cut -c1-200 $OUT/bench.json; python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("ms_per_step",d["ms_per_step"],"roofline",{k:d["roofline"][k] for k in ("kernel","avg_ms","frac")})
PY
if [ "${2:-}" = "pmc" ]; then
  R=$PWD
  cd /tmp
  run() {
    timeout 200 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/$OUT/pmc_$1 -o $1 -- \
      python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $R/$OUT/pmc_$1.log 2>&1
  }
  run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA"
  run b "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES"
  run c "FETCH_SIZE GRBM_GUI_ACTIVE"
  run d "WRITE_SIZE"
  run e "TCC_HIT_sum TCC_MISS_sum"
  cd $R
  python tools/pmc_summary.py $OUT 2>/dev/null | grep -A40 "blur_fused" | head -60
fi

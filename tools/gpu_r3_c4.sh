#!/bin/bash
# C4: parity of the three-launch Lab + ContrastStretch, then its timing
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r3_c4}
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_batch.py -m gpu -q -x -k "stretch or c4 or equalize or lab or colorspace or colourspace" ) 2>&1 | tail -8 | tee $OUT/pytest.txt
timeout 300 python - <<'PY' 2>&1 | tee $OUT/times.txt
import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
import imagemagick_amd as im
import bench
im.load()
im.set_precision(im.PRECISION_FAST)
gen = torch.Generator(device="cuda").manual_seed(3)
for label, env in (("three launches", {}), ("general route", {"MAGICKHIP_NO_STRETCH_LEVELS": "1"})):
    os.environ.pop("MAGICKHIP_NO_STRETCH_LEVELS", None)
    os.environ.update(env)
    r = bench.c4_config(im, torch, gen)
    print(label, json.dumps({k: r[k] for k in ("ms", "kernel_only_ms", "Mpixels_per_s")}), {k: round(v["avg_ms"], 4) for k, v in r["kernels"].items()})
PY

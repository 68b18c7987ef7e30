#!/bin/bash
# round 4, final: full GPU suite + default bench line (after the fold, table colourspaces, compose operators, tie slots)
mkdir -p gpurun_out/r4final
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r4final/gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -6 gpurun_out/r4final/gpu_tests.log
timeout 900 python bench.py > gpurun_out/r4final/bench.json 2> gpurun_out/r4final/bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r4final/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4final/bench.json").read().strip().splitlines()[-1])
print("value",d["value"],"ms",d["ms_per_step"],"roofline",d["roofline"])
print("exact",d.get("value_exact"))
print("c4",json.dumps(d["configs"]["c4_lab_contrast_stretch"].get("batch")), d["configs"]["c4_lab_contrast_stretch"]["ms"], d["configs"]["c4_lab_contrast_stretch"]["kernel_only_ms"])
print("resize",d["resize"]["ms"],json.dumps(d["resize"].get("modes")))
print({k:(v.get("ms") if isinstance(v,dict) else v) for k,v in d["configs"].items()})
print({k:v for k,v in d["extra"].items() if k.startswith("shim")})
print("cpu",d.get("cpu_baseline"), d["resize"].get("cpu_baseline"))
PY

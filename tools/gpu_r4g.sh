#!/bin/bash
# round 4: full GPU suite + default bench line after the hybrid blur, shim arbitration, resize non-finite handling, batch deferral
mkdir -p gpurun_out/r4g
timeout 1200 python -m pytest tests -q -m gpu -x > gpurun_out/r4g/gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -6 gpurun_out/r4g/gpu_tests.log
timeout 900 python bench.py > gpurun_out/r4g/bench.json 2> gpurun_out/r4g/bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r4g/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4g/bench.json").read().strip().splitlines()[-1])
print("value",d["value"],"ms",d["ms_per_step"],"roofline",d["roofline"])
print("exact",d.get("value_exact"))
print("c4",json.dumps(d["configs"]["c4_lab_contrast_stretch"].get("batch")), d["configs"]["c4_lab_contrast_stretch"]["ms"], d["configs"]["c4_lab_contrast_stretch"]["kernel_only_ms"])
print("resize",d["resize"]["ms"],d["resize"]["kernels"].keys())
print({k:v for k,v in d["extra"].items() if k.startswith("shim")})
PY

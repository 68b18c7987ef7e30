#!/bin/bash
# round 4, first GPU call: the new shim tests, the whole GPU suite, a default bench line
mkdir -p gpurun_out/r4a
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout 600 python -m pytest tests/test_magickcore_shim.py -x -q -m gpu > gpurun_out/r4a/shim_tests.log 2>&1
echo "shim tests rc=$?" | tee -a gpurun_out/r4a/summary.txt
tail -15 gpurun_out/r4a/shim_tests.log
timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_magickcore_shim.py > gpurun_out/r4a/gpu_tests.log 2>&1
echo "gpu tests rc=$?" | tee -a gpurun_out/r4a/summary.txt
tail -8 gpurun_out/r4a/gpu_tests.log
timeout 600 python bench.py > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
echo "bench rc=$?" | tee -a gpurun_out/r4a/summary.txt
tail -c 3000 gpurun_out/r4a/bench.json
tail -5 gpurun_out/r4a/bench.err

#!/bin/bash
O=gpurun_out/r5j; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "c2_blur" > $O/fullsize.log 2>&1; tail -12 $O/fullsize.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "page_locked or blur" > $O/tests.log 2>&1; tail -4 $O/tests.log
for v in "" "MAGICKHIP_NO_GIVE_UP=1"; do echo "== $v"; env $v python tools/time_blur_modes.py exact 8192 10 20 2>&1 | grep -v amdgpu | tail -3; done

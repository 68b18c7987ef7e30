#!/bin/bash
# the whole GPU suite + the randomised differential run
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/final_r3
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/ -q -m gpu > $OUT/tests.log 2>&1
tail -6 $OUT/tests.log | cut -c1-300
timeout 400 python tests/stress_parity.py 100 53 > $OUT/stress.log 2>&1; tail -4 $OUT/stress.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2

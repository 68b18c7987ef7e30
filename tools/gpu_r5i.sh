#!/bin/bash
# full GPU suite + the bench line on the round-5 tree
O=gpurun_out/r5i; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json; tail -5 $O/bench.err

#!/bin/bash
# round 4 call x: the hybrid blur's walk as two teams of eight waves (MAGICKHIP_HYBRID_TEAMS): timing, then parity
mkdir -p gpurun_out/r4x
echo "== lockstep" > gpurun_out/r4x/teams.log
KNOCK_MASKS=0 timeout 120 python tools/time_hybrid_knock.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4x/teams.log
echo "== teams" >> gpurun_out/r4x/teams.log
MAGICKHIP_HYBRID_TEAMS=1 KNOCK_MASKS=0 timeout 120 python tools/time_hybrid_knock.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4x/teams.log
cat gpurun_out/r4x/teams.log
MAGICKHIP_HYBRID_TEAMS=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "blur and fast" > gpurun_out/r4x/tests_teams.log 2>&1; tail -8 gpurun_out/r4x/tests_teams.log

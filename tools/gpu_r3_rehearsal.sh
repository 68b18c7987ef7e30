#!/bin/bash
# bench.py as the driver launches it for N > 1, rehearsed with two ranks on the one GPU of the box
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/rehearsal
mkdir -p $OUT
cd $R
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo > $OUT/n2.json 2> $OUT/n2.err
tail -c 600 $OUT/n2.json; echo; tail -3 $OUT/n2.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 \
  bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --config c5 > $OUT/n2_c5.json 2> $OUT/n2_c5.err
tail -c 400 $OUT/n2_c5.json; echo; tail -3 $OUT/n2_c5.err
timeout 300 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/n1_c5.json 2> $OUT/n1_c5.err
tail -c 500 $OUT/n1_c5.json; echo; tail -3 $OUT/n1_c5.err

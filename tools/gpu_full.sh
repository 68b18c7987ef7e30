#!/bin/bash
# Full GPU pass: the whole -m gpu suite, the default bench line, the other bench configs at N=1
# and their 2-rank control flow rehearsed with gloo on the one GPU.   tools/gpu_full.sh <tag>
TAG=${1:-full}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
( time timeout 600 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python - <<PY
import json
for line in open("$OUT/bench.json"):
    if line.startswith("{"):
        d=json.loads(line)
        print("value",d["value"],"ms",d["ms_per_step"],"roofline",{k:d["roofline"][k] for k in ("kernel","avg_ms","frac","traffic")})
        for k in ("modes","sustained","resize","configs","extra","cpu_baseline"):
            print(k, json.dumps(d.get(k))[:1500])
PY
for cfg in c4 c5 equalize; do
  timeout 300 python bench.py --config $cfg --steps 3 --warmup 1 > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  cut -c1-400 $OUT/bench_$cfg.json; tail -2 $OUT/bench_$cfg.err
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
     bench.py --gpus 2 --config $cfg --steps 2 --warmup 1 --backend gloo > $OUT/bench_${cfg}_2ranks.json 2> $OUT/bench_${cfg}_2ranks.err
  cut -c1-300 $OUT/bench_${cfg}_2ranks.json; tail -2 $OUT/bench_${cfg}_2ranks.err
done

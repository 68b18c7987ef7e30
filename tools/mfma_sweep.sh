for m in 2 3 4; do
  echo -n "per_cu=$m "; MAGICKHIP_MFMA=1 MAGICKHIP_MFMA_PER_CU=$m timeout 200 python bench.py --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernels_ms'])"
done
MAGICKHIP_MFMA=1 timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "fast or blur or unsharp" 2>&1 | tail -2

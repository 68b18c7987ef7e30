for m in 0 1 3; do
  echo -n "skip=$m "; MAGICKHIP_MFMA=1 MAGICKHIP_MFMA_SKIP=$m timeout 200 python bench.py --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernels_ms'])"
done

for d in 0 1 2 3 4; do
  echo "DBG=$d"; QMX_M16_DBG=$d python bench.py --batch 32 --steps 10 --warmup 2 --hnsw-rows 0 --no-cpu --verify 0 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['kernel_ms'], j['value'])"
done

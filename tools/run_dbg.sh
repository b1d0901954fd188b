for b in 1 2 4; do
  python bench.py --batch $b --steps 30 --warmup 3 --hnsw-rows 0 --no-cpu --verify 0 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('batch', $b, r['kernel_ms'], j['value'], r['hbm']['frac'])"
done

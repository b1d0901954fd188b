for d in 1024 1536; do
for b in 16 32 64; do
  python bench.py --dim $d --batch $b --steps 10 --warmup 2 --hnsw-rows 0 --no-cpu --verify 0 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('dim', $d, 'batch', $b, r['kernel_ms'], j['ms_per_step'], j['value'], r['hbm']['frac'], r['mfma_f32']['frac'])"
done; done

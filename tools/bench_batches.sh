for b in 1 4 8 16 32 64 128; do
  python bench.py --batch $b --steps 20 --warmup 3 --hnsw-rows 0 --no-cpu --no-hbm-point --verify 0 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print($b, r['kernel_ms'], j['ms_per_step'], j['value'], r['bound'], r['frac'], r['hbm']['achieved_GBps'], r['mfma_f32']['achieved_TFLOPs'])"
done

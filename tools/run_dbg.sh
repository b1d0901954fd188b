for sh in 5 6 7 8 9 10; do
  QMX_PRESCAN_SHIFT=$sh python bench.py --batch 64 --steps 10 --warmup 2 --hnsw-rows 0 --no-cpu --verify 0 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('shift', $sh, r['kernel_ms'], j['ms_per_step'], j['value'])"
done

for q in 1 8 16 32 64; do python bench.py --batch $q --steps 40 --warmup 5 --configs "" --no-cpu --no-hbm-point --split-copy none 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print(d['config']['batch'], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['kernel'][:60])
"; done

for q in 1 4 16 32 64 128 512; do python bench.py --batch $q --steps 40 --warmup 5 --configs "" --no-cpu --no-hbm-point 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print(d['config']['batch'], d['value'], d['ms_per_step'], r['kernel_ms'], d.get('prefilter_equals_exact_scan_whole_block'), r['kernel'][:40])
"; done

for q in 1 64 128 256 512 1024; do python bench.py --batch $q --steps 30 --warmup 5 --configs "" --no-cpu --no-hbm-point 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print(d['config']['batch'], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['launches_per_pass'], d.get('prefilter_equals_exact_scan_whole_block'), r['kernel'][:44])
"; done

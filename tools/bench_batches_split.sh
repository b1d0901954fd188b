for c in half pair; do for q in 128 256; do echo "copy $c"; python bench.py --batch $q --split-copy $c --steps 40 --warmup 5 --configs "" --no-cpu --no-hbm-point 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print(d['config']['batch'], d['value'], d['ms_per_step'], r['kernel_ms'], r['kernel'][:50])
"; done; done

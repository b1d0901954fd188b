for q in 64 128 256; do python bench.py --batch $q --steps 40 --warmup 5 --configs "" --no-cpu --no-hbm-point 2>/dev/null; done

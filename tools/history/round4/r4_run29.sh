#!/bin/bash
# round 4, run 29: the table-free PQ build as the default: build tests (both options), then the C4 leg of bench.py at full size (10 M x 1536, m = 96)
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_hnsw_build.py tests/test_gpu_multivector.py -m gpu -q -x -k "pq" 2>&1 | tail -5 > gpurun_out/r4ac_tests.log
cat gpurun_out/r4ac_tests.log
timeout 900 python bench.py --rows 1000000 --configs c4 --no-sweep --no-robustness --no-cpu --no-other-copy-point --no-hbm-point --fanout-rows 0 --config-rows 10000000 --verify 1 > gpurun_out/r4ac_c4_10m.json 2> gpurun_out/r4ac_c4_10m.err
python - gpurun_out/r4ac_c4_10m.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = d["configs"]["C4"]["hnsw_pq_walk"]
print("build_s", h["build_s"], "pts/s", h["build_points_per_s"], {k: (w.get("kernel_ms"), w.get("recall_at_10_vs_exact")) for k, w in h["walks"].items()}, h.get("oracle_walk_check"))
PY
tail -2 gpurun_out/r4ac_c4_10m.err

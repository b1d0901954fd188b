#!/bin/bash
# round 5, call 7: the PQ walk's hop prefilter (8-bit LUT image in LDS) - tests, then C4-shaped walks at 2 M points with it on / off; the SQ walk once more
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_pq.py tests/test_gpu_hnsw_reference_order.py tests/test_gpu_pq_block_walk.py -x -q 2>&1 | tail -12 > gpurun_out/r5g_tests.log
timeout 600 python tools/walk_variants.py --rows 1000000 --c4-rows 2000000 --variants hnsw_per_cu=0 hnsw_no_pq_prefilter=1 hnsw_per_cu=4 hnsw_per_cu=8 > gpurun_out/r5g_walk_variants.jsonl 2> gpurun_out/r5g_walk_variants.err
cat gpurun_out/r5g_tests.log
python - <<'PY'
import json
for l in open("gpurun_out/r5g_walk_variants.jsonl"):
    d = json.loads(l)
    print({k: d[k] for k in d if k in ("walk", "rows", "variant", "kernel_ms", "frac_of_hbm", "equals_first_variant", "build_s")}, d.get("kernel", "")[30:80])
PY
tail -3 gpurun_out/r5g_walk_variants.err

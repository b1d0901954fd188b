#!/bin/bash
# round 5, call 4: the walk with its visited set in LDS (hnsw.hpp LdsVisited) against the bitmap-only walk, at full size; every walk test; the staged LUT kernel
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_hnsw_reference_order.py tests/test_gpu_hnsw.py tests/test_gpu_sq.py tests/test_gpu_pq.py tests/test_gpu_bq.py tests/test_gpu_tq.py \
   tests/test_gpu_multivector.py tests/test_gpu_custom_queries.py tests/test_gpu_custom_quantized.py tests/test_gpu_pq_block_walk.py -x -q 2>&1 | tail -8 > gpurun_out/r5d_tests.log
timeout 900 python tools/walk_variants.py --rows 10000000 --c4-rows 2000000 --variants hnsw_spec=0 hnsw_spec=1 hnsw_spec=0,hnsw_row_u4=3 hnsw_spec=1,hnsw_row_u4=3 \
   hnsw_spec=0,hnsw_no_lds_visited=1 hnsw_spec=0,hnsw_per_cu=6 > gpurun_out/r5d_walk_variants.jsonl 2> gpurun_out/r5d_walk_variants.err
cat gpurun_out/r5d_tests.log
python - <<'PY'
import json
for l in open("gpurun_out/r5d_walk_variants.jsonl"):
    d = json.loads(l)
    print({k: d[k] for k in d if k in ("walk", "variant", "kernel_ms", "frac_of_hbm", "equals_first_variant", "build_s")}, d.get("kernel", "")[30:95])
PY
tail -3 gpurun_out/r5d_walk_variants.err

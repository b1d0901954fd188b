#!/bin/bash
# round 5, call 6: the walk's searches handed out from a counter (dynamic slots) against the static stride, searches per CU, at full size; C4 walks at 2 M
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_hnsw_reference_order.py tests/test_gpu_hnsw.py tests/test_gpu_pq.py tests/test_gpu_multivector.py tests/test_gpu_custom_quantized.py -x -q 2>&1 | tail -4 > gpurun_out/r5f_tests.log
timeout 600 python tools/walk_variants.py --rows 10000000 --c4-rows 2000000 --variants hnsw_per_cu=0 hnsw_per_cu=6 hnsw_per_cu=7 hnsw_per_cu=8 hnsw_static_slots=1 hnsw_static_slots=1,hnsw_per_cu=6 hnsw_no_lds_visited=1 > gpurun_out/r5f_walk_variants.jsonl 2> gpurun_out/r5f_walk_variants.err
cat gpurun_out/r5f_tests.log
python - <<'PY'
import json
for l in open("gpurun_out/r5f_walk_variants.jsonl"):
    d = json.loads(l)
    print({k: d[k] for k in d if k in ("walk", "variant", "kernel_ms", "frac_of_hbm", "equals_first_variant", "build_s")}, d.get("kernel", "")[30:80])
PY
tail -3 gpurun_out/r5f_walk_variants.err

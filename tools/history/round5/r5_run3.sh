#!/bin/bash
# round 5, call 3: the random-gather roof (tools/micro/gather_roof) and the SQ walk under {steps in flight 4 | 3} x {searches per CU 8 | 12 | 16} x {read-ahead 0 | 1}
set -u
mkdir -p gpurun_out
timeout 300 tools/micro/gather_roof 32 > gpurun_out/r5c_gather_roof.txt 2>&1
timeout 900 python tools/walk_variants.py --rows 10000000 --variants hnsw_spec=0 hnsw_spec=1 hnsw_spec=0,hnsw_row_u4=3 hnsw_spec=1,hnsw_row_u4=3 \
   hnsw_spec=0,hnsw_per_cu=8 hnsw_spec=0,hnsw_per_cu=4 hnsw_spec=0,hnsw_row_u4=3,hnsw_per_cu=12 hnsw_spec=0,hnsw_row_u4=3,hnsw_per_cu=8 > gpurun_out/r5c_walk_variants.jsonl 2> gpurun_out/r5c_walk_variants.err
cat gpurun_out/r5c_gather_roof.txt
python - <<'PY'
import json
for l in open("gpurun_out/r5c_walk_variants.jsonl"):
    d = json.loads(l)
    print({k: d[k] for k in d if k in ("walk", "variant", "kernel_ms", "frac_of_hbm", "equals_first_variant", "build_s")}, d.get("kernel", "")[40:95])
PY
tail -3 gpurun_out/r5c_walk_variants.err

#!/bin/bash
# round 5, call 8: searches per CU of the PQ walks after the rounding rule (>= 8 only)
set -u
mkdir -p gpurun_out
timeout 600 python tools/walk_variants.py --rows 1000000 --c4-rows 2000000 --variants hnsw_per_cu=0 hnsw_per_cu=4 hnsw_per_cu=5 > gpurun_out/r5h_walk_variants.jsonl 2> gpurun_out/r5h_walk_variants.err
python - <<'PY'
import json
for l in open("gpurun_out/r5h_walk_variants.jsonl"):
    d = json.loads(l)
    print({k: d[k] for k in d if k in ("walk", "rows", "variant", "kernel_ms", "frac_of_hbm", "equals_first_variant", "build_s")}, d.get("kernel", "")[30:80])
PY
tail -3 gpurun_out/r5h_walk_variants.err

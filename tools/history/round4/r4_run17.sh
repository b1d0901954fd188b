#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/bench_configs.py --configs c4 --batches 32,64,128,256 --reps 5 > gpurun_out/r4q_c4.jsonl 2> gpurun_out/r4q_c4.err
timeout 600 python tools/bench_configs.py --configs c4 --batches 128 --reps 5 --c4-zero-query 77 > gpurun_out/r4q_c4_straggler.jsonl 2>> gpurun_out/r4q_c4.err
cat gpurun_out/r4q_c4_straggler.jsonl >> gpurun_out/r4q_c4.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r4q_c4.jsonl"):
    d = json.loads(l)
    print({k: d[k] for k in d if k in ('batch','scan_kernel_ms','ms_per_search_wall','qps','prefilter_queries','fallback_queries','verified_rows','prefilter_candidates','topk_on_sample_matches_oracle')})
PY
tail -2 gpurun_out/r4q_c4.err

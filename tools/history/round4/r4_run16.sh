#!/bin/bash
# round 4, run 16: (a) the exact pass behind the PQ prefilter shares its grid among the queries listed; (b) the copy kernels write in the output's order
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_pq_prefilter.py tests/test_gpu_pq.py tests/test_gpu_i8_copy.py tests/test_gpu_split_scan.py tests/test_gpu_full_size.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r4p_tests.log
cat gpurun_out/r4p_tests.log
timeout 600 python tools/bench_configs.py --configs c4 --batches 32,128,256 --reps 10 > gpurun_out/r4p_c4.jsonl 2> gpurun_out/r4p_c4.err
python - <<'PY'
import json
for l in open("gpurun_out/r4p_c4.jsonl"):
    d = json.loads(l)
    print({k: d[k] for k in d if k in ('batch','scan_kernel_ms','ms_per_search_wall','qps','fallback_queries','verified_rows','topk_on_sample_matches_oracle')})
PY
timeout 300 bash tools/pmc_traffic.sh r4p --what c2i8 --reps 3 > /dev/null 2>&1
cat gpurun_out/pmc_r4p/traffic.md | tail -8

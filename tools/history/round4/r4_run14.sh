#!/bin/bash
# round 4, run 14: the PQ prefilter over 16-bit codes against the 8-bit copy: tests, then C4 brute force at 10 M x 1536 (m = 96), 32 and 128 queries
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pq_prefilter.py tests/test_gpu_pq.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r4n_tests_pqf.log
cat gpurun_out/r4n_tests_pqf.log
for w in 0 1; do
  QMX_PQ_PREFILTER_NO_W16=$w timeout 600 python tools/bench_configs.py --configs c4 --batches 4,32,128 --reps 10 > gpurun_out/r4n_c4_now16_$w.jsonl 2> gpurun_out/r4n_c4_now16_$w.err
  echo "no_w16=$w"; cat gpurun_out/r4n_c4_now16_$w.jsonl | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print({k: d[k] for k in d if k in ('config','batch','scan_kernel_ms','launches_per_search','ms_per_search_wall','qps','topk_on_sample_matches_oracle','pq_kmeans_train_s')})
"
  tail -3 gpurun_out/r4n_c4_now16_$w.err
done

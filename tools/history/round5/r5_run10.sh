#!/bin/bash
# round 5, call 10: the table-free PQ build with the hop prefilter in its insertion searches: parity tests, then the C4-shaped build at 2 M points on / off
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_hnsw_build.py tests/test_gpu_pq.py tests/test_gpu_threads.py -x -q -k "pq or PQ or quantized or thread" 2>&1 | tail -30 > gpurun_out/r5j_tests.log
cat gpurun_out/r5j_tests.log
for OFF in 0 1; do
  QMX_HNSW_NO_PQ_PREFILTER=$OFF timeout 400 python tools/walk_variants.py --rows 200000 --c4-rows 2000000 --variants hnsw_per_cu=0 2>&1 | grep -E '"walk": "C4"|C4 PQ LUT"|rror' | cut -c1-300
done

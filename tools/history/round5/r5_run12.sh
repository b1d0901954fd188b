#!/bin/bash
# round 5, call 12: the walk tests on the last library (LdsVisited::unset marks instead of emptying; the visited-table stress test), the walks' times once more
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_hnsw_reference_order.py tests/test_gpu_pq.py tests/test_gpu_sq.py tests/test_gpu_bq.py tests/test_gpu_tq.py \
   tests/test_gpu_multivector.py tests/test_gpu_custom_queries.py tests/test_gpu_custom_quantized.py tests/test_gpu_pq_block_walk.py tests/test_gpu_threads.py -q 2>&1 | tail -6 > gpurun_out/r5l_tests.log
timeout 150 python __graft_entry__.py smoke 2>&1 | tail -4 >> gpurun_out/r5l_tests.log
cat gpurun_out/r5l_tests.log
timeout 300 python tools/walk_variants.py --rows 2000000 --c4-rows 2000000 --variants hnsw_per_cu=0 hnsw_no_lds_visited=1 2>&1 | grep -E 'build_s|kernel_ms' | cut -c1-60,150-330

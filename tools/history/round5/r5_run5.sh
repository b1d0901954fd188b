#!/bin/bash
# round 5, call 5: the walk without the read-ahead code, searches per CU 4..12, the LDS table on / off; then the PMC passes over the default walk at 10 M points
set -u
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_hnsw_reference_order.py tests/test_gpu_hnsw.py tests/test_gpu_pq.py -x -q 2>&1 | tail -4 > gpurun_out/r5e_tests.log
timeout 1200 bash tools/pmc_walk10m.sh r5e --rows 10000000 --variants hnsw_per_cu=0 hnsw_per_cu=6 hnsw_per_cu=5 hnsw_per_cu=4 hnsw_no_lds_visited=1 hnsw_no_lds_visited=1,hnsw_per_cu=8 hnsw_row_u4=3,hnsw_per_cu=6 > gpurun_out/r5e_pmc.log 2>&1
cat gpurun_out/r5e_tests.log
cat gpurun_out/r5e_pmc.log | cut -c1-300

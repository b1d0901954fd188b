#!/bin/bash
# round 5, call 2: the walk's read-ahead (option hnsw_spec) and the 4-waves-per-SIMD row loop at full size; the hnsw tests on the rebuilt library
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_hnsw_reference_order.py tests/test_gpu_hnsw.py tests/test_gpu_sq.py -x -q 2>&1 | tail -5 > gpurun_out/r5b_tests.log
timeout 900 python tools/walk_variants.py --rows 10000000 --c4-rows 2000000 --variants hnsw_spec=0 hnsw_spec=1 hnsw_spec=2 > gpurun_out/r5b_walk_variants.jsonl 2> gpurun_out/r5b_walk_variants.err
cat gpurun_out/r5b_tests.log
cut -c1-400 gpurun_out/r5b_walk_variants.jsonl
tail -5 gpurun_out/r5b_walk_variants.err

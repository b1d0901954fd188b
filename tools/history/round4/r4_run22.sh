#!/bin/bash
# round 4, run 22: tests of the direct PQ walk (opt-in), TurboQuant / Manhattan build + custom walks, prefilter, and the walks' suites
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_pq.py tests/test_gpu_pq_block_walk.py tests/test_gpu_hnsw_build.py tests/test_gpu_custom_quantized.py tests/test_gpu_tq.py tests/test_gpu_pq_prefilter.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r4u_tests.log
cat gpurun_out/r4u_tests.log

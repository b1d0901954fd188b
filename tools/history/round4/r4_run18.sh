#!/bin/bash
# round 4, run 18: HNSW build and custom walks through a TurboQuant storage over Manhattan
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_hnsw_build.py tests/test_gpu_custom_quantized.py tests/test_gpu_tq.py tests/test_gpu_custom_queries.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r4r_tests.log
cat gpurun_out/r4r_tests.log

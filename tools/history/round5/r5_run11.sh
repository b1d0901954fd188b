#!/bin/bash
# round 5, call 11: the build's insertions drawn from a counter (dynamic slots): build tests, then the C3 build (10 M x 768 through SQ) and the C4-shaped build
# (2 M x 1536 through PQ) with the counter and with the static stride
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_hnsw_build.py tests/test_gpu_threads.py tests/test_gpu_multivector.py -x -q 2>&1 | tail -5 > gpurun_out/r5k_tests.log
cat gpurun_out/r5k_tests.log
for ST in 0 1; do
  echo "static_slots=$ST"
  QMX_HNSW_STATIC_SLOTS=$ST timeout 500 python tools/walk_variants.py --rows 10000000 --c4-rows 2000000 --variants hnsw_per_cu=0 2>&1 | grep -E 'build_s|kernel_ms' | cut -c1-120
done

set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_hnsw_build.py tests/test_gpu_large_top.py tests/test_gpu_merge.py tests/test_gpu_multivector.py tests/test_gpu_pq.py tests/test_gpu_pq_block_walk.py tests/test_gpu_pq_prefilter.py tests/test_gpu_scan_mfma.py tests/test_gpu_sharded_cabi.py tests/test_gpu_split_scan.py tests/test_gpu_sq.py tests/test_gpu_threads.py tests/test_gpu_tq.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r4g_gpu_rest.log
cat gpurun_out/r4g_gpu_rest.log

set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_i8_copy.py tests/test_gpu_split_scan.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r4l_tests.log
cat gpurun_out/r4l_tests.log

#!/bin/bash
# round 4, run 23: multi-vector points over PQ inner rows: the MaxSim walk and the build through the LUTs of the original inner vectors
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_multivector.py tests/test_gpu_custom_quantized.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r4v_tests.log
cat gpurun_out/r4v_tests.log

#!/bin/bash
# round 5, call 9: the int8 prefilter's first-launch sample stride (16 | 32 | 64 | 24) on the headline, one batch and two in flight; its tests
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_i8_copy.py tests/test_gpu_split_scan.py -x -q 2>&1 | tail -4 > gpurun_out/r5i_tests.log
for S in 16 32 64 24 48; do
  QMX_I8_SAMPLE_STRIDE=$S timeout 200 python bench.py --no-sweep --no-robustness --no-cpu --no-other-copy-point --no-hbm-point --configs "" --fanout-rows 0 --steps 200 --warmup 10 \
      --details gpurun_out/r5i_stride$S.details.json 2> /dev/null | tail -1 > gpurun_out/r5i_stride$S.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r5i_stride$S.details.json"))
print("stride $S", "qps", d["value"], "ms", d["ms_per_step"], "std", d["value_stddev"], "kernel_ms", d["roofline"]["kernel_ms"], "equal", d.get("prefilter_equals_exact_scan_whole_block"), d["roofline"].get("prefilter_per_batch"))
PY
done
QMX_I8_SAMPLE_STRIDE=32 timeout 300 python -m pytest tests/test_gpu_i8_copy.py tests/test_gpu_split_scan.py -x -q 2>&1 | tail -3
cat gpurun_out/r5i_tests.log

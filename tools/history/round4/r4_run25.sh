#!/bin/bash
# round 4, run 25: the int8 scan with wave-owned rows (no stage hand-over) against the three-stage kernel
set -u
mkdir -p gpurun_out
QMX_I8_SCAN_DEEP=3 timeout 1200 python -m pytest tests/test_gpu_i8_copy.py tests/test_gpu_split_scan.py tests/test_gpu_full_size.py -m gpu -q -x 2>&1 | tail -12 > gpurun_out/r4x_tests.log
cat gpurun_out/r4x_tests.log
for mode in 3 0; do
  for fl in 1 2; do
    QMX_I8_SCAN_DEEP=$mode timeout 300 python bench.py --no-sweep --no-robustness --no-cpu --no-other-copy-point --no-hbm-point --configs "" --fanout-rows 0 --in-flight $fl > gpurun_out/r4x_bench_${mode}_$fl.json 2> gpurun_out/r4x_bench_${mode}_$fl.err
    python - gpurun_out/r4x_bench_${mode}_$fl.json $mode $fl <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("shape", sys.argv[2], "in flight", sys.argv[3], "value", d["value"], "ms", d["ms_per_step"], "kernel", d["roofline"]["kernel"][:34], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "equal", d.get("prefilter_equals_exact_scan_whole_block"))
PY
  done
done

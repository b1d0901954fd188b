#!/bin/bash
# round 4, run 24: the int8 scan with the deeper half-stage pipeline against round 3's kernel: tests, then C2 at 128 queries with one and two batches in flight
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_i8_copy.py tests/test_gpu_split_scan.py tests/test_gpu_full_size.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r4w_tests.log
cat gpurun_out/r4w_tests.log
for mode in deep classic; do
  for fl in 1 2; do
    if [ $mode = classic ]; then export QMX_I8_SCAN_CLASSIC=1; else unset QMX_I8_SCAN_CLASSIC; fi
    timeout 300 python bench.py --no-sweep --no-robustness --no-cpu --no-other-copy-point --no-hbm-point --configs "" --fanout-rows 0 --in-flight $fl > gpurun_out/r4w_bench_${mode}_$fl.json 2> gpurun_out/r4w_bench_${mode}_$fl.err
    python - gpurun_out/r4w_bench_${mode}_$fl.json $mode $fl <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "in flight", sys.argv[3], "value", d["value"], "ms", d["ms_per_step"], "kernel", d["roofline"]["kernel"][:40], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "equal", d.get("prefilter_equals_exact_scan_whole_block"), d.get("gpu_matches_oracle_on_sample_bit_exact"))
PY
  done
done
unset QMX_I8_SCAN_CLASSIC

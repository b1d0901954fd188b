set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_i8_copy.py tests/test_gpu_pq_block_walk.py tests/test_gpu_hnsw.py tests/test_gpu_threads.py -m gpu -q 2>&1 | tail -15 > gpurun_out/r4e_tests.log
for f in 1 2; do
  timeout 600 python bench.py --no-cpu --no-sweep --no-hbm-point --no-other-copy-point --no-robustness --configs "" --fanout-rows 0 --steps 100 --in-flight $f > gpurun_out/r4e_bench_inflight$f.json 2> gpurun_out/r4e_bench_inflight$f.err
done
timeout 600 python bench.py --no-cpu --no-sweep --no-hbm-point --no-other-copy-point --no-robustness --configs "" --steps 20 > gpurun_out/r4e_bench_fanout.json 2> gpurun_out/r4e_bench_fanout.err
cat gpurun_out/r4e_tests.log
python - <<'PY'
import json
for f in ("inflight1", "inflight2", "fanout"):
    try:
        d = json.loads(open("gpurun_out/r4e_bench_%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value", d["value"], "ms", d["ms_per_step"], "std", d.get("value_stddev"), d.get("step_groups"), "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"],
              "equal", d.get("prefilter_equals_exact_scan_whole_block"))
        if "one_process_fanout" in d: print(json.dumps(d["one_process_fanout"])[:1500])
    except Exception as e:
        print(f, "ERR", repr(e)); print(open("gpurun_out/r4e_bench_%s.err" % f).read()[-1500:])
PY

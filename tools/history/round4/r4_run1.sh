set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_i8_copy.py tests/test_gpu_threads.py tests/test_gpu_split_scan.py tests/test_gpu_sharded_cabi.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r4a_tests1.log
timeout 600 python -m pytest tests/test_gpu_full_size.py -m gpu -q -k "i8" 2>&1 | tail -15 > gpurun_out/r4a_tests_full_i8.log
timeout 600 python -m pytest tests/test_gpu_pq_prefilter.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r4a_tests_pqf.log
timeout 200 python __graft_entry__.py smoke > gpurun_out/r4a_smoke.log 2>&1
timeout 600 python bench.py --no-cpu --no-sweep --no-hbm-point --no-other-copy-point --configs "" --steps 50 > gpurun_out/r4a_bench_robust.json 2> gpurun_out/r4a_bench_robust.err
cat gpurun_out/r4a_tests1.log gpurun_out/r4a_tests_full_i8.log gpurun_out/r4a_tests_pqf.log
tail -3 gpurun_out/r4a_smoke.log
tail -c 600 gpurun_out/r4a_bench_robust.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4a_bench_robust.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "std", d.get("value_stddev"), d["config"]["derived_copy"])
r = d.get("robustness", {})
for k, v in r.items():
    if "qps" in v: print(k, v["qps"], v["copy"], v["verified_rows_per_query"], v["candidates_per_query"], v["fallback_rate"], v["equals_exact_scan_whole_block"])
    else:
        for kk, vv in v.items():
            if isinstance(vv, dict) and "qps" in vv: print(k, kk, vv["qps"], vv["copy"], vv["i8_scale_balance"], vv["verified_rows_per_query"], vv["candidates_per_query"], vv["fallback_rate"], vv["equals_exact_scan_whole_block"], vv["trial"])
            else: print(k, kk, vv)
PY

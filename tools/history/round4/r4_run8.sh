set -u
mkdir -p gpurun_out
( time timeout 900 python bench.py > gpurun_out/r4h_bench_full_default.json 2> gpurun_out/r4h_bench_full_default.err ) 2> gpurun_out/r4h_bench_time.txt
tail -3 gpurun_out/r4h_bench_time.txt
tail -c 500 gpurun_out/r4h_bench_full_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4h_bench_full_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "std", d.get("value_stddev"), "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"], "equal", d.get("prefilter_equals_exact_scan_whole_block"))
print("copy", d["config"]["derived_copy"])
for k, v in d.get("robustness", {}).items():
    if "qps" in v: print(k, v["qps"], v["copy"], "ver", v["verified_rows_per_query"], "cand", v["candidates_per_query"], "fb", v["fallback_rate"], v["equals_exact_scan_whole_block"])
    else:
        for kk, vv in v.items():
            if isinstance(vv, dict) and "qps" in vv: print(k, kk, vv["qps"], vv["copy"], "G", vv["i8_scale_balance"], "ver", vv["verified_rows_per_query"], "cand", vv["candidates_per_query"], "fb", vv["fallback_rate"], vv["equals_exact_scan_whole_block"], vv["trial"])
            else: print(k, kk, vv)
print("fanout", json.dumps(d.get("one_process_fanout"))[:700])
print("cpu", d.get("cpu_baseline"))
for c, v in d.get("configs", {}).items():
    print(c, json.dumps(v)[:1800])
PY

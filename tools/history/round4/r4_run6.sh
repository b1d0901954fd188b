set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r4f_gpu_suite.log
timeout 600 python tools/bench_hnsw.py --rows 2000000 --dim 768 --scorer f32,sq --nq 8192 --check 32 --cpu-queries 0 > gpurun_out/r4f_walk_2m_d768.jsonl 2> gpurun_out/r4f_walk_2m_d768.err
for f in 3; do
  timeout 600 python bench.py --no-cpu --no-sweep --no-hbm-point --no-other-copy-point --no-robustness --configs "" --fanout-rows 0 --steps 100 --in-flight $f > gpurun_out/r4f_bench_inflight$f.json 2> gpurun_out/r4f_bench_inflight$f.err
done
cat gpurun_out/r4f_gpu_suite.log
python - <<'PY'
import json
for line in open("gpurun_out/r4f_walk_2m_d768.jsonl"):
    if line.startswith("{"):
        d = json.loads(line); print({k: d[k] for k in d if any(t in k for t in ("scorer", "kernel_ms", "qps_kernel", "recall", "oracle", "scored", "useful"))})
for f in ("inflight3",):
    try:
        d = json.loads(open("gpurun_out/r4f_bench_%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value", d["value"], "ms", d["ms_per_step"], "std", d.get("value_stddev"), d.get("step_groups"), "equal", d.get("prefilter_equals_exact_scan_whole_block"))
    except Exception as e:
        print(f, "ERR", repr(e)); print(open("gpurun_out/r4f_bench_%s.err" % f).read()[-800:])
PY

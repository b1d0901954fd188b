set -u
mkdir -p gpurun_out
for d in 1536 1024; do
timeout 600 python bench.py --dim $d --no-cpu --no-sweep --no-robustness --no-hbm-point --no-other-copy-point --configs "" --fanout-rows 0 --steps 50 > gpurun_out/r4m_bench_dim$d.json 2> gpurun_out/r4m_bench_dim$d.err
python - $d <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4m_bench_dim%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], "kernel", d["roofline"]["kernel"][:40], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "equal", d.get("prefilter_equals_exact_scan_whole_block"), d["config"]["derived_copy"]["derived_copy"], d["roofline"].get("prefilter_per_batch"))
PY
tail -c 300 gpurun_out/r4m_bench_dim$d.err
done

set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pq.py tests/test_gpu_hnsw.py tests/test_gpu_hnsw_build.py tests/test_gpu_pq_block_walk.py tests/test_gpu_custom_quantized.py -m gpu -q -k "pq or PQ or walk" 2>&1 | tail -6 > gpurun_out/r4i_tests_pq.log
cat gpurun_out/r4i_tests_pq.log
timeout 900 python bench.py --configs c4 --no-cpu --no-sweep --no-robustness --no-hbm-point --no-other-copy-point --fanout-rows 0 --steps 20 > gpurun_out/r4i_bench_c4.json 2> gpurun_out/r4i_bench_c4.err
tail -c 300 gpurun_out/r4i_bench_c4.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4i_bench_c4.json").read().strip().splitlines()[-1])
c4 = d["configs"]["C4"]
hn = c4["hnsw_pq_walk"]
print("build_s", hn["build_s"], hn["build_points_per_s"])
for k, v in hn["walks"].items():
    print(k, v["kernel_ms"], v.get("recall_at_10_vs_exact"), v.get("points_scored_per_query"))
print(json.dumps(hn.get("oracle_walk_check")))
print(json.dumps(c4.get("brute_force_Q32_oversampling2_rescore"))[:400])
PY

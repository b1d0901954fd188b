set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_sq.py tests/test_gpu_tq.py tests/test_gpu_bq.py tests/test_gpu_dense_f16_u8.py tests/test_gpu_large_top.py -m gpu -q 2>&1 | tail -6 > gpurun_out/r4k_tests_mfma_scans.log
cat gpurun_out/r4k_tests_mfma_scans.log
for ll in 0 1; do
  QMX_SQ_MFMA_NO_LLIST=$ll timeout 600 python tools/bench_configs.py --configs c3 --batches 4,8,16,32 --reps 20 > gpurun_out/r4k_c3_scans_nollist$ll.jsonl 2> gpurun_out/r4k_c3_scans_nollist$ll.err
  echo "== QMX_SQ_MFMA_NO_LLIST=$ll"
  python - gpurun_out/r4k_c3_scans_nollist$ll.jsonl <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line); print({k: d[k] for k in d if any(t in k for t in ("batch", "scan_kernel_ms", "frac", "oracle"))})
PY
done
timeout 900 python bench.py --configs c3,tq --no-cpu --no-sweep --no-robustness --no-hbm-point --no-other-copy-point --fanout-rows 0 --steps 20 > gpurun_out/r4k_bench_c3_tq.json 2> gpurun_out/r4k_bench_c3_tq.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4k_bench_c3_tq.json").read().strip().splitlines()[-1])
for c in ("C3", "TQ4"):
    v = d["configs"][c]
    bf = v["brute_force_oversampling2_rescore"]
    for q in ("Q1", "Q32"):
        print(c, q, bf[q]["kernel"][:70], bf[q]["kernel_ms"], bf[q]["roofline"]["frac"], bf[q]["recall_at_10_vs_exact"])
    if "hnsw_sq_walk_rescore" in v: print(c, "walk", v["hnsw_sq_walk_rescore"]["kernel_ms"], v["hnsw_sq_walk_rescore"]["roofline"]["frac"], v["hnsw_sq_walk_rescore"].get("oracle_walk_check"))
    print(c, v.get("oracle_check"))
PY
timeout 500 bash tools/pmc_walk.sh r4_sq --rows 2000000 --dim 768 --scorer sq --nq 8192 --check 0 --cpu-queries 0 --reps 2 > /dev/null 2>&1
cat gpurun_out/pmc_walk_r4_sq/summary.txt | cut -c1-200

set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sq.py tests/test_gpu_full_size.py -m gpu -q -k "sq or SQ or c3" 2>&1 | tail -6 > gpurun_out/r4j_tests_sq.log
cat gpurun_out/r4j_tests_sq.log
for st in 0 1; do
  QMX_SQ_MFMA_NO_STAGE=$st timeout 600 python tools/bench_configs.py --configs c3 --batches 4,8,16,32 --reps 20 > gpurun_out/r4j_c3_scans_nostage$st.jsonl 2> gpurun_out/r4j_c3_scans_nostage$st.err
  echo "== QMX_SQ_MFMA_NO_STAGE=$st"
  python - gpurun_out/r4j_c3_scans_nostage$st.jsonl <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line); print({k: d[k] for k in d if any(t in k for t in ("batch", "kernel", "ms", "frac", "GBs", "gbps", "exact", "oracle"))})
PY
done

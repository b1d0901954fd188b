set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_i8_copy.py tests/test_gpu_threads.py tests/test_gpu_split_scan.py tests/test_gpu_sharded_cabi.py tests/test_gpu_pq_prefilter.py -m gpu -q 2>&1 | tail -40 > gpurun_out/r4b_tests1.log
timeout 900 python -m pytest tests/test_gpu_pq_block_walk.py tests/test_gpu_hnsw.py tests/test_gpu_pq.py -m gpu -q -x 2>&1 | tail -40 > gpurun_out/r4b_tests_walk.log
for mode in block old; do
  if [ $mode = old ]; then export QMX_NO_HNSW_PQ_BLOCK=1; else unset QMX_NO_HNSW_PQ_BLOCK; fi
  timeout 600 python tools/bench_hnsw.py --rows 2000000 --dim 1536 --scorer pq --nq 8192 --check 32 --cpu-queries 0 > gpurun_out/r4b_pqwalk_2m_$mode.jsonl 2> gpurun_out/r4b_pqwalk_2m_$mode.err
done
unset QMX_NO_HNSW_PQ_BLOCK
for w in 3 5; do
  QMX_HNSW_PQ_BLOCK_WAVES=$w timeout 600 python tools/bench_hnsw.py --rows 2000000 --dim 1536 --scorer pq --nq 8192 --check 0 --cpu-queries 0 > gpurun_out/r4b_pqwalk_2m_w$w.jsonl 2> gpurun_out/r4b_pqwalk_2m_w$w.err
done
cat gpurun_out/r4b_tests1.log | tail -25
cat gpurun_out/r4b_tests_walk.log | tail -30
for f in gpurun_out/r4b_pqwalk_2m_*.jsonl; do echo $f; python - "$f" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"): continue
    d = json.loads(line)
    keep = {k: d[k] for k in d if any(t in k for t in ("ms", "qps", "recall", "kernel", "scored", "oracle", "check", "build_s"))}
    print(json.dumps(keep)[:900])
PY
tail -3 ${f%.jsonl}.err; done

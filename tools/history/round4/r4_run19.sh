#!/bin/bash
# round 4, run 19: the LUT-free PQ walk (pq.hip HopPQDirect) against the LUT walk: tests, then 8 192 searches over a 2 M x 1536 graph (m = 96)
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pq.py tests/test_gpu_hnsw.py tests/test_gpu_pq_block_walk.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r4s_tests_walk.log
cat gpurun_out/r4s_tests_walk.log | tail -12
timeout 600 python -m pytest tests/test_gpu_hnsw_build.py -m gpu -q -x -k "tq_manhattan or pq" 2>&1 | tail -8 > gpurun_out/r4s_tests_build.log
cat gpurun_out/r4s_tests_build.log
for mode in direct lut; do
  if [ $mode = lut ]; then export QMX_HNSW_PQ_LUT_WALK=1; else unset QMX_HNSW_PQ_LUT_WALK; fi
  timeout 600 python tools/bench_hnsw.py --rows 2000000 --dim 1536 --scorer pq --nq 8192 --check 32 --cpu-queries 0 > gpurun_out/r4s_pqwalk_2m_$mode.jsonl 2> gpurun_out/r4s_pqwalk_2m_$mode.err
done
unset QMX_HNSW_PQ_LUT_WALK
for f in gpurun_out/r4s_pqwalk_2m_*.jsonl; do echo $f; python - "$f" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"): continue
    d = json.loads(line)
    keep = {k: d[k] for k in d if any(t in k for t in ("ms", "qps", "recall", "kernel", "scored", "oracle", "check", "build_s"))}
    print(json.dumps(keep)[:900])
PY
tail -3 ${f%.jsonl}.err; done

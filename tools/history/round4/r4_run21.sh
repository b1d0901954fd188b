#!/bin/bash
# round 4, run 21: the LUT-free PQ walk, 10 x lanes per SSE lane + chunk steps per round; 2 M x 1536, m = 96, 8 192 searches
set -u
mkdir -p gpurun_out
for r in 23 22 26 43 42 46 13; do
  QMX_HNSW_PQ_DIRECT_R=$r timeout 600 python tools/bench_hnsw.py --rows 2000000 --dim 1536 --scorer pq --nq 8192 --check 16 --cpu-queries 0 --reps 3 > gpurun_out/r4t_pqwalk_2m_r$r.jsonl 2> gpurun_out/r4t_pqwalk_2m_r$r.err
  echo "R=$r"; python - gpurun_out/r4t_pqwalk_2m_r$r.jsonl <<'PY'
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"): continue
    d = json.loads(line)
    print({k: d[k] for k in d if k in ("kernel_ms", "points_scored_per_query", "oracle_walk_same_ids", "oracle_walk_same_score_bits")})
PY
  tail -2 gpurun_out/r4t_pqwalk_2m_r$r.err
done

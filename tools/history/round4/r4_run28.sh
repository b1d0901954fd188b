#!/bin/bash
# round 4, run 28: the PQ build without tables (HopPQDirectBuild + HopPQInternalDirect) against the LUT / pair-table build
set -u
mkdir -p gpurun_out
QMX_HNSW_PQ_DIRECT_WALK=2 timeout 600 python -m pytest tests/test_gpu_hnsw_build.py -m gpu -q -x -k "pq" 2>&1 | tail -6 > gpurun_out/r4ab_tests.log
cat gpurun_out/r4ab_tests.log
for mode in 2 0; do
  QMX_HNSW_PQ_DIRECT_WALK=$mode timeout 600 python bench.py --rows 1000000 --configs c4 --config-rows 2000000 --no-sweep --no-robustness --no-cpu --no-other-copy-point --no-hbm-point --fanout-rows 0 --verify 1 > gpurun_out/r4ab_c4_2m_$mode.json 2> gpurun_out/r4ab_c4_2m_$mode.err
  python - gpurun_out/r4ab_c4_2m_$mode.json $mode <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = d["configs"]["C4"]["hnsw_pq_walk"]
print("mode", sys.argv[2], "build_s", h["build_s"], "pts/s", h["build_points_per_s"], {k: (w.get("kernel_ms"), w.get("recall_at_10_vs_exact")) for k, w in h["walks"].items()}, h.get("oracle_walk_check"))
PY
  tail -2 gpurun_out/r4ab_c4_2m_$mode.err
done

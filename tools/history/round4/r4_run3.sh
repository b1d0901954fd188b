set -u
mkdir -p gpurun_out
for w in 3 8; do
  QMX_HNSW_PQ_BLOCK_WAVES=$w timeout 600 python tools/bench_hnsw.py --rows 2000000 --dim 1536 --scorer pq --nq 8192 --check 0 --cpu-queries 0 --reps 2 > gpurun_out/r4c_prof_w$w.log 2>&1
  grep "pqb\]" gpurun_out/r4c_prof_w$w.log | sort | uniq -c | sort -rn | head -12
  grep kernel_ms gpurun_out/r4c_prof_w$w.log | cut -c1-300
done

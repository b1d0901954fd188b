#!/bin/bash
# round 4, run 20: counters of the two PQ walks (2 M x 1536, m = 96, 8 192 searches)
set -u
mkdir -p gpurun_out
timeout 900 bash tools/pmc_walk.sh r4_pq_direct --rows 2000000 --dim 1536 --scorer pq --nq 8192 --check 0 --cpu-queries 0 --reps 2 > /dev/null 2>&1
QMX_HNSW_PQ_LUT_WALK=1 timeout 900 bash tools/pmc_walk.sh r4_pq_lut --rows 2000000 --dim 1536 --scorer pq --nq 8192 --check 0 --cpu-queries 0 --reps 2 > /dev/null 2>&1
cat gpurun_out/pmc_walk_r4_pq_direct/summary.txt | cut -c60-200
cat gpurun_out/pmc_walk_r4_pq_lut/summary.txt | cut -c60-200

#!/bin/bash
# Runs on the GPU box (via gpurun): the scalar-int8 128-query pass (scan_sqw.hip) alone - bench, kernel trace, PMC (the SQ legs of round6_tqw_artifacts.sh).
#   usage: bash tools/round6_sqw_artifacts.sh
set -u
R=$PWD
mkdir -p gpurun_out
timeout 400 python tools/tq_wide_bench.py --storage sq --reps 5 --out gpurun_out/r6_sqw_bench_dot.json > /dev/null 2> gpurun_out/r6_sqw_bench.err
timeout 400 python tools/tq_wide_bench.py --storage sq --reps 5 --distance euclid --out gpurun_out/r6_sqw_bench_euclid.json > /dev/null 2>> gpurun_out/r6_sqw_bench.err
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sqw -o s --output-format csv -- python $R/tools/tq_wide_bench.py --storage sq --reps 5 > /dev/null 2>&1
cd $R
head -30 gpurun_out/prof_sqw/s_kernel_stats.csv > gpurun_out/r6_sqw_kernel_stats.csv
rm -rf gpurun_out/prof_sqw
timeout 900 bash tools/pmc_tqw.sh final_sq --storage sq > /dev/null 2>&1
cat gpurun_out/r6_sqw_bench_dot.json gpurun_out/r6_sqw_bench_euclid.json | cut -c1-1800
grep scan_sqw_kernel gpurun_out/pmc_tqw_final_sq/summary.txt
head -6 gpurun_out/r6_sqw_kernel_stats.csv | cut -c1-200

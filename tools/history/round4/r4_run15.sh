#!/bin/bash
# round 4, run 15: where do 12 ms of a 128-query brute-force pass over the C4 block (10 M x 96 PQ codes) go?  kernel stats of that pass alone
mkdir -p gpurun_out
R=$PWD
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r4o -o s --output-format csv -- python $R/tools/bench_configs.py --configs c4 --batches 128 --reps 5 > $R/gpurun_out/r4o_c4_q128.jsonl 2> /dev/null
cd $R
head -25 gpurun_out/prof_r4o/s_kernel_stats.csv | cut -c1-200 > gpurun_out/r4o_c4_q128_kernel_stats.csv
rm -rf gpurun_out/prof_r4o
cat gpurun_out/r4o_c4_q128_kernel_stats.csv
tail -1 gpurun_out/r4o_c4_q128.jsonl

#!/bin/bash
# round 5, call 1: the reference-heap-order tests, the default bench line (compact headline + bench_details.json), the rocprofv3 kernel trace of the
# SURVEY 8(d) exact stream at 16 queries and at 1, and `--gpus 2` on a one-GPU box (must fail loudly).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/round5/r5_run1.sh'
set -u
R=$PWD
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_hnsw_reference_order.py tests/test_gpu_hnsw.py -x -q 2>&1 | tail -15 > gpurun_out/r5a_hnsw_tests.log
timeout 700 python bench.py > gpurun_out/r5a_bench.out 2> gpurun_out/r5a_bench.err
echo "bench rc=$? stdout lines=$(wc -l < gpurun_out/r5a_bench.out) last line bytes=$(tail -1 gpurun_out/r5a_bench.out | wc -c)"
cp bench_details.json gpurun_out/r5a_bench_details.json 2>/dev/null
python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r5a_gpus2.out 2> gpurun_out/r5a_gpus2.err; echo "gpus2 rc=$?" >> gpurun_out/r5a_gpus2.err
export TMPDIR=/tmp
cd /tmp
for Q in 16 1; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r5a_q$Q -o s --output-format csv -- python $R/bench.py --split-copy none --batch $Q --steps 40 --warmup 5 \
      --in-flight 1 --no-sweep --no-robustness --no-cpu --no-other-copy-point --no-hbm-point --verify 0 --configs "" --fanout-rows 0 --details /tmp/d$Q.json \
      > $R/gpurun_out/r5a_exact_q${Q}_traced_bench.json 2> /dev/null
  head -12 $R/gpurun_out/prof_r5a_q$Q/s_kernel_stats.csv > $R/gpurun_out/r5a_exact_q${Q}_kernel_stats.csv
  rm -rf $R/gpurun_out/prof_r5a_q$Q
done
cd $R
cat gpurun_out/r5a_hnsw_tests.log
tail -c 1500 gpurun_out/r5a_bench.out
tail -3 gpurun_out/r5a_gpus2.err
head -3 gpurun_out/r5a_exact_q16_kernel_stats.csv | cut -c1-200
head -3 gpurun_out/r5a_exact_q1_kernel_stats.csv | cut -c1-200

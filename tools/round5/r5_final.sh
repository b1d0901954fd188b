#!/bin/bash
# Everything profiles/ wants from the final build of round 5, in one gpurun call (~14 GPU-minutes):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/round5/r5_final.sh'
set -u
R=$PWD
mkdir -p gpurun_out
if [ "${SKIP_SUITE:-0}" = 1 ]; then echo "(suite skipped)" > gpurun_out/r5z_gpu_suite.log; else timeout 1000 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r5z_gpu_suite.log; fi
timeout 150 python __graft_entry__.py smoke > gpurun_out/r5z_smoke.log 2>&1
timeout 700 python bench.py > gpurun_out/r5z_bench_headline.json 2> gpurun_out/r5z_bench.err
cp bench_details.json gpurun_out/r5z_bench_details.json
export TMPDIR=/tmp
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r5z -o s --output-format csv -- python $R/bench.py --no-sweep --no-robustness --no-cpu \
    --no-other-copy-point --no-hbm-point --verify 0 --configs "" --fanout-rows 0 --steps 30 --warmup 3 --details /tmp/dz.json > $R/gpurun_out/r5z_c2_traced_bench.json 2> /dev/null
cd $R
python tools/step_from_trace.py gpurun_out/prof_r5z/s_kernel_trace.csv > gpurun_out/r5z_c2_step_timeline.txt 2>&1
head -40 gpurun_out/prof_r5z/s_kernel_stats.csv > gpurun_out/r5z_c2_kernel_stats.csv
rm -rf gpurun_out/prof_r5z
cat gpurun_out/r5z_gpu_suite.log
tail -4 gpurun_out/r5z_smoke.log
echo "headline bytes: $(tail -1 gpurun_out/r5z_bench_headline.json | wc -c), stdout lines: $(wc -l < gpurun_out/r5z_bench_headline.json)"
tail -1 gpurun_out/r5z_bench_headline.json
head -4 gpurun_out/r5z_c2_kernel_stats.csv | cut -c1-170
tail -3 gpurun_out/r5z_c2_step_timeline.txt

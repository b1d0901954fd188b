#!/bin/bash
# Runs on the GPU box (via gpurun): the round's bench lines and traces from the final build.   usage: bash tools/round6_artifacts.sh [tag]
set -u
TAG=${1:-r6}
R=$PWD
mkdir -p gpurun_out
T0=$(date +%s); timeout 600 python bench.py > gpurun_out/${TAG}_bench_headline.json 2> gpurun_out/${TAG}_bench.err && cp bench_details.json gpurun_out/${TAG}_bench_details.json
T1=$(date +%s); echo "default bench: $((T1 - T0)) s" > gpurun_out/${TAG}_bench_wall_seconds.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_form_headline.json 2> gpurun_out/${TAG}_bench_driver_form.err && cp bench_details.json gpurun_out/${TAG}_bench_driver_form_details.json
T2=$(date +%s); echo "driver-form bench: $((T2 - T1)) s" >> gpurun_out/${TAG}_bench_wall_seconds.txt
export TMPDIR=/tmp
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -o s --output-format csv -- python $R/bench.py --no-sweep --no-robustness --no-cpu \
    --no-other-copy-point --no-hbm-point --verify 0 --configs "" --fanout-rows 0 --steps 40 --warmup 4 > $R/gpurun_out/${TAG}_c2_traced_bench.json 2> /dev/null
cd $R
python tools/step_from_trace.py gpurun_out/prof_${TAG}/s_kernel_trace.csv > gpurun_out/${TAG}_c2_step_timeline.txt
head -40 gpurun_out/prof_${TAG}/s_kernel_stats.csv > gpurun_out/${TAG}_c2_kernel_stats.csv
rm -rf gpurun_out/prof_${TAG}
timeout 200 bash tools/pmc_traffic.sh ${TAG} --what c2i8 --reps 3 > /dev/null 2>&1
(cd tools/micro && timeout 200 ./gather_roof 1 | grep -E "^#|lut4" > $R/gpurun_out/${TAG}_gather_roof_lut4.txt)
cat gpurun_out/${TAG}_bench_wall_seconds.txt
tail -c 300 gpurun_out/${TAG}_bench.err
for f in gpurun_out/${TAG}_bench_headline.json gpurun_out/${TAG}_bench_driver_form_headline.json; do tail -1 $f | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'ms', d['ms_per_step'], 'sd', d.get('value_stddev'), 'roof', d['roofline']['frac'], 'timed', d['roofline']['timed_kernel'].get('kernel_ms'), d['roofline']['timed_kernel'].get('kernel_ms_in_timed_region'), d['roofline']['timed_kernel'].get('frac'), d.get('checks'), len(json.dumps(d)))
print(json.dumps(d.get('configs'))[:1800])
"; done
head -4 gpurun_out/${TAG}_c2_kernel_stats.csv | cut -c1-200
tail -3 gpurun_out/${TAG}_c2_step_timeline.txt
tail -3 gpurun_out/pmc_${TAG}/traffic.md
tail -8 gpurun_out/${TAG}_gather_roof_lut4.txt
# the N > 1 path of bench.py end to end on this one GPU: two ranks share it, gloo over device tensors (not a scaling measurement: a test that the path runs)
timeout 600 python bench.py --gpus 2 --backend gloo --ranks-share-gpu --steps 20 --warmup 5 --configs= > gpurun_out/${TAG}_bench_two_ranks_one_gpu.json 2> gpurun_out/${TAG}_bench_two_ranks_one_gpu.err
tail -1 gpurun_out/${TAG}_bench_two_ranks_one_gpu.json | cut -c1-1500

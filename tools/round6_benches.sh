#!/bin/bash
# Runs on the GPU box (via gpurun): the two bench lines of the final tree.   usage: bash tools/round6_benches.sh [tag]
set -u
TAG=${1:-r6}
mkdir -p gpurun_out
T0=$(date +%s); timeout 600 python bench.py > gpurun_out/${TAG}_bench_headline.json 2> gpurun_out/${TAG}_bench.err && cp bench_details.json gpurun_out/${TAG}_bench_details.json
T1=$(date +%s); echo "default bench: $((T1 - T0)) s" > gpurun_out/${TAG}_bench_wall_seconds.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_form_headline.json 2> gpurun_out/${TAG}_bench_driver_form.err && cp bench_details.json gpurun_out/${TAG}_bench_driver_form_details.json
T2=$(date +%s); echo "driver-form bench: $((T2 - T1)) s" >> gpurun_out/${TAG}_bench_wall_seconds.txt
timeout 600 python bench.py --gpus 2 --backend gloo --ranks-share-gpu --steps 20 --warmup 5 --configs= > gpurun_out/${TAG}_bench_two_ranks_one_gpu.json 2> gpurun_out/${TAG}_bench_two_ranks_one_gpu.err
cat gpurun_out/${TAG}_bench_wall_seconds.txt
for f in gpurun_out/${TAG}_bench_headline.json gpurun_out/${TAG}_bench_driver_form_headline.json gpurun_out/${TAG}_bench_two_ranks_one_gpu.json; do tail -1 $f | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'ms', d['ms_per_step'], 'sd', d.get('value_stddev'), 'roof', d['roofline']['frac'], 'timed', d['roofline']['timed_kernel'].get('kernel_ms'), d['roofline']['timed_kernel'].get('frac'), d.get('checks'), len(json.dumps(d)))
"; done

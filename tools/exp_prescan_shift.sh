#!/bin/bash
# round 6, last session: the size of the sample whose exact k-th best score admits the first launch's candidates (option prescan_shift: sample = rows >> shift, at least 8 192)
out=gpurun_out/r6_prescan_shift_experiment.txt
: > $out
common="--steps 120 --warmup 12 --configs= --no-sweep --no-robustness --no-cpu --no-other-copy-point --fanout-rows 0 --no-hbm-point"
for v in ${1:-10 9 8 7 11 10 9 8}; do
    line=$(env QMX_PRESCAN_SHIFT=$v python bench.py $common --details /tmp/exp_details.json 2>/tmp/exp_err.txt | tail -1)
    echo "QMX_PRESCAN_SHIFT=$v $(echo "$line" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); t=d['roofline']['timed_kernel']
print(d['value'], d['ms_per_step'], 'kernel_ms', t.get('kernel_ms'), 'verified/q', t.get('verified_rows_per_query'), 'fallback', t.get('fallback_queries'), d.get('checks'))" 2>&1 | tail -1)" | tee -a $out
done

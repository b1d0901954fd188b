#!/bin/bash
# round 6, last session: the int8 prefilter scan with the queries' image resident in LDS (option i8_resident = stages of loads ahead; 0 = the staged kernel)
out=gpurun_out/r6_i8_resident_experiment.txt
: > $out
common="--steps 120 --warmup 12 --configs= --no-sweep --no-robustness --no-cpu --no-other-copy-point --fanout-rows 0 --no-hbm-point"
for v in ${1:-0 1 0 1}; do
  for lanes in ${2:-4}; do
    line=$(env QMX_I8_RESIDENT=$v python bench.py $common --in-flight $lanes --details /tmp/exp_details.json 2>/tmp/exp_err.txt | tail -1)
    echo "QMX_I8_RESIDENT=$v in_flight=$lanes $(echo "$line" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); t=d['roofline']['timed_kernel']
print(d['value'], d['ms_per_step'], 'kernel_ms', t.get('kernel_ms'), 'in_region', t.get('kernel_ms_in_timed_region'), 'frac', t.get('frac'), t.get('kernel','')[:60], d.get('checks'))" 2>&1 | tail -1)" | tee -a $out
    tail -2 /tmp/exp_err.txt | cut -c1-300 >> $out
  done
done

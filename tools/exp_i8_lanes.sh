#!/bin/bash
# round 6: batches in flight x LDS left free by the int8 scan (the side kernels of the other batches run beside the scan only if they find LDS)
out=gpurun_out/r6_i8_lanes_experiment_b.txt
: > $out
common="--steps 120 --warmup 12 --configs= --no-sweep --no-robustness --no-cpu --no-other-copy-point --fanout-rows 0 --no-hbm-point"
for v in "QMX_I8_SCAN_LDS160=0"; do
  for lanes in 4 5 6 8; do
    line=$(env $v python bench.py $common --in-flight $lanes --details /tmp/exp_details.json 2>/dev/null | tail -1)
    echo "$v in_flight=$lanes $(echo "$line" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('checks'))")" | tee -a $out
  done
done

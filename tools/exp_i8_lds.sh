#!/bin/bash
# round 6: does the headline step shorten when the other batch's small kernels can run BESIDE the int8 scan (LDS left free on the CU)?
# variants: lds160 = rounds 3-5 (the scan holds the CU's whole LDS); lds144 = three query stages (16 KiB free);
# +1 = the sample's exact scores through 4-query tiles (12.5 KiB of LDS); +2 = no conditional fallback launches; +3 = both
out=gpurun_out/r6_i8_lds_experiment.txt
: > $out
common="--steps 100 --warmup 10 --configs= --no-sweep --no-robustness --no-cpu --no-other-copy-point --fanout-rows 0 --no-hbm-point"
for v in "QMX_I8_SCAN_LDS160=1" "QMX_I8_SCAN_LDS160=0" "QMX_EXPERIMENT=1" "QMX_EXPERIMENT=2" "QMX_EXPERIMENT=3" "QMX_I8_SCAN_LDS160=1 QMX_EXPERIMENT=3"; do
  for rep in 1 2; do
    line=$(env $v python bench.py $common --details /tmp/exp_details.json 2>/dev/null | tail -1)
    echo "$v rep$rep $(echo "$line" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['timed_kernel']['kernel_ms'], d.get('checks'))")" | tee -a $out
  done
done

timeout 400 python -m pytest tests/test_gpu_split_scan.py tests/test_gpu_sq_wide.py tests/test_gpu_i8_copy.py -m gpu -q 2>&1 | tail -3
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export QMX_EXPERIMENT_SQW_NT=1; else unset QMX_EXPERIMENT_SQW_NT; fi
  echo "SQW_NT=$v $(timeout 200 python tools/tq_wide_bench.py --storage sq --reps 5 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); w=d['wide_128_per_pass']; print(w['kernel_ms_per_launch'], w['wall_ms_per_search'], w['hbm_frac_of_8TBps'], d['lists_equal'])")"
done
unset QMX_EXPERIMENT_SQW_NT
common="--steps 60 --warmup 6 --configs= --no-sweep --no-robustness --no-cpu --no-other-copy-point --fanout-rows 0 --no-hbm-point"
for i in 1 2; do
python bench.py $common --split-copy half --details /tmp/exp_details.json 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); t=d['roofline']['timed_kernel']; print('half copy', d['value'], d['ms_per_step'], t.get('kernel'), t.get('kernel_ms'), t.get('frac'), d.get('checks'))"
done

#!/bin/bash
# Everything profiles/ wants from the final build of round 4, in one gpurun call (~12 GPU-minutes):
#   usage: /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/round4/r4_final.sh'
set -u
R=$PWD
mkdir -p gpurun_out
if [ "${SKIP_SUITE:-0}" = 1 ]; then echo "(suite skipped: run separately)" > gpurun_out/r4z_gpu_suite.log; else timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r4z_gpu_suite.log; fi
timeout 100 python __graft_entry__.py smoke > gpurun_out/r4z_smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/r4z_bench_full_default.json 2> gpurun_out/r4z_bench_full_default.err
export TMPDIR=/tmp
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r4z -o s --output-format csv -- python $R/bench.py --no-sweep --no-robustness --no-cpu \
    --no-other-copy-point --no-hbm-point --verify 0 --configs "" --fanout-rows 0 --steps 30 --warmup 3 > $R/gpurun_out/r4z_c2_traced_bench.json 2> /dev/null
cd $R
python tools/step_from_trace.py gpurun_out/prof_r4z/s_kernel_trace.csv > gpurun_out/r4z_c2_step_timeline.txt 2>&1
head -40 gpurun_out/prof_r4z/s_kernel_stats.csv > gpurun_out/r4z_c2_kernel_stats.csv
rm -rf gpurun_out/prof_r4z
timeout 200 bash tools/pmc_traffic.sh r4z --what c2i8 --reps 3 > /dev/null 2>&1
cat gpurun_out/r4z_gpu_suite.log
tail -2 gpurun_out/r4z_smoke.log
tail -c 300 gpurun_out/r4z_bench_full_default.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r4z_bench_full_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "std", d.get("value_stddev"), "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"], "equal", d.get("prefilter_equals_exact_scan_whole_block"))
for c in ("C3", "TQ4", "C4"):
    v = d["configs"].get(c, {})
    bf = v.get("brute_force_oversampling2_rescore") or {}
    for q in ("Q1", "Q32"):
        if q in bf: print(c, q, bf[q]["kernel_ms"], bf[q]["roofline"]["frac"])
    if "hnsw_sq_walk_rescore" in v: print(c, "walk", v["hnsw_sq_walk_rescore"]["kernel_ms"], v["hnsw_sq_walk_rescore"].get("oracle_walk_check"))
    if "hnsw_pq_walk" in v: print(c, "walk", {k: w.get("kernel_ms", w) for k, w in v["hnsw_pq_walk"]["walks"].items()}, "build_s", v["hnsw_pq_walk"]["build_s"], v["hnsw_pq_walk"].get("oracle_walk_check"))
    if "brute_force_Q32_oversampling2_rescore" in v: print(c, "bfQ32", v["brute_force_Q32_oversampling2_rescore"]["kernel_ms"])
PY
head -4 gpurun_out/r4z_c2_kernel_stats.csv | cut -c1-170
tail -3 gpurun_out/r4z_c2_step_timeline.txt
tail -4 gpurun_out/pmc_r4z/traffic.md

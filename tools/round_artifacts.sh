#!/bin/bash
# Runs on the GPU box (via gpurun, ~10 GPU-minutes): everything a round's profiles/ wants from the final build, in one call.
#   usage: /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/round_artifacts.sh r4'
#   -> gpurun_out/<tag>_gpu_suite.log            pytest tests -m gpu
#      gpurun_out/<tag>_bench_full_default.json  the JSON line of `python bench.py` as the driver runs it
#      gpurun_out/<tag>_c2_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the C2 step alone (the dominant kernel's calls / average duration)
#      gpurun_out/<tag>_c2_step_timeline.txt     every dispatch of one step (tools/step_from_trace.py)
#      gpurun_out/pmc_<tag>/traffic.{md,json}    the two PMC passes (FETCH_SIZE, WRITE_SIZE) over the int8 C2 search (tools/pmc_traffic.sh)
# Copy what is to be judged into profiles/ afterwards (gpurun_out/ is scratch).
set -u
TAG=${1:-rX}
R=$PWD
mkdir -p gpurun_out
timeout 420 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/${TAG}_gpu_suite.log
timeout 420 python bench.py > gpurun_out/${TAG}_bench_full_default.json 2> gpurun_out/${TAG}_bench_full_default.err
export TMPDIR=/tmp
cd /tmp
timeout 90 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -o s --output-format csv -- python $R/bench.py --no-sweep --no-robustness --no-cpu \
    --no-other-copy-point --no-hbm-point --verify 0 --configs "" --steps 30 --warmup 3 > $R/gpurun_out/${TAG}_c2_traced_bench.json 2> /dev/null
cd $R
python tools/step_from_trace.py gpurun_out/prof_${TAG}/s_kernel_trace.csv > gpurun_out/${TAG}_c2_step_timeline.txt
head -40 gpurun_out/prof_${TAG}/s_kernel_stats.csv > gpurun_out/${TAG}_c2_kernel_stats.csv
rm -rf gpurun_out/prof_${TAG}
timeout 150 bash tools/pmc_traffic.sh ${TAG} --what c2i8 --reps 3 > /dev/null 2>&1
cat gpurun_out/${TAG}_gpu_suite.log
tail -c 400 gpurun_out/${TAG}_bench_full_default.err
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_full_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"],
      "equal", d.get("prefilter_equals_exact_scan_whole_block"))
PY
head -3 gpurun_out/${TAG}_c2_kernel_stats.csv | cut -c1-160
tail -2 gpurun_out/${TAG}_c2_step_timeline.txt
tail -3 gpurun_out/pmc_${TAG}/traffic.md

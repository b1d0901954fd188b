#!/bin/bash
# Runs on the GPU box: kernel trace (csv) of a short bench run and the timeline of one step in the middle of the timed region
# (kernel, start offset and duration in us, gap to the previous kernel).   usage: tools/step_timeline.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/timeline_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $REPO/bench.py --steps 12 --warmup 3 --no-cpu --no-hbm-point --verify 0 --configs "" "$@" > $OUT/bench.json 2> $OUT/err.log
cd $REPO
CSV=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 - "$CSV" > $OUT/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = [i for i, r in enumerate(rows) if ("scan_f16pair" in r["Kernel_Name"] or "scan_i8copy" in r["Kernel_Name"] or "scan_f32_mfma16" in r["Kernel_Name"]) and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 1_000_000]
if len(big) < 6:
    print("too few main-kernel launches", len(big)); sys.exit(0)
a, b = big[-4], big[-3]
t0 = int(rows[a]["End_Timestamp"])
prev = t0
print("one step = from the end of one main scan to the end of the next (us)")
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f  +%7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Kernel_Name"][:90]))
    prev = e
print("step total %.1f us" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3))
PY
rm -rf $OUT/trace
cat $OUT/timeline.txt

#!/bin/bash
# Runs on the GPU box (via gpurun): the clock each scan kernel sustains - GRBM_GUI_ACTIVE / the dispatch's duration (MI355X guide, "DVFS give-back": the chip clocks to
# its power budget) - over tools/traffic_workloads.py.  One PMC pass with --kernel-trace only.   usage: tools/pmc_clock.sh <tag> [traffic_workloads args...]
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/clk -o clk -- python $REPO/tools/traffic_workloads.py "$@" > $OUT/clk.out 2> $OUT/clk.err
cd $REPO
python - "$OUT" "$*" <<'PY' > $OUT/clock.md
import csv, glob, os, sys
out, args = sys.argv[1], sys.argv[2]
cc = glob.glob(os.path.join(out, "clk", "**", "*counter_collection.csv"), recursive=True)
kt = glob.glob(os.path.join(out, "clk", "**", "*kernel_trace.csv"), recursive=True)
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        dur[r.get("Dispatch_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = {}
for f in cc:
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        d = None
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        elif r.get("Dispatch_Id") in dur:
            d = dur[r["Dispatch_Id"]][0]
        if not d or d < 200_000:          # dispatches of 0.2 ms and more
            continue
        acc.setdefault(r["Kernel_Name"], []).append((float(r["Counter_Value"]), d))
print("# Clock sustained per kernel: GRBM_GUI_ACTIVE / dispatch duration (rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace; tools/traffic_workloads.py %s)\n" % args)
print("(the counter is summed over the chip's 8 XCDs: the figure below divides by 8; dispatches of 0.2 ms and more)\n")
print("| kernel | dispatches | mean duration ms | GRBM_GUI_ACTIVE / 8 / duration = GHz |")
print("|---|---|---|---|")
for k, v in sorted(acc.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    ms = sum(d for _, d in v) / len(v) / 1e6
    ghz = sum(c for c, _ in v) / 8.0 / sum(d for _, d in v)
    print("| `%s` | %d | %.3f | %.3f |" % (k[:110], len(v), ms, ghz))
PY
for f in $(find $OUT/clk -name '*.csv'); do echo "== $f"; head -3 $f | cut -c1-600; done > $OUT/clk_heads.txt 2>&1
rm -rf $OUT/clk
cat $OUT/clock.md

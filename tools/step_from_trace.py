#!/usr/bin/env python3
"""One step of bench.py out of a rocprofv3 --kernel-trace CSV: the dispatches between the ends of two consecutive main-scan launches (start offset,
duration, gap to the previous kernel, in us).   usage: step_from_trace.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
main = ("scan_f16pair", "scan_i8copy", "scan_f32_mfma16")
big = [i for i, r in enumerate(rows) if any(m in r["Kernel_Name"] for m in main) and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 1_000_000]
if len(big) < 6:
    print("too few main-kernel launches", len(big))
    sys.exit(0)
a, b = big[len(big) // 2 - 1], big[len(big) // 2]        # (the middle of the run: the timed region, not the one-batch-in-flight steps behind it)
t0 = int(rows[a]["End_Timestamp"])
prev = t0
print("one step = from the end of one main scan to the end of the next (us)")
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f  +%7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Kernel_Name"][:90]))
    prev = e
print("step total %.1f us" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3))

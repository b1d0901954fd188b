#!/usr/bin/env python3
"""Prints the timeline of the last dispatches of a rocprofv3 --kernel-trace CSV: start offset, duration and the idle gap before each kernel (µs).
usage: step_timeline.py <dir with *_kernel_trace.csv> [n_last]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[-n_last:]
t0 = rows[0][0]
prev_end = t0
for s, e, name in rows:
    print("%9.1f  dur %8.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name[:110]))
    prev_end = max(prev_end, e)

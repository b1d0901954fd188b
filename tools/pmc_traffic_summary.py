#!/usr/bin/env python3
"""Summarises the two PMC passes of tools/pmc_traffic.sh: per kernel symbol, the mean FETCH_SIZE / WRITE_SIZE of its LARGE launches (within 2x of its
largest: a symbol also serves small pre-scans) -> HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024.  The factor 2 is the gfx950
correction of /opt/skills/guides/MI355X_MICROARCH.md (HBM section: FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes);
gather kernels (HNSW walks) issue narrower requests, for them the figure is an upper bound.  Writes traffic.json next to the markdown it prints."""
import glob
import json
import os
import sqlite3
import sys

out_dir, args = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
vals = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    for db in glob.glob(os.path.join(out_dir, counter, "**", "*.db"), recursive=True):
        c = sqlite3.connect(db)
        try:
            rows = list(c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)))
        except Exception as e:       # noqa: BLE001
            print("cannot read", db, e)
            continue
        for name, v in rows:
            vals.setdefault(name, {}).setdefault(counter, []).append(float(v))
res = {}
for name, d in vals.items():
    f = d.get("FETCH_SIZE", [])
    if not f or max(f) * 2 * 1024 < 50e6:       # below 50 MB per launch: not a streaming kernel of interest
        continue
    big = [x for x in f if x >= max(f) / 2]
    w = d.get("WRITE_SIZE", [])
    wbig = sorted(w)[-len(big):] if w else [0.0]
    fetch_kib, write_kib = sum(big) / len(big), sum(wbig) / len(wbig)
    res[name] = {"launches": len(big), "fetch_size_kib": round(fetch_kib, 1), "write_size_kib": round(write_kib, 1),
                 "bytes": int((2 * fetch_kib + write_kib) * 1024)}
json.dump({"_args": args, "by_kernel": res}, open(os.path.join(out_dir, "traffic.json"), "w"), indent=1)
print("# HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, two passes) over tools/traffic_workloads.py %s\n" % args)
print("bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (gfx950 correction for wide coalesced reads); mean over a symbol's large launches\n")
print("| kernel | launches | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes / launch |")
print("|---|---|---|---|---|")
for name, r in sorted(res.items(), key=lambda kv: -kv[1]["bytes"]):
    print("| `%s` | %d | %.0f | %.0f | %.4g |" % (name[:110], r["launches"], r["fetch_size_kib"], r["write_size_kib"], r["bytes"]))

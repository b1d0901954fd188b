#!/usr/bin/env python3
"""Merges the walk kernels of a tools/pmc_traffic.sh run (`--what hnsw`) into profiles/pmc_traffic.json: bytes per launch from traffic.json, the useful bytes
from the run's own "hnsw <name> rows= searches= scored/query= <kernel symbol>" lines (scored rows x 772 B for the SQ walk, x 96 B for the PQ walk).

  python tools/pmc_traffic_update.py gpurun_out/pmc_<tag> profiles/<file the numbers are kept in>.md
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir, profile = sys.argv[1], sys.argv[2]
traffic = json.load(open(os.path.join(out_dir, "traffic.json")))["by_kernel"]
table_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
table = json.load(open(table_path))
norm = lambda s: "".join(s.replace("void ", "").split()).split("(")[0]      # noqa: E731
for line in open(os.path.join(out_dir, "FETCH_SIZE.out")):
    m = re.match(r"hnsw (\w+) rows=(\d+) searches=(\d+) scored/query=([\d.]+) (.*)", line.strip())
    if not m:
        continue
    name, rows, searches, spq, sym = m.group(1), int(m.group(2)), int(m.group(3)), float(m.group(4)), m.group(5)
    hit = [k for k in traffic if norm(k) == norm(sym)]
    if not hit:
        print("no counters for", sym)
        continue
    row_bytes = 772 if name == "sq" else 96
    alg = int(searches * spq * row_bytes)
    for k in [k for k in table["by_kernel"] if "hnsw_search_kernel" in k and (("RowSQ" in k) == (name == "sq"))]:
        del table["by_kernel"][k]           # (the symbol of the round before)
    table["by_kernel"][sym] = {"bytes": traffic[hit[0]]["bytes"], "rows": rows, "searches": searches, "scored_per_query": spq, "algorithmic_bytes": alg,
                               "over_algorithmic": round(traffic[hit[0]]["bytes"] / alg, 3), "profile": profile,
                               "workload": "%s walk, ef 128, %d searches over a %d-point graph" % (name.upper(), searches, rows)}
    print(sym, table["by_kernel"][sym])
json.dump(table, open(table_path, "w"), indent=1)

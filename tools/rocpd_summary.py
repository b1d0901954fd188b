#!/usr/bin/env python3
"""Summarises rocprofv3 rocpd databases (kernel trace + PMC passes) into the small text files
committed under profiles/.

  python tools/rocpd_summary.py OUT.md label=path/to/results.db [label=...]

Kernel-trace DBs give the `--stats` table (calls, total, average per kernel); PMC DBs give the
per-kernel counter averages.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
128-byte requests as 64 bytes for wide coalesced streams (MI355X_MICROARCH.md, HBM section), so
the corrected read traffic is 2 x FETCH_SIZE x 1024 bytes.
"""
import sqlite3
import sys


def main():
    out = sys.argv[1]
    lines = []
    for spec in sys.argv[2:]:
        label, path = spec.split("=", 1)
        c = sqlite3.connect(path)
        lines.append(f"## {label}\n")
        rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        if rows:
            lines.append("| kernel | calls | total_ms | avg_ms | % |\n|---|---|---|---|---|")
            keep = rows[:12] + [r for r in rows[12:] if any(k in r[0] for k in ("scan", "rows_kernel", "hnsw", "pair_kernel"))][:40]   # the top 12 + every scan / walk kernel
            for n, calls, tot, avg, pct in keep:
                lines.append(f"| `{n[:110]}` | {calls} | {tot / 1e3:.1f} | {avg / 1e3:.3f} | {pct:.2f} |")
            lines.append("")
        try:
            pmc = list(c.execute(
                "select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
                "group by kernel_name, counter_name order by avg(value) desc"))
        except sqlite3.Error:
            pmc = []
        if pmc:
            lines.append("| kernel | counter | dispatches | avg | min | max |\n|---|---|---|---|---|---|")
            for n, cn, k, avg, mn, mx in pmc[:12]:
                lines.append(f"| `{n[:110]}` | {cn} | {k} | {avg:.1f} | {mn:.1f} | {mx:.1f} |")
            lines.append("")
    open(out, "w").write("\n".join(lines) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main()

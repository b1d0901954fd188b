"""Roofline arithmetic of bench.py (host logic, no GPU): the bytes a launch is priced at, the roof it is compared with, the PMC traffic attached
from profiles/pmc_traffic.json.  Imported by bench.py (which re-exports it) and tools/bench_sections.py."""
import json
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32-input MFMA peak (same guide: v_mfma_f32_16x16x4_f32 / 32x32x2, 64 FLOP/clk/SIMD)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16 / bf16 MFMA peak (same guide); 2377 measured in a bare loop (profiles/r2_mfma_issue_rates.txt)
MFMA_I8_PEAK_TOPS = 5000.0  # dense int8 MFMA peak (same guide: twice the f16 rate; v_mfma_i32_16x16x64_i8)


def _roofline(n, dim, kernel_ms, alg_bytes, achieved_gbps, launches, launches_per_step, Q, kernel_symbol, launches_per_pass=1.0):
    """The dominant kernel against BOTH ceilings; `bound` is the one it sits closer to.  Up to 16 queries per pass the scan is
    an HBM stream (every row byte read once: SURVEY 8d, 3072 B / row at d = 768); the 32- / 64-query passes of scan_mfma16.hip
    do 2 * dim flops per (row, query) on the f32 matrix cores and cross over to the MFMA ceiling; the prefilter of scan_split.hip
    (more than 64 queries) streams a derived f16 copy of the block and multiplies on the f16 matrix cores (1 or 3 products per element)."""
    per_pass = Q / max(1.0, round(launches_per_step / launches_per_pass))   # queries one pass over the block serves: MEASURED launches per step, not a dispatch guess
    half256 = "scan_f16half256_kernel" in kernel_symbol
    i8 = "scan_i8copy_kernel" in kernel_symbol
    split = "scan_f16pair_kernel" in kernel_symbol or "scan_f32_split_kernel" in kernel_symbol or half256 or i8
    products = 1 if ("scan_f16pair_kernel<true>" in kernel_symbol or half256 or i8) else 3 if split else 1
    flops = 2.0 * n * dim * ((256 if half256 else 128) if split else per_pass) * products / launches_per_pass     # (the prefilter multiplies a padded 128- / 256-query tile)
    tflops = flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    mfma_peak = MFMA_I8_PEAK_TOPS if i8 else MFMA_F16_PEAK_TFLOPS if split else MFMA_F32_PEAK_TFLOPS
    hbm_frac, mfma_frac = achieved_gbps / HBM_PEAK_GBPS, tflops / mfma_peak
    traffic, traffic_src = _pmc_traffic(n, dim, Q, kernel_symbol)
    common = {"traffic": traffic, "traffic_source": traffic_src, "traffic_over_algorithmic": round(traffic / float(alg_bytes), 4) if traffic and alg_bytes else None, "kernel": kernel_symbol, "kernel_ms": round(kernel_ms, 4), "launches_timed": launches,
              "queries_per_pass": per_pass, "launches_per_pass": launches_per_pass, "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_flops_per_launch": flops,
              "hbm": {"achieved_GBps": round(achieved_gbps, 1), "peak_GBps": HBM_PEAK_GBPS, "frac": round(hbm_frac, 4)},
              "mfma": {"dtype": "int8 (column- and query-scaled codes, i32 accumulate; results re-scored exactly in f32)" if i8 else
                                "f16 (x = h + l prefilter, f32 accumulate; results re-scored exactly in f32)" if split else "f32",
                       "achieved_TFLOPs": round(tflops, 2), "peak_TFLOPs": mfma_peak, "frac": round(mfma_frac, 4)},
              "f32_block_equivalent_GBps": round(n * dim * 4 / (kernel_ms * launches_per_pass * 1e-3) / 1e9, 1) if kernel_ms > 0 else 0.0}
    if split:
        eq = common["f32_block_equivalent_GBps"]
        common["frac_of_copy_stream"] = round(hbm_frac, 4)
        common["f32_block_equivalent"] = {"GBps": eq, "frac_of_peak": round(eq / HBM_PEAK_GBPS, 4),
                                          "note": "SURVEY 8(d) counts 4 B / element of the stored f32 block per scan; those bytes are NOT streamed by this kernel: it streams a derived "
                                                  + ("int8" if i8 else "f16") + " copy (achieved / frac above are bytes of the copy / kernel time) and re-scores the survivors from the f32 rows.  The 8(d)-conformant "
                                                  "figure (the f32 block itself streamed once) is roofline_hbm_point_q16 / batch_sweep.Q*_exact."}
    if mfma_frac > hbm_frac:
        return dict({"bound": "mfma", "achieved": round(tflops, 2), "peak": mfma_peak, "unit": "TFLOP/s", "frac": round(mfma_frac, 4)}, **common)
    return dict({"bound": "hbm", "achieved": round(achieved_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(hbm_frac, 4)}, **common)


def _stddev(xs):
    if len(xs) < 2:
        return 0.0
    m = sum(xs) / len(xs)
    return math.sqrt(sum((x - m) ** 2 for x in xs) / (len(xs) - 1))


def _timed_path(kernel_symbol):
    """What the timed step is, in words, from the symbol of the kernel that ran (config.timed_path of the JSON line)."""
    what = None
    if "scan_i8copy_kernel" in kernel_symbol:
        what = "prefilter over an int8 copy of the block (1 B / element, int8 matrix cores)"
    elif "scan_f16pair_kernel<true>" in kernel_symbol or "scan_f16half256_kernel" in kernel_symbol:
        what = "prefilter over an f16 copy of the block (2 B / element, f16 matrix cores)"
    elif "scan_f16pair_kernel" in kernel_symbol:
        what = "prefilter over an f16-pair copy of the block (4 B / element, f16 matrix cores)"
    elif "scan_f32_split_kernel" in kernel_symbol:
        what = "prefilter converting the f32 rows to f16 pairs on the fly (f16 matrix cores)"
    if what is None:
        return "exact f32 scan"
    return what + " + exact f32 re-scoring of the survivors: the returned lists are the exact f32 scan's, bit for bit (checked in the run)"


def _pmc_entry(kernel_symbol):
    """The entry of profiles/pmc_traffic.json for this kernel symbol (template arguments included), or None.  HBM bytes per launch come from
    separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over tools/traffic_workloads.py (counters cannot be read from inside the
    process)."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        for k, e in json.load(open(p)).get("by_kernel", {}).items():
            if _same_kernel(k, kernel_symbol):
                return e
    except Exception:
        pass
    return None


def _pmc_traffic(n, dim, Q, kernel_symbol):
    """(bytes per launch, source) when the table holds THIS kernel symbol at THIS row count; otherwise (None, None): a stale number is worse than none."""
    e = _pmc_entry(kernel_symbol)
    if e and e.get("rows") == n and "over_algorithmic" not in e:
        return e["bytes"], "%s (rocprofv3 --pmc passes on this kernel: %s)" % (e.get("profile", "profiles/pmc_traffic.json"), e.get("workload", ""))
    return None, None


def _attach_traffic(roof, kernel_symbol, n):
    """fills roof['traffic'] (+ the ratio to the algorithmic bytes) from the PMC table; graph walks carry the ratio measured on a smaller graph"""
    e = _pmc_entry(kernel_symbol)
    if not e:
        return roof
    if "over_algorithmic" in e:
        roof["traffic_measured_elsewhere"] = {"rows": e["rows"], "searches": e.get("searches"), "bytes_per_launch": e["bytes"],
                                              "algorithmic_bytes_per_launch": e.get("algorithmic_bytes"), "over_algorithmic": e["over_algorithmic"],
                                              "source": e.get("profile"), "note": e.get("workload")}
    elif e.get("rows") == n:
        roof["traffic"] = e["bytes"]
        roof["traffic_source"] = e.get("profile")
        alg = roof.get("algorithmic_bytes_per_launch")
        if alg:
            roof["traffic_over_algorithmic"] = round(e["bytes"] / float(alg), 4)
    return roof


def _same_kernel(a, b):
    norm = lambda s: "".join(str(s).replace("void ", "").split())   # noqa: E731
    a, b = norm(a), norm(b)
    return a.split("(")[0] == b.split("(")[0]


def _human(n):
    return ("%dM" % (n // 1_000_000)) if n % 1_000_000 == 0 else ("%dk" % (n // 1000)) if n % 1000 == 0 else str(n)


"""TurboQuant 4-bit brute force at C3's size: the 32-query kernel (scan_sq_mfma.hip TqOps<4>) against the 128-query pass (scan_tq4w.hip), the same
queries through both - kernel time from HIP events around the scan (qmx_query_set_timing), wall time per search, list equality.
    python tools/tq_wide_bench.py [--rows 10000000] [--dim 768] [--queries 128] [--reps 5] [--distance dot|euclid]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=128)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--distance", default="dot")
    ap.add_argument("--storage", default="tq4", help="tq4 | sq")
    ap.add_argument("--clusters", type=int, default=0, help="rows around this many centres (0: iid unit rows)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import numpy as np
    import torch
    import qdrant_amd as qa
    from qdrant_amd import _ffi as F
    lib = F.lib()
    dev = torch.device("cuda:0")
    n, dim, nq, top = args.rows, args.dim, args.queries, 10
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED0007)
    rows = torch.empty((n, dim), dtype=torch.float32, device=dev)
    step = 1 << 20
    centres = torch.randn((max(args.clusters, 1), dim), generator=g, device=dev)
    for r0 in range(0, n, step):      # unit rows around 256 centres: scores that crowd a little, as embeddings do
        r1 = min(n, r0 + step)
        rows[r0:r1] = torch.randn((r1 - r0, dim), generator=g, device=dev)
        if args.clusters:
            rows[r0:r1] = 0.5 * rows[r0:r1] + centres[torch.randint(0, args.clusters, (r1 - r0,), generator=g, device=dev)]
    rows /= rows.norm(dim=1, keepdim=True)
    queries = torch.randn((nq, dim), generator=g, device=dev)
    if args.clusters:
        queries = 0.5 * queries + centres[torch.randint(0, args.clusters, (nq,), generator=g, device=dev)]
    queries /= queries.norm(dim=1, keepdim=True)
    dist = {"dot": qa.Distance.Dot, "euclid": qa.Distance.Euclid, "cosine": qa.Distance.Cosine}[args.distance]
    if args.storage == "sq":
        quant = qa.ScalarQuantizer.fit(rows, dim, dist)
        p = quant.params()
        row_bytes = quant.quantized_vector_size()
        codes = torch.empty((n, row_bytes), dtype=torch.uint8, device=dev)
        F.check(lib.qmx_sq_encode(0, int(dist), C.byref(p), F.ptr(rows), n, dim, F.ptr(codes)))
        torch.cuda.synchronize(dev)
        del rows
        enc = qa.EncodedVectorsU8(codes, quant)
        opt = "sq_wide_min_queries"
    else:
        quant = qa.TurboQuantizer(dim, dist, 0)
        p = quant.params()
        row_bytes = quant.quantized_vector_size()
        codes = torch.empty((n, row_bytes), dtype=torch.uint8, device=dev)
        F.check(lib.qmx_tq_encode(0, int(dist), dim, C.byref(p), F.ptr(rows), n, F.ptr(codes)))
        torch.cuda.synchronize(dev)
        del rows
        enc = qa.EncodedVectorsTQ(codes, quant)
        opt = "tq_wide_min_queries"
    del codes
    digits = 1 if args.storage == "sq" else 2
    out = {"storage": args.storage, "rows": n, "dim": dim, "queries": nq, "row_bytes": row_bytes, "distance": args.distance}
    lists = {}
    for name, optv in (("narrow_32_per_pass", 0), ("wide_128_per_pass", 33)):
        qa.set_option(opt, optv)
        s = qa.BatchFilteredSearcher(queries.cpu().numpy(), enc, top)
        F.check(lib.qmx_query_set_timing(s.scorer._h, 1))
        res = s.peek_top_all()      # warm-up
        ms, nl = C.c_float(), C.c_uint32()
        F.check(lib.qmx_query_timing(s.scorer._h, C.byref(ms), C.byref(nl)))
        t0 = time.perf_counter()
        for _ in range(args.reps):
            res = s.peek_top_all()
        wall = (time.perf_counter() - t0) / args.reps
        F.check(lib.qmx_query_timing(s.scorer._h, C.byref(ms), C.byref(nl)))
        c = s.counters
        launches = max(1, int(nl.value))
        kms = ms.value / launches
        per_search = ms.value / args.reps
        out[name] = {"kernel": F.last_kernel(s.scorer._h)[:90], "kernel_ms_per_launch": round(kms, 4), "launches_per_search": launches / args.reps,
                     "scan_ms_per_search": round(per_search, 4), "wall_ms_per_search": round(wall * 1e3, 3), "qps_wall": round(nq / wall, 1),
                     "qps_scan": round(nq / (per_search * 1e-3), 1),
                     "hbm_frac_of_8TBps": round(n * row_bytes * (launches / args.reps) / (per_search * 1e-3) / 8e12, 4),
                     "int8_mfma_frac_of_5POPS": round(2.0 * digits * n * dim * nq / (per_search * 1e-3) / 5.0e15, 4),
                     "fallback_queries": int(c.fallback_queries), "candidates": int(c.prefilter_candidates), "verified_rows": int(c.verified_rows)}
        lists[name] = res
    qa.set_option(opt, -1)
    a, b = lists["narrow_32_per_pass"], lists["wide_128_per_pass"]
    out["lists_equal"] = bool(all(x["idx"].tolist() == y["idx"].tolist() and np.array_equal(x["score"].view(np.uint32), y["score"].view(np.uint32))
                                  for x, y in zip(a, b)))
    line = json.dumps(out)
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()

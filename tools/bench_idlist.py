#!/usr/bin/env python3
"""Filtered brute force (peek_top_iter over a candidate id list, point_scorer.rs:423-472) on C2-shaped data: 10 M x 768 f32 cosine,
a random half of the points as candidates.  One JSON line per batch size; QMX_NO_MFMA16=1 gives the 4x4x1 kernel for comparison."""
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
    ap.add_argument("--fraction", type=float, default=0.5)
    ap.add_argument("--batches", default="16,32,64")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import numpy as np
    import torch
    import qdrant_amd as qa
    from qdrant_amd import _ffi as F
    lib = F.lib()
    dev = torch.device("cuda", 0)
    n, dim, top = args.rows, args.dim, 10
    rows = torch.empty((n, dim), dtype=torch.float32, device=dev)
    F.check(lib.qmx_synth_fill_f32(0, 0x5EED0002, 0, n, dim, F.ptr(rows)))
    F.check(lib.qmx_preprocess_f32(0, int(qa.Distance.Cosine), F.ptr(rows), n, dim, F.ptr(rows)))
    torch.cuda.synchronize()
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    m = int(n * args.fraction)
    ids = torch.randperm(n, device=dev)[:m].to(torch.int32).sort().values      # a payload index yields ascending offsets
    for Q in [int(x) for x in args.batches.split(",")]:
        q = torch.randn((Q, dim), device=dev, dtype=torch.float32)
        qh = C.c_void_p()
        F.check(lib.qmx_query_create(vs._h, F.ptr(q), Q, C.byref(qh)))
        F.check(lib.qmx_query_set_timing(qh, 1))
        out = torch.zeros((Q, top, 2), dtype=torch.int32, device=dev)
        counts = torch.zeros((Q,), dtype=torch.int32, device=dev)
        ms, nl = C.c_float(), C.c_uint32()
        F.check(lib.qmx_search_topk_async(qh, top, F.ptr(ids), m, F.ptr(out), F.ptr(counts)))
        F.check(lib.qmx_query_synchronize(qh))
        F.check(lib.qmx_query_timing(qh, C.byref(ms), C.byref(nl)))
        t0 = time.perf_counter()
        for _ in range(args.reps):
            F.check(lib.qmx_search_topk_async(qh, top, F.ptr(ids), m, F.ptr(out), F.ptr(counts)))
        F.check(lib.qmx_query_synchronize(qh))
        wall = (time.perf_counter() - t0) / args.reps
        F.check(lib.qmx_query_timing(qh, C.byref(ms), C.byref(nl)))
        kms = ms.value / max(nl.value, 1)
        F.check(lib.qmx_query_destroy(qh))
        print(json.dumps({"workload": "filtered brute force: %d of %d x %d f32 cosine, top-10" % (m, n, dim), "batch": Q,
                          "scan_kernel_ms": round(kms, 3), "launches_per_search": nl.value / args.reps, "ms_per_search_wall": round(wall * 1e3, 3),
                          "qps": round(Q / wall, 1), "gathered_GBps": round(m * dim * 4 / (kms * 1e-3) / 1e9, 1),
                          "chain_major": qa.get_option("no_mfma16") == 0}), flush=True)


if __name__ == "__main__":
    main()

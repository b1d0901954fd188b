#!/usr/bin/env python3
"""HNSW search on the device vs the CPU oracle's restatement of GraphLayers::search, on one graph.

Not the round's headline bench (bench.py = C2 brute force): this measures the C3 / C4 style path of
BASELINE.json at a size whose graph the CPU oracle can build in a minute or two on the GPU box's host
cores (the 10 M-point graphs of C3 / C4 need a device-side builder).

  rows: N x dim f32, N(0,1) synthetic, cosine-normalised; graph: oracle HNSW (m, ef_construct), parallel build
  scorer: f32 | sq (EncodedVectorsU8, dot) | pq (EncodedVectorsPQ chunk 16, dot)
  device: qmx_hnsw_search of `nq` queries in one launch (one wavefront per search)
  checks: first `--check` queries against the oracle's walk (ids + score bits), recall@10 vs exact search
  cpu:    oracle search, single thread, `--cpu-queries` queries (the reference runs one search per thread)

Prints one JSON line.  Usage: python tools/bench_hnsw.py --rows 1000000 --dim 768 --scorer sq
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--scorer", default="sq", help="f32 | sq | pq, or a comma list: one graph, one JSON line per scorer")
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--ef-construct", type=int, default=100)
    ap.add_argument("--ef", type=int, default=128)
    ap.add_argument("--top", type=int, default=10)
    ap.add_argument("--nq", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", type=int, default=64)
    ap.add_argument("--cpu-queries", type=int, default=256)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    args = ap.parse_args()

    import numpy as np
    import torch  # noqa: F401  (HIP runtime load order, see tests/conftest.py)
    import qdrant_amd as qa
    from qdrant_amd import _ffi as F
    import oracle_ffi as O   # the checker and the CPU baseline

    n, dim, nq = args.rows, args.dim, args.nq
    t0 = time.time()
    rows = O.preprocess(O.COSINE, O.synth(0x5EED0003, 0, n, dim))
    queries = O.synth(0x5EED0004, 0, nq, dim)
    qpre = O.preprocess(O.COSINE, queries)
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    t_data = time.time() - t0
    t0 = time.time()
    g = O.Hnsw(st, m=args.m, ef_construct=args.ef_construct, seed=42, threads=args.threads)
    t_build = time.time() - t0
    plain = g.export_plain()

    def run_one(which):
        vs = vs_f32
        # ---- device side ----
        oracle_search = None
        if which == "f32":
            enc, row_bytes = vs, dim * 4
            oracle_search = lambda qs: g.search_dense(st, qs, args.top, args.ef)                      # noqa: E731
            dev_queries = queries
        elif which == "sq":
            quant = qa.ScalarQuantizer.from_min_max(rows, dim, qa.Distance.Dot)
            codes = quant.encode(rows)
            enc, row_bytes = qa.EncodedVectorsU8(codes, quant), codes.shape[1]
            osq = O.SqOracle(O.DOT, dim, quant.alpha, quant.offset)
            osq.rows = codes
            oracle_search = lambda qs: g.search_sq(st, osq, O.preprocess(O.COSINE, qs), args.top, args.ef)   # noqa: E731
            dev_queries = qpre            # the quantized storage is a Dot storage over normalised vectors
        else:
            chunk = 16
            cen = O.PqOracle.train(rows[:20000], dim, chunk, 256, iters=5)
            quant = qa.ProductQuantizer(dim, qa.Distance.Dot, chunk, cen)
            codes = quant.encode(rows)
            enc, row_bytes = qa.EncodedVectorsPQ(codes, quant), codes.shape[1]
            opq = O.PqOracle(O.DOT, dim, chunk, cen)
            opq.codes = codes
            oracle_search = lambda qs: g.search_pq(st, opq, O.preprocess(O.COSINE, qs), args.top, args.ef)   # noqa: E731
            dev_queries = qpre
        graph = qa.GraphLayers.from_plain(plain)
        scorer = qa.new_raw_scorer(dev_queries, enc)
        lib = F.lib()
        F.check(lib.qmx_query_set_timing(scorer._h, 1))
        got = graph.search(args.top, args.ef, scorer)       # warm-up (allocates the visited bitmaps)
        ms, nl = C.c_float(), C.c_uint32()
        F.check(lib.qmx_query_timing(scorer._h, C.byref(ms), C.byref(nl)))
        wall, scored = [], 0
        for _ in range(args.reps):
            t0 = time.time()
            got, scored = graph.search(args.top, args.ef, scorer, with_scored=True)
            wall.append(time.time() - t0)
        F.check(lib.qmx_query_timing(scorer._h, C.byref(ms), C.byref(nl)))
        kernel_ms = ms.value / max(nl.value, 1)

        # ---- parity with the oracle's walk, recall vs exact ----
        want = oracle_search(queries[:args.check])
        same_ids = sum(int(a["idx"].tolist() == b["idx"].tolist()) for a, b in zip(got, want))
        same_scores = sum(int(np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32))) for a, b in zip(got, want))
        exact = qa.BatchFilteredSearcher(queries[:256], vs, args.top).peek_top_all()
        recall = sum(len(set(a["idx"].tolist()) & set(b["idx"].tolist())) for a, b in zip(got[:256], exact)) / (256.0 * args.top)

        # ---- CPU baseline: the oracle's search, one thread ----
        t0 = time.time()
        cpu_res = oracle_search(queries[:args.cpu_queries])
        t_cpu = time.time() - t0
        cpu_recall = sum(len(set(a["idx"].tolist()) & set(b["idx"].tolist())) for a, b in zip(cpu_res[:256], exact)) / (
            min(256, args.cpu_queries) * float(args.top))

        out = {
            "metric": "HNSW search QPS (device-resident walk)", "scorer": which, "rows": n, "dim": dim, "m": args.m,
            "ef_construct": args.ef_construct, "ef": args.ef, "top": args.top, "nq": nq,
            "qps_kernel": round(nq / (kernel_ms * 1e-3), 1), "kernel_ms": round(kernel_ms, 3),
            "qps_wall_incl_copies": round(nq / min(wall), 1),
            "points_scored_per_query": round(scored / nq, 1),
            "gather_GBps": round(scored * row_bytes / (kernel_ms * 1e-3) / 1e9, 1), "row_bytes": int(row_bytes),
            "recall_at_10": round(recall, 4), "cpu_recall_at_10": round(cpu_recall, 4),
            "oracle_walk_same_ids": f"{same_ids}/{len(want)}", "oracle_walk_same_score_bits": f"{same_scores}/{len(want)}",
            "cpu_baseline": {"qps_one_thread": round(args.cpu_queries / t_cpu, 1), "kind": "port", "cores": 1,
                             "host_cores": os.cpu_count()},
            "build_s": round(t_build, 1), "data_s": round(t_data, 1), "build_threads": args.threads,
        }
        print(json.dumps(out), flush=True)

    vs_f32 = qa.VectorStorage(rows, qa.Distance.Cosine)
    for which in args.scorer.split(","):
        run_one(which)


if __name__ == "__main__":
    main()

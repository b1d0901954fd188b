#!/usr/bin/env python3
"""The HNSW walk of C3 (10 M x 768 SQ-int8, built through the SQ scorer, ef = 128, 8192 searches per launch) - and optionally of C4 (PQ m = 96 over d = 1536) -
under the walk's options, one JSON line per variant: kernel time (HIP events on the kernel's stream), scored points, useful HBM fraction, and whether the
lists equal the first variant's bit for bit (every option here is result-identical by construction; this checks it at full size).

  python tools/walk_variants.py --rows 10000000 --variants hnsw_spec=0 hnsw_spec=1 hnsw_spec=2 [--c4-rows 2000000]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--c4-rows", type=int, default=0)
    ap.add_argument("--nq", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--graph-cache", default="", help="npz file: the C3 graph is loaded from it when it exists, built and saved otherwise (PMC passes re-run the process)")
    ap.add_argument("--variants", nargs="+", default=["hnsw_spec=0", "hnsw_spec=1", "hnsw_spec=2"], help="name=value[,name=value] per variant")
    args = ap.parse_args()
    import numpy as np
    import torch
    import qdrant_amd as qa
    from qdrant_amd import _ffi as F
    import bench_sections as S
    lib = F.lib()
    dev = torch.device("cuda", 0)
    ctx = dict(args=argparse.Namespace(config_rows=args.rows, rows=args.rows), dev=dev, lib=lib, F=F, qa=qa, np=np, torch=torch)
    top, ef = 20, 128

    def run(tag, graph, scorer, row_bytes, n_rows):
        first = None
        for v in args.variants:
            pairs = [kv.split("=") for kv in v.split(",")]
            for k, val in pairs:
                qa.set_option(k, int(val))
            try:
                F.check(lib.qmx_query_set_timing(scorer._h, 1))
                res = graph.search(top, ef, scorer)      # warm-up
                ms, nl = C.c_float(), C.c_uint32()
                F.check(lib.qmx_query_timing(scorer._h, C.byref(ms), C.byref(nl)))
                scored = 0
                t0 = time.perf_counter()
                for _ in range(args.reps):
                    res, sc = graph.search(top, ef, scorer, with_scored=True)
                    scored += sc
                wall = (time.perf_counter() - t0) / args.reps
                F.check(lib.qmx_query_timing(scorer._h, C.byref(ms), C.byref(nl)))
                kernel_ms = ms.value / max(1, nl.value)
                per_launch = scored / float(args.reps)
                gbps = per_launch * row_bytes / (kernel_ms * 1e-3) / 1e9
                same = None
                if first is None:
                    first = res
                else:
                    same = all(a["idx"].tolist() == b["idx"].tolist() and np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32)) for a, b in zip(res, first))
                print(json.dumps({"walk": tag, "rows": n_rows, "variant": v, "kernel": F.last_kernel(scorer._h), "kernel_ms": round(kernel_ms, 4), "wall_ms": round(wall * 1e3, 3),
                                  "scored_per_query": round(per_launch / scorer.nq, 1), "useful_GBps": round(gbps, 1), "frac_of_hbm": round(gbps / 8000.0, 4),
                                  "equals_first_variant": same}), flush=True)
            finally:
                for k, _ in pairs:
                    qa.set_option(k, -1)

    # ---- C3 ----
    n, dim = args.rows, 768
    rows = S._latent(ctx, 0x5EED0003, 0, n, dim)
    queries = S._latent(ctx, 0x5EED0003, S.QUERY_ROW0, args.nq, dim)
    quant = qa.ScalarQuantizer.fit(rows, dim, qa.Distance.Dot)
    p = quant.params()
    codes = torch.empty((n, quant.quantized_vector_size()), dtype=torch.uint8, device=dev)
    F.check(lib.qmx_sq_encode(0, int(qa.Distance.Dot), C.byref(p), F.ptr(rows), n, dim, F.ptr(codes)))
    torch.cuda.synchronize(dev)
    enc = qa.EncodedVectorsU8(codes, quant)
    del codes
    t0 = time.perf_counter()
    if args.graph_cache and os.path.exists(args.graph_cache):
        z = np.load(args.graph_cache)

        class Plain:
            pass
        pl = Plain()
        for k in ("reindex", "level_offsets", "offsets", "neighbors", "ep_ids", "ep_levels", "xp_ids", "xp_levels"):
            setattr(pl, k, z[k])
        pl.m, pl.m0 = int(z["m"]), int(z["m0"])
        graph = qa.GraphLayers.from_plain(pl)
        print(json.dumps({"walk": "C3", "loaded_s": round(time.perf_counter() - t0, 2)}), flush=True)
    else:
        graph = qa.GraphLayers.build(enc, m=16, ef_construct=100, seed=42)
        print(json.dumps({"walk": "C3", "build_s": round(time.perf_counter() - t0, 2)}), flush=True)
        if args.graph_cache:
            pl = graph.export_plain()
            np.savez(args.graph_cache, m=pl.m, m0=pl.m0, reindex=pl.reindex, level_offsets=pl.level_offsets, offsets=pl.offsets, neighbors=pl.neighbors,
                     ep_ids=pl.ep_ids, ep_levels=pl.ep_levels, xp_ids=np.asarray(getattr(pl, "xp_ids", ()), dtype=np.uint32),
                     xp_levels=np.asarray(getattr(pl, "xp_levels", ()), dtype=np.uint32))
    run("C3 SQ", graph, qa.new_raw_scorer(queries.contiguous(), enc), quant.quantized_vector_size(), n)
    graph.close()
    del enc, rows, graph
    torch.cuda.empty_cache()
    # ---- C4 (smaller graph: the relative effect) ----
    if args.c4_rows:
        n, dim, chunk = args.c4_rows, 1536, 16
        rows = S._latent(ctx, 0x5EED0004, 0, n, dim)
        queries = S._latent(ctx, 0x5EED0004, S.QUERY_ROW0, args.nq, dim)
        vs = qa.VectorStorage(rows, qa.Distance.Cosine)
        sample = rows[::max(1, n // 10000)][:10000].contiguous()
        cen = torch.zeros((256, dim), dtype=torch.float32, device=dev)
        iters = np.zeros(dim // chunk, dtype=np.uint32)
        F.check(lib.qmx_pq_train(0, F.ptr(sample), sample.shape[0], dim, chunk, 256, 100, 1e-5, 1, F.ptr(cen), F.ptr(iters)))
        quant = qa.ProductQuantizer(dim, qa.Distance.Dot, chunk, cen.cpu().numpy(), lut_mfma=True)
        pp = quant.params()
        codes = torch.empty((n, quant.m), dtype=torch.uint8, device=dev)
        F.check(lib.qmx_pq_encode(0, C.byref(pp), F.ptr(rows), n, dim, F.ptr(codes)))
        enc = qa.EncodedVectorsPQ(codes, quant)
        t0 = time.perf_counter()
        graph = qa.GraphLayers.build(enc, m=16, ef_construct=100, seed=42, original=vs)
        print(json.dumps({"walk": "C4", "build_s": round(time.perf_counter() - t0, 2)}), flush=True)
        scorer = qa.new_raw_scorer(queries.contiguous(), enc)
        run("C4 PQ LUT", graph, scorer, quant.m, n)
        qa.set_option("hnsw_pq_direct_walk", 1)
        try:
            run("C4 PQ LUT-free", graph, scorer, quant.m, n)
        finally:
            qa.set_option("hnsw_pq_direct_walk", -1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Device side of the HNSW quality comparison: the same rows and queries as tools/hnsw_quality_cpu.py
(qmx_synth_fill_latent_f32 == qo_synth_fill_latent_f32 bit for bit), graph built by qmx_hnsw_build, walked by qmx_hnsw_search,
recall@10 against exact brute force on the device, for a range of ef and of `max_batch` (the cap on how many points one build
batch inserts concurrently; a batch never exceeds 1/32 of the points already linked).  One JSON line per (rows, max_batch)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
QUERY_ROW0 = 1 << 40


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="1000000")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--noise", type=float, default=1.0)
    ap.add_argument("--seed", type=lambda x: int(x, 0), default=0x5EED0003)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--ef-construct", type=int, default=100)
    ap.add_argument("--efs", default="16,32,64,128,256,512")
    ap.add_argument("--max-batches", default="0,4096,1024")
    ap.add_argument("--nq", type=int, default=1000)
    ap.add_argument("--build-over", default="f32", help="f32 | sq")
    args = ap.parse_args()
    import ctypes as C
    import numpy as np
    import torch
    import qdrant_amd as qa
    from qdrant_amd import _ffi as F
    lib = F.lib()
    dev = torch.device("cuda", 0)
    dim, top = args.dim, 10
    for n in [int(x) for x in args.rows.split(",")]:
        rows = torch.empty((n, dim), dtype=torch.float32, device=dev)
        F.check(lib.qmx_synth_fill_latent_f32(0, args.seed, 0, n, dim, args.latent, args.noise, F.ptr(rows)))
        F.check(lib.qmx_preprocess_f32(0, int(qa.Distance.Cosine), F.ptr(rows), n, dim, F.ptr(rows)))
        queries = torch.empty((args.nq, dim), dtype=torch.float32, device=dev)
        F.check(lib.qmx_synth_fill_latent_f32(0, args.seed, QUERY_ROW0, args.nq, dim, args.latent, args.noise, F.ptr(queries)))
        F.check(lib.qmx_preprocess_f32(0, int(qa.Distance.Cosine), F.ptr(queries), args.nq, dim, F.ptr(queries)))
        vs = qa.VectorStorage(rows, qa.Distance.Cosine)
        exact = qa.BatchFilteredSearcher(queries.cpu().numpy(), vs, top).peek_top_all()
        build_storage = vs
        if args.build_over == "sq":
            quant = qa.ScalarQuantizer.fit(rows, dim, qa.Distance.Dot)
            p = quant.params()
            codes = torch.empty((n, quant.quantized_vector_size()), dtype=torch.uint8, device=dev)
            F.check(lib.qmx_sq_encode(0, int(qa.Distance.Dot), C.byref(p), F.ptr(rows), n, dim, F.ptr(codes)))
            build_storage = qa.EncodedVectorsU8(codes, quant)
            del codes
        scorer = qa.new_raw_scorer(queries, vs)
        for mb in [int(x) for x in args.max_batches.split(",")]:
            torch.cuda.synchronize()
            t0 = time.time()
            g = qa.GraphLayers.build(build_storage, m=args.m, ef_construct=args.ef_construct, seed=42, max_batch=mb)
            t_build = time.time() - t0
            curve = {}
            for ef in [int(x) for x in args.efs.split(",")]:
                res, scored = g.search(top, ef, scorer, with_scored=True)
                rec = sum(len(set(a["idx"].tolist()) & set(b["idx"].tolist())) for a, b in zip(res, exact)) / float(args.nq * top)
                curve[str(ef)] = {"recall_at_10": round(rec, 4), "points_scored_per_query": round(scored / args.nq, 1)}
            print(json.dumps({"what": "device HNSW build (qmx_hnsw_build over %s) + device walk (f32 cosine)" % args.build_over, "rows": n, "dim": dim,
                              "latent_dim": args.latent, "noise": args.noise, "seed": args.seed, "m": args.m, "ef_construct": args.ef_construct,
                              "max_batch": mb if mb else 16384, "nq": args.nq, "build_s": round(t_build, 2), "build_points_per_s": round(n / t_build, 1),
                              "recall_vs_ef": curve}), flush=True)
            del g
        del scorer, vs, rows, build_storage
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

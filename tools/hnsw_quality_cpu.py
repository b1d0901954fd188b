#!/usr/bin/env python3
"""CPU side of the HNSW quality comparison: the oracle's PARALLEL builder (GraphLayersBuilder restated,
rayon-style concurrent insertions under per-point locks, oracle/qdrant_oracle_hnsw.c) over rows of low intrinsic dimension
(qo_synth_fill_latent_f32: bit-identical to what qmx_synth_fill_latent_f32 generates on the device), then recall@10 of the
oracle's walk against exact brute force for a range of ef.  Needs no GPU; the device side is tools/hnsw_quality_gpu.py, which
generates the same rows and queries, builds with qmx_hnsw_build and walks with qmx_hnsw_search.  One JSON document."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

QUERY_ROW0 = 1 << 40   # queries = rows of the same generator (same latent basis), far past the stored range


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--noise", type=float, default=1.0)
    ap.add_argument("--seed", type=lambda x: int(x, 0), default=0x5EED0003)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--ef-construct", type=int, default=100)
    ap.add_argument("--efs", default="16,32,64,128,256,512")
    ap.add_argument("--nq", type=int, default=1000)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import numpy as np
    import oracle_ffi as O

    n, dim, top = args.rows, args.dim, 10
    t0 = time.time()
    rows = O.preprocess(O.COSINE, O.synth_latent(args.seed, 0, n, dim, args.latent, args.noise))
    queries = O.preprocess(O.COSINE, O.synth_latent(args.seed, QUERY_ROW0, args.nq, dim, args.latent, args.noise))
    t_data = time.time() - t0
    t0 = time.time()
    gt = np.zeros((args.nq, top), dtype=np.int64)
    for q0 in range(0, args.nq, 100):
        s = rows @ queries[q0:q0 + 100].T
        part = np.argpartition(-s, top, axis=0)[:top]
        for j in range(part.shape[1]):
            gt[q0 + j] = part[np.argsort(-s[part[:, j], j]), j]
    t_gt = time.time() - t0
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    t0 = time.time()
    g = O.Hnsw(st, m=args.m, ef_construct=args.ef_construct, seed=42, threads=args.threads)
    t_build = time.time() - t0
    curve = {}
    for ef in [int(x) for x in args.efs.split(",")]:
        t0 = time.time()
        res, stats = g.search_dense(st, queries, top, ef, with_stats=True)
        el = time.time() - t0
        rec = sum(len(set(r["idx"].tolist()) & set(gt[i].tolist())) for i, r in enumerate(res)) / float(args.nq * top)
        curve[str(ef)] = {"recall_at_10": round(rec, 4), "points_scored_per_query": round(sum(stats) / args.nq, 1), "qps_one_thread": round(args.nq / el, 1)}
        print("ef", ef, curve[str(ef)], file=sys.stderr, flush=True)
    doc = {"what": "oracle (CPU) parallel HNSW build + oracle walk, f32 cosine", "rows": n, "dim": dim, "latent_dim": args.latent, "noise": args.noise,
           "seed": args.seed, "m": args.m, "ef_construct": args.ef_construct, "threads": args.threads, "nq": args.nq, "top": top,
           "data_s": round(t_data, 1), "ground_truth_s": round(t_gt, 1), "build_s": round(t_build, 1), "build_points_per_s": round(n / t_build, 1),
           "recall_vs_ef": curve}
    text = json.dumps(doc)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""MaxSim brute force over a multi-dense storage (ColBERT-shaped: 128-d token vectors), one JSON line.
points x ~tokens inner rows on the device; a batch of multi-queries of `--qtokens` tokens each; qmx_multi_search_topk = dense
score-mode scan of every (query token, stored token) pair + the MaxSim reduction + top-k.  First query checked against the oracle."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=100_000)
    ap.add_argument("--tokens", type=int, default=16, help="mean inner vectors per point")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--queries", type=int, default=8)
    ap.add_argument("--qtokens", type=int, default=32)
    ap.add_argument("--top", type=int, default=10)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import numpy as np
    import qdrant_amd as qa
    import oracle_ffi as O

    rng = np.random.default_rng(1)
    lens = rng.integers(max(1, args.tokens // 2), args.tokens * 3 // 2 + 1, args.points)
    offsets = np.zeros(args.points + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    n_rows = int(offsets[-1])
    inner = O.preprocess(O.COSINE, rng.standard_normal((n_rows, args.dim)).astype(np.float32))
    st = qa.MultiDenseVectorStorage(inner, offsets, qa.Distance.Cosine)
    queries = [rng.standard_normal((args.qtokens, args.dim)).astype(np.float32) for _ in range(args.queries)]
    st.peek_top_all(queries, args.top)                      # warm-up (allocations)
    walls = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        res = st.peek_top_all(queries, args.top)
        walls.append(time.perf_counter() - t0)
    # oracle check of the first query on a sample of points
    sample = np.arange(min(args.points, 2000), dtype=np.uint32)
    ost = O.DenseStorage(O.F32, O.COSINE, inner[: int(offsets[len(sample)])])
    want = O.multi_scores(ost, queries[0], np.array([0, args.qtokens], dtype=np.uint32), offsets, sample)[0]
    got = st.score_points(queries[:1], sample)[0]
    w = min(walls)
    pairs = args.queries * args.qtokens * n_rows
    print(json.dumps({"metric": "MaxSim brute-force (multi-dense vectors)", "points": args.points, "inner_rows": n_rows, "dim": args.dim,
                      "queries": args.queries, "query_tokens": args.qtokens, "top": args.top, "wall_s": round(w, 4),
                      "multi_queries_per_s": round(args.queries / w, 1), "token_pairs_per_s": round(pairs / w, 1),
                      "inner_block_GB": round(n_rows * args.dim * 4 / 1e9, 3), "sim_matrix_GB": round(pairs * 4 / 1e9, 3),
                      "scores_match_oracle_bit_for_bit_on_sample": bool(np.array_equal(got.view(np.uint32), want.view(np.uint32))),
                      "top1": [int(r["idx"][0]) for r in res[:3]]}), flush=True)


if __name__ == "__main__":
    main()

"""Triage of the int8-copy prefilter on the GPU box: one small search with a synchronisation after every stage (qmx_set_option("debug", 2)),
compared with the exact scan of the same block; prints what differs and what the pass cost.
usage: python tools/i8_triage.py [dim] [nq] [top] [dot|cosine] [rows]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_ffi as O      # noqa: E402  (test infrastructure: the checker)
import qdrant_amd as qa     # noqa: E402


def main():
    arg = sys.argv[1:]
    dim = int(arg[0]) if len(arg) > 0 else 128
    nq = int(arg[1]) if len(arg) > 1 else 128
    top = int(arg[2]) if len(arg) > 2 else 10
    dot = len(arg) > 3 and arg[3] == "dot"
    n = int(arg[4]) if len(arg) > 4 else 300_000
    rows = O.synth(0x5EED0700 + dim, 0, n, dim)
    if not dot:
        rows = O.preprocess(O.COSINE, rows)
    queries = O.synth(0x5EED0701 + nq, 0, nq, dim)
    t0 = time.time()
    vs = qa.VectorStorage(rows, qa.Distance.Dot if dot else qa.Distance.Cosine, flags=qa._ffi.SEG_I8_COPY)
    print("segment with int8 copy: %.2f s" % (time.time() - t0), flush=True)
    qa.set_option("debug", 2 if os.environ.get("I8_STAGES") else 0)
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    qa.set_option("debug", -1)
    print("kernel:", qa._ffi.last_kernel(s.scorer._h))
    c = s.counters
    print("prefilter_queries %d candidates/q %.1f verified/q %.1f fallback %d" % (c.prefilter_queries, c.prefilter_candidates / nq, c.verified_rows / nq,
                                                                                 c.fallback_queries), flush=True)
    qa.set_option("no_split_scan", 1)
    want = qa.BatchFilteredSearcher(queries, vs, top).peek_top_all()
    qa.set_option("no_split_scan", -1)
    bad = []
    for j, (g, w) in enumerate(zip(got, want)):
        if g["idx"].tolist() != w["idx"].tolist() or not np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32)):
            bad.append(j)
            if len(bad) <= 4:
                missing = [(int(i), k, float(w["score"][k])) for k, i in enumerate(w["idx"]) if i not in set(g["idx"].tolist())]
                extra = [int(i) for i in g["idx"] if i not in set(w["idx"].tolist())]
                print("query", j, "len", len(g), len(w), "missing (id, rank, score)", missing, "extra", extra, "k-th score", float(w["score"][-1]))
    print("queries that differ from the exact scan: %d / %d: %s" % (len(bad), nq, bad[:40]), flush=True)


if __name__ == "__main__":
    main()

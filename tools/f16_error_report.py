#!/usr/bin/env python3
"""Worst observed error of the f16 scorers against the oracle (AVX2 F16C leaf), reported three ways: relative to
sum|terms| (the test's bar), relative to the score itself over ALL pairs, and relative to the score over the returned top-10 of a C2-shaped
search (unit vectors, d = 768: the scores a user sees).  One JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import qdrant_amd as qa
    import oracle_ffi as O
    out = {}
    worst_terms, worst_score_all = 0.0, 0.0
    for dist, qd in ((O.DOT, qa.Distance.Dot), (O.COSINE, qa.Distance.Cosine)):
        for dim in (32, 100, 768, 1536):
            for nq in (4, 16, 32):
                rng = np.random.default_rng(dim + nq)
                rows16 = O.to_f16(O.preprocess(dist, rng.standard_normal((2000, dim)).astype(np.float32)))
                queries = rng.standard_normal((nq, dim)).astype(np.float32)
                st = qa.VectorStorage(rows16.view(np.float16), qd, qa.VectorStorageDatatype.Float16)
                ost = O.DenseStorage(O.F16, dist, rows16)
                ids = np.arange(2000, dtype=np.uint32)
                got = qa.new_raw_scorer(queries, st).score_points(ids).astype(np.float64)
                want = ost.score_points(queries, ids).astype(np.float64)
                q16 = ost.encode_queries(queries)
                scale = np.abs(O.f16_to_f32(q16).astype(np.float64)[:, None, :] * O.f16_to_f32(rows16).astype(np.float64)[None, :, :]).sum(-1)
                err = np.abs(got - want)
                worst_terms = max(worst_terms, float((err / scale).max()))
                nz = np.abs(want) > 0
                worst_score_all = max(worst_score_all, float((err[nz] / np.abs(want[nz])).max()))
    out["worst_err_over_sum_abs_terms"] = worst_terms
    out["worst_err_relative_to_score_all_pairs_incl_cancellations"] = worst_score_all
    # the scores a search returns: top-10 of 200 k unit rows, d = 768
    n, dim = 200_000, 768
    rows = O.preprocess(O.COSINE, O.synth(0x5EED0002, 0, n, dim))
    rows16 = O.to_f16(rows)
    queries = O.synth(0x5EED0012, 0, 32, dim)
    st = qa.VectorStorage(rows16.view(np.float16), qa.Distance.Cosine, qa.VectorStorageDatatype.Float16)
    ost = O.DenseStorage(O.F16, O.COSINE, rows16)
    got = qa.BatchFilteredSearcher(queries, st, 10).peek_top_all()
    want = ost.peek_top(queries, 10)
    rel, same = 0.0, 0
    for g, w in zip(got, want):
        same += int(g["idx"].tolist() == w["idx"].tolist())
        k = min(len(g), len(w))
        rel = max(rel, float(np.max(np.abs(g["score"][:k].astype(np.float64) - w["score"][:k]) / np.abs(w["score"][:k]))))
    out["top10_of_200k_unit_rows_d768"] = {"worst_relative_to_score": rel, "identical_id_lists": "%d/32" % same}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

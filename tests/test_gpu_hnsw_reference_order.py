"""The order among EQUAL scores inside the HNSW walk (VERDICT r4 missing #3 / weak #2).

The reference's walk is deterministic on a given graph: `candidates` is a std BinaryHeap, `nearest` a FixedLengthPriorityQueue
(search_context.rs:8-40, graph_layers.rs:108-149, fixed_length_priority_queue.rs:47-59) and ScoredPointOffset orders by score alone, so which
of two equal scores is popped / evicted first is a property of the heaps' arrays.  The oracle restates those heaps.  The device's default
walk keeps one sorted register list (equal scores: ascending id), so on integer-score storages (SQ, u8, BQ, 1-bit TurboQuant) - where ties
are the norm - its lists may differ from the reference's inside runs of equal scores.  Two things are pinned here:

  1. option "hnsw_reference_heap_order": the device keeps the reference's two heaps in std's sift order (hnsw.hpp RefHeaps).  In that mode
     the walk IS the oracle's walk on every storage: the same lists (ids and score bits), the same pop sequence (ids and score bits), the
     same number of scored points - ties or not.
  2. the default mode: per search, either the pop sequence equals the oracle's, or the FIRST position where the two sequences differ holds two
     candidates with bit-equal scores (`tie_explained`): the walks part at a tie and nowhere else.
"""
import zlib

import numpy as np
import pytest

import oracle_ffi as O
from parity_asserts import first_divergence_is_a_tie

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    import qdrant_amd
    assert qdrant_amd.device_count() >= 1
    return qdrant_amd


def _bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def _clustered(rng, n, dim, k=24, noise=0.5):
    centers = rng.standard_normal((k, dim)).astype(np.float32)
    return (centers[rng.integers(0, k, n)] + noise * rng.standard_normal((n, dim))).astype(np.float32)


def _world(qa, kind):
    """-> (graph on the device, device scorer, oracle walk function(top, ef) -> lists [the oracle graph collects pops], oracle graph, nq)"""
    rng = np.random.default_rng(zlib.crc32(kind.encode()) & 0xFFFF)
    n, m, nq = 3000, 8, 32
    if kind == "f32_duplicated_rows":          # every row exists twice: every score ties
        dim = 32
        base = O.preprocess(O.DOT, O.synth(0x5EED03B0, 0, n // 2, dim))
        rows = np.concatenate([base, base])
        st = O.DenseStorage(O.F32, O.DOT, rows)
        queries = O.synth(0x5EED03B1, 0, nq, dim)
        g = O.Hnsw(st, m=m, ef_construct=64, seed=3, threads=0)
        scorer = qa.new_raw_scorer(queries, qa.VectorStorage(rows, qa.Distance.Dot))
        walk = lambda top, ef: g.search_dense(st, queries, top, ef, with_stats=True)      # noqa: E731
    elif kind == "u8_euclid":                  # integer scores
        dim = 48
        stored = rng.integers(0, 24, size=(n, dim)).astype(np.uint8)      # a small alphabet: many equal distances
        st = O.DenseStorage(O.U8, O.EUCLID, stored)
        queries = rng.integers(0, 24, size=(nq, dim)).astype(np.float32)
        g = O.Hnsw(st, m=m, ef_construct=64, seed=3, threads=0)
        scorer = qa.new_raw_scorer(queries, qa.VectorStorage(stored, qa.Distance.Euclid, qa.VectorStorageDatatype.Uint8))
        walk = lambda top, ef: g.search_dense(st, queries, top, ef, with_stats=True)      # noqa: E731
    elif kind in ("sq_manhattan", "sq_dot"):
        distance = O.MANHATTAN if kind == "sq_manhattan" else O.DOT
        dim = 64
        rows = O.preprocess(distance, _clustered(rng, n, dim))
        st = O.DenseStorage(O.F32, distance, rows)
        queries = _clustered(rng, nq, dim)
        g = O.Hnsw(st, m=m, ef_construct=64, seed=3, threads=0)
        qd = qa.Distance.Manhattan if distance == O.MANHATTAN else qa.Distance.Dot
        quant = qa.ScalarQuantizer.from_min_max(rows, dim, qd)
        osq = O.SqOracle(distance, dim, quant.alpha, quant.offset)
        osq.encode_rows(rows)
        scorer = qa.new_raw_scorer(queries, qa.EncodedVectorsU8(quant.encode(rows), quant))
        qpre = O.preprocess(distance, queries)
        walk = lambda top, ef: (g.search_sq(st, osq, qpre, top, ef), None)      # noqa: E731
    elif kind == "bq":                         # small integer scores: ties everywhere
        dim = 128
        rows = O.preprocess(O.COSINE, _clustered(rng, n, dim, noise=0.6))
        st = O.DenseStorage(O.F32, O.COSINE, rows)
        queries = O.preprocess(O.COSINE, _clustered(rng, nq, dim, noise=0.6))
        g = O.Hnsw(st, m=m, ef_construct=64, seed=3, threads=0)
        quant = qa.BinaryQuantizer(dim, qa.Distance.Cosine)
        obq = O.BqOracle(O.COSINE, dim)
        obq.encode_rows(rows)
        scorer = qa.new_raw_scorer(queries, qa.EncodedVectorsBin(quant.encode(rows), quant))
        walk = lambda top, ef: (g.search_bq(st, obq, queries, top, ef), None)      # noqa: E731
    elif kind == "tq_1bit":
        dim = 128
        vecs = O.preprocess(O.COSINE, rng.uniform(-1.0, 1.0, (n, dim)).astype(np.float32))
        otq = O.TqOracle(O.COSINE, dim, O.TQ_BITS1)
        codes = otq.encode_rows(vecs)
        quant = qa.TurboQuantizer(dim, qa.Distance.Cosine, O.TQ_BITS1)
        st = O.DenseStorage(O.F32, O.COSINE, vecs)
        queries = rng.uniform(-1.0, 1.0, (nq, dim)).astype(np.float32)
        g = O.Hnsw(st, m=m, ef_construct=48, seed=5, threads=0)
        scorer = qa.new_raw_scorer(queries, qa.EncodedVectorsTQ(codes, quant))
        qpre = O.preprocess(O.COSINE, queries)
        walk = lambda top, ef: (g.search_tq(st, otq, qpre, top, ef), None)      # noqa: E731
    elif kind == "pq":                         # sums of few LUT entries tie now and then
        dim, chunk = 64, 4
        rows = O.preprocess(O.DOT, _clustered(rng, n, dim))
        st = O.DenseStorage(O.F32, O.DOT, rows)
        queries = _clustered(rng, nq, dim)
        g = O.Hnsw(st, m=m, ef_construct=64, seed=3, threads=0)
        cen = O.PqOracle.train(rows[:2000], dim, chunk, 256, iters=3)
        opq = O.PqOracle(O.DOT, dim, chunk, cen)
        codes = opq.encode(rows)
        quant = qa.ProductQuantizer(dim, qa.Distance.Dot, chunk, cen)
        scorer = qa.new_raw_scorer(queries, qa.EncodedVectorsPQ(codes, quant))
        qpre = O.preprocess(O.DOT, queries)
        walk = lambda top, ef: g.search_pq(st, opq, qpre, top, ef, with_stats=True)      # noqa: E731
    else:
        raise AssertionError(kind)
    return qa.GraphLayers.from_plain(g.export_plain()), scorer, walk, g, nq


KINDS = ["f32_duplicated_rows", "u8_euclid", "sq_manhattan", "sq_dot", "bq", "tq_1bit", "pq"]


@pytest.mark.parametrize("kind", KINDS)
def test_reference_heap_order_is_the_oracles_walk_among_equal_scores(qa, kind):
    graph, scorer, walk, g, nq = _world(qa, kind)
    qa.set_option("hnsw_reference_heap_order", 1)
    try:
        n_tied_lists = 0
        for top, ef in ((10, 64), (30, 30), (5, 600)):             # 600: a `nearest` heap of 600 entries in LDS
            g.pops = []
            want, stats = walk(top, ef)
            want_pops, g.pops = g.pops, None
            (got, got_pops), scored = graph.search_traced(top, ef, scorer), None
            assert "hnsw_search_kernel" in qa._ffi.last_kernel(scorer._h) and ", -1, " in qa._ffi.last_kernel(scorer._h), qa._ffi.last_kernel(scorer._h)
            for qi in range(nq):
                assert got[qi]["idx"].tolist() == want[qi]["idx"].tolist(), (kind, top, ef, qi)
                assert np.array_equal(_bits(got[qi]["score"]), _bits(want[qi]["score"]))
                assert got_pops[qi]["idx"].tolist() == want_pops[qi]["idx"].tolist(), (kind, top, ef, qi)
                assert np.array_equal(_bits(got_pops[qi]["score"]), _bits(want_pops[qi]["score"]))
                n_tied_lists += int(len(np.unique(want[qi]["score"])) < len(want[qi]))
            # the plain entry points take the mode too (same lists, + the reference's hardware counter)
            plain, scored = graph.search(top, ef, scorer, with_scored=True)
            for a, b in zip(plain, want):
                assert a["idx"].tolist() == b["idx"].tolist() and np.array_equal(_bits(a["score"]), _bits(b["score"]))
            if stats is not None:
                assert scored == sum(stats)
        if kind != "pq" and kind != "sq_dot":
            assert n_tied_lists > 0, "the world was meant to tie"
    finally:
        qa.set_option("hnsw_reference_heap_order", -1)


@pytest.mark.parametrize("kind", KINDS)
def test_default_walk_parts_from_the_oracle_only_at_equal_scores(qa, kind):
    """the default (sorted register list) walk: every search either pops the oracle's sequence or first differs from it at a pair of bit-equal scores;
    its score lists stay the oracle's wherever the sequences agree"""
    graph, scorer, walk, g, nq = _world(qa, kind)
    top = ef = 64                                    # (top = ef: the list returned is the whole `nearest`, its last score the bound at the end of the walk)
    g.pops = []
    want, _ = walk(top, ef)
    want_pops, g.pops = g.pops, None
    got, got_pops = graph.search_traced(top, ef, scorer)
    assert ", -1, " not in qa._ffi.last_kernel(scorer._h)
    same = explained = 0
    for qi in range(nq):
        verdict = first_divergence_is_a_tie(got_pops[qi], want_pops[qi], bound_score=got[qi]["score"][-1] if len(got[qi]) == ef else None)
        assert verdict in ("same", "tie"), (kind, qi, verdict)
        same += verdict == "same"
        explained += verdict == "tie"
        if verdict == "same":
            assert np.array_equal(_bits(got[qi]["score"]), _bits(want[qi]["score"]))
            assert sorted(zip(got[qi]["score"].tolist(), got[qi]["idx"].tolist())) == sorted(zip(want[qi]["score"].tolist(), want[qi]["idx"].tolist())) or \
                len(np.unique(want[qi]["score"])) < len(want[qi])
    assert same + explained == nq
    if kind in ("f32_duplicated_rows", "bq", "u8_euclid"):
        assert explained > 0, "ties were meant to change some walks"
    # and the traced call returns what the plain call returns
    for a, b in zip(graph.search(top, ef, scorer), got):
        assert a["idx"].tolist() == b["idx"].tolist() and np.array_equal(_bits(a["score"]), _bits(b["score"]))


def test_reference_heap_order_leaves_other_walks_alone_and_refuses_nothing_silently(qa):
    graph, scorer, walk, g, nq = _world(qa, "sq_dot")
    qa.set_option("hnsw_reference_heap_order", 1)
    try:
        # ACORN keeps its own loop (the option is the plain walk's): results unchanged by the option
        a = graph.search(10, 64, scorer, acorn=True)
        qa.set_option("hnsw_reference_heap_order", 0)
        b = graph.search(10, 64, scorer, acorn=True)
        for x, y in zip(a, b):
            assert x["idx"].tolist() == y["idx"].tolist()
    finally:
        qa.set_option("hnsw_reference_heap_order", -1)
    assert qa._ffi.get_option("hnsw_reference_heap_order") == 0

"""Custom queries (Recommend best-score / sum-scores, Discover, Context, naive Feedback) beyond dense brute force:
  * over quantized storages - `QuantizedCustomQueryScorer` (vector_storage/quantized/quantized_custom_query_scorer.rs:13-113: every example encoded as
    the storage's query, score_by over the quantized scores) for SQ / PQ / BQ and `TurboCustomQueryScorer` (query_scorer/turbo_custom_query_scorer.rs:17-113);
  * over multi-vector points - `MultiCustomQueryScorer` (query_scorer/multi_custom_query_scorer.rs:19-130) and its quantized twin
    (quantized/quantized_multi_custom_query_scorer.rs:19-96): similarity(example, point) = MaxSim;
  * as the scorer of the HNSW walk (raw_scorer.rs:228-333 builds them for any storage, graph_layers.rs:108-149 walks with whatever it gets).
Oracle: qo_scorer kind 6 = score_by over example scorers of the storage's own kind.  Bars: scores bit-exact, top-k lists identical, the device walk ==
the oracle's walk of the same graph (ids, score bits, points scored)."""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    import qdrant_amd
    assert qdrant_amd.device_count() >= 1
    return qdrant_amd


def _dist(qa, d):
    return {O.COSINE: qa.Distance.Cosine, O.DOT: qa.Distance.Dot, O.EUCLID: qa.Distance.Euclid, O.MANHATTAN: qa.Distance.Manhattan}[d]


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _storages(qa, kind, distance, rows, dim):
    """(device storage, oracle ScorerFactory) of one kind over the same preprocessed rows"""
    flags = O.DenseStorage(O.F32, distance, rows)
    if kind == "dense":
        return qa.VectorStorage(rows, _dist(qa, distance)), O.ScorerFactory("dense", flags), flags
    if kind == "sq":
        quant = qa.ScalarQuantizer.from_min_max(rows, dim, _dist(qa, distance))
        oq = O.SqOracle(distance, dim, quant.alpha, quant.offset)
        oq.rows = oq.encode_rows(rows)
        return qa.EncodedVectorsU8(quant.encode(rows), quant), O.ScorerFactory("sq", flags, oq), flags
    if kind == "pq":
        cen = O.PqOracle.train(rows[:2000], dim, 8, 256, iters=3)
        oq = O.PqOracle(distance, dim, 8, cen)
        oq.encode(rows)
        quant = qa.ProductQuantizer(dim, _dist(qa, distance), 8, cen)
        return qa.EncodedVectorsPQ(quant.encode(rows), quant), O.ScorerFactory("pq", flags, oq), flags
    if kind == "bq":
        quant = qa.BinaryQuantizer(dim, _dist(qa, distance))
        oq = O.BqOracle(distance, dim)
        oq.rows = oq.encode_rows(rows)
        return qa.EncodedVectorsBin(quant.encode(rows), quant), O.ScorerFactory("bq", flags, oq), flags
    otq = O.TqOracle(distance, dim, O.TQ_BITS4)
    otq.rows = otq.encode_rows(rows)
    quant = qa.TurboQuantizer(dim, _dist(qa, distance), O.TQ_BITS4)
    return qa.EncodedVectorsTQ(otq.rows, quant), O.ScorerFactory("tq", flags, otq), flags


def _queries(qa, rng, dim, centers=None):
    def V(k):
        if centers is None:
            return [rng.standard_normal(dim).astype(np.float32) for _ in range(k)]
        return [(centers[rng.integers(len(centers))] + 0.5 * rng.standard_normal(dim)).astype(np.float32) for _ in range(k)]
    return [
        qa.CustomQuery.recommend_best_score(V(3), V(2)),
        qa.CustomQuery.recommend_best_score(V(1), []),
        qa.CustomQuery.recommend_sum_scores(V(4), V(3)),
        qa.CustomQuery.discover(V(1)[0], [tuple(V(2)) for _ in range(3)]),
        qa.CustomQuery.context([tuple(V(2)) for _ in range(4)]),
        qa.CustomQuery.feedback_naive(V(1)[0], list(zip(V(4), [0.9, 0.1, 0.5, 0.7])), a=0.8, b=1.5, c=0.3),
    ]


def _oracle_scorer(fac, distance, q):
    return fac.custom([O.preprocess(distance, e[None, :])[0] for e in q.examples], q.kind, q.n_a, q.n_b, q.coefs)


@pytest.mark.parametrize("kind", ["sq", "pq", "bq", "tq"])
@pytest.mark.parametrize("distance", [O.DOT, O.EUCLID])
def test_custom_queries_over_quantized_storages(qa, kind, distance):
    rng = np.random.default_rng(7 + distance)
    n, dim = 2500, 64
    rows = O.preprocess(distance, rng.standard_normal((n, dim)).astype(np.float32))
    st, fac, flags = _storages(qa, kind, distance, rows, dim)
    queries = _queries(qa, rng, dim)
    scorer = qa.CustomRawScorer(queries, st)
    ids = rng.permutation(n)[:300].astype(np.uint32)
    got = scorer.score_points(ids)
    all_ids = np.arange(n, dtype=np.uint32)
    res = scorer.peek_top(20)
    for qi, q in enumerate(queries):
        s, keep = _oracle_scorer(fac, distance, q)
        want = O.scorer_score_points(s, ids)
        assert np.array_equal(_bits(got[qi]), _bits(want)), (kind, qi)
        sc = O.scorer_score_points(s, all_ids)
        order = np.lexsort((all_ids, -sc.astype(np.float64)))[:20]
        assert np.array_equal(_bits(res[qi]["score"]), _bits(sc[order])), (kind, qi)
        uniq = np.array([(res[qi]["score"] == x).sum() == 1 for x in res[qi]["score"]])
        assert np.array_equal(res[qi]["idx"][uniq], all_ids[order][uniq])


@pytest.mark.parametrize("kind", ["dense", "sq", "pq", "tq", "bq", "tq_l1"])
def test_custom_walk_equals_the_oracle_walk(qa, kind):
    """The graph comes from the oracle (built over the original rows); both sides walk it with the custom scorer of the storage `kind`.
    tq_l1: TurboQuant over Manhattan - every hop is scored against every example by the wave (HopCustom over HopTQL1, round 4)."""
    distance, dim, n, m = {"dense": O.COSINE, "tq_l1": O.MANHATTAN}.get(kind, O.DOT), 64, 4000, 8
    rng = np.random.default_rng(31)
    centers = rng.standard_normal((30, dim)).astype(np.float32) * 2
    rows = O.preprocess(distance, (centers[rng.integers(30, size=n)] + rng.standard_normal((n, dim))).astype(np.float32))
    st, fac, flags = _storages(qa, kind, distance, rows, dim)
    g = O.Hnsw(flags, m=m, ef_construct=48, seed=11)
    graph = qa.GraphLayers.from_plain(g.export_plain())
    queries = _queries(qa, rng, dim, centers)
    scorer = qa.CustomRawScorer(queries, st)
    for top, ef in ((10, 64), (5, 16), (10, 200), (10, 700)):          # (700: the LDS beam, hnsw.hpp Beam<0>)
        got, scored = scorer.search_hnsw(graph, top, ef, with_scored=True)
        total, ties = 0, False
        for qi, q in enumerate(queries):
            s, keep = _oracle_scorer(fac, distance, q)
            want, ns = g.search_scorer(s, top, ef)
            total += ns
            # integer-valued example scores (bq) and context queries (every point on the right side of all pairs scores exactly 0.0) tie in the
            # beam; the order among equals inside the reference's BinaryHeaps is unpinned: true scores, sorted
            if kind == "bq" or q.kind == qa._ffi.CUSTOM_CONTEXT:
                sc = O.scorer_score_points(s, got[qi]["idx"])
                assert np.array_equal(_bits(got[qi]["score"]), _bits(sc)) and np.all(np.diff(got[qi]["score"]) <= 0)
                ties = True
                continue
            assert got[qi]["idx"].tolist() == want["idx"].tolist(), (kind, qi, top, ef)
            assert np.array_equal(_bits(got[qi]["score"]), _bits(want["score"])), (kind, qi)
        if not ties:
            assert scored == total


def test_custom_walk_with_deleted_points_and_many_examples(qa):
    """16 positives + 16 negatives of 768 floats do not fit the walk's LDS share: the examples are then read through L2; deleted points are skipped."""
    distance, dim, n, m = O.COSINE, 768, 1500, 8
    rng = np.random.default_rng(5)
    rows = O.preprocess(distance, rng.standard_normal((n, dim)).astype(np.float32))
    deleted = rng.random(n) < 0.2
    flags = O.DenseStorage(O.F32, distance, rows, point_deleted=deleted)
    st = qa.VectorStorage(rows, qa.Distance.Cosine)
    st.set_deleted(deleted, None)
    fac = O.ScorerFactory("dense", flags)
    g = O.Hnsw(flags, m=m, ef_construct=32, seed=3)
    graph = qa.GraphLayers.from_plain(g.export_plain())
    V = lambda k: [rng.standard_normal(dim).astype(np.float32) for _ in range(k)]       # noqa: E731
    queries = [qa.CustomQuery.recommend_best_score(V(16), V(16)), qa.CustomQuery.recommend_sum_scores(V(2), V(1))]
    scorer = qa.CustomRawScorer(queries, st)
    got = scorer.search_hnsw(graph, 10, 64)
    for qi, q in enumerate(queries):
        s, keep = _oracle_scorer(fac, distance, q)
        want, _ = g.search_scorer(s, 10, 64)
        assert got[qi]["idx"].tolist() == want["idx"].tolist() and np.array_equal(_bits(got[qi]["score"]), _bits(want["score"]))
        assert not deleted[got[qi]["idx"]].any()


@pytest.mark.parametrize("inner_kind", ["dense", "sq"])
def test_custom_queries_over_multivector_points(qa, inner_kind):
    distance, dim, n_points = O.DOT, 48, 300
    rng = np.random.default_rng(13)
    lens = rng.integers(1, 9, n_points)
    offsets = np.zeros(n_points + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    inner = O.preprocess(distance, rng.standard_normal((int(offsets[-1]), dim)).astype(np.float32))
    deleted = rng.random(n_points) < 0.1
    ost = O.DenseStorage(O.F32, distance, inner)
    if inner_kind == "dense":
        st = qa.MultiDenseVectorStorage(inner, offsets, _dist(qa, distance))
        mo = O.MultiOracle(("dense", ost), offsets, point_deleted=deleted)
    else:
        quant = qa.ScalarQuantizer.from_min_max(inner, dim, _dist(qa, distance))
        osq = O.SqOracle(distance, dim, quant.alpha, quant.offset)
        osq.rows = osq.encode_rows(inner)
        st = qa.QuantizedMultivectorStorage(qa.EncodedVectorsU8(quant.encode(inner), quant), offsets)
        mo = O.MultiOracle(("sq", ost, osq), offsets, point_deleted=deleted)
    st.set_deleted(deleted)
    fac = O.ScorerFactory("multi", mo)
    MV = lambda k: [rng.standard_normal((int(rng.integers(1, 6)), dim)).astype(np.float32) for _ in range(k)]       # noqa: E731
    queries = [
        qa.CustomQuery.recommend_best_score(MV(2), MV(2)),
        qa.CustomQuery.recommend_sum_scores(MV(3), MV(1)),
        qa.CustomQuery.discover(MV(1)[0], [tuple(MV(2)) for _ in range(2)]),
        qa.CustomQuery.context([tuple(MV(2)) for _ in range(3)]),
    ]
    for q in queries:      # (CustomQuery keeps examples as given; multi-vector examples are 2-d)
        q.examples = [np.atleast_2d(e) for e in q.examples]
    ids = rng.permutation(n_points)[:120].astype(np.uint32)
    got = st.custom_score_points(queries, ids)
    top = st.custom_peek_top(queries, 15)
    all_ids = np.arange(n_points, dtype=np.uint32)
    for qi, q in enumerate(queries):
        s, keep = fac.custom([O.preprocess(distance, e) for e in q.examples], q.kind, q.n_a, q.n_b, q.coefs)
        assert np.array_equal(_bits(got[qi]), _bits(O.scorer_score_points(s, ids))), qi
        sc = O.scorer_score_points(s, all_ids)
        live = ~deleted
        order = np.lexsort((all_ids[live], -sc[live].astype(np.float64)))[:15]
        assert np.array_equal(_bits(top[qi]["score"]), _bits(sc[live][order])), qi
        uniq = np.array([(top[qi]["score"] == x).sum() == 1 for x in top[qi]["score"]])
        assert np.array_equal(top[qi]["idx"][uniq], all_ids[live][order][uniq])
        assert not deleted[top[qi]["idx"]].any()


@pytest.mark.parametrize("inner_kind", ["dense", "sq"])
def test_multivector_custom_walk_equals_the_oracle_walk(qa, inner_kind):
    """GraphLayers::search over multi-vector points with a MultiCustomQueryScorer: the oracle builds the graph (score_internal_max_similarity) and walks it
    with qo_scorer kind 6 over kind-4 example scorers; the device walks the same graph with HopCustom over HopMaxSim."""
    distance, dim, n_points = O.DOT, 64, 900
    rng = np.random.default_rng(41)
    centers = rng.standard_normal((20, dim)).astype(np.float32) * 2
    lens = rng.integers(1, 7, n_points)
    offsets = np.zeros(n_points + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    inner = O.preprocess(distance, (centers[rng.integers(20, size=int(offsets[-1]))] + rng.standard_normal((int(offsets[-1]), dim))).astype(np.float32))
    deleted = rng.random(n_points) < 0.15
    ost = O.DenseStorage(O.F32, distance, inner)
    if inner_kind == "dense":
        st = qa.MultiDenseVectorStorage(inner, offsets, _dist(qa, distance))
        mo = O.MultiOracle(("dense", ost), offsets, point_deleted=deleted)
    else:
        quant = qa.ScalarQuantizer.from_min_max(inner, dim, _dist(qa, distance))
        osq = O.SqOracle(distance, dim, quant.alpha, quant.offset)
        osq.rows = osq.encode_rows(inner)
        st = qa.QuantizedMultivectorStorage(qa.EncodedVectorsU8(quant.encode(inner), quant), offsets)
        mo = O.MultiOracle(("sq", ost, osq), offsets, point_deleted=deleted)
    st.set_deleted(deleted)
    graph_o = mo.build(m=8, ef_construct=32)
    graph = qa.GraphLayers.from_plain(graph_o.export_plain())
    fac = O.ScorerFactory("multi", mo)
    MV = lambda k: [(centers[rng.integers(20)] + rng.standard_normal((int(rng.integers(1, 6)), dim))).astype(np.float32) for _ in range(k)]       # noqa: E731
    queries = [
        qa.CustomQuery.recommend_best_score(MV(2), MV(2)),
        qa.CustomQuery.recommend_sum_scores(MV(3), MV(1)),
        qa.CustomQuery.discover(MV(1)[0], [tuple(MV(2)) for _ in range(2)]),
        qa.CustomQuery.feedback_naive(MV(1)[0], list(zip(MV(3), [0.9, 0.2, 0.6])), a=0.7, b=1.2, c=0.5),
    ]
    for q in queries:
        q.examples = [np.atleast_2d(e) for e in q.examples]
    for top, ef in ((10, 64), (5, 16)):
        got, ctr = st.custom_search_hnsw(graph, queries, top, ef, with_counters=True)
        total = 0
        for qi, q in enumerate(queries):
            s, keep = fac.custom([O.preprocess(distance, e) for e in q.examples], q.kind, q.n_a, q.n_b, q.coefs)
            want, ns = graph_o.search_scorer(s, top, ef)
            total += ns
            assert np.array_equal(_bits(got[qi]["score"]), _bits(want["score"])), (inner_kind, qi)
            # (SQ scores are integers times a constant: two points can tie, and the order among equals is unpinned - compare tie groups as sets)
            for sc in np.unique(want["score"]):
                assert set(got[qi]["idx"][got[qi]["score"] == sc].tolist()) == set(want["idx"][want["score"] == sc].tolist()), (inner_kind, qi, top, ef)
            assert not deleted[got[qi]["idx"]].any()
        assert ctr.vectors_scored == total

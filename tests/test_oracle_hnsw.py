"""CPU tests of the HNSW / merge part of the oracle (oracle/qdrant_oracle_hnsw.c).

Pins the restatement against the reference's literal known-answer test
lib/segment/src/index/hnsw_index/links_container.rs:312-391 (`test_connect_new_point`) and checks
the structural invariants the reference's own tests check (graph_layers_builder.rs tests: link
counts <= level_m, search recall against exact search)."""
import numpy as np
import pytest

import oracle_ffi as O

# links_container.rs:349-361: target + 10 points; + = selected by the heuristic
POINTS = np.array([
    [21.79, 7.18],   # target
    [20.58, 5.46],   # + 1 B
    [21.19, 4.51],   #   2 C
    [24.73, 8.24],   # + 3 D
    [24.55, 9.98],   #   4 E
    [26.11, 6.85],   #   5 F
    [17.64, 11.14],  # + 6 G
    [14.97, 11.52],  #   7 I
    [14.97, 9.60],   #   8 J
    [16.23, 14.32],  #   9 H
    [12.69, 19.13],  #  10 K
], dtype=np.float32)


def _score_table():
    # scorer(a, b) = -sqrt((ax-bx)^2 + (ay-by)^2) in f32 (links_container.rs:363-367)
    d = POINTS[:, None, :] - POINTS[None, :, :]
    return (-np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(np.float32))).astype(np.float32)


def test_reference_heuristic_known_answer():
    t = _score_table()
    m = 6
    ids = list(range(1, len(POINTS)))
    cand = O.topk_push_all([(i, t[0, i]) for i in ids], len(ids))          # FixedLengthPriorityQueue -> into_iter_sorted
    assert O.links_heuristic(cand, m, t) == [1, 3, 6]                       # links_container.rs:381


@pytest.mark.parametrize("seed", [0, 1, 42, 7])
def test_reference_connect_known_answer(seed):
    t = _score_table()
    m = 6
    ids = list(range(1, len(POINTS)))
    np.random.default_rng(seed).shuffle(ids)
    links = []
    for i in ids:
        links = O.links_connect(links, i, 0, m, t)
    assert links == [1, 2, 3, 4, 5, 6]                                      # links_container.rs:390


def _data(n, dim, seed, distance=O.COSINE):
    raw = O.synth(seed, 0, n, dim)
    return O.preprocess(distance, raw)


def _recall(got, want):
    hit = sum(len(set(g["idx"].tolist()) & set(w["idx"].tolist())) for g, w in zip(got, want))
    return hit / sum(len(w) for w in want)


@pytest.mark.parametrize("distance", [O.COSINE, O.EUCLID, O.DOT])
@pytest.mark.parametrize("heuristic", [True, False])
def test_build_invariants_and_recall(distance, heuristic):
    n, dim, m = 1500, 24, 8
    rows = _data(n, dim, 0x5EED0101, distance)
    st = O.DenseStorage(O.F32, distance, rows)
    g = O.Hnsw(st, m=m, ef_construct=64, use_heuristic=heuristic, seed=7)
    levels = np.array([g.point_level(i) for i in range(n)])
    assert levels.max() >= 1 and 0.55 < (levels == 0).mean() < 0.75      # P(round(-ln U / ln m) = 0) = 1 - m^-0.5 = 0.65 at m = 8
    for i in range(0, n, 13):
        for lv in range(levels[i] + 1):
            ln = g.links(i, lv)
            assert len(ln) <= (2 * m if lv == 0 else m)
            assert len(set(ln.tolist())) == len(ln) and i not in ln
            assert all(levels[j] >= lv for j in ln)
    ep_ids, ep_levels = g.entry_points()
    assert len(ep_ids) >= 1 and ep_levels[0] == levels.max()
    queries = O.synth(0x5EED0102, 0, 20, dim)
    exact = st.peek_top(queries, 10)
    got = g.search_dense(st, queries, 10, 64)
    for r in got:
        assert len(r) == 10 and np.all(np.diff(r["score"]) <= 0)
    assert _recall(got, exact) > (0.9 if heuristic else 0.8)


def test_build_is_deterministic_and_export_matches():
    n, dim = 800, 16
    st = O.DenseStorage(O.F32, O.EUCLID, _data(n, dim, 3, O.EUCLID))
    g1 = O.Hnsw(st, m=8, ef_construct=32, seed=5)
    g2 = O.Hnsw(st, m=8, ef_construct=32, seed=5)
    p1, p2 = g1.export_plain(), g2.export_plain()
    assert np.array_equal(p1.neighbors, p2.neighbors) and np.array_equal(p1.offsets, p2.offsets)
    assert np.array_equal(p1.reindex, p2.reindex) and np.array_equal(p1.level_offsets, p2.level_offsets)
    # plain view == builder links (graph_links/view.rs:211-218 offset_idx)
    assert int(p1.level_offsets[1]) == n
    for i in range(n):
        for lv in range(g1.point_level(i) + 1):
            assert p1.links(i, lv).tolist() == g1.links(i, lv).tolist()
    # reindex is a permutation ordered by descending level
    assert sorted(p1.reindex.tolist()) == list(range(n))
    lv = np.array([g1.point_level(i) for i in range(n)])
    order = np.argsort(p1.reindex)
    assert np.all(np.diff(lv[order]) <= 0)


def test_parallel_build_gives_a_searchable_graph():
    n, dim = 3000, 16
    st = O.DenseStorage(O.F32, O.COSINE, _data(n, dim, 11))
    g = O.Hnsw(st, m=8, ef_construct=64, seed=1, threads=4)
    queries = O.synth(12, 0, 16, dim)
    assert _recall(g.search_dense(st, queries, 10, 64), st.peek_top(queries, 10)) > 0.9


def test_search_respects_deleted_flags_and_ef_semantics():
    n, dim = 1200, 16
    rows = _data(n, dim, 21)
    deleted = np.zeros(n, dtype=bool)
    deleted[::3] = True
    st_all = O.DenseStorage(O.F32, O.COSINE, rows)
    g = O.Hnsw(st_all, m=8, ef_construct=64, seed=2)
    st_del = O.DenseStorage(O.F32, O.COSINE, rows, point_deleted=deleted)
    queries = O.synth(22, 0, 10, dim)
    got = g.search_dense(st_del, queries, 10, 128)
    for r in got:
        assert not deleted[r["idx"]].any()
    assert _recall(got, st_del.peek_top(queries, 10)) > 0.85
    # ef = max(ef, top) (graph_layers.rs:551): ef=1, top=10 still returns 10
    assert all(len(r) == 10 for r in g.search_dense(st_all, queries, 10, 1))


def test_sq_and_pq_scorers_drive_the_same_graph():
    n, dim = 1000, 32
    rows = _data(n, dim, 31, O.DOT)
    st = O.DenseStorage(O.F32, O.DOT, rows)
    g = O.Hnsw(st, m=8, ef_construct=64, seed=3)
    queries = O.synth(32, 0, 8, dim)
    exact = st.peek_top(queries, 10)
    sq = O.SqOracle(O.DOT, dim, (rows.max() - rows.min()) / 127.0, rows.min())
    sq.encode_rows(rows)
    r_sq = g.search_sq(st, sq, queries, 10, 64)
    assert _recall(r_sq, exact) > 0.4            # hnsw_quantized_search_test.rs:248-330 asks for > 40 %
    cen = O.PqOracle.train(rows, dim, 4, 256, iters=3)
    pq = O.PqOracle(O.DOT, dim, 4, cen)
    pq.encode(rows)
    r_pq = g.search_pq(st, pq, queries, 10, 64)
    assert _recall(r_pq, exact) > 0.4


def test_merge_matches_a_single_queue():
    rng = np.random.default_rng(5)
    n_lists, nq, k = 5, 7, 10
    lists = np.zeros((n_lists, nq, k), dtype=O.ScoredPointOffset)
    counts = rng.integers(0, k + 1, size=(n_lists, nq)).astype(np.uint32)
    base = (np.arange(n_lists) * 1000).astype(np.uint32)
    for l in range(n_lists):
        for q in range(nq):
            s = np.sort(rng.standard_normal(k).astype(np.float32))[::-1]
            lists[l, q]["score"] = s
            lists[l, q]["idx"] = rng.permutation(1000)[:k]
    got = O.merge_topk(lists, counts, k, base)
    for q in range(nq):
        allp = [(int(lists[l, q, i]["idx"]) + int(base[l]), float(lists[l, q, i]["score"]))
                for l in range(n_lists) for i in range(counts[l, q])]
        want = O.topk_push_all(allp, k)
        assert got[q]["idx"].tolist() == want["idx"].tolist()
        assert got[q]["score"].tolist() == want["score"].tolist()
    # duplicates of an id are dropped after the first occurrence (search_result_aggregator.rs:33-36)
    dup = np.zeros((2, 1, 3), dtype=O.ScoredPointOffset)
    dup[0, 0] = [(5, 3.0), (6, 2.0), (7, 1.0)]
    dup[1, 0] = [(5, 9.0), (8, 2.5), (9, 0.5)]
    r = O.merge_topk(dup, None, 3)[0]
    assert r["idx"].tolist() == [5, 8, 6] and r["score"].tolist() == [3.0, 2.5, 2.0]


def test_import_of_exported_plain_links_walks_identically():
    """qo_hnsw_import_plain (GraphLayers::load of the plain arrays) == the graph it came from, for the search."""
    n, dim = 1500, 24
    st = O.DenseStorage(O.F32, O.EUCLID, _data(n, dim, 31, O.EUCLID))
    g = O.Hnsw(st, m=8, ef_construct=48, seed=9)
    g2 = O.Hnsw.from_plain(g.export_plain(), n)
    assert [g2.point_level(i) for i in range(n)] == [g.point_level(i) for i in range(n)]
    for i in range(0, n, 17):
        for lv in range(g.point_level(i) + 1):
            assert g2.links(i, lv).tolist() == g.links(i, lv).tolist()
    queries = O.synth(32, 0, 12, dim)
    a, sa = g.search_dense(st, queries, 10, 64, with_stats=True)
    b, sb = g2.search_dense(st, queries, 10, 64, with_stats=True)
    assert sa == sb
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_discover_queries_walk_the_graph_like_the_reference_asks():
    """hnsw_discover_precision (lib/segment/tests/integration/hnsw_discover_test.rs:57-168): 5 000 points of 8 uniform coordinates, cosine, m = 16,
    ef_construct = 64, built single-threaded; 100 DiscoverQuery searches (a target + 1..2 context pairs), top 3 with ef 32 through the graph against the
    plain search of the same query: at most 5 of 100 may differ."""
    rng = np.random.default_rng(42)
    dim, n, top, ef, attempts, max_failures = 8, 5000, 3, 32, 100, 5
    rows = O.preprocess(O.COSINE, rng.random((n, dim), dtype=np.float32))
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    graph = O.Hnsw(st, m=16, ef_construct=64, seed=42)
    factory = O.ScorerFactory("dense", st)
    ids = np.arange(n, dtype=np.uint32)
    hits = 0
    for _ in range(attempts):
        pairs = int(rng.integers(1, 3))                                           # rng.random_range(1..MAX_EXAMPLE_PAIRS), MAX_EXAMPLE_PAIRS = 3
        examples = O.preprocess(O.COSINE, rng.random((1 + 2 * pairs, dim), dtype=np.float32))      # target, then (positive, negative) pairs
        scorer, keep = factory.custom(list(examples), 2, 1, pairs)
        walked, _ = graph.search_scorer(scorer, top, ef)
        scores = O.scorer_score_points(scorer, ids)
        plain = np.lexsort((ids, -scores.astype(np.float64)))[:top]
        same = walked["idx"].tolist() == plain.tolist() and np.array_equal(walked["score"].view(np.uint32), scores[plain].view(np.uint32))
        hits += int(same)
    assert attempts - hits <= max_failures, hits


def test_search_on_level_of_a_hand_made_graph():
    """test_search_on_level (graph_layers.rs:920-978): ten points of 8 coordinates, dot; point 0 links to 1..6 on level 0, nobody else has links; a walk from
    entry point 0 with ef 32 for the stored vector of point 7 meets exactly point 0 and its six links, every one with the score of (7, that point)."""
    rng = np.random.default_rng(42)
    n, dim = 10, 8
    rows = rng.random((n, dim), dtype=np.float32)
    st = O.DenseStorage(O.F32, O.DOT, rows)
    links0 = [[1, 2, 3, 4, 5, 6]] + [[] for _ in range(n - 1)]
    offsets = np.cumsum([0] + [len(x) for x in links0]).astype(np.uint64)
    plain = O.PlainLinks(8, 16, reindex=np.arange(n, dtype=np.uint32), level_offsets=np.array([0, n], dtype=np.uint64), offsets=offsets,
                         neighbors=np.array(sum(links0, []), dtype=np.uint32), ep_ids=[0], ep_levels=[0])
    graph = O.Hnsw.from_plain(plain, n)
    got = graph.search_dense(st, rows[7:8], 32, 32)[0]
    assert sorted(got["idx"].tolist()) == [0, 1, 2, 3, 4, 5, 6]
    want = st.score_points(rows[7:8], got["idx"])[0]
    assert np.array_equal(got["score"].view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("threads", [0, 4])
def test_every_point_has_an_inbound_link(threads):
    """test_graph_connectivity (hnsw_index/tests/test_graph_connectivity.rs:25-112): 1 000 points of 32 coordinates uniform in -1 .. 1, cosine, m = 16,
    ef_construct = 100, built on 4 threads there: no point of the level-0 graph is without an inbound link."""
    rng = np.random.default_rng(5)
    n, dim = 1000, 32
    rows = O.preprocess(O.COSINE, rng.uniform(-1.0, 1.0, (n, dim)).astype(np.float32))
    graph = O.Hnsw(O.DenseStorage(O.F32, O.COSINE, rows), m=16, ef_construct=100, seed=42, threads=threads)
    inbound = np.zeros(n, dtype=np.int64)
    for p in range(n):
        inbound[np.asarray(graph.links(p, 0), dtype=np.int64)] += 1
    assert (inbound > 0).all(), np.flatnonzero(inbound == 0)[:10]


@pytest.mark.parametrize("kind", ["pq", "tq"])
def test_multivector_build_over_rows_that_are_not_queries_falls_back_like_the_single_vector_build(kind):
    """`QuantizedMultivectorStorage::encode_internal_vector` (quantized_multivector_storage/mod.rs:458-470) is None as soon as one inner row's is - PQ and
    TurboQuant rows - so `FilteredScorer::new_internal` (point_scorer.rs:183-218) scores the searches of an insertion through the query scorer of the point's
    ORIGINAL multi-vector and only stored <-> stored pairs through score_internal_max_similarity.  Pinned against the path that already does this for single
    vectors: with ONE inner vector per point MaxSim is the plain score (0.0 + max(-inf, x) = x, bit for bit), so the multi-vector build must produce the graph
    of `Hnsw.build_pq` / `build_tq` link for link - which it does not if the searches go through score_internal (what the oracle did before round 4)."""
    n, dim, m, efc, seed = 400, 32, 6, 24, 9
    rng = np.random.default_rng(3)
    centers = rng.standard_normal((12, dim)).astype(np.float32) * 2
    rows = O.preprocess(O.DOT, (centers[rng.integers(12, size=n)] + rng.standard_normal((n, dim))).astype(np.float32))
    st = O.DenseStorage(O.F32, O.DOT, rows)
    offsets = np.arange(n + 1, dtype=np.uint64)
    if kind == "pq":
        cen = O.PqOracle.train(rows, dim, 4, 256, iters=3)
        q = O.PqOracle(O.DOT, dim, 4, cen)
        q.codes = q.encode(rows)
        single = O.Hnsw.build_pq(st, q, m=m, ef_construct=efc, seed=seed, entry_points_num=4)
    else:
        q = O.TqOracle(O.DOT, dim, O.TQ_BITS4)
        q.rows = q.encode_rows(rows)
        single = O.Hnsw.build_tq(st, q, m=m, ef_construct=efc, seed=seed, entry_points_num=4)
    multi = O.MultiOracle((kind, st, q), offsets).build(m=m, ef_construct=efc, seed=seed, entry_points_num=4)
    a, b = single.export_plain(), multi.export_plain()
    assert np.array_equal(a.reindex, b.reindex) and np.array_equal(a.offsets, b.offsets) and np.array_equal(a.neighbors, b.neighbors)
    assert a.ep_ids.tolist() == b.ep_ids.tolist()
    # ... and the internal-only build (every score stored <-> stored) is a different graph: the fallback is not vacuous
    s, keep = O.MultiOracle((kind, st, q), offsets).template()
    lo, hi = 0, 0
    for p in range(40, 60):
        for o in range(5):
            lo += O.MultiOracle((kind, st, q), offsets).score_internal(p, o) != np.float32(O.MultiOracle((kind, st, q), offsets).score_points([rows[p:p + 1]], [o])[0, 0])
            hi += 1
    assert lo > hi // 2          # a stored row scored as a query differs from its original's query scorer almost always


def test_traced_search_returns_the_plain_search_and_a_consistent_pop_sequence():
    """qo_hnsw_search_traced (round 5): the same lists as qo_hnsw_search + the candidates search_on_level popped AND expanded (graph_layers.rs:120-147).  The
    sequence never repeats a point, and every returned point is a popped point or a link of one (nothing enters `nearest` unscored; the level-0 entry is the
    first pop)."""
    import numpy as np
    import oracle_ffi as O
    n, dim, m, nq, top, ef = 3000, 24, 8, 16, 10, 48
    rows = O.preprocess(O.COSINE, O.synth(0x5EED0720, 0, n, dim))
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    g = O.Hnsw(st, m=m, ef_construct=48, seed=9, threads=0)
    queries = O.synth(0x5EED0721, 0, nq, dim)
    plain = g.search_dense(st, queries, top, ef)
    g.pops = []
    traced = g.search_dense(st, queries, top, ef)
    pops, g.pops = g.pops, None
    pl = g.export_plain()
    off, nb = np.asarray(pl.offsets), np.asarray(pl.neighbors)
    for a, b, p in zip(plain, traced, pops):
        assert a["idx"].tolist() == b["idx"].tolist() and np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32))
        assert len(p) >= 1 and len(set(p["idx"].tolist())) == len(p)
        reach = set(p["idx"].tolist())
        for c in p["idx"].tolist():
            reach.update(nb[int(off[c]):int(off[c + 1])].tolist())
        assert set(b["idx"].tolist()) <= reach

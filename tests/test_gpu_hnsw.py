"""GPU parity of the device-resident HNSW search (`qmx_hnsw_search`) against the CPU oracle's
restatement of `GraphLayers::search` (lib/segment/src/index/hnsw_index/graph_layers.rs:530-562) on
the SAME graph (built by the oracle, exported as plain GraphLinks).

Bars: f32 / SQ / PQ scorers are bit-exact, so with distinct scores the walk is the same walk:
identical id lists, identical score bits and the identical NUMBER of scored points (the reference's
hardware counter).  f16 scores differ in summation order (<= 1e-5), u8 scores tie (integers): there the
bar is the reference's own (`hnsw_quantized_search_test.rs`-style recall against exact search).
"""
import os

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
    return {O.COSINE: qa.Distance.Cosine, O.DOT: qa.Distance.Dot, O.EUCLID: qa.Distance.Euclid,
            O.MANHATTAN: qa.Distance.Manhattan}[d]


def _same(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g["idx"].tolist() == w["idx"].tolist()
        assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32))


def _same_modulo_ties(got, want):
    """Score lists bit-equal; ids equal wherever the score is unique in the list and above the last (boundary) score."""
    n_ids = 0
    for g, w in zip(got, want):
        assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32))
        sc = w["score"]
        for i in range(len(w)):
            if (sc == sc[i]).sum() == 1 and sc[i] != sc[-1]:
                assert g["idx"][i] == w["idx"][i]
                n_ids += 1
    return n_ids


_GRAPHS = {}


def _graph(distance, n, dim, m, seed):
    key = (distance, n, dim, m, seed)
    if key not in _GRAPHS:
        rows = O.preprocess(distance, O.synth(seed, 0, n, dim))
        st = O.DenseStorage(O.F32, distance, rows)
        g = O.Hnsw(st, m=m, ef_construct=64, seed=seed & 0xFF, threads=0)
        _GRAPHS[key] = (rows, st, g, g.export_plain())
    return _GRAPHS[key]


@pytest.mark.parametrize("distance", [O.COSINE, O.EUCLID, O.DOT, O.MANHATTAN])
@pytest.mark.parametrize("dim", [24, 48, 100])          # 24: SSE/scalar leaf (one lane per row); 48/100: AVX leaf (+ scalar tail)
def test_f32_search_is_the_reference_walk(qa, distance, dim):
    n, m, nq = 2500, 8, 40
    rows, st, g, plain = _graph(distance, n, dim, m, 0x5EED0300 + dim)
    queries = O.synth(0x5EED0301 + distance, 0, nq, dim)
    vs = qa.VectorStorage(rows, _dist(qa, distance))
    graph = qa.GraphLayers.from_plain(plain)
    scorer = qa.new_raw_scorer(queries, vs)
    for top, ef in [(10, 64), (10, 16), (5, 128), (10, 200), (64, 8), (1, 1)]:
        want, stats = g.search_dense(st, queries, top, ef, with_stats=True)
        got, scored = graph.search(top, ef, scorer, with_scored=True)
        _same(got, want)
        if distance == O.MANHATTAN:
            # the synthetic generator is integer-based (Irwin-Hall of 16-bit uniforms): L1 distances tie now and
            # then, and a candidate that ties with the lower bound is expanded or not depending on BinaryHeap
            # order in the reference (unpinned) -- same results, a handful of scored points apart
            assert abs(scored - sum(stats)) <= sum(stats) // 1000
        else:
            assert scored == sum(stats)


def test_deleted_points_and_entry_point_fallback(qa):
    n, dim, m, nq = 2000, 32, 8, 24
    rows, st_all, g, plain = _graph(O.COSINE, n, dim, m, 0x5EED0310)
    queries = O.synth(0x5EED0311, 0, nq, dim)
    rng = np.random.default_rng(5)
    deleted = rng.random(n) < 0.3
    ep_ids, _ = g.entry_points()
    deleted[ep_ids[0]] = True                      # the best entry point is gone: get_entry_point moves on
    vec_deleted = rng.random(n) < 0.05
    st_del = O.DenseStorage(O.F32, O.COSINE, rows, point_deleted=deleted, vec_deleted=vec_deleted)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    vs.set_deleted(deleted, vec_deleted)
    graph = qa.GraphLayers.from_plain(plain)
    scorer = qa.new_raw_scorer(queries, vs)
    want, stats = g.search_dense(st_del, queries, 10, 64, with_stats=True)
    got, scored = graph.search(10, 64, scorer, with_scored=True)
    _same(got, want)
    assert scored == sum(stats)
    for r in got:
        assert not deleted[r["idx"]].any() and not vec_deleted[r["idx"]].any()
    # everything deleted: no entry point -> empty result (graph_layers.rs:539-542)
    vs.set_deleted(np.ones(n, dtype=bool), None)
    assert all(len(r) == 0 for r in graph.search(10, 64, scorer))


def test_more_queries_than_slots_and_repeatability(qa):
    """Slots are reused across queries: the visited bitmap must come back clean (log and full clear)."""
    n, dim, m = 1500, 32, 8
    rows, st, g, plain = _graph(O.DOT, n, dim, m, 0x5EED0320)
    base = O.synth(0x5EED0321, 0, 50, dim)
    queries = np.tile(base, (120, 1))               # 6000 searches, 50 distinct
    vs = qa.VectorStorage(rows, qa.Distance.Dot)
    graph = qa.GraphLayers.from_plain(plain)
    want = g.search_dense(st, base, 10, 48)
    for cap in (None, 4):                         # "4": the log overflows at once -> whole-bitmap clear path
        if cap is None:
            qa.set_option("hnsw_log_cap", -1)
        else:
            qa.set_option("hnsw_log_cap", cap)
        try:
            scorer = qa.new_raw_scorer(queries, vs)
            for _ in range(2):                      # second launch reuses the same scratch
                got = graph.search(10, 48, scorer)
                _same(got[:50], want)
                for rep in range(1, 120):
                    for j in (0, 17, 49):
                        assert np.array_equal(got[rep * 50 + j], got[j])
        finally:
            qa.set_option("hnsw_log_cap", -1)


@pytest.mark.parametrize("distance", [O.DOT, O.EUCLID, O.MANHATTAN])
def test_sq_scorer_walk_bit_exact_and_rescoring(qa, distance):
    n, dim, m, nq = 3000, 96, 8, 32
    rows, st, g, plain = _graph(distance, n, dim, m, 0x5EED0330 + distance)
    queries = O.synth(0x5EED0331, 0, nq, dim)
    qpre = O.preprocess(distance, queries)
    quant = qa.ScalarQuantizer.from_min_max(rows, dim, _dist(qa, distance))
    osq = O.SqOracle(distance, dim, quant.alpha, quant.offset)
    sq_rows = osq.encode_rows(rows)
    enc = qa.EncodedVectorsU8(quant.encode(rows), quant)
    graph = qa.GraphLayers.from_plain(plain)
    scorer = qa.new_raw_scorer(queries, enc)
    want = g.search_sq(st, osq, qpre, 20, 64)
    got = graph.search(20, 64, scorer)
    if distance == O.MANHATTAN:
        # alpha * sum|q - v| over integer codes: scores tie, the walk may legitimately take another branch among equals
        # (unpinned in the reference).  Every returned pair must still be a true SQ score, bit-exact, and the
        # walks must be of the same quality.
        exact_sq = [set(np.argsort(-osq.score_points(qpre[i:i + 1], np.arange(n))[0], kind="stable")[:20].tolist()) for i in range(nq)]
        rg = sum(len(set(r["idx"].tolist()) & e) for r, e in zip(got, exact_sq))
        rw = sum(len(set(r["idx"].tolist()) & e) for r, e in zip(want, exact_sq))
        assert abs(rg - rw) <= 0.02 * 20 * nq
        for i, r in enumerate(got):
            w = osq.score_points(qpre[i:i + 1], r["idx"])[0]
            assert np.array_equal(r["score"].view(np.uint32), w.view(np.uint32)) and np.all(np.diff(r["score"]) <= 0)
    else:
        _same(got, want)
    assert sq_rows.shape[0] == n
    # hnsw/read_view/search.rs: oversampled quantized search, then rescoring with the original vectors
    vs = qa.VectorStorage(rows, _dist(qa, distance))
    raw = qa.new_raw_scorer(queries, vs)
    ids = np.zeros((nq, 20), dtype=np.uint32)
    cnt = np.zeros(nq, dtype=np.uint32)
    for i, r in enumerate(got):
        ids[i, :len(r)] = r["idx"]
        cnt[i] = len(r)
    res = raw.rescore(ids, 10, cnt)
    exact = st.peek_top(queries, 10)
    hit = sum(len(set(a["idx"].tolist()) & set(b["idx"].tolist())) for a, b in zip(res, exact))
    assert hit / (10 * nq) > 0.6      # hnsw_quantized_search_test.rs:248-330 asks for > 0.4
    for i, r in enumerate(res):                     # rescored scores are the exact f32 scores
        w = st.score_points(queries[i:i + 1], r["idx"])[0]
        assert np.array_equal(r["score"].view(np.uint32), w.view(np.uint32))


@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("distance,dim,chunk", [(O.DOT, 64, 4), (O.EUCLID, 96, 16), (O.COSINE, 70, 8), (O.MANHATTAN, 96, 16), (O.EUCLID, 72, 8), (O.DOT, 320, 16)])
def test_pq_scorer_walk_bit_exact(qa, distance, dim, chunk, direct):
    """direct: the walk without LUTs (option hnsw_pq_direct_walk, pq.hip HopPQDirect) - every LUT entry recomputed from the codebook where it is needed, in
    pq_lut_kernel's order: the same bits.  Codebooks it does not take (a ragged last chunk: 70 = 8 x 8 + 6) keep the LUT walk."""
    if direct:
        qa.set_option("hnsw_pq_direct_walk", 1)
    try:
        _pq_walk(qa, distance, dim, chunk, direct)
    finally:
        qa.set_option("hnsw_pq_direct_walk", -1)


def _pq_walk(qa, distance, dim, chunk, direct):
    n, m, nq = 3000, 8, 24
    rows, st, g, plain = _graph(distance, n, dim, m, 0x5EED0340 + dim)
    queries = O.synth(0x5EED0341, 0, nq, dim)
    qpre = O.preprocess(distance, queries)
    cen = O.PqOracle.train(rows[:2000], dim, chunk, 256, iters=3)
    opq = O.PqOracle(distance, dim, chunk, cen)
    codes = opq.encode(rows)
    quant = qa.ProductQuantizer(dim, _dist(qa, distance), chunk, cen)
    enc = qa.EncodedVectorsPQ(codes, quant)
    graph = qa.GraphLayers.from_plain(plain)
    scorer = qa.new_raw_scorer(queries, enc)
    # PQ scores tie now and then (sums of few LUT entries): compare the walks where the oracle's result has distinct scores
    want = g.search_pq(st, opq, qpre, 10, 64)
    got = graph.search(10, 64, scorer)
    kernel = qa._ffi.last_kernel(scorer._h)
    assert ("HopPQDirect" in kernel) == (direct and dim % chunk == 0), kernel
    n_cmp = 0
    for gq, wq in zip(got, want):
        assert np.array_equal(gq["score"].view(np.uint32), wq["score"].view(np.uint32))
        if len(np.unique(wq["score"])) == len(wq):
            assert gq["idx"].tolist() == wq["idx"].tolist()
            n_cmp += 1
    assert n_cmp >= nq // 2
    if direct:      # the wide beam (LDS list) and the walk that returns the expanded points take the same scorer
        want = g.search_pq(st, opq, qpre[:6], 10, 700)
        got = graph.search(10, 700, qa.new_raw_scorer(queries[:6], enc))
        for gq, wq in zip(got, want):
            assert np.array_equal(gq["score"].view(np.uint32), wq["score"].view(np.uint32))


@pytest.mark.parametrize("dtype", ["f16", "u8"])
def test_f16_u8_search_recall(qa, dtype):
    n, dim, m, nq = 3000, 64, 8, 32
    rng = np.random.default_rng(9)
    if dtype == "f16":
        raw = O.preprocess(O.COSINE, O.synth(0x5EED0350, 0, n, dim))
        stored = O.to_f16(raw)
        st = O.DenseStorage(O.F16, O.COSINE, stored)
        queries = O.synth(0x5EED0351, 0, nq, dim)
        vs = qa.VectorStorage(stored.view(np.float16), qa.Distance.Cosine, qa.VectorStorageDatatype.Float16)
    else:
        stored = rng.integers(0, 256, size=(n, dim)).astype(np.uint8)
        st = O.DenseStorage(O.U8, O.EUCLID, stored)
        queries = rng.integers(0, 256, size=(nq, dim)).astype(np.float32)
        vs = qa.VectorStorage(stored, qa.Distance.Euclid, qa.VectorStorageDatatype.Uint8)
    g = O.Hnsw(st, m=m, ef_construct=64, seed=3)
    graph = qa.GraphLayers.from_plain(g.export_plain())
    scorer = qa.new_raw_scorer(queries, vs)
    got = graph.search(10, 96, scorer)
    want = g.search_dense(st, queries, 10, 96)
    exact = st.peek_top(queries, 10)

    def recall(res):
        return sum(len(set(a["idx"].tolist()) & set(b["idx"].tolist())) for a, b in zip(res, exact)) / (10.0 * nq)
    assert recall(got) > 0.9 and abs(recall(got) - recall(want)) < 0.03
    for r in got:
        assert len(r) == 10 and np.all(np.diff(r["score"]) <= 0)


@pytest.mark.parametrize("kind", ["f32", "sq", "acorn"])
def test_wide_searches_use_the_lds_beam_and_stay_the_reference_walk(qa, kind):
    """max(top, ef) > 512: the list of the walk lives in LDS (hnsw.hpp Beam<0>) - the same walk, decision for decision: ids, score bits and the number
    of scored points equal the oracle's at ef 600 / 1000 / 2500 and top up to 700, through the f32 and SQ scorers and the ACORN expansion (custom queries,
    multi-vectors, the other quantized scorers: a wide case in their own walk tests)."""
    n, dim, m, nq = 6000, 48, 8, 12
    rows, st, g, plain = _graph(O.DOT, n, dim, m, 0x5EED0390)
    queries = O.synth(0x5EED0391, 0, nq, dim)
    vs = qa.VectorStorage(rows, qa.Distance.Dot)
    graph = qa.GraphLayers.from_plain(plain)
    cases = [(10, 600), (100, 1000), (700, 700), (10, 2500), (513, 16)]
    if kind == "f32":
        scorer = qa.new_raw_scorer(queries, vs)
        for top, ef in cases:
            want, stats = g.search_dense(st, queries, top, ef, with_stats=True)
            got, scored = graph.search(top, ef, scorer, with_scored=True)
            _same(got, want)
            assert scored == sum(stats)
        assert "hnsw_search_kernel" in qa._ffi.last_kernel(scorer._h)
    elif kind == "sq":
        quant = qa.ScalarQuantizer.from_min_max(rows, dim, qa.Distance.Dot)
        osq = O.SqOracle(O.DOT, dim, quant.alpha, quant.offset)
        osq.encode_rows(rows)
        scorer = qa.new_raw_scorer(queries, qa.EncodedVectorsU8(quant.encode(rows), quant))
        for top, ef in cases[:3]:
            want = g.search_sq(st, osq, queries, top, ef)
            _same_modulo_ties(graph.search(top, ef, scorer), want)
    elif kind == "acorn":
        rng = np.random.default_rng(5)
        allowed = rng.random(n) < 0.3
        scorer = qa.new_raw_scorer(queries, vs)
        scorer.set_filter(allowed)
        ost = O.DenseStorage(O.F32, O.DOT, rows, point_deleted=~allowed)
        g.algorithm = 1
        try:
            for top, ef in cases[:3]:
                want, stats = g.search_dense(ost, queries, top, ef, with_stats=True)
                got, scored = graph.search(top, ef, scorer, with_scored=True, acorn=True)
                _same(got, want)
                assert scored == sum(stats)
        finally:
            g.algorithm = 0


def test_tiny_graphs_and_argument_errors(qa):
    dim = 32
    rows = O.preprocess(O.COSINE, O.synth(1, 0, 3, dim))
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    g = O.Hnsw(st, m=4, ef_construct=8, seed=1)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    graph = qa.GraphLayers.from_plain(g.export_plain())
    queries = O.synth(2, 0, 5, dim)
    scorer = qa.new_raw_scorer(queries, vs)
    _same(graph.search(10, 4, scorer), g.search_dense(st, queries, 10, 4))     # fewer points than top
    _same(graph.search(10, 1000, scorer), g.search_dense(st, queries, 10, 1000))   # ef beyond the register beam: the LDS list
    with pytest.raises(qa.QmxError) as e:
        graph.search(10, 5000, scorer)                                          # ... which ends at 4096
    assert e.value.status == qa._ffi.ERR_NOT_SUPPORTED
    with pytest.raises(qa.QmxError) as e:
        graph.search(0, 10, scorer)
    assert e.value.status == qa._ffi.ERR_BAD_ARG
    assert all(len(r) == 3 for r in graph.search(10, 16, scorer, is_stopped=False))
    with pytest.raises(qa.QmxError) as e:
        graph.search(10, 16, scorer, is_stopped=True)
    assert e.value.status == qa._ffi.ERR_CANCELLED
    # a graph over more points than the segment holds is refused
    big = O.preprocess(O.COSINE, O.synth(3, 0, 50, dim))
    gb = O.Hnsw(O.DenseStorage(O.F32, O.COSINE, big), m=4, ef_construct=8, seed=1)
    graph_big = qa.GraphLayers.from_plain(gb.export_plain())
    with pytest.raises(qa.QmxError) as e:
        graph_big.search(5, 8, scorer)
    assert e.value.status == qa._ffi.ERR_OUT_OF_BOUNDS
    # corrupt links are rejected at create time
    p = g.export_plain()
    bad = p.offsets.copy()
    bad[1] = bad[-1] + 5
    with pytest.raises(qa.QmxError):
        qa.GraphLayers(p.m, p.m0, p.reindex, p.level_offsets, bad, p.neighbors, p.ep_ids, p.ep_levels)


def test_plain_links_file_ingestion(qa):
    """The reference's plain graph-links file (graph_links/serializer.rs) uploads to the same graph as the arrays."""
    n, dim, m, nq = 2000, 32, 8, 16
    rows, st, g, plain = _graph(O.COSINE, n, dim, m, 0x5EED0310)
    queries = O.synth(0x5EED0361, 0, nq, dim)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    scorer = qa.new_raw_scorer(queries, vs)
    from_file = qa.GraphLayers.from_plain_file(O.plain_links_file(plain), plain.m, plain.m0, plain.ep_ids, plain.ep_levels,
                                               plain.xp_ids, plain.xp_levels)
    assert from_file.n_points == n
    _same(from_file.search(10, 64, scorer), g.search_dense(st, queries, 10, 64))


def test_compressed_links_file_ingestion(qa):
    """The reference's COMPRESSED graph-links files (GraphLinksFormat::Compressed / CompressedWithVectors,
    graph_links/serializer.rs:23-243) upload through qmx_hnsw_create_from_file; the device walk equals the oracle's walk of
    the same (re-ordered: first level_m links ascending) lists."""
    from qdrant_amd.hnsw import decode_links_file
    n, dim, m, nq = 2000, 32, 8, 16
    rows, st, g, plain = _graph(O.COSINE, n, dim, m, 0x5EED0390)
    queries = O.synth(0x5EED0391, 0, nq, dim)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    scorer = qa.new_raw_scorer(queries, vs)
    rng = np.random.default_rng(1)
    files = [O.compressed_links_file(plain),
             O.compressed_links_file(plain, rng.integers(0, 256, (n, 16), dtype=np.uint8), rng.integers(0, 256, (n, 8), dtype=np.uint8), 8, 8)]
    for data in files:
        d = decode_links_file(data)
        twin = O.Hnsw.from_plain(O.PlainLinks(d.m, d.m0, d.reindex, d.level_offsets, d.offsets, d.neighbors, plain.ep_ids,
                                              plain.ep_levels, plain.xp_ids, plain.xp_levels), n)
        graph = qa.GraphLayers.from_file(data, plain.ep_ids, plain.ep_levels, plain.xp_ids, plain.xp_levels)
        assert (graph.n_points, graph.m, graph.m0) == (n, plain.m, plain.m0)
        _same(graph.search(10, 64, scorer), twin.search_dense(st, queries, 10, 64))


def test_payload_filter_bitmap_brute_force_and_walk(qa):
    """ScorerFilters' payload filter as an allow bitmap (qmx_query_set_filter): for the oracle a rejected point is a
    deleted point (`check_vector` = not deleted AND filter), so both must agree — brute force and filtered walk."""
    n, dim, m, nq = 3000, 32, 8, 20
    rows, st_all, g, plain = _graph(O.COSINE, n, dim, m, 0x5EED0370)
    queries = O.synth(0x5EED0371, 0, nq, dim)
    rng = np.random.default_rng(8)
    allowed = rng.random(n) < 0.4
    deleted = rng.random(n) < 0.1
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    vs.set_deleted(deleted, None)
    st_f = O.DenseStorage(O.F32, O.COSINE, rows, point_deleted=deleted | ~allowed)
    s = qa.BatchFilteredSearcher(queries, vs, 10)
    s.scorer.set_filter(allowed)
    _same(s.peek_top_all(), st_f.peek_top(queries, 10))
    graph = qa.GraphLayers.from_plain(plain)
    want, stats = g.search_dense(st_f, queries, 10, 64, with_stats=True)
    got, scored = graph.search(10, 64, s.scorer, with_scored=True)
    _same(got, want)
    assert scored == sum(stats)
    for r in got:
        assert allowed[r["idx"]].all() and not deleted[r["idx"]].any()
    s.scorer.set_filter(None)                                   # cleared: back to the deleted flags alone
    _same(s.peek_top_all(), O.DenseStorage(O.F32, O.COSINE, rows, point_deleted=deleted).peek_top(queries, 10))


@pytest.mark.parametrize("selectivity", [0.03, 0.15, 0.5, 1.0])
@pytest.mark.parametrize("m", [8, 16, 32, 48, 64])         # m0 = 64: up to 4160 ids in one scoring batch; m0 = 96 / 128: scored in several in-order batches
def test_acorn_walk_is_the_reference_walk(qa, selectivity, m):
    """SearchAlgorithm::Acorn (search_on_level_acorn, graph_layers.rs:154-243) on device == the oracle's restatement on the same
    graph: ids, score bits and the number of scored points, for filters from 3 % to 100 % of the points (+ deleted points).
    The filtered plain walk strands on low selectivity; ACORN explores through rejected points."""
    n, dim, nq, top, ef = 4000, 32, 24, 10, 48
    rows, st_all, g, plain = _graph(O.COSINE, n, dim, m, 0x5EED0390 + m)
    queries = O.synth(0x5EED0391, 0, nq, dim)
    rng = np.random.default_rng(int(selectivity * 100) + m)
    allowed = rng.random(n) < selectivity
    deleted = rng.random(n) < 0.05
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    vs.set_deleted(deleted, None)
    st_f = O.DenseStorage(O.F32, O.COSINE, rows, point_deleted=deleted | ~allowed)
    scorer = qa.new_raw_scorer(queries, vs)
    scorer.set_filter(allowed)
    graph = qa.GraphLayers.from_plain(plain)
    g.algorithm = 1
    try:
        want, stats = g.search_dense(st_f, queries, top, ef, with_stats=True)
    finally:
        g.algorithm = 0
    got, scored = graph.search(top, ef, scorer, with_scored=True, acorn=True)
    _same(got, want)
    assert scored == sum(stats)
    for r in got:
        assert allowed[r["idx"]].all() and not deleted[r["idx"]].any()
    # the visited bitmaps come back clean: a second run (plain walk, then ACORN again) gives the same lists
    plain_walk = graph.search(top, ef, scorer)
    again = graph.search(top, ef, scorer, acorn=True)
    _same(again, want)
    if selectivity <= 0.15:      # what ACORN is for: it finds more of the filtered neighbours than the plain filtered walk
        exact = st_f.peek_top(queries, top)
        hit = lambda res: sum(len(set(a["idx"].tolist()) & set(b["idx"].tolist())) for a, b in zip(res, exact))   # noqa: E731
        assert hit(got) >= hit(plain_walk)


def test_acorn_with_sq_scorer_and_log_overflow(qa):
    n, dim, m, nq = 3000, 96, 8, 16
    rows, st, g, plain = _graph(O.DOT, n, dim, m, 0x5EED0330 + O.DOT)
    queries = O.synth(0x5EED0392, 0, nq, dim)
    qpre = O.preprocess(O.DOT, queries)
    quant = qa.ScalarQuantizer.from_min_max(rows, dim, qa.Distance.Dot)
    osq = O.SqOracle(O.DOT, dim, quant.alpha, quant.offset)
    osq.encode_rows(rows)
    enc = qa.EncodedVectorsU8(quant.encode(rows), quant)
    rng = np.random.default_rng(5)
    allowed = rng.random(n) < 0.2
    st_f = O.DenseStorage(O.F32, O.DOT, rows, point_deleted=~allowed)
    scorer = qa.new_raw_scorer(queries, enc)
    scorer.set_filter(allowed)
    graph = qa.GraphLayers.from_plain(plain)
    g.algorithm = 1
    try:
        want = g.search_sq(st_f, osq, qpre, 10, 40)
    finally:
        g.algorithm = 0
    _same(graph.search(10, 40, scorer, acorn=True), want)
    # the one-call pipeline with the request's SearchAlgorithm: ACORN walk (oversampled) + rescoring == the two calls by hand
    vs = qa.VectorStorage(rows, qa.Distance.Dot)
    raw = qa.new_raw_scorer(queries, vs)
    raw.set_filter(allowed)
    fused = qa.search_quantized(scorer, raw, 10, oversampling=2.0, rescore=True, graph=graph, hnsw_ef=40, acorn=True)
    walk = graph.search(20, 40, scorer, acorn=True)
    ids = np.zeros((nq, 20), dtype=np.uint32)
    cnt = np.zeros(nq, dtype=np.uint32)
    for i, r in enumerate(walk):
        ids[i, :len(r)] = r["idx"]
        cnt[i] = len(r)
    for a_, b_ in zip(fused, raw.rescore(ids, 10, cnt)):
        assert np.array_equal(a_, b_)
    qa.set_option("hnsw_log_cap", 8)                      # the visited log overflows: whole-bitmap clear of both lists
    try:
        _same(graph.search(10, 40, scorer, acorn=True), want)
        _same(graph.search(10, 40, scorer, acorn=True), want)
    finally:
        qa.set_option("hnsw_log_cap", -1)


def test_walk_survives_a_graph_whose_levels_are_inconsistent(qa):
    """a decodable links file can name, on level L, a node without a slot on level L, or an entry point with a level it does not
    have.  Host-visible arrays are refused by qmx_hnsw_create (tests/test_oracle_links.py); arrays handed over as DEVICE memory are not
    inspected, so the kernel itself must bound the slot: the search returns (some) valid result instead of faulting."""
    import ctypes as C
    import torch
    from qdrant_amd import _ffi as F
    n, dim, m, nq = 3000, 32, 8, 64
    rows, st, g, plain = _graph(O.DOT, n, dim, m, 0x5EED0390)
    lo, off, re_ = np.asarray(plain.level_offsets), np.asarray(plain.offsets), np.asarray(plain.reindex)
    assert len(lo) - 1 >= 2
    top_level = len(lo) - 2
    size1 = int(lo[2] - lo[1])
    low_nodes = np.flatnonzero(re_ >= size1)                       # points that exist on level 0 only
    nb = np.array(plain.neighbors, dtype=np.uint32)
    rng = np.random.default_rng(9)
    for s in range(int(lo[1]), int(lo[-1])):                       # every upper-level list gets one link to a level-0-only node
        if off[s + 1] > off[s]:
            nb[int(off[s])] = low_nodes[int(rng.integers(0, len(low_nodes)))]
    dev = torch.device("cuda", 0)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(np.int64 if dt == np.uint64 else np.int32)).to(dev)  # noqa: E731
    keep = [t(np.asarray(plain.reindex, dtype=np.uint32), np.uint32), t(np.asarray(lo, dtype=np.uint64), np.uint64),
            t(np.asarray(off, dtype=np.uint64), np.uint64), t(nb, np.uint32),
            t(np.array([int(low_nodes[0])], dtype=np.uint32), np.uint32), t(np.array([top_level], dtype=np.uint32), np.uint32)]
    d = F.HnswDesc()
    d.m, d.m0, d.n_points, d.n_levels = plain.m, plain.m0, n, len(lo) - 1
    d.reindex, d.level_offsets, d.offsets, d.n_offsets = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), len(off)
    d.neighbors, d.n_neighbors = keep[3].data_ptr(), len(nb)
    d.entry_point_ids, d.entry_point_levels, d.n_entry_points = keep[4].data_ptr(), keep[5].data_ptr(), 1   # an entry point far above its level
    h = C.c_void_p()
    F.check(F.lib().qmx_hnsw_create(C.byref(d), C.byref(h)))
    try:
        vs = qa.VectorStorage(rows, qa.Distance.Dot)
        scorer = qa.new_raw_scorer(O.synth(0x5EED0391, 0, nq, dim), vs)
        out = np.zeros((nq, 10), dtype=O.ScoredPointOffset)
        cnt = np.zeros(nq, dtype=np.uint32)
        for acorn in (False, True):
            fn = F.lib().qmx_hnsw_search_acorn if acorn else F.lib().qmx_hnsw_search
            F.check(fn(h, scorer._h, 10, 64, F.ptr(out), F.ptr(cnt), None, None))
            assert (cnt >= 1).all() and (cnt <= 10).all()
            for i in range(nq):
                assert (out["idx"][i, :cnt[i]] < n).all()
        torch.cuda.synchronize()
    finally:
        F.lib().qmx_hnsw_destroy(h)


def test_equal_scores_explain_every_oracle_walk_difference(qa):
    """at 1 M rows the oracle's walk and the device's returned different ids for 1 of 64 (f32) and 4 of 64 (SQ) queries
    with IDENTICAL score bits.  profiles/r2_hnsw_1m_oracle_walk_ties.jsonl has every such query of a 256-query re-run: each differing
    position holds two points whose scores to the query are bit-identical (SQ: quantized scores collide often; f32: two of ~250 same-cluster
    rows 6e-8 apart, once in 256 queries) - the device orders equal scores by ascending id, the reference by BinaryHeap insertion history
    (unpinned).  Here ties are forced: every row exists twice, so EVERY score ties.  The device's lists must carry the oracle's score bits
    and differ from its ids only inside runs of equal scores; among equal scores the device returns ascending ids."""
    n_half, dim, m, nq = 1500, 32, 8, 40
    base = O.preprocess(O.DOT, O.synth(0x5EED03A0, 0, n_half, dim))
    rows = np.concatenate([base, base])                        # row i == row i + n_half
    st = O.DenseStorage(O.F32, O.DOT, rows)
    g = O.Hnsw(st, m=m, ef_construct=64, seed=3, threads=0)
    queries = O.synth(0x5EED03A1, 0, nq, dim)
    vs = qa.VectorStorage(rows, qa.Distance.Dot)
    graph = qa.GraphLayers.from_plain(g.export_plain())
    got = graph.search(10, 64, qa.new_raw_scorer(queries, vs))
    want = g.search_dense(st, queries, 10, 64)
    n_pairs = 0
    for gq, wq in zip(got, want):
        assert np.array_equal(gq["score"].view(np.uint32), wq["score"].view(np.uint32))
        sc = gq["score"]
        for i in range(len(gq) - 1):
            if sc[i] == sc[i + 1]:
                assert gq["idx"][i] < gq["idx"][i + 1]            # equal scores: ascending id on the device
                n_pairs += 1
        # ids agree with the oracle as multisets of (score, id mod n_half): the same points up to the choice among equals
        assert sorted(zip(sc.tolist(), (gq["idx"] % n_half).tolist()))[:-1] == sorted(zip(wq["score"].tolist(), (wq["idx"] % n_half).tolist()))[:-1] or \
            sorted(zip(sc.tolist(), (gq["idx"] % n_half).tolist())) == sorted(zip(wq["score"].tolist(), (wq["idx"] % n_half).tolist()))
    assert n_pairs > nq                                        # the ties were really there


# ---------------------------------------------------------------------------------------------------------------------------------
# GraphLayers::search_with_vectors (graph_layers.rs:564-596): graphs with inline storage - the walk is steered by the quantized link
# vectors, every popped candidate is scored on its full base vector, the result comes from the base scores
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,distance,dim", [("sq", O.COSINE, 96), ("sq", O.DOT, 128), ("sq", O.EUCLID, 64), ("pq", O.DOT, 64), ("bq", O.COSINE, 256)])
def test_search_with_vectors_equals_the_oracle(qa, kind, distance, dim):
    n, m, nq = 4000, 8, 32
    rng = np.random.default_rng(dim + distance)
    centers = rng.standard_normal((40, dim)).astype(np.float32) * 1.5
    rows = O.preprocess(distance, (centers[rng.integers(40, size=n)] + rng.standard_normal((n, dim))).astype(np.float32))
    st = O.DenseStorage(O.F32, distance, rows)
    g = O.Hnsw(st, m=m, ef_construct=48, seed=11)
    graph = qa.GraphLayers.from_plain(g.export_plain())
    queries = (centers[rng.integers(40, size=nq)] + rng.standard_normal((nq, dim))).astype(np.float32)
    vs = qa.VectorStorage(rows, _dist(qa, distance))
    if kind == "sq":
        quant = qa.ScalarQuantizer.from_min_max(rows, dim, _dist(qa, distance))
        oq = O.SqOracle(distance, dim, quant.alpha, quant.offset)
        oq.rows = oq.encode_rows(rows)
        qs = qa.EncodedVectorsU8(quant.encode(rows), quant)
    elif kind == "pq":
        cen = O.PqOracle.train(rows[:2000], dim, 8, 256, iters=3)
        oq = O.PqOracle(distance, dim, 8, cen)
        oq.encode(rows)
        quant = qa.ProductQuantizer(dim, _dist(qa, distance), 8, cen)
        qs = qa.EncodedVectorsPQ(quant.encode(rows), quant)
    else:
        quant = qa.BinaryQuantizer(dim, _dist(qa, distance))
        oq = O.BqOracle(distance, dim)
        oq.rows = oq.encode_rows(rows)
        qs = qa.EncodedVectorsBin(quant.encode(rows), quant)
    links_scorer, base_scorer = qa.new_raw_scorer(queries, qs), qa.new_raw_scorer(queries, vs)
    exact = st.peek_top(queries, 10)
    for top, ef in ((10, 64), (5, 16), (10, 200)):
        want, n_links, n_base = g.search_with_vectors(st, (kind, oq), queries, top, ef)
        got, scored = graph.search_with_vectors(top, ef, links_scorer, base_scorer, with_scored=True)
        if kind == "bq":
            # integer-valued link scores tie in the beam (BinaryHeap order among equals is unpinned): the base scores returned must
            # be true scores, sorted, and the result as good as the oracle's
            for i, gq in enumerate(got):
                assert np.all(np.diff(gq["score"]) <= 0)
                assert np.array_equal(st.score_points(queries[i:i + 1], gq["idx"])[0].view(np.uint32), gq["score"].view(np.uint32))
            sc_got = np.mean([gq["score"][: min(len(gq), len(wq))].mean() for gq, wq in zip(got, want)])
            sc_want = np.mean([wq["score"][: min(len(gq), len(wq))].mean() for gq, wq in zip(got, want)])
            assert sc_got >= sc_want - 0.01 * abs(sc_want)
            continue
        _same(got, want)
        # (a candidate whose link score TIES with the lower bound is popped-and-expanded or not depending on BinaryHeap order: quantized scores
        # tie now and then; same results, a handful of scored vectors apart)
        assert abs(scored - (sum(n_links) + sum(n_base))) <= max(2, (sum(n_links) + sum(n_base)) // 1000)
    # what the fused rescoring buys: the walk's quantized order is replaced by the exact base scores of everything it expanded
    got = graph.search_with_vectors(10, 128, links_scorer, base_scorer)
    plain_walk = graph.search(10, 128, links_scorer)
    def recall(res):
        return np.mean([len(set(r["idx"].tolist()) & set(e["idx"].tolist())) / 10 for r, e in zip(res, exact)])
    assert recall(got) >= recall(plain_walk) - 0.02


@pytest.mark.parametrize("distance,dim,chunk", [(O.DOT, 768, 8), (O.EUCLID, 384, 4), (O.COSINE, 1536, 16)])
def test_pq_walk_hop_prefilter_is_the_same_walk(qa, distance, dim, chunk):
    """HopPQ::prefilter (round 5): a hop's candidates meet an 8-bit image of the search's LUT in LDS first, and those whose upper bound stays below the beam's
    worst score are dropped without their exact score.  The reference would have rejected them on the exact score, so the walk must not change at all: the same
    lists (ids and score bits), the same pop sequence, the same number of scored points as with the option off - and the oracle's walk of the same graph.
    m = 96 chunks (a 96 KiB LUT read through L2): the shape of BASELINE's C4."""
    n, m, nq = 4000, 8, 24
    rows, st, g, plain = _graph(distance, n, dim, m, 0x5EED03C0 + dim)
    queries = O.synth(0x5EED03C1, 0, nq, dim)
    qpre = O.preprocess(distance, queries)
    cen = O.PqOracle.train(rows[:2000], dim, chunk, 256, iters=2)
    opq = O.PqOracle(distance, dim, chunk, cen)
    assert opq.m == 96
    codes = opq.encode(rows)
    quant = qa.ProductQuantizer(dim, _dist(qa, distance), chunk, cen)
    graph = qa.GraphLayers.from_plain(plain)
    scorer = qa.new_raw_scorer(queries, qa.EncodedVectorsPQ(codes, quant))
    for top, ef in ((10, 64), (20, 128), (5, 600)):
        (got, pops), (_, scored) = graph.search_traced(top, ef, scorer), graph.search(top, ef, scorer, with_scored=True)
        assert "HopPQ," in qa._ffi.last_kernel(scorer._h)
        qa.set_option("hnsw_no_pq_prefilter", 1)
        try:
            (plain_lists, plain_pops), (_, plain_scored) = graph.search_traced(top, ef, scorer), graph.search(top, ef, scorer, with_scored=True)
        finally:
            qa.set_option("hnsw_no_pq_prefilter", -1)
        assert scored == plain_scored
        for a, b, pa, pb in zip(got, plain_lists, pops, plain_pops):
            assert a["idx"].tolist() == b["idx"].tolist() and np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32))
            assert pa["idx"].tolist() == pb["idx"].tolist() and np.array_equal(pa["score"].view(np.uint32), pb["score"].view(np.uint32))
        want, stats = g.search_pq(st, opq, qpre, top, ef, with_stats=True)
        assert scored == sum(stats)
        for a, w in zip(got, want):
            assert np.array_equal(a["score"].view(np.uint32), w["score"].view(np.uint32))
            if len(np.unique(w["score"])) == len(w):
                assert a["idx"].tolist() == w["idx"].tolist()


@pytest.mark.parametrize("lut_mfma", [False, True])
@pytest.mark.parametrize("distance,dim,chunk", [(O.DOT, 768, 8), (O.COSINE, 1536, 16), (O.EUCLID, 384, 4)])
def test_lut_free_pq_walk_with_the_hop_prefilter_is_the_same_walk(qa, distance, dim, chunk, lut_mfma):
    """Round 6: the LUT-free walk (option hnsw_pq_direct_walk, HopPQDirect: entries recomputed from the codebook in pq_lut_kernel's order) prefilters its hops
    on the 8-bit image of the EXACT-ORDER LUT - also when the batch's own LUTs came from the matrix cores (lut_mfma: exact-order LUTs are then made for the
    image alone).  Same walk as with the prefilter off: lists, pop sequences, scored points; and the oracle's walk (exact-order scores) of the same graph."""
    n, m, nq = 4000, 8, 24
    rows, st, g, plain = _graph(distance, n, dim, m, 0x5EED03D0 + dim)
    queries = O.synth(0x5EED03D1, 0, nq, dim)
    qpre = O.preprocess(distance, queries)
    cen = O.PqOracle.train(rows[:2000], dim, chunk, 256, iters=2)
    opq = O.PqOracle(distance, dim, chunk, cen)
    codes = opq.encode(rows)
    quant = qa.ProductQuantizer(dim, _dist(qa, distance), chunk, cen, lut_mfma=lut_mfma)
    graph = qa.GraphLayers.from_plain(plain)
    scorer = qa.new_raw_scorer(queries, qa.EncodedVectorsPQ(codes, quant))
    qa.set_option("hnsw_pq_direct_walk", 1)
    try:
        for top, ef in ((10, 64), (20, 128)):
            (got, pops), (_, scored) = graph.search_traced(top, ef, scorer), graph.search(top, ef, scorer, with_scored=True)
            assert "HopPQDirect" in qa._ffi.last_kernel(scorer._h)
            assert graph.counters.prefilter_candidates > graph.counters.verified_rows > 0          # the prefilter ran and dropped candidates
            qa.set_option("hnsw_no_pq_prefilter", 1)
            try:
                (plain_lists, plain_pops), (_, plain_scored) = graph.search_traced(top, ef, scorer), graph.search(top, ef, scorer, with_scored=True)
                assert graph.counters.prefilter_candidates == 0
            finally:
                qa.set_option("hnsw_no_pq_prefilter", -1)
            assert scored == plain_scored
            for a, b, pa, pb in zip(got, plain_lists, pops, plain_pops):
                assert a["idx"].tolist() == b["idx"].tolist() and np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32))
                assert pa["idx"].tolist() == pb["idx"].tolist() and np.array_equal(pa["score"].view(np.uint32), pb["score"].view(np.uint32))
            want, stats = g.search_pq(st, opq, qpre, top, ef, with_stats=True)
            assert scored == sum(stats)
            for a, w in zip(got, want):
                assert np.array_equal(a["score"].view(np.uint32), w["score"].view(np.uint32))
                if len(np.unique(w["score"])) == len(w):
                    assert a["idx"].tolist() == w["idx"].tolist()
    finally:
        qa.set_option("hnsw_pq_direct_walk", -1)


def test_lds_visited_table_answers_like_the_bitmap_with_full_buckets_and_over_long_lists(qa):
    """hnsw.hpp LdsVisited against the per-slot HBM bitmap alone (option hnsw_no_lds_visited), where the table works hardest: searches wide enough to fill its
    buckets (24 000 points, ef up to 3 000: ids sharing their low ten bits overflow into the bitmap) over a graph whose level-0 lists are LONGER than the m0 it
    declares (the links behind the limit are taken back out of the set: the slot is marked, never emptied, so a bucket that was full stays full).  Same lists,
    same score bits, same number of scored points, search after search on the same slots."""
    import types
    n, dim, m, nq = 24000, 16, 16, 48
    rows, st, g, plain = _graph(O.DOT, n, dim, m, 0x5EED03D0)
    short = types.SimpleNamespace(**{k: getattr(plain, k) for k in ("reindex", "level_offsets", "offsets", "neighbors", "ep_ids", "ep_levels", "xp_ids", "xp_levels")})
    short.m, short.m0 = 6, 10                                     # the lists hold up to 32 links on level 0, up to 16 above
    assert int(np.max(np.diff(np.asarray(plain.offsets)[:n + 1]))) > short.m0
    vs = qa.VectorStorage(rows, qa.Distance.Dot)
    scorer = qa.new_raw_scorer(O.synth(0x5EED03D1, 0, nq, dim), vs)
    for graph in (qa.GraphLayers.from_plain(short), qa.GraphLayers.from_plain(plain)):
        for top, ef in ((10, 64), (50, 500), (100, 3000)):
            for _ in range(2):                                    # (the second round runs on slots the first one left behind)
                got, scored = graph.search(top, ef, scorer, with_scored=True)
                qa.set_option("hnsw_no_lds_visited", 1)
                try:
                    want, want_scored = graph.search(top, ef, scorer, with_scored=True)
                finally:
                    qa.set_option("hnsw_no_lds_visited", -1)
                assert scored == want_scored
                _same(got, want)

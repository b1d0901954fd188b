"""The 128-query pass over 4-bit TurboQuant blocks (qdrant_amd/csrc/scan_tq4w.hip): brute-force top-k of 33 and more queries over a block of 2^18 rows
and more decodes the codes once per pass, in the registers of the lanes whose matrix-core operands they are, and multiplies 128 queries against them:
both digits of them (exact integers in the pass), or - option tq_wide_high_digit - their HIGH digits only (the pass's scores are within a stated band of
the exact ones; what could hide a result row is re-scored exactly by the pair kernel).  Either way the lists must be the 32-query scan's and the
oracle's (oracle/qdrant_oracle_tq.c) - ids, score bits, tie order.  A query whose candidate lists overflow (masses of equal scores) takes the 32-query
scan alone (qmx_counters.fallback_queries)."""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu

N = 262_144 + 1_111      # >= 2^18 rows: the wide pass applies; not a multiple of the 256-row tile


@pytest.fixture(scope="module")
def qa():
    import qdrant_amd
    assert qdrant_amd.device_count() >= 1
    return qdrant_amd


def _dist(qa, d):
    return {O.COSINE: qa.Distance.Cosine, O.DOT: qa.Distance.Dot, O.EUCLID: qa.Distance.Euclid}[d]


def _same(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g["idx"].tolist() == w["idx"].tolist()
        assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32))


def _kernel(qa, searcher):
    return qa._ffi.last_kernel(searcher.scorer._h)


def _narrow(qa, queries, st, top):
    qa.set_option("tq_wide_min_queries", 0)
    try:
        s = qa.BatchFilteredSearcher(queries, st, top)
        res = s.peek_top_all()
        assert "scan_tq4w_kernel" not in _kernel(qa, s)
        return res
    finally:
        qa.set_option("tq_wide_min_queries", -1)


_CACHE = {}


def _segment(qa, dist, dim, n, seed, plus=False, ties=0):
    key = (dist, dim, n, seed, plus, ties)
    if key in _CACHE:
        return _CACHE[key]
    _CACHE.clear()
    rng = np.random.default_rng(seed)
    if ties:          # a handful of distinct rows, repeated: every score occurs n / ties times
        base = rng.uniform(-1.0, 1.0, (ties, dim)).astype(np.float32)
        vecs = base[rng.integers(0, ties, n)]
    else:
        centres = rng.standard_normal((64, dim)).astype(np.float32)
        vecs = (centres[rng.integers(0, 64, n)] * rng.uniform(0.5, 2.0, (n, 1)) + 0.5 * rng.standard_normal((n, dim))).astype(np.float32)
    vecs = O.preprocess(dist, vecs)
    otq = O.TqOracle(dist, dim, O.TQ_BITS4)
    quant = qa.TurboQuantizer(dim, _dist(qa, dist), O.TQ_BITS4)
    rows = quant.encode(vecs)                                 # (byte-exact against the oracle's encoder: tests/test_gpu_tq.py)
    assert np.array_equal(rows[:64], otq.encode_rows(vecs[:64]))
    st = qa.EncodedVectorsTQ(rows, quant)
    _CACHE[key] = (vecs, otq, rows, st)
    return _CACHE[key]


def _oracle_top(otq, rows, qpre, n, top, dead=None):
    otq.rows = rows
    sc = otq.score_points(qpre, np.arange(n, dtype=np.uint32))
    out = []
    for i in range(len(qpre)):
        s = sc[i].copy()
        if dead is not None:
            s[dead] = -np.inf
        order = np.lexsort((np.arange(n), -s.astype(np.float64)))[:top]      # descending score, ties -> lower id
        out.append((order, s[order]))
    return out


@pytest.mark.parametrize("nq,top", [(33, 10), (128, 10), (150, 1), (300, 64)])
@pytest.mark.parametrize("dist,dim", [(O.DOT, 256), (O.COSINE, 768), (O.EUCLID, 768), (O.EUCLID, 1536)])
def test_tq_wide_pass_returns_the_narrow_scan_and_the_oracle(qa, dist, dim, nq, top):
    n = N
    vecs, otq, rows, st = _segment(qa, dist, dim, n, seed=dim * 3 + dist)
    rng = np.random.default_rng(nq * 7 + top)
    queries = (vecs[rng.integers(0, n, nq)] + 0.3 * rng.standard_normal((nq, dim))).astype(np.float32)
    queries[nq // 2] = 0.0                                    # a zero query: every score equal
    s = qa.BatchFilteredSearcher(queries, st, top)
    got = s.peek_top_all()
    assert "scan_tq4w_kernel" in _kernel(qa, s), _kernel(qa, s)
    c = s.counters
    assert c.prefilter_queries == nq and c.prefilter_candidates >= c.verified_rows >= top * (nq - c.fallback_queries)
    assert c.fallback_queries <= 1                            # (the zero query, if its bound cannot be derived)
    _same(got, _narrow(qa, queries, st, top))
    qa.set_option("tq_wide_high_digit", 1)                    # ... and the pass over the high digits alone
    try:
        s2 = qa.BatchFilteredSearcher(queries, st, top)
        got2 = s2.peek_top_all()
        assert "scan_tq4w_kernel<true>" in _kernel(qa, s2), _kernel(qa, s2)
    finally:
        qa.set_option("tq_wide_high_digit", -1)
    assert "scan_tq4w_kernel<false>" in _kernel(qa, s)
    _same(got2, got)
    k = 2
    want = _oracle_top(otq, rows, O.preprocess(dist, queries[:k]), n, top)
    for g, (ids, sc) in zip(got[:k], want):
        assert g["idx"].tolist() == ids.tolist()
        assert np.array_equal(g["score"].view(np.uint32), sc.view(np.uint32))


@pytest.mark.parametrize("dist", [O.DOT, O.EUCLID])
def test_tq_wide_pass_with_deleted_rows_and_a_filter(qa, dist):
    n, dim, nq, top = N, 256, 140, 10
    vecs, otq, rows, st = _segment(qa, dist, dim, n, seed=dim * 3 + dist)
    rng = np.random.default_rng(5)
    queries = (vecs[rng.integers(0, n, nq)] + 0.3 * rng.standard_normal((nq, dim))).astype(np.float32)
    deleted = rng.random(n) < 0.3
    deleted[vecs.shape[0] - 300:] = True                      # the last tile: all deleted
    st.set_deleted(deleted)
    try:
        s = qa.BatchFilteredSearcher(queries, st, top)
        got = s.peek_top_all()
        assert "scan_tq4w_kernel" in _kernel(qa, s)
        _same(got, _narrow(qa, queries, st, top))
        want = _oracle_top(otq, rows, O.preprocess(dist, queries[:3]), n, top, dead=deleted)
        for g, (ids, sc) in zip(got[:3], want):
            assert g["idx"].tolist() == ids.tolist() and not deleted[g["idx"]].any()
            assert np.array_equal(g["score"].view(np.uint32), sc.view(np.uint32))
    finally:
        st.set_deleted(np.zeros(n, dtype=bool))


def test_tq_wide_pass_masses_of_equal_scores_take_the_narrow_scan(qa):
    """one row repeated 263 k times: every score of a query is the same number, every candidate list overflows, every query takes the conditional
    32-query scan; the lists are the linear scan's (the lowest ids)"""
    n, dim, nq, top = N, 256, 70, 10
    vecs, otq, rows, st = _segment(qa, O.DOT, dim, n, seed=11, ties=1)
    rng = np.random.default_rng(6)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    s = qa.BatchFilteredSearcher(queries, st, top)
    got = s.peek_top_all()
    assert "scan_tq4w_kernel" in _kernel(qa, s)
    assert s.counters.fallback_queries == nq
    _same(got, _narrow(qa, queries, st, top))
    want = _oracle_top(otq, rows, O.preprocess(O.DOT, queries[:2]), n, top)
    for g, (ids, sc) in zip(got[:2], want):
        assert g["idx"].tolist() == ids.tolist() == list(range(top))
        assert np.array_equal(g["score"].view(np.uint32), sc.view(np.uint32))


def test_tq_wide_pass_few_distinct_rows(qa):
    """five distinct rows repeated 52 k times each: 52 k candidates tie at a query's best score; whether a query verifies them all or takes the narrow
    scan is the pool's business - the lists are the linear scan's either way"""
    n, dim, nq, top = N, 256, 40, 10
    vecs, otq, rows, st = _segment(qa, O.DOT, dim, n, seed=12, ties=5)
    rng = np.random.default_rng(7)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    s = qa.BatchFilteredSearcher(queries, st, top)
    got = s.peek_top_all()
    assert "scan_tq4w_kernel" in _kernel(qa, s)
    _same(got, _narrow(qa, queries, st, top))
    want = _oracle_top(otq, rows, O.preprocess(O.DOT, queries[:2]), n, top)
    for g, (ids, sc) in zip(got[:2], want):
        assert g["idx"].tolist() == ids.tolist()
        assert np.array_equal(g["score"].view(np.uint32), sc.view(np.uint32))


def test_tq_wide_pass_threshold_option(qa):
    """below tq_wide_min_queries (33 by default) the 32-query kernel serves; the option moves the switch"""
    n, dim = N, 256
    vecs, otq, rows, st = _segment(qa, O.DOT, dim, n, seed=dim * 3 + O.DOT)
    rng = np.random.default_rng(8)
    queries = rng.standard_normal((32, dim)).astype(np.float32)
    s = qa.BatchFilteredSearcher(queries, st, 10)
    narrow = s.peek_top_all()
    assert "scan_tq4w_kernel" not in _kernel(qa, s)
    qa.set_option("tq_wide_min_queries", 8)
    try:
        s = qa.BatchFilteredSearcher(queries, st, 10)
        wide = s.peek_top_all()
        assert "scan_tq4w_kernel" in _kernel(qa, s)
    finally:
        qa.set_option("tq_wide_min_queries", -1)
    _same(wide, narrow)

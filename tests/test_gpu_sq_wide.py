"""The 128-query pass over scalar-int8 blocks (qdrant_amd/csrc/scan_sqw.hip): brute-force top-k of 33 and more queries over a block of 2^18 rows and more
streams the codes once per 128 queries - a wave's lanes fetch exactly their matrix-core operand pieces, the integer dots are exact, the f32 expression of
postprocess_score (encoded_vectors_u8.rs:100-103) finishes them - so the lists must be the 32-query scan's and the oracle's (ids, score bits, tie order).
A query whose candidate lists overflow (masses of equal scores) takes the 32-query scan alone (qmx_counters.fallback_queries)."""
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
    return {O.COSINE: qa.Distance.Cosine, O.DOT: qa.Distance.Dot, O.EUCLID: qa.Distance.Euclid, O.MANHATTAN: qa.Distance.Manhattan}[d]


def _same(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g["idx"].tolist() == w["idx"].tolist()
        assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32))


def _kernel(qa, searcher):
    return qa._ffi.last_kernel(searcher.scorer._h)


def _narrow(qa, queries, st, top):
    qa.set_option("sq_wide_min_queries", 0)
    try:
        s = qa.BatchFilteredSearcher(queries, st, top)
        res = s.peek_top_all()
        assert "scan_sqw_kernel" not in _kernel(qa, s)
        return res
    finally:
        qa.set_option("sq_wide_min_queries", -1)


_CACHE = {}


def _segment(qa, dist, dim, n, seed, ties=0):
    key = (dist, dim, n, seed, ties)
    if key in _CACHE:
        return _CACHE[key]
    _CACHE.clear()
    rng = np.random.default_rng(seed)
    if ties:
        base = rng.standard_normal((ties, dim)).astype(np.float32)
        vecs = base[rng.integers(0, ties, n)]
    else:
        centres = rng.standard_normal((64, dim)).astype(np.float32)
        vecs = (centres[rng.integers(0, 64, n)] * rng.uniform(0.5, 2.0, (n, 1)) + 0.5 * rng.standard_normal((n, dim))).astype(np.float32)
    vecs = O.preprocess(dist, vecs)
    quant = qa.ScalarQuantizer.from_min_max(vecs[:20000], dim, _dist(qa, dist))
    osq = O.SqOracle(dist, dim, quant.alpha, quant.offset)
    rows = quant.encode(vecs)                                 # (byte-exact against the oracle's encoder: tests/test_gpu_sq.py)
    assert np.array_equal(rows[:64], osq.encode_rows(vecs[:64]))
    st = qa.EncodedVectorsU8(rows, quant)
    _CACHE[key] = (vecs, osq, rows, st)
    return _CACHE[key]


def _oracle_top(osq, rows, qpre, n, top, dead=None):
    osq.rows = rows
    sc = osq.score_points(qpre, np.arange(n, dtype=np.uint32))
    out = []
    for i in range(len(qpre)):
        s = sc[i].copy()
        if dead is not None:
            s[dead] = -np.inf
        order = np.lexsort((np.arange(n), -s.astype(np.float64)))[:top]      # descending score, ties -> lower id
        out.append((order, s[order]))
    return out


@pytest.mark.parametrize("nq,top", [(33, 10), (128, 10), (150, 1), (300, 64)])
@pytest.mark.parametrize("dist,dim", [(O.DOT, 128), (O.COSINE, 768), (O.EUCLID, 768), (O.DOT, 1024)])
def test_sq_wide_pass_returns_the_narrow_scan_and_the_oracle(qa, dist, dim, nq, top):
    n = N
    vecs, osq, rows, st = _segment(qa, dist, dim, n, seed=dim * 3 + dist)
    rng = np.random.default_rng(nq * 7 + top)
    queries = (vecs[rng.integers(0, n, nq)] + 0.3 * rng.standard_normal((nq, dim))).astype(np.float32)
    queries[nq // 2] = 0.0
    s = qa.BatchFilteredSearcher(queries, st, top)
    got = s.peek_top_all()
    assert "scan_sqw_kernel" in _kernel(qa, s), _kernel(qa, s)
    c = s.counters
    assert c.prefilter_queries == nq and c.prefilter_candidates >= c.verified_rows >= top * (nq - c.fallback_queries)
    _same(got, _narrow(qa, queries, st, top))
    k = 2
    want = _oracle_top(osq, rows, O.preprocess(dist, queries[:k]), n, top)
    for g, (ids, sc) in zip(got[:k], want):
        assert g["idx"].tolist() == ids.tolist()
        assert np.array_equal(g["score"].view(np.uint32), sc.view(np.uint32))


@pytest.mark.parametrize("dist", [O.DOT, O.EUCLID])
def test_sq_wide_pass_with_deleted_rows(qa, dist):
    n, dim, nq, top = N, 128, 140, 10
    vecs, osq, rows, st = _segment(qa, dist, dim, n, seed=dim * 3 + dist)
    rng = np.random.default_rng(5)
    queries = (vecs[rng.integers(0, n, nq)] + 0.3 * rng.standard_normal((nq, dim))).astype(np.float32)
    deleted = rng.random(n) < 0.3
    deleted[n - 300:] = True                                   # the last tile: all deleted
    st.set_deleted(deleted)
    try:
        s = qa.BatchFilteredSearcher(queries, st, top)
        got = s.peek_top_all()
        assert "scan_sqw_kernel" in _kernel(qa, s)
        _same(got, _narrow(qa, queries, st, top))
        want = _oracle_top(osq, rows, O.preprocess(dist, queries[:2]), n, top, dead=deleted)
        for g, (ids, sc) in zip(got[:2], want):
            assert g["idx"].tolist() == ids.tolist() and not deleted[g["idx"]].any()
            assert np.array_equal(g["score"].view(np.uint32), sc.view(np.uint32))
    finally:
        st.set_deleted(np.zeros(n, dtype=bool))


def test_sq_wide_pass_masses_of_equal_scores_take_the_narrow_scan(qa):
    """one row repeated 263 k times: every score of a query is the same number, every candidate list overflows, every query takes the conditional
    32-query scan; the lists are the linear scan's (the lowest ids)"""
    n, dim, nq, top = N, 128, 70, 10
    vecs, osq, rows, st = _segment(qa, O.DOT, dim, n, seed=11, ties=1)
    rng = np.random.default_rng(6)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    s = qa.BatchFilteredSearcher(queries, st, top)
    got = s.peek_top_all()
    assert "scan_sqw_kernel" in _kernel(qa, s)
    assert s.counters.fallback_queries == nq
    _same(got, _narrow(qa, queries, st, top))
    for g in got[:3]:
        assert g["idx"].tolist() == list(range(top))


def test_sq_wide_pass_few_distinct_rows_and_manhattan_keeps_the_narrow_kernel(qa):
    n, dim, nq, top = N, 128, 40, 10
    vecs, osq, rows, st = _segment(qa, O.DOT, dim, n, seed=12, ties=5)
    rng = np.random.default_rng(7)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    s = qa.BatchFilteredSearcher(queries, st, top)
    got = s.peek_top_all()
    assert "scan_sqw_kernel" in _kernel(qa, s)
    _same(got, _narrow(qa, queries, st, top))
    # Manhattan (sad, not a dot) and rows that are not a multiple of 128 codes: the 32-query / VALU kernels serve
    for dist, d2 in ((O.MANHATTAN, 128), (O.DOT, 96)):
        vecs, osq, rows, st = _segment(qa, dist, d2, n, seed=13)
        q2 = rng.standard_normal((nq, d2)).astype(np.float32)
        s = qa.BatchFilteredSearcher(q2, st, top)
        s.peek_top_all()
        assert "scan_sqw_kernel" not in _kernel(qa, s)

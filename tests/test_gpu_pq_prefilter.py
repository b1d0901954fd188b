"""The PQ prefilter (qdrant_amd/csrc/pq_prefilter.hip): brute-force top-k over PQ codes for 4 and more queries scans a rotated copy of the code
block with 6-bit tables (four queries per LDS gather, conflict-free by layout), keeps every row whose integer score is within a rigorous band
of the running k-th best and re-scores the survivors with the exact kernel (score_point_sse's order).  The integer scores never leave the
library: the lists must be the exact scan's - ids, score bits, tie order - and the oracle's.  A query whose lists overflow, or whose LUT is
degenerate, takes the exact scan alone (qmx_counters.fallback_queries)."""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu

N = 270_000          # >= 2^18 rows: the prefilter applies


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


def _exact(qa, queries, st, top):
    qa.set_option("no_pq_prefilter", 1)
    try:
        s = qa.BatchFilteredSearcher(queries, st, top)
        res = s.peek_top_all()
        assert "pq_prefilter_kernel" not in _kernel(qa, s)
        return res
    finally:
        qa.set_option("no_pq_prefilter", -1)


_CACHE = {}


def _segment(qa, dist, dim, chunk, n, seed, ncent=256, clustered=True, cache=False):
    """(rng, rows, quantizer, oracle quantizer with the codes, device storage); `cache`: the parametrised cases of one shape share one block"""
    key = (dist, dim, chunk, n, seed, ncent, clustered)
    if cache and key in _CACHE:
        rng, vecs, quant, opq, st = _CACHE[key]
        return np.random.default_rng(seed + 1), vecs, quant, opq, st
    if cache:
        _CACHE.clear()           # one block at a time (the parametrisation iterates shapes in the outer loop)
    out = _segment_build(qa, dist, dim, chunk, n, seed, ncent, clustered)
    if cache:
        _CACHE[key] = out
    return out


def _segment_build(qa, dist, dim, chunk, n, seed, ncent, clustered):
    rng = np.random.default_rng(seed)
    if clustered:      # rows around 64 centres: a codebook that means something, scores that crowd
        centres = rng.standard_normal((64, dim)).astype(np.float32)
        vecs = centres[rng.integers(0, 64, n)] + 0.3 * rng.standard_normal((n, dim)).astype(np.float32)
    else:
        vecs = rng.standard_normal((n, dim)).astype(np.float32)
    vecs = O.preprocess(dist, vecs.astype(np.float32))
    cen = O.PqOracle.train(vecs[:3000], dim, chunk, ncent, iters=3)
    opq = O.PqOracle(dist, dim, chunk, cen)
    quant = qa.ProductQuantizer(dim, _dist(qa, dist), chunk, cen)
    codes = quant.encode(vecs)
    opq.codes = codes
    return rng, vecs, quant, opq, qa.EncodedVectorsPQ(codes, quant)


def _oracle_top(opq, qpre, n, top, dead=None):
    sc = opq.score_points(qpre, np.arange(n, dtype=np.uint32))
    out = []
    for i in range(len(qpre)):
        s = sc[i].copy()
        if dead is not None:
            s[dead] = -np.inf
        # descending score, ties -> lower id (the linear scan keeps the first of equal scores)
        order = np.lexsort((np.arange(n), -s.astype(np.float64)))[:top]
        out.append((order, s[order]))
    return out


@pytest.mark.parametrize("nq,top", [(4, 10), (5, 1), (32, 10), (70, 64), (261, 10)])
@pytest.mark.parametrize("dist,dim,chunk", [(O.DOT, 768, 8), (O.COSINE, 128, 8), (O.EUCLID, 160, 4), (O.MANHATTAN, 66, 2), (O.DOT, 96, 1), (O.COSINE, 1536, 16)])
def test_pq_prefilter_returns_the_exact_scan(qa, dist, dim, chunk, nq, top):
    n = N if dim <= 768 else 262_200
    rng, vecs, quant, opq, st = _segment(qa, dist, dim, chunk, n, seed=dim * 7 + chunk + dist, cache=True)
    assert quant.m <= 96
    queries = (vecs[rng.integers(0, n, nq)] + 0.2 * rng.standard_normal((nq, dim))).astype(np.float32)
    s = qa.BatchFilteredSearcher(queries, st, top)
    got = s.peek_top_all()
    assert "pq_prefilter_kernel" in _kernel(qa, s), _kernel(qa, s)
    c = s.counters
    assert c.prefilter_queries == nq and c.verified_rows >= top * (nq - c.fallback_queries) and c.prefilter_candidates >= c.verified_rows
    _same(got, _exact(qa, queries, st, top))
    # ... and the oracle on the first queries: score bits and ids (ties -> lower id)
    k = min(nq, 3)
    want = _oracle_top(opq, O.preprocess(dist, queries[:k]), n, top)
    for i in range(k):
        assert np.array_equal(got[i]["score"].view(np.uint32), want[i][1].view(np.uint32))
        assert got[i]["idx"].tolist() == want[i][0].tolist()


def test_pq_prefilter_with_deleted_rows_and_filter(qa):
    dist, dim, chunk, nq, top = O.DOT, 256, 8, 37, 10
    rng, vecs, quant, opq, st = _segment(qa, dist, dim, chunk, N, seed=91)
    queries = (vecs[rng.integers(0, N, nq)] + 0.2 * rng.standard_normal((nq, dim))).astype(np.float32)
    pdel = rng.random(N) < 0.4
    vdel = rng.random(N) < 0.05
    st.set_deleted(pdel, vdel)
    s = qa.BatchFilteredSearcher(queries, st, top)
    got = s.peek_top_all()
    assert "pq_prefilter_kernel" in _kernel(qa, s)
    _same(got, _exact(qa, queries, st, top))
    for r in got:
        assert not pdel[r["idx"]].any() and not vdel[r["idx"]].any()
    want = _oracle_top(opq, O.preprocess(dist, queries[:2]), N, top, dead=pdel | vdel)
    for i in range(2):
        assert got[i]["idx"].tolist() == want[i][0].tolist()
    allowed = rng.random(N) < 0.3
    s.scorer.set_filter(allowed)
    got = s.peek_top_all()
    want = _oracle_top(opq, O.preprocess(dist, queries[:2]), N, top, dead=pdel | vdel | ~allowed)
    for i in range(2):
        assert got[i]["idx"].tolist() == want[i][0].tolist()
        assert np.array_equal(got[i]["score"].view(np.uint32), want[i][1].view(np.uint32))


def test_only_the_overflowing_queries_take_the_exact_pq_scan(qa):
    """5000 rows carry the same codes: a query near them has 5000 equal scores at the top of its list - more than a query may verify here (option
    verify_max_per_query = 2048) -, so that query is re-scanned exactly, the others keep the prefilter's verified lists."""
    dist, dim, chunk, nq, top, n_dup = O.DOT, 128, 8, 40, 10, 5000
    rng, vecs, quant, opq, st0 = _segment(qa, dist, dim, chunk, N, seed=17, clustered=False)
    codes = opq.codes.copy()
    dup_at = rng.choice(N, n_dup, replace=False)
    codes[dup_at] = codes[dup_at[0]]
    opq.codes = codes
    st = qa.EncodedVectorsPQ(codes, quant)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    hot = np.sort(rng.choice(nq, 6, replace=False))
    queries[hot] = (3.0 * vecs[dup_at[0]] + 0.05 * rng.standard_normal((6, dim))).astype(np.float32)
    qa.set_option("verify_max_per_query", 2048)        # (by default a query takes what it needs from the batch's pool: 5000 tied rows would simply be verified)
    try:
        s = qa.BatchFilteredSearcher(queries, st, top)
        got = s.peek_top_all()
    finally:
        qa.set_option("verify_max_per_query", -1)
    assert "pq_prefilter_kernel" in _kernel(qa, s)
    assert s.counters.fallback_queries == len(hot), (s.counters.fallback_queries, hot)
    _same(got, _exact(qa, queries, st, top))
    for i in hot:
        assert set(got[i]["idx"].tolist()) <= set(dup_at.tolist()) and got[i]["idx"].tolist() == sorted(got[i]["idx"].tolist())


@pytest.mark.parametrize("nq", [8, 131])
def test_degenerate_luts_take_the_exact_scan(qa, nq):
    """A zero query under Dot: every LUT entry is 0 (no step to quantise with); a NaN query: non-finite entries.  Both fall back, alone - and the
    exact pass, launched for the worst case, divides its blocks among the two (pq.hip: one straggler of 128 queries used to be scanned by 1 / 128th
    of the launch)."""
    dist, dim, chunk, top = O.DOT, 128, 8, 5
    rng, vecs, quant, opq, st = _segment(qa, dist, dim, chunk, N, seed=23)
    queries = (vecs[rng.integers(0, N, nq)] + 0.1 * rng.standard_normal((nq, dim))).astype(np.float32)
    queries[2] = 0.0
    queries[5, 7] = np.nan
    s = qa.BatchFilteredSearcher(queries, st, top)
    got = s.peek_top_all()
    assert "pq_prefilter_kernel" in _kernel(qa, s)
    assert s.counters.fallback_queries == 2
    want = _exact(qa, queries, st, top)
    for i in range(nq):
        assert got[i]["idx"].tolist() == want[i]["idx"].tolist()
        assert np.array_equal(got[i]["score"].view(np.uint32), want[i]["score"].view(np.uint32))
    assert got[2]["idx"].tolist() == list(range(top))          # all scores equal: the first rows


def test_small_blocks_and_candidate_lists_keep_the_exact_kernel(qa):
    dist, dim, chunk = O.DOT, 128, 8
    rng, vecs, quant, opq, st = _segment(qa, dist, dim, chunk, 20_000, seed=5)
    queries = rng.standard_normal((8, dim)).astype(np.float32)
    s = qa.BatchFilteredSearcher(queries, st, 10)
    s.peek_top_all()
    assert "pq_prefilter_kernel" not in _kernel(qa, s) and s.counters.prefilter_queries == 0

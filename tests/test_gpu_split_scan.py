"""The split prefilter (qdrant_amd/csrc/scan_split.hip): f32 dot / cosine top-k of more than 64 queries over a large block runs 128
queries per pass on the f16 matrix cores (x = h + l, three products), keeps every row whose approximate score is within a rigorous band
of the running k-th best, and re-scores the survivors with the exact gather kernel.  The approximate scores never leave the library:
the result must be the exact scan's — the oracle's — ids and score bits, ties included.  A query whose lists do not fit the buffers (masses
of equal scores, a sample without live rows) raises its own device flag and the exact scan of THOSE queries runs behind the prefilter."""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    import qdrant_amd
    assert qdrant_amd.device_count() >= 1
    return qdrant_amd


def _same(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g["idx"].tolist() == w["idx"].tolist()
        assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32))


def _kernel(qa, searcher):
    return qa._ffi.last_kernel(searcher.scorer._h)


N = 300_000          # >= 2^18: the prefilter applies


@pytest.mark.parametrize("distance,dim", [(O.COSINE, 128), (O.DOT, 256), (O.COSINE, 768)])
@pytest.mark.parametrize("nq,top", [(65, 10), (128, 1), (200, 10), (130, 64), (256, 10), (300, 5)])
@pytest.mark.parametrize("copy", [0, 1, 2])           # 1: QMX_SEG_SPLIT_COPY (f16 pairs, three products), 2: QMX_SEG_HALF_COPY (high parts, one product)
def test_split_scan_returns_the_exact_scan(qa, distance, dim, nq, top, copy):
    n = N if dim < 768 else 270_000
    rows = O.preprocess(distance, O.synth(0x5EED0500 + dim, 0, n, dim))
    queries = O.synth(0x5EED0501 + nq, 0, nq, dim)
    st = O.DenseStorage(O.F32, distance, rows)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine if distance == O.COSINE else qa.Distance.Dot, flags=[0, qa._ffi.SEG_SPLIT_COPY, qa._ffi.SEG_HALF_COPY][copy])
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    if nq % 128 == 0 or nq % 128 > 64:
        want_kernel = ["scan_f32_split_kernel", "scan_f16pair_kernel<false>", "scan_f16pair_kernel<true>"][copy]
        if copy == 2 and nq > 128 and (nq % 256 == 0 or nq % 256 > 128):     # the LAST tile of the batch takes the 256-query shape of the half copy
            want_kernel = "scan_f16half256_kernel"
        assert want_kernel in _kernel(qa, s), _kernel(qa, s)
    _same(got, st.peek_top(queries, top, threads=8))
    qa.set_option("no_split_scan", 1)                    # ... and the exact kernels agree (they are what the fallback runs)
    try:
        s2 = qa.BatchFilteredSearcher(queries, vs, top)
        _same(s2.peek_top_all(), got)
        assert "scan_f32_split_kernel" not in _kernel(qa, s2) and "scan_f16pair_kernel" not in _kernel(qa, s2) and "half256" not in _kernel(qa, s2)
    finally:
        qa.set_option("no_split_scan", -1)


def test_split_scan_with_deleted_rows_and_filter(qa):
    n, dim, nq, top = N, 128, 100, 10
    rng = np.random.default_rng(3)
    rows = O.preprocess(O.COSINE, O.synth(0x5EED0510, 0, n, dim))
    queries = O.synth(0x5EED0511, 0, nq, dim)
    deleted = rng.random(n) < 0.4
    vdel = rng.random(n) < 0.05
    st = O.DenseStorage(O.F32, O.COSINE, rows, point_deleted=deleted, vec_deleted=vdel)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine, flags=qa._ffi.SEG_HALF_COPY)
    vs.set_deleted(deleted, vdel)
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    assert "scan_f16pair_kernel" in _kernel(qa, s)
    _same(got, st.peek_top(queries, top, threads=8))
    for r in got:
        assert not deleted[r["idx"]].any() and not vdel[r["idx"]].any()
    # payload filter as an allow bitmap on top of the deleted flags
    allowed = rng.random(n) < 0.3
    s.scorer.set_filter(allowed)
    st_f = O.DenseStorage(O.F32, O.COSINE, rows, point_deleted=deleted | ~allowed, vec_deleted=vdel)
    _same(s.peek_top_all(), st_f.peek_top(queries, top, threads=8))


def _same_up_to_equal_scores(got, want):
    """Score bits identical; ids identical as sets; among equal scores the lower id first (the linear scan keeps the FIRST of equal scores:
    strict <, fixed_length_priority_queue.rs:53-57)."""
    for g, w in zip(got, want):
        assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32))
        assert sorted(g["idx"].tolist()) == sorted(w["idx"].tolist())
        for i in range(len(g) - 1):
            if g["score"][i] == g["score"][i + 1]:
                assert g["idx"][i] < g["idx"][i + 1]


@pytest.fixture
def max_2048_rows_per_query(qa):
    """A query may verify at most 2048 rows (option verify_max_per_query; by default it takes what it needs from the batch's pool of 16 384 per query):
    3000-fold ties then send exactly the tied queries to the exact scan - the per-query fallback these tests are about."""
    qa.set_option("verify_max_per_query", 2048)
    yield
    qa.set_option("verify_max_per_query", -1)


@pytest.mark.parametrize("copy", [0, 1, 2])
def test_split_scan_falls_back_when_scores_tie_in_masses(qa, copy, max_2048_rows_per_query):
    """Every row exists 3000 times: the verification band holds at least 3000 rows per query, more than a query may verify -> every query's overflow flag
    -> the exact scan of those queries runs behind the prefilter in the same stream.  Equal scores come back in ascending id order, like the oracle's."""
    dim, nq, top, rep = 128, 70, 10, 3000
    base = O.preprocess(O.COSINE, O.synth(0x5EED0520, 0, N // rep, dim))
    rows = np.tile(base, (rep, 1))                       # row i == row i + len(base)
    queries = O.synth(0x5EED0521, 0, nq, dim)
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine, flags=[0, qa._ffi.SEG_SPLIT_COPY, qa._ffi.SEG_HALF_COPY][copy])
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    assert ["scan_f32_split_kernel", "scan_f16pair_kernel<false>", "scan_f16pair_kernel<true>"][copy] in _kernel(qa, s)
    assert s.counters.prefilter_queries == nq and s.counters.fallback_queries == nq      # the prefilter ran, every query overflowed
    want = st.peek_top(queries, top)                      # sequential
    _same_up_to_equal_scores(got, want)
    for g, w in zip(got, want):
        assert sorted((g["idx"] % len(base)).tolist()) == sorted((w["idx"] % len(base)).tolist())


@pytest.mark.parametrize("copy", [0, 2])
def test_split_scan_when_the_sample_is_all_deleted(qa, copy):
    """The strided sample is deleted entirely: no threshold -> every row is a candidate -> the candidate buffers
    overflow -> the exact scan takes over for every query.  Same result as ever."""
    n, dim, nq, top = N, 128, 80, 5
    rows = O.preprocess(O.COSINE, O.synth(0x5EED0530, 0, n, dim))
    queries = O.synth(0x5EED0531, 0, nq, dim)
    S = max(n >> (11 if copy else 8), 8192)              # (api_search.hip search_enqueue: the sample of a block without a copy is eight times denser)
    step = n // S
    deleted = np.zeros(n, dtype=bool)
    deleted[::step] = True
    st = O.DenseStorage(O.F32, O.COSINE, rows, point_deleted=deleted)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine, flags=[0, 0, qa._ffi.SEG_HALF_COPY][copy])
    vs.set_deleted(deleted, None)
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    assert ["scan_f32_split_kernel", "", "scan_f16pair_kernel<true>"][copy] in _kernel(qa, s)
    assert s.counters.fallback_queries == nq
    _same(got, st.peek_top(queries, top, threads=8))


@pytest.mark.parametrize("copy", [0, 1, 2])
@pytest.mark.parametrize("nq,n_hot", [(150, 5), (128, 1), (100, 20), (300, 70), (256, 17)])
def test_only_the_overflowing_queries_take_the_exact_scan(qa, copy, nq, n_hot, max_2048_rows_per_query):
    """3000 rows of the block are copies of one vector v.  A query near v has 3000 equal scores at the top of its list: more than its
    verification list takes, so THAT query is re-scanned exactly; the other queries of the batch keep the prefilter's (verified) lists.
    qmx_counters.fallback_queries says how many took the exact scan.  nq = 150 without a copy: the last 22 queries take the regular exact
    path, whose pre-scan bounds once overwrote those of queries 0..21 before their deferred exact pass read them (round-2 advisor finding)."""
    n, dim, top, n_dup = N, 128, 10, 3000
    rng = np.random.default_rng(11)
    raw = O.synth(0x5EED0550, 0, n, dim)
    v = O.synth(0x5EED0551, 0, 1, dim)[0]
    dup_at = rng.choice(n, n_dup, replace=False)
    raw[dup_at] = v
    rows = O.preprocess(O.COSINE, raw)
    queries = O.synth(0x5EED0552 + nq, 0, nq, dim)
    hot = rng.choice(nq, n_hot, replace=False)
    queries[hot] = v + 0.05 * O.synth(0x5EED0553, 0, n_hot, dim)
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine, flags=[0, qa._ffi.SEG_SPLIT_COPY, qa._ffi.SEG_HALF_COPY][copy])
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    c = s.counters
    # without a copy a last tile of <= 64 queries takes the regular exact path: its hot queries never see the prefilter
    n_split = nq if copy or nq % 128 == 0 or nq % 128 > 64 else nq - nq % 128
    assert c.prefilter_queries == n_split
    assert c.fallback_queries == int((hot < n_split).sum()), (c.fallback_queries, sorted(hot.tolist()))
    assert c.verified_rows >= top * (n_split - c.fallback_queries) and c.prefilter_candidates >= c.verified_rows
    want = st.peek_top(queries, top)
    cold = [i for i in range(nq) if i not in set(hot.tolist())]
    _same([got[i] for i in cold], [want[i] for i in cold])
    _same_up_to_equal_scores([got[i] for i in hot], [want[i] for i in hot])
    for i in hot:
        assert set(got[i]["idx"].tolist()) <= set(dup_at.tolist())         # (the copies of v are the best rows of a query near v)


@pytest.mark.parametrize("row_mag,query_mag", [(1000.0, 1e-3), (1e-4, 1e4), (37.0, 1.0)])
def test_split_scan_dot_with_any_magnitudes(qa, row_mag, query_mag):
    """Dot distance, un-normalised rows: the power-of-two scales come from max |x| of the block and of the batch."""
    n, dim, nq, top = N - 77, 128, 90, 10            # (a partial last 256-row tile)
    rows = (O.synth(0x5EED0540, 0, n, dim) * np.float32(row_mag)).astype(np.float32)
    queries = (O.synth(0x5EED0541, 0, nq, dim) * np.float32(query_mag)).astype(np.float32)
    st = O.DenseStorage(O.F32, O.DOT, rows)
    for copy in (0, 1, 2):
        vs = qa.VectorStorage(rows, qa.Distance.Dot, flags=[0, qa._ffi.SEG_SPLIT_COPY, qa._ffi.SEG_HALF_COPY][copy])
        s = qa.BatchFilteredSearcher(queries, vs, top)
        got = s.peek_top_all()
        assert ["scan_f32_split_kernel", "scan_f16pair_kernel<false>", "scan_f16pair_kernel<true>"][copy] in _kernel(qa, s)
        _same(got, st.peek_top(queries, top, threads=8))


@pytest.mark.parametrize("copy", [1, 2])
def test_the_verification_pool_serves_long_tie_lists_and_overflows_query_by_query(qa, copy):
    """By default the batch shares one pool of 16 384 rows per query: a query with 3000 tied rows at the top verifies them all (no exact scan), and when
    the ties are so many that the pool runs out - every row exists 30 000 times: 70 queries want 2.1 M rows of a pool of 1.15 M - the queries that
    come too late take the exact scan, one by one, while the others keep their verified lists.  Lists: the exact scan's either way."""
    dim, nq, top = 128, 70, 10
    flag = [0, qa._ffi.SEG_SPLIT_COPY, qa._ffi.SEG_HALF_COPY][copy]
    for rep, want_fallback in ((3000, False), (30_000, True)):
        base = O.preprocess(O.COSINE, O.synth(0x5EED0560, 0, N // rep, dim))
        rows = np.tile(base, (rep, 1))
        queries = O.synth(0x5EED0561, 0, nq, dim)
        vs = qa.VectorStorage(rows, qa.Distance.Cosine, flags=flag)
        s = qa.BatchFilteredSearcher(queries, vs, top)
        got = s.peek_top_all()
        c = s.counters
        assert c.prefilter_queries == nq
        if want_fallback:
            assert 0 < c.fallback_queries < nq, c.fallback_queries
        else:
            assert c.fallback_queries == 0 and c.verified_rows >= 3000 * nq
        qa.set_option("no_split_scan", 1)
        try:
            exact = qa.BatchFilteredSearcher(queries, vs, top).peek_top_all()
        finally:
            qa.set_option("no_split_scan", -1)
        for g, e in zip(got, exact):
            assert np.array_equal(g, e)
        for g in got:       # the lowest offsets of the tied copies, ascending
            assert g["idx"].tolist() == sorted(g["idx"].tolist()) and (g["idx"] < 10 * len(base)).all()

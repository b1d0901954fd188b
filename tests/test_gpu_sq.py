"""GPU parity: EncodedVectorsU8 (scalar int8 quantization) — encode, encode_query, score, score_internal,
brute-force top-k on codes, oversampling + rescoring — through the C-ABI against the CPU oracle.
Integer / quantized work: every comparison is BIT-EXACT against the x86 AVX2 path; the oracle's
integer leaves are themselves pinned to the reference's own C kernels (oracle/_ref, built from
lib/quantization/cpp/avx2.c) in tests/test_oracle_golden.py and again here.
Error-bound tests restate lib/quantization/tests/integration/test_avx2.rs:16-57.
"""
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


def _setup(qa, dist, dim, n, seed, isa=O.ISA_AUTO, scale=1.0):
    rng = np.random.default_rng(seed)
    raw = (rng.standard_normal((n, dim)) * scale).astype(np.float32)
    vecs = O.preprocess(dist, raw)                       # stored vectors are preprocessed at insert
    quant = qa.ScalarQuantizer.from_min_max(vecs, dim, _dist(qa, dist))
    osq = O.SqOracle(dist, dim, quant.alpha, quant.offset, isa)
    assert np.float32(osq.sq.multiplier) == quant.multiplier and osq.sq.actual_dim == quant.actual_dim
    return rng, vecs, quant, osq


@pytest.mark.parametrize("dist", [O.DOT, O.COSINE, O.EUCLID, O.MANHATTAN])
@pytest.mark.parametrize("dim", [3, 16, 65, 128, 768, 1041, 1536, 2064])
def test_sq_encode_score_bit_exact(qa, dist, dim):
    n, nq = 300, 4
    rng, vecs, quant, osq = _setup(qa, dist, dim, n, seed=dim * 3 + dist)
    want_rows = osq.encode_rows(vecs)
    got_rows = quant.encode(vecs)
    assert np.array_equal(got_rows, want_rows)                             # codes AND the f32 vector_offset bytes
    st = qa.EncodedVectorsU8(got_rows, quant)
    assert np.array_equal(st.get_quantized_vector([0, n - 1, 7]), want_rows[[0, n - 1, 7]])
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    qpre = O.preprocess(dist, queries)
    scorer = qa.new_raw_scorer(queries, st)
    for i in range(nq):                                                    # EncodedQueryU8 {offset, codes}
        codes, off = osq.encode_query(qpre[i])
        enc = scorer.encoded_query(i)
        assert np.array_equal(enc[4:], codes)
        assert enc[:4].view(np.float32)[0].view(np.uint32) == np.float32(off).view(np.uint32)
    ids = rng.permutation(n).astype(np.uint32)[:200]
    got = scorer.score_points(ids)
    want = osq.score_points(qpre, ids)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # score_internal (encoded_vectors_u8.rs:675-705)
    a, b = ids[:50], ids[50:100]
    assert np.array_equal(scorer.score_internal(a, b).view(np.uint32), osq.score_internal(a, b).view(np.uint32))
    # score_bytes == score_points on the same rows (query_scorer/mod.rs:48-68)
    assert np.array_equal(scorer.score_bytes(want_rows[ids[:64]]).view(np.uint32), got[:, :64].view(np.uint32))


def test_sq_matches_reference_c_kernels(qa):
    # the integer leaves of the reference itself (oracle/_ref = lib/quantization/cpp/avx2.c compiled unchanged)
    if O.load_ref_quant() is None:
        pytest.skip("oracle/_ref not present")
    for dist, dim in ((O.DOT, 768), (O.MANHATTAN, 768), (O.EUCLID, 1536), (O.DOT, 2064), (O.DOT, 48)):
        n = 200
        rng, vecs, quant, osq = _setup(qa, dist, dim, n, seed=dim + dist, isa=O.ISA_REF)
        rows = osq.encode_rows(vecs)
        st = qa.EncodedVectorsU8(rows, quant)
        queries = rng.standard_normal((3, dim)).astype(np.float32)
        ids = np.arange(n, dtype=np.uint32)
        got = qa.new_raw_scorer(queries, st).score_points(ids)
        want = osq.score_points(O.preprocess(dist, queries), ids)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # worst case codes (all 127): lane sums beyond 2^24 at dim 2064 exercise the f32 hsum order
    dim, n = 2064, 8
    quant = qa.ScalarQuantizer(dim, qa.Distance.Dot, 1.0 / 127.0, 0.0)
    osq = O.SqOracle(O.DOT, dim, quant.alpha, quant.offset, O.ISA_REF)
    vecs = np.ones((n, dim), dtype=np.float32)
    rows = osq.encode_rows(vecs)
    assert rows[:, 4:].min() == 127
    st = qa.EncodedVectorsU8(rows, quant)
    got = qa.new_raw_scorer(np.ones((1, dim), dtype=np.float32), st).score_points(np.arange(n, dtype=np.uint32))
    want = osq.score_points(np.ones((1, dim), dtype=np.float32), np.arange(n))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("dist", [O.DOT, O.EUCLID, O.MANHATTAN])
def test_sq_error_bound_like_reference_tests(qa, dist):
    # lib/quantization/tests/integration/test_avx2.rs:16-57: dim 65, 129 vectors in [0,1), |quantized - exact| < dim * 0.1
    rng = np.random.default_rng(42)
    dim, n = 65, 129
    vecs = rng.random((n, dim)).astype(np.float32)
    query = rng.random(dim).astype(np.float32)
    quant = qa.ScalarQuantizer.from_min_max(vecs, dim, _dist(qa, dist))
    st = qa.EncodedVectorsU8(quant.encode(vecs), quant)
    got = qa.new_raw_scorer(query, st).score_points(np.arange(n, dtype=np.uint32))[0]
    exact = np.array([O.similarity(O.F32, dist, query, vecs[i]) for i in range(n)])
    assert np.all(np.abs(got - exact) < dim * 0.1)


@pytest.mark.parametrize("dist", [O.COSINE, O.EUCLID])
def test_sq_brute_force_oversampling_and_rescore(qa, dist):
    # PlainVectorIndexReadView::search with a quantized scorer, oversampling 2.0, rescore = true
    # (plain_vector_index/read_view/search.rs:54-135, vector_index_search_common.rs:39-44,73-90)
    n, dim, nq, top = 20000, 128, 8, 10
    rng, vecs, quant, osq = _setup(qa, dist, dim, n, seed=7 + dist)
    rows = quant.encode(vecs)
    osq.rows = rows
    qst = qa.EncodedVectorsU8(rows, quant)
    ost = qa.VectorStorage(vecs, _dist(qa, dist))
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    qpre = O.preprocess(dist, queries)
    oversampled = int(2.0 * top)
    got_q = qa.BatchFilteredSearcher(queries, ost, oversampled, quantized_vectors=qst).peek_top_all()
    all_q = osq.score_points(qpre, np.arange(n))
    for i, g in enumerate(got_q):
        order = np.argsort(-all_q[i], kind="stable")[:oversampled]
        assert np.array_equal(g["score"].view(np.uint32), all_q[i][order].view(np.uint32))
        kth = all_q[i][order[-1]]
        assert set(g["idx"][g["score"] > kth]) == set(order[all_q[i][order] > kth].tolist())
    # rescoring with the ORIGINAL vectors: sort desc, truncate(top)
    ids = np.stack([g["idx"] for g in got_q])
    resc = qa.new_raw_scorer(queries, ost).rescore(ids, top)
    exact = O.DenseStorage(O.F32, dist, vecs)
    for i, r in enumerate(resc):
        s = exact.score_points(queries[i], ids[i])[0]
        order = np.argsort(-s, kind="stable")[:top]
        assert r["idx"].tolist() == ids[i][order].tolist()
        assert np.array_equal(r["score"].view(np.uint32), s[order].view(np.uint32))
    # recall of quantized + rescored search against the exact search (hnsw_quantized_search_test.rs:248-330 asks > 40 %)
    truth = exact.peek_top(queries, top)
    hits = sum(len(set(r["idx"]) & set(t["idx"])) for r, t in zip(resc, truth))
    assert hits / (nq * top) > 0.8


def test_ragged_score_points_and_counts(qa):
    # HNSW hops of many searches batched into one launch: query qi scores its own <= m0 ids
    rng = np.random.default_rng(3)
    n, dim, nq = 4000, 96, 13
    for dtype in ("f32", "sq"):
        vecs = O.preprocess(O.COSINE, rng.standard_normal((n, dim)).astype(np.float32))
        queries = rng.standard_normal((nq, dim)).astype(np.float32)
        lists = [rng.integers(0, n, rng.integers(0, 33)).astype(np.uint32) for _ in range(nq)]
        if dtype == "f32":
            st = qa.VectorStorage(vecs, qa.Distance.Cosine)
            want = [O.DenseStorage(O.F32, O.COSINE, vecs).score_points(queries[i], lists[i])[0] for i in range(nq)]
        else:
            quant = qa.ScalarQuantizer.from_min_max(vecs, dim, qa.Distance.Cosine)
            osq = O.SqOracle(O.COSINE, dim, quant.alpha, quant.offset)
            rows = osq.encode_rows(vecs)
            st = qa.EncodedVectorsU8(rows, quant)
            qpre = O.preprocess(O.COSINE, queries)
            want = [osq.score_points(qpre[i], lists[i])[0] for i in range(nq)]
        scorer = qa.new_raw_scorer(queries, st)
        got = scorer.score_points_ragged(lists)
        for g, w in zip(got, want):
            assert np.array_equal(g.view(np.uint32), np.asarray(w, dtype=np.float32).view(np.uint32))
        # rescore with per-query counts (fewer results than slots)
        ids = rng.integers(0, n, (nq, 20)).astype(np.uint32)
        counts = rng.integers(0, 21, nq).astype(np.uint32)
        res = scorer.rescore(ids, 5, counts)
        for i, r in enumerate(res):
            assert len(r) == min(5, counts[i])
            assert set(r["idx"]) <= set(ids[i][:counts[i]])
            assert np.all(np.diff(r["score"]) <= 0)


def test_sq_fit_min_max_on_device(qa):
    """`quantile = None` fit (find_min_max_from_iter, quantile.rs:19-33): exact, NaN never wins, all derived fields."""
    import ctypes as C
    from qdrant_amd import _ffi as F
    rng = np.random.default_rng(77)
    for dist in (O.DOT, O.EUCLID, O.MANHATTAN, O.COSINE):
        dim = 70
        data = (rng.standard_normal((5000, dim)) * 3).astype(np.float32)
        data[17, 3] = np.nan
        ref = qa.ScalarQuantizer.from_min_max(data[~np.isnan(data).any(axis=1)], dim, _dist(qa, dist))
        fit = qa.ScalarQuantizer.fit(data, dim, _dist(qa, dist))
        assert fit.alpha == ref.alpha and fit.offset == ref.offset and fit.multiplier == ref.multiplier and fit.actual_dim == ref.actual_dim
        p = F.SqParams()
        F.check(F.lib().qmx_sq_fit_min_max(0, int(_dist(qa, dist)), F.ptr(data), len(data), dim, C.byref(p)))
        osq = O.SqOracle(dist, dim, p.alpha, p.offset)
        assert np.float32(osq.sq.multiplier) == np.float32(p.multiplier) and osq.sq.actual_dim == p.actual_dim and bool(osq.sq.invert) == bool(p.invert)


def test_search_quantized_pipeline_matches_the_reference_flow(qa):
    """get_oversampled_top + quantized search + postprocess_search_result (vector_index_search_common.rs:27-91) in one
    call == the same steps done by hand with the oracle, for the plain index and for the graph."""
    rng = np.random.default_rng(91)
    n, dim, nq, top = 4000, 64, 6, 10
    centers = rng.standard_normal((16, dim)).astype(np.float32) * 2
    vecs = O.preprocess(O.COSINE, (centers[rng.integers(0, 16, n)] + 0.7 * rng.standard_normal((n, dim))).astype(np.float32))
    queries = O.preprocess(O.COSINE, rng.standard_normal((nq, dim)).astype(np.float32))
    quant = qa.ScalarQuantizer.fit(vecs, dim, qa.Distance.Dot)
    osq = O.SqOracle(O.DOT, dim, quant.alpha, quant.offset)
    codes = osq.encode_rows(vecs)
    enc = qa.EncodedVectorsU8(codes, quant)
    vs = qa.VectorStorage(vecs, qa.Distance.Cosine)
    sq_scorer, raw_scorer = qa.new_raw_scorer(queries, enc), qa.new_raw_scorer(queries, vs)
    truth = O.DenseStorage(O.F32, O.COSINE, vecs)
    sq_scores = osq.score_points(queries, np.arange(n))
    for oversampling, otop in [(2.5, 25), (1.0, 10), (0.0, 10), (7.3, 73)]:          # 73 > 64: two bounded passes inside
        got = qa.search_quantized(sq_scorer, raw_scorer, top, oversampling=oversampling, rescore=True)
        for qi in range(nq):
            cand = np.argsort(-sq_scores[qi], kind="stable")[:otop]
            exact = truth.score_points(queries[qi:qi + 1], cand)[0]
            order = np.argsort(-exact, kind="stable")[:top]
            assert np.array_equal(got[qi]["score"].view(np.uint32), exact[order].view(np.uint32))
            assert set(got[qi]["idx"].tolist()) == set(cand[order].tolist())
        plain = qa.search_quantized(sq_scorer, None, top, oversampling=oversampling, rescore=False)   # truncate(top) of the quantized list
        for qi in range(nq):
            want = np.sort(sq_scores[qi])[::-1][:top]
            assert np.array_equal(plain[qi]["score"].view(np.uint32), want.view(np.uint32))
    # graph arm: == qmx_hnsw_search(oversampled top, max(ef, oversampled top)) + qmx_rescore by hand
    graph = qa.GraphLayers.build(vs, m=8, ef_construct=64, seed=4)
    got = qa.search_quantized(sq_scorer, raw_scorer, top, oversampling=3.0, rescore=True, graph=graph, hnsw_ef=16)
    walk = graph.search(30, 30, sq_scorer)
    ids = np.zeros((nq, 30), dtype=np.uint32)
    cnt = np.zeros(nq, dtype=np.uint32)
    for i, r in enumerate(walk):
        ids[i, :len(r)] = r["idx"]
        cnt[i] = len(r)
    want = raw_scorer.rescore(ids, top, cnt)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    with pytest.raises(qa.QmxError):
        qa.search_quantized(sq_scorer, None, top, rescore=True)                       # rescoring needs the original batch


@pytest.mark.parametrize("n,dim,q", [(5000, 64, 0.99), (1000, 96, 0.95), (200, 17, 0.5), (130, 3, 0.999), (5000, 768, 0.99)])
def test_sq_fit_quantile_is_the_reference_order_statistic(qa, n, dim, q):
    """qmx_sq_fit_quantile == find_quantile_interval (quantile.rs:35-84) on the same sample: an exact order statistic."""
    rng = np.random.default_rng(n + dim)
    sample = (rng.standard_normal((n, dim)) * rng.uniform(0.5, 3.0)).astype(np.float32)
    want = O.sq_quantile_interval(sample, n, q)
    assert want is not None
    quant = qa.ScalarQuantizer.fit_quantile(sample, dim, qa.Distance.Dot, q)
    assert np.float32(quant.offset).view(np.uint32) == want[0].view(np.uint32)
    assert np.float32(quant.alpha).view(np.uint32) == ((want[1] - want[0]) / np.float32(127.0)).view(np.uint32)
    srt = np.sort(sample.ravel())
    cut = max(1, min((srt.size - 1) // 2, int(np.float32(n) * (np.float32(1.0) - np.float32(q)) / np.float32(2.0))))
    assert want[0] == srt[cut + 1] and want[1] == srt[srt.size - cut - 1]


def test_sq_fit_quantile_fallbacks_like_the_reference(qa):
    rng = np.random.default_rng(3)
    small = rng.standard_normal((100, 8)).astype(np.float32)                       # count < 127 -> None -> min / max fit
    assert O.sq_quantile_interval(small, 100, 0.99) is None
    a, b = qa.ScalarQuantizer.fit_quantile(small, 8, qa.Distance.Dot, 0.99), qa.ScalarQuantizer.fit(small, 8, qa.Distance.Dot)
    assert (a.alpha, a.offset) == (b.alpha, b.offset)
    data = rng.standard_normal((500, 8)).astype(np.float32)                        # quantile >= 1 -> None
    assert O.sq_quantile_interval(data, 500, 1.0) is None
    a, b = qa.ScalarQuantizer.fit_quantile(data, 8, qa.Distance.Dot, 1.0), qa.ScalarQuantizer.fit(data, 8, qa.Distance.Dot)
    assert (a.alpha, a.offset) == (b.alpha, b.offset)
    # a sample of the storage (count says how many vectors the storage has)
    q2 = qa.ScalarQuantizer.fit_quantile(data, 8, qa.Distance.Euclid, 0.9, sample=data[::5], count=500)
    want = O.sq_quantile_interval(data[::5], 500, 0.9)
    assert np.float32(q2.offset) == want[0] and q2.invert

"""GPU parity: EncodedVectorsPQ — encode (argmin), LUT build (exact order and MFMA), score in
score_point_sse order, score_internal, brute-force top-k on codes — through the C-ABI against the
CPU oracle.  Codes and exact-order LUT / scores are BIT-EXACT; the MFMA LUT (Dot/Cosine, an fmaf
chain instead of mul+add) is within 1e-5 relative to sum(abs(terms)).
Error-bound test restates lib/quantization/tests/integration/test_pq.rs:14-55.
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


def _setup(qa, dist, dim, chunk, n, ncent, seed, mfma=False):
    rng = np.random.default_rng(seed)
    vecs = O.preprocess(dist, rng.standard_normal((n, dim)).astype(np.float32))
    cen = O.PqOracle.train(vecs[: min(n, 2000)], dim, chunk, ncent, iters=3)
    opq = O.PqOracle(dist, dim, chunk, cen)
    quant = qa.ProductQuantizer(dim, _dist(qa, dist), chunk, cen, lut_mfma=mfma)
    assert quant.m == opq.m
    return rng, vecs, quant, opq


@pytest.mark.parametrize("dist", [O.DOT, O.COSINE, O.EUCLID, O.MANHATTAN])
@pytest.mark.parametrize("dim,chunk,ncent", [(64, 1, 256), (65, 2, 256), (96, 16, 256), (128, 8, 100), (70, 16, 256), (1536, 16, 256)])
def test_pq_encode_lut_score_bit_exact(qa, dist, dim, chunk, ncent):
    n, nq = 700, 3
    rng, vecs, quant, opq = _setup(qa, dist, dim, chunk, n, ncent, seed=dim * 5 + chunk + dist)
    want_codes = opq.encode(vecs)
    got_codes = quant.encode(vecs)
    assert np.array_equal(got_codes, want_codes)                 # integer output: exact, first minimum wins
    st = qa.EncodedVectorsPQ(got_codes, quant)
    assert np.array_equal(st.get_quantized_vector([3, n - 1]), want_codes[[3, n - 1]])
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    qpre = O.preprocess(dist, queries)
    scorer = qa.new_raw_scorer(queries, st)
    for i in range(nq):
        assert np.array_equal(scorer.encoded_query(i).view(np.uint32), opq.lut(qpre[i]).view(np.uint32))
    ids = rng.permutation(n).astype(np.uint32)[:300]
    got = scorer.score_points(ids)
    want = opq.score_points(qpre, ids)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    a, b = ids[:64], ids[64:128]
    assert np.array_equal(scorer.score_internal(a, b).view(np.uint32), opq.score_internal(a, b).view(np.uint32))
    # ragged (HNSW hop) scoring == dense scoring
    lists = [ids[:7], ids[7:40], ids[40:41]]
    rag = scorer.score_points_ragged(lists)
    assert np.array_equal(rag[0].view(np.uint32), want[0, :7].view(np.uint32))
    assert np.array_equal(rag[1].view(np.uint32), want[1, 7:40].view(np.uint32))
    assert np.array_equal(rag[2].view(np.uint32), want[2, 40:41].view(np.uint32))
    # PQ has no internal query encoding (encode_internal_vector returns None)
    with pytest.raises(qa.QmxError) as e:
        qa.new_raw_scorer_internal([1, 2], st)
    assert e.value.status == qa._ffi.ERR_NOT_SUPPORTED


@pytest.mark.parametrize("dist", [O.DOT, O.COSINE])
@pytest.mark.parametrize("nq", [1, 5, 32, 45])
def test_pq_lut_mfma_within_tolerance(qa, dist, nq):
    dim, chunk, ncent, n = 1536, 16, 256, 300
    rng, vecs, quant, opq = _setup(qa, dist, dim, chunk, n, ncent, seed=11 + nq, mfma=True)
    st = qa.EncodedVectorsPQ(opq.encode(vecs), quant)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    qpre = O.preprocess(dist, queries)
    scorer = qa.new_raw_scorer(queries, st)
    cen = opq.centroids.reshape(ncent, opq.m, chunk).astype(np.float64)
    for i in range(nq):
        got = scorer.encoded_query(i)
        want = opq.lut(qpre[i])
        sub = qpre[i].reshape(opq.m, chunk).astype(np.float64)
        scale = np.abs(sub[:, None, :] * cen.transpose(1, 0, 2)).sum(-1)        # [m][ncent] sum(abs(terms))
        assert np.all(np.abs(got.astype(np.float64) - want) <= 1e-5 * scale + 1e-30)
    ids = np.arange(n, dtype=np.uint32)
    got = scorer.score_points(ids)
    want = opq.score_points(qpre, ids)
    for i in range(nq):   # a score is a sum of m LUT entries: tolerance relative to sum(abs(entries))
        lut = np.abs(opq.lut(qpre[i]).astype(np.float64))
        scale = lut[np.arange(opq.m)[None, :], opq.codes[ids]].sum(-1)
        assert np.all(np.abs(got[i].astype(np.float64) - want[i]) <= 1e-5 * scale + 1e-30)


@pytest.mark.parametrize("dim,chunk,ncent,nq", [(1536, 16, 256, 300), (70, 16, 256, 130), (96, 8, 100, 33), (65, 2, 256, 1), (320, 32, 256, 129), (96, 3, 17, 260)])
def test_pq_lut_mfma_staged_in_lds_carries_the_bits_of_the_global_operand_kernel(qa, dim, chunk, ncent, nq):
    """pq_lut_mfma_lds_kernel (both operands staged in LDS, 128 queries per block) against pq_lut_mfma_kernel (option pq_lut_no_lds: every operand element read
    from global memory per instruction): the same instruction in the same k order - every LUT entry must carry the same bits, for ragged last chunks, query
    counts that do not fill a block and codebooks of fewer than 256 centroids; and both stay within 1e-5 of the exact-order LUT"""
    rng, vecs, quant, opq = _setup(qa, O.DOT, dim, chunk, 200, ncent, seed=dim + nq, mfma=True)
    st = qa.EncodedVectorsPQ(opq.encode(vecs), quant)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    qpre = O.preprocess(O.DOT, queries)
    staged = qa.new_raw_scorer(queries, st)
    qa.set_option("pq_lut_no_lds", 1)
    try:
        plain = qa.new_raw_scorer(queries, st)
    finally:
        qa.set_option("pq_lut_no_lds", -1)
    for i in sorted({0, 1, 31, 32, 127, 128, nq - 1} & set(range(nq))):
        a, b = staged.encoded_query(i), plain.encoded_query(i)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), i
        want = opq.lut(qpre[i]).astype(np.float64)
        m = opq.m
        sub = np.zeros((m, chunk)); cenp = np.zeros((ncent, m, chunk))
        sub.reshape(-1)[:dim] = qpre[i]
        cenp.reshape(ncent, -1)[:, :dim] = opq.centroids.reshape(ncent, dim)
        scale = np.abs(sub[:, None, :] * cenp.transpose(1, 0, 2)).sum(-1)
        assert np.all(np.abs(a.astype(np.float64).reshape(m, ncent) - want.reshape(m, ncent)) <= 1e-5 * scale + 1e-30)


def test_pq_error_bound_like_reference_tests(qa):
    # lib/quantization/tests/integration/test_pq.rs:14-55: dim 65, 513 vectors in [0,1), chunk 1, |pq - exact| < dim * 0.05
    rng = np.random.default_rng(42)
    dim, n = 65, 513
    vecs = rng.random((n, dim)).astype(np.float32)
    query = rng.random(dim).astype(np.float32)
    for dist in (O.DOT, O.EUCLID, O.MANHATTAN):
        cen = O.PqOracle.train(vecs, dim, 1, 256, iters=20)
        quant = qa.ProductQuantizer(dim, _dist(qa, dist), 1, cen)
        st = qa.EncodedVectorsPQ(quant.encode(vecs), quant)
        got = qa.new_raw_scorer(query, st).score_points(np.arange(n, dtype=np.uint32))[0]
        exact = np.array([O.similarity(O.F32, dist, query, vecs[i]) for i in range(n)])
        assert np.all(np.abs(got - exact) < dim * 0.05)


@pytest.mark.parametrize("dist", [O.COSINE, O.EUCLID])
def test_pq_brute_force_topk(qa, dist):
    n, dim, chunk, nq, top = 30000, 128, 8, 20, 10
    rng, vecs, quant, opq = _setup(qa, dist, dim, chunk, n, 256, seed=21 + dist)
    codes = quant.encode(vecs)
    opq.codes = codes
    st = qa.EncodedVectorsPQ(codes, quant)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    qpre = O.preprocess(dist, queries)
    pdel = rng.random(n) < 0.2
    st.set_deleted(pdel, None)
    got = qa.BatchFilteredSearcher(queries, st, top).peek_top_all()
    allq = opq.score_points(qpre[:4], np.arange(n))
    for i in range(4):
        s = allq[i].copy()
        s[pdel] = -np.inf
        order = np.argsort(-s, kind="stable")[:top]
        assert np.array_equal(got[i]["score"].view(np.uint32), s[order].view(np.uint32))
        kth = s[order[-1]]
        assert set(got[i]["idx"][got[i]["score"] > kth]) == set(order[s[order] > kth].tolist())
    # candidate list + score_bytes
    ids = rng.permutation(n).astype(np.uint32)[:999]
    got = qa.BatchFilteredSearcher(queries[:2], st, top).peek_top_iter(ids)
    for i in range(2):
        s = opq.score_points(qpre[i], ids)[0]
        s[pdel[ids]] = -np.inf
        order = np.argsort(-s, kind="stable")[:top]
        assert np.array_equal(got[i]["score"].view(np.uint32), s[order].view(np.uint32))
    sc = qa.new_raw_scorer(queries[:2], st)
    assert np.array_equal(sc.score_bytes(codes[:50]).view(np.uint32), sc.score_points(np.arange(50, dtype=np.uint32)).view(np.uint32))


@pytest.mark.parametrize("dim,chunk,ncent,n,threads", [(64, 8, 256, 3000, 1), (70, 16, 100, 1500, 3), (32, 1, 16, 500, 8), (48, 16, 256, 200, 1)])
def test_pq_train_kmeans_bit_exact(qa, dim, chunk, ncent, n, threads):
    """k-means on a given sample (kmeans.rs): centroids and per-chunk iteration counts equal the oracle's restatement,
    bit for bit, for any number of accumulation ranges (`max_kmeans_threads`)."""
    rng = np.random.default_rng(dim + n)
    centers = rng.standard_normal((20, dim)).astype(np.float32) * 2
    sample = (centers[rng.integers(0, 20, n)] + rng.standard_normal((n, dim)).astype(np.float32)).astype(np.float32)
    want, witers = O.PqOracle.train_ex(sample, dim, chunk, ncent, max_iters=25, accuracy=1e-5, threads=threads)
    got, giters = qa.pq_train(sample, dim, chunk, ncent, max_iterations=25, accuracy=1e-5, threads=threads)
    assert giters.tolist() == witers.tolist()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if n > ncent:       # a usable codebook: encoding with it beats encoding with the first-k init
        quant = qa.ProductQuantizer(dim, qa.Distance.Euclid, chunk, got)
        codes = quant.encode(sample)
        m = quant.m
        recon = np.concatenate([got[codes[:, c], c * chunk:min((c + 1) * chunk, dim)] for c in range(m)], axis=1)
        init = np.zeros_like(got)
        init[:] = sample[:ncent] if n >= ncent else 0
        codes0 = qa.ProductQuantizer(dim, qa.Distance.Euclid, chunk, init).encode(sample)
        recon0 = np.concatenate([init[codes0[:, c], c * chunk:min((c + 1) * chunk, dim)] for c in range(m)], axis=1)
        assert ((sample - recon) ** 2).sum() < ((sample - recon0) ** 2).sum()

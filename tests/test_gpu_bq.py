"""GPU parity: EncodedVectorsBin<u128> (binary quantization, Encoding::OneBit, QueryEncoding::SameAsStorage) through the
C-ABI against the CPU oracle: encode, encoded query, score_point, score_internal, ragged hop scoring, brute-force top-k,
HNSW walk, oversampled search + rescoring.  Everything here is integer work: BIT-EXACT.
Reference: lib/quantization/src/encoded_vectors_binary.rs; its tests lib/quantization/tests/integration/test_binary.rs.
"""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu

DIMS = [1, 8, 33, 65, 127, 128, 129, 3 * 129, 768, 1000, 1536, 4096]


@pytest.fixture(scope="module")
def qa():
    import qdrant_amd
    assert qdrant_amd.device_count() >= 1
    return qdrant_amd


def _dist(qa, d):
    return {O.COSINE: qa.Distance.Cosine, O.DOT: qa.Distance.Dot, O.EUCLID: qa.Distance.Euclid,
            O.MANHATTAN: qa.Distance.Manhattan}[d]


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("dist", [O.DOT, O.COSINE, O.EUCLID, O.MANHATTAN])
@pytest.mark.parametrize("dim", DIMS)
def test_bq_encode_and_scores_bit_exact(qa, dist, dim):
    n, nq = 600, 5
    rng = np.random.default_rng(dim * 7 + dist)
    vecs = O.preprocess(dist, rng.standard_normal((n, dim)).astype(np.float32))
    vecs[3, : min(dim, 4)] = 0.0                                     # > 0.0 only
    quant = qa.BinaryQuantizer(dim, _dist(qa, dist))
    obq = O.BqOracle(dist, dim)
    assert quant.quantized_vector_size() == obq.row_bytes and int(quant.invert) == obq.invert
    want_rows = obq.encode_rows(vecs)
    got_rows = quant.encode(vecs)
    assert np.array_equal(got_rows, want_rows)
    st = qa.EncodedVectorsBin(got_rows, quant)
    assert np.array_equal(st.get_quantized_vector([3, n - 1]), want_rows[[3, n - 1]])
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    qpre = O.preprocess(dist, queries)
    scorer = qa.new_raw_scorer(queries, st)
    for i in range(nq):
        assert np.array_equal(scorer.encoded_query(i), obq.encode(qpre[i])[0])
    ids = rng.permutation(n).astype(np.uint32)[:300]
    want = obq.score_points(qpre, ids)
    assert np.array_equal(_bits(scorer.score_points(ids)), _bits(want))
    a, b = ids[:64], ids[64:128]
    assert np.array_equal(_bits(scorer.score_internal(a, b)), _bits(obq.score_internal(a, b)))
    lists = [ids[:7], ids[7:40], ids[40:41], ids[:0], ids[41:60]]
    for r, (lo, hi), qi in zip(scorer.score_points_ragged(lists), [(0, 7), (7, 40), (40, 41), (0, 0), (41, 60)], range(5)):
        assert np.array_equal(_bits(r), _bits(want[qi, lo:hi]))
    # the stored row IS the internal query (encode_internal_vector :923-934)
    internal = qa.new_raw_scorer_internal([1, 2, n - 1], st)
    for i, pid in enumerate([1, 2, n - 1]):
        assert np.array_equal(internal.encoded_query(i), want_rows[pid])
    wi = np.stack([obq.score_internal(np.full(len(ids), pid), ids) for pid in [1, 2, n - 1]])
    assert np.array_equal(_bits(internal.score_points(ids)), _bits(wi))


@pytest.mark.parametrize("dist,invert", [(O.DOT, True), (O.COSINE, True), (O.EUCLID, False), (O.MANHATTAN, False)])
def test_bq_toggled_invert_like_the_reference_tests(qa, dist, invert):
    """test_binary.rs:77-127 (dot inverted), :238-292 (l1 not inverted): +-1 vectors, score == -+ dot exactly."""
    n, dim = 128, 3 * 129
    rng = np.random.default_rng(42)
    vecs = np.where(rng.uniform(-1, 1, (n, dim)) >= 0, 1.0, -1.0).astype(np.float32)
    query = np.where(rng.uniform(-1, 1, (1, dim)) >= 0, 1.0, -1.0).astype(np.float32)
    quant = qa.BinaryQuantizer(dim, _dist(qa, dist), invert=invert)
    st = qa.EncodedVectorsBin(quant.encode(vecs), quant)
    scorer = qa.new_raw_scorer(query * 3.0, st)                      # any positive scale: same bits (cosine normalises)
    got = scorer.score_points(np.arange(n, dtype=np.uint32))[0]
    dot = (vecs.astype(np.float64) @ query[0].astype(np.float64)).astype(np.float32)
    assert np.array_equal(got, -dot)
    obq = O.BqOracle(dist, dim, invert=invert)
    obq.encode_rows(vecs)
    assert np.array_equal(_bits(got), _bits(obq.score_points(query, np.arange(n))[0]))


@pytest.mark.parametrize("dist", [O.COSINE, O.EUCLID])
@pytest.mark.parametrize("nq,top", [(1, 10), (7, 100), (33, 5)])
def test_bq_brute_force_topk(qa, dist, nq, top):
    n, dim = 20000, 256
    rng = np.random.default_rng(nq + top)
    vecs = O.preprocess(dist, rng.standard_normal((n, dim)).astype(np.float32))
    quant = qa.BinaryQuantizer(dim, _dist(qa, dist))
    st = qa.EncodedVectorsBin(quant.encode(vecs), quant)
    deleted = rng.permutation(n)[:500]
    mask = np.zeros(n, dtype=bool)
    mask[deleted] = True
    st.set_deleted(mask)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    obq = O.BqOracle(dist, dim)
    obq.encode_rows(vecs)
    got = qa.BatchFilteredSearcher(queries, st, top).peek_top_all()
    live = np.setdiff1d(np.arange(n), deleted)
    want = obq.score_points(O.preprocess(dist, queries), live)
    for qi in range(nq):
        # integer scores tie heavily: the score list is pinned, the id set only above the boundary score
        ws = np.sort(want[qi])[::-1][:top]
        assert np.array_equal(_bits(got[qi]["score"]), _bits(ws))
        sure = set(live[want[qi] > ws[-1]].tolist())
        assert sure <= set(got[qi]["idx"].tolist())
        assert not (set(got[qi]["idx"].tolist()) & set(deleted.tolist()))
        by_id = dict(zip(live.tolist(), want[qi].tolist()))
        assert all(by_id[int(i)] == s for i, s in zip(got[qi]["idx"], got[qi]["score"]))


def test_bq_hnsw_walk_and_oversampled_rescoring(qa):
    """hnsw_quantized_search_test.rs flow with binary quantization: walk the graph with the BQ scorer (oversampled), rescore
    with the original vectors.  BQ scores are small integers: ties everywhere, so (as for Manhattan SQ) the walk is compared
    on what is pinned: returned scores are true BQ scores in descending order and the recall matches the oracle's walk."""
    n, dim, m, nq, top = 4000, 256, 8, 32, 10
    rng = np.random.default_rng(77)
    centers = rng.standard_normal((32, dim)).astype(np.float32)
    rows = O.preprocess(O.COSINE, (centers[rng.integers(0, 32, n)] + 0.6 * rng.standard_normal((n, dim))).astype(np.float32))
    queries = O.preprocess(O.COSINE, (centers[rng.integers(0, 32, nq)] + 0.6 * rng.standard_normal((nq, dim))).astype(np.float32))
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    g = O.Hnsw(st, m=m, ef_construct=64, seed=3, threads=0)
    graph = qa.GraphLayers.from_plain(g.export_plain())
    quant = qa.BinaryQuantizer(dim, qa.Distance.Cosine)
    enc = qa.EncodedVectorsBin(quant.encode(rows), quant)
    obq = O.BqOracle(O.COSINE, dim)
    obq.encode_rows(rows)
    scorer = qa.new_raw_scorer(queries, enc)
    got = graph.search(30, 64, scorer)
    want = g.search_bq(st, obq, queries, 30, 64)
    all_scores = obq.score_points(queries, np.arange(n))
    exact_bq = [set(np.argsort(-all_scores[i], kind="stable")[:30].tolist()) for i in range(nq)]
    rg = sum(len(set(r["idx"].tolist()) & e) for r, e in zip(got, exact_bq))
    rw = sum(len(set(r["idx"].tolist()) & e) for r, e in zip(want, exact_bq))
    assert abs(rg - rw) <= 0.05 * 30 * nq
    for i, r in enumerate(got):
        assert len(r) == 30 and np.all(np.diff(r["score"]) <= 0) and len(set(r["idx"].tolist())) == 30
        assert np.array_equal(_bits(r["score"]), _bits(all_scores[i][r["idx"]]))
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    raw = qa.new_raw_scorer(queries, vs)
    res = qa.search_quantized(scorer, raw, top, oversampling=3.0, rescore=True, graph=graph, hnsw_ef=64)
    exact = st.peek_top(queries, top)
    hit = sum(len(set(a["idx"].tolist()) & set(b["idx"].tolist())) for a, b in zip(res, exact))
    assert hit / (top * nq) > 0.5                                  # hnsw_quantized_search_test.rs asks for > 0.4
    for i, r in enumerate(res):
        w = st.score_points(queries[i:i + 1], r["idx"])[0]
        assert np.array_equal(_bits(r["score"]), _bits(w))


def test_bq_argument_errors(qa):
    quant = qa.BinaryQuantizer(100, qa.Distance.Dot)
    with pytest.raises(AssertionError):
        qa.EncodedVectorsBin(np.zeros((4, 13), dtype=np.uint8), quant)
    st = qa.EncodedVectorsBin(np.zeros((0, 16), dtype=np.uint8), quant)    # empty storage
    got = qa.BatchFilteredSearcher(np.ones((2, 100), dtype=np.float32), st, 5).peek_top_all()
    assert all(len(r) == 0 for r in got)


@pytest.mark.parametrize("encoding", [1, 2])                      # Encoding::TwoBits, Encoding::OneAndHalfBits
@pytest.mark.parametrize("dist", [O.DOT, O.EUCLID])
@pytest.mark.parametrize("dim", [1, 7, 64, 65, 129, 768])
@pytest.mark.parametrize("with_stats", [True, False])
def test_bq_two_bit_encodings(qa, encoding, dist, dim, with_stats):
    """encode_two_bits_vector / encode_one_and_half_bits_vector (encoded_vectors_binary.rs:570-672) with given VectorStats, the
    row sizes of :829-840, queries encoded the same way (SameAsStorage), scores = the one-bit metric over the longer rows with the
    original dim.  Bit-exact vs the oracle; brute-force top-k on top."""
    n, nq = 500, 4
    rng = np.random.default_rng(dim * 3 + encoding + dist)
    vecs = (rng.standard_normal((n, dim)) * rng.uniform(0.2, 2.0, dim) + rng.uniform(-0.5, 0.5, dim)).astype(np.float32)
    mean = vecs.mean(axis=0).astype(np.float32) if with_stats else None
    stddev = vecs.std(axis=0).astype(np.float32) if with_stats else None
    if with_stats and dim > 2:
        stddev[1] = 0.0                                              # sd < EPSILON: plain sign bit, no second bit
    quant = qa.BinaryQuantizer(dim, _dist(qa, dist), encoding=encoding, mean=mean, stddev=stddev)
    obq = O.BqOracle(dist, dim, encoding=encoding, mean=mean, stddev=stddev)
    ext = 2 * dim if encoding == 1 else (3 * dim + 1) // 2
    assert quant.quantized_vector_size() == obq.row_bytes == (max(ext, 1) + 127) // 128 * 16
    want_rows = obq.encode_rows(vecs)
    got_rows = quant.encode(vecs)
    assert np.array_equal(got_rows, want_rows)
    st = qa.EncodedVectorsBin(got_rows, quant)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    scorer = qa.new_raw_scorer(queries, st)
    for i in range(nq):
        assert np.array_equal(scorer.encoded_query(i), obq.encode(queries[i])[0])
    ids = rng.permutation(n).astype(np.uint32)[:200]
    want = obq.score_points(queries, ids)
    assert np.array_equal(_bits(scorer.score_points(ids)), _bits(want))
    got = qa.BatchFilteredSearcher(queries, st, 5).peek_top_all()
    allsc = obq.score_points(queries, np.arange(n))
    for qi in range(nq):
        assert np.array_equal(_bits(got[qi]["score"]), _bits(np.sort(allsc[qi])[::-1][:5]))


@pytest.mark.parametrize("encoding", [0, 1, 2])
@pytest.mark.parametrize("qenc,bits", [(1, 4), (2, 8)])
@pytest.mark.parametrize("dist,dim", [(O.DOT, 33), (O.COSINE, 129), (O.EUCLID, 768), (O.MANHATTAN, 1000), (O.DOT, 1536)])
def test_bq_scalar_query_encodings_bit_exact(qa, dist, dim, qenc, bits, encoding):
    """QueryEncoding::Scalar4bits / Scalar8bits (encoded_vectors_binary.rs:692-756 encode, :337-409 xor_popcnt_scalar, :783-810
    calculate_metric): the encoded query's bytes, score_points, ragged hop scoring, brute-force top-k; stored <-> stored scores
    stay one-bit (score_internal :892-917)."""
    n, nq = 700, 6
    rng = np.random.default_rng(dim * 3 + bits + encoding)
    vecs = O.preprocess(dist, rng.standard_normal((n, dim)).astype(np.float32))
    mean = vecs.mean(axis=0).astype(np.float32) if encoding else None
    stddev = vecs.std(axis=0).astype(np.float32) if encoding else None
    quant = qa.BinaryQuantizer(dim, _dist(qa, dist), encoding=encoding, mean=mean, stddev=stddev, query_encoding=qenc)
    obq = O.BqOracle(dist, dim, encoding=encoding, mean=mean, stddev=stddev)
    rows = obq.encode_rows(vecs)
    st = qa.EncodedVectorsBin(quant.encode(vecs), quant)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    queries[1] = 0.0                                                   # delta = 0
    queries[2, : min(dim, 3)] *= 50.0                                  # one dominant value: most others land mid-range
    qpre = O.preprocess(dist, queries)
    scorer = qa.new_raw_scorer(queries, st)
    want_q = obq.encode_scalar_queries(qpre, bits)
    for i in range(nq):
        assert np.array_equal(scorer.encoded_query(i), want_q[i])
    ids = rng.permutation(n).astype(np.uint32)[:400]
    want = obq.score_points_scalar(qpre, ids, bits)
    assert np.array_equal(_bits(scorer.score_points(ids)), _bits(want))
    lists = [ids[:9], ids[9:50], ids[:0], ids[50:51], ids[51:90], ids[90:130]]
    for r, (lo, hi), qi in zip(scorer.score_points_ragged(lists), [(0, 9), (9, 50), (0, 0), (50, 51), (51, 90), (90, 130)], range(6)):
        assert np.array_equal(_bits(r), _bits(want[qi, lo:hi]))
    a, b = ids[:64], ids[64:128]
    assert np.array_equal(_bits(scorer.score_internal(a, b)), _bits(obq.score_internal(a, b)))
    internal = qa.new_raw_scorer_internal([5, n - 1], st)
    assert np.array_equal(internal.encoded_query(1), rows[n - 1])
    wi = np.stack([obq.score_internal(np.full(len(ids), pid), ids) for pid in [5, n - 1]])
    assert np.array_equal(_bits(internal.score_points(ids)), _bits(wi))
    # brute-force top-k over all rows, 6 queries (and a 20-query batch: two tiles)
    full = obq.score_points_scalar(qpre, np.arange(n), bits)
    for res, sc in zip(qa.BatchFilteredSearcher(queries, st, 10).peek_top_all(), full):
        assert np.array_equal(_bits(res["score"]), _bits(np.sort(sc)[::-1][:10]))
        assert np.array_equal(_bits(sc[res["idx"]]), _bits(res["score"]))
    many = np.tile(queries, (4, 1))[:20]
    for i, res in enumerate(qa.BatchFilteredSearcher(many, st, 7).peek_top_all()):
        assert np.array_equal(_bits(res["score"]), _bits(np.sort(full[i % nq])[::-1][:7]))


def test_bq_scalar_query_hnsw_walk_and_rescoring(qa):
    """The asymmetric query through the device HNSW walk and the one-call oversampled search + rescoring: returned scores are
    true scalar-query BQ scores in descending order; an 8-bit query finds more of the exact neighbours than the 1-bit query."""
    n, dim, m, nq, top = 4000, 256, 8, 32, 10
    rng = np.random.default_rng(78)
    centers = rng.standard_normal((32, dim)).astype(np.float32)
    rows = O.preprocess(O.COSINE, (centers[rng.integers(0, 32, n)] + 0.6 * rng.standard_normal((n, dim))).astype(np.float32))
    queries = O.preprocess(O.COSINE, (centers[rng.integers(0, 32, nq)] + 0.6 * rng.standard_normal((nq, dim))).astype(np.float32))
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    g = O.Hnsw(st, m=m, ef_construct=64, seed=3, threads=0)
    graph = qa.GraphLayers.from_plain(g.export_plain())
    obq = O.BqOracle(O.COSINE, dim)
    obq.encode_rows(rows)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    raw = qa.new_raw_scorer(queries, vs)
    exact = st.peek_top(queries, top)
    hits = {}
    for qenc, bits in [(0, 1), (2, 8)]:
        quant = qa.BinaryQuantizer(dim, qa.Distance.Cosine, query_encoding=qenc)
        enc = qa.EncodedVectorsBin(quant.encode(rows), quant)
        scorer = qa.new_raw_scorer(queries, enc)
        got = graph.search(30, 64, scorer)
        all_scores = obq.score_points(queries, np.arange(n)) if bits == 1 else obq.score_points_scalar(queries, np.arange(n), bits)
        for i, r in enumerate(got):
            assert len(r) == 30 and np.all(np.diff(r["score"]) <= 0) and len(set(r["idx"].tolist())) == 30
            assert np.array_equal(_bits(r["score"]), _bits(all_scores[i][r["idx"]]))
        res = qa.search_quantized(scorer, raw, top, oversampling=3.0, rescore=True, graph=graph, hnsw_ef=64)
        hits[bits] = sum(len(set(a["idx"].tolist()) & set(b["idx"].tolist())) for a, b in zip(res, exact))
    assert hits[8] >= hits[1] and hits[8] / (top * nq) > 0.5


def test_vector_stats_on_device_equals_the_oracle(qa):
    """qmx_vector_stats (VectorStats::build: the statistics of the 2-bit / 1.5-bit encodings) == the oracle's streaming Welford, bit for bit, for
    row counts around the kernel's 8-row unroll; BinaryQuantizer.fit encodes like a quantizer given those statistics."""
    rng = np.random.default_rng(11)
    for n, dim in ((0, 5), (1, 7), (7, 64), (8, 65), (1003, 130), (20000, 96)):
        x = (rng.standard_normal((n, dim)) * 2.0 + rng.standard_normal(dim)).astype(np.float32)
        want = O.vector_stats(x)
        got = qa.vector_stats(x, dim)
        for g, w in zip(got, want):
            assert np.array_equal(g.view(np.uint32), w.view(np.uint32))
    quant = qa.BinaryQuantizer.fit(x, dim, qa.Distance.Dot, 1)
    ref = qa.BinaryQuantizer(dim, qa.Distance.Dot, encoding=1, mean=want[2], stddev=want[3])
    assert np.array_equal(quant.encode(x[:200]), ref.encode(x[:200]))
    obq = O.BqOracle(O.DOT, dim, encoding=1, mean=want[2], stddev=want[3])
    assert np.array_equal(quant.encode(x[:200]), obq.encode_rows(x[:200]))

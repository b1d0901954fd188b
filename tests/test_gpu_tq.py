"""GPU parity: EncodedVectorsTQ (TurboQuant, lib/quantization/src/turboquant/ behind encoded_vectors_tq.rs) through the C-ABI against the CPU oracle
(oracle/qdrant_oracle_tq.c; what pins the oracle: tests/test_oracle_tq.py).  The query rotation runs in f64 with the reference's operation order,
everything after it is integer arithmetic: the encoded query, every score and every top-k list must equal the oracle's BIT FOR BIT."""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu

BITS = [O.TQ_BITS4, O.TQ_BITS2, O.TQ_BITS1_5, O.TQ_BITS1]


@pytest.fixture(scope="module")
def qa():
    import qdrant_amd
    assert qdrant_amd.device_count() >= 1
    return qdrant_amd


def _dist(qa, d):
    return {O.COSINE: qa.Distance.Cosine, O.DOT: qa.Distance.Dot, O.EUCLID: qa.Distance.Euclid, O.MANHATTAN: qa.Distance.Manhattan}[d]


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _world(qa, distance, dim, bits, n, seed, unpadded=False):
    rng = np.random.default_rng(seed)
    vecs = O.preprocess(distance, rng.uniform(-1.0, 1.0, (n, dim)).astype(np.float32))
    otq = O.TqOracle(distance, dim, bits, rotation_unpadded=unpadded)
    rows = otq.encode_rows(vecs)
    quant = qa.TurboQuantizer(dim, _dist(qa, distance), bits, rotation_unpadded=unpadded)
    assert quant.quantized_vector_size() == otq.row_bytes and quant.padded_dim == otq.padded_dim and quant.invert == otq.invert
    st = qa.EncodedVectorsTQ(rows, quant)
    return rng, vecs, otq, rows, st


@pytest.mark.parametrize("bits", BITS)
@pytest.mark.parametrize("distance", [O.DOT, O.COSINE, O.EUCLID])
@pytest.mark.parametrize("dim", [16, 65, 128, 384, 700, 768, 1536])
def test_tq_scores_bit_exact(qa, distance, dim, bits):
    n, nq = 300, 5
    rng, vecs, otq, rows, st = _world(qa, distance, dim, bits, n, seed=dim * 11 + bits * 3 + distance)
    assert np.array_equal(st.get_quantized_vector([0, 7, n - 1]), rows[[0, 7, n - 1]])
    queries = rng.uniform(-1.0, 1.0, (nq, dim)).astype(np.float32)
    queries[3] = 0.0                                                      # test_tq_zero_query_*
    scorer = qa.new_raw_scorer(queries, st)
    ids = rng.permutation(n).astype(np.uint32)[:200]
    got = scorer.score_points(ids)
    want = otq.score_points(O.preprocess(distance, queries), ids)
    assert np.array_equal(_bits(got), _bits(want))
    # ragged (HNSW hop) scoring == dense scoring
    rag = scorer.score_points_ragged([ids[:9], ids[9:40], ids[40:41], ids[:0], ids[41:60]])
    assert np.array_equal(_bits(rag[1]), _bits(want[1, 9:40])) and np.array_equal(_bits(rag[4]), _bits(want[4, 41:60]))
    # score_symmetric
    a, b = ids[:64], ids[64:128]
    assert np.array_equal(_bits(scorer.score_internal(a, b)), _bits(otq.score_internal(a, b)))
    # no internal query encoding (encode_internal_vector -> None)
    with pytest.raises(qa.QmxError) as e:
        qa.new_raw_scorer_internal([1, 2], st)
    assert e.value.status == qa._ffi.ERR_NOT_SUPPORTED


@pytest.mark.parametrize("bits", BITS)
@pytest.mark.parametrize("distance", [O.DOT, O.EUCLID])
def test_tq_brute_force_topk_and_deleted(qa, distance, bits):
    n, dim, nq = 20000, 256, 19
    rng, vecs, otq, rows, st = _world(qa, distance, dim, bits, n, seed=1000 + bits + distance)
    queries = rng.uniform(-1.0, 1.0, (nq, dim)).astype(np.float32)
    deleted = rng.random(n) < 0.2
    st.set_deleted(deleted)
    res = qa.BatchFilteredSearcher(queries, st, 10).peek_top_all()
    sample = rng.permutation(n)[:3000]
    want = otq.score_points(O.preprocess(distance, queries), np.arange(n)) if n <= 20000 else None
    for qi, r in enumerate(res):
        sc = want[qi].copy()
        sc[deleted] = -np.inf
        assert not deleted[r["idx"]].any()
        assert np.array_equal(_bits(r["score"]), _bits(np.sort(sc)[::-1][:10]))
        assert np.array_equal(_bits(want[qi][r["idx"]]), _bits(r["score"]))


@pytest.mark.parametrize("bits", BITS)
@pytest.mark.parametrize("distance", [O.DOT, O.COSINE, O.EUCLID])
def test_tq_encode_on_device_byte_exact(qa, distance, bits):
    """TurboQuantizer::quantize on the device: codes and extras equal the oracle's bytes (rotation, rescale, the f64 sums in the reference's order)"""
    for dim, unpadded in ((65, False), (384, False), (700, False), (1536, False), (100, True)):
        if unpadded and bits == O.TQ_BITS1_5:
            continue
        rng = np.random.default_rng(dim + bits)
        vecs = O.preprocess(distance, rng.uniform(-1.0, 1.0, (40, dim)).astype(np.float32))
        vecs[7] = 0.0                                                         # zero vector: the length-rescale guard, the cosine centroid-norm guard
        otq = O.TqOracle(distance, dim, bits, rotation_unpadded=unpadded)
        quant = qa.TurboQuantizer(dim, _dist(qa, distance), bits, rotation_unpadded=unpadded)
        assert np.array_equal(quant.encode(vecs), otq.encode_rows(vecs))


def test_tq_rotation_kernels_agree(qa):
    """The rotation runs one vector per wave where the rotated length is a multiple of 16 (up to 1024), 32 (2048) or 64 (4096) coordinates, one block per
    vector otherwise (option tq_rotate_block forces it): the same adds in the same order - the encoded rows of both are the oracle's bytes."""
    for dim, bits, unpadded in ((768, O.TQ_BITS4, False), (1024, O.TQ_BITS2, False), (1536, O.TQ_BITS4, False), (3072, O.TQ_BITS1, False), (96, O.TQ_BITS4, True),
                                (80, O.TQ_BITS4, True), (72, O.TQ_BITS4, True), (2080, O.TQ_BITS2, False)):
        rng = np.random.default_rng(dim)
        vecs = rng.uniform(-1.0, 1.0, (70, dim)).astype(np.float32)
        otq = O.TqOracle(O.DOT, dim, bits, rotation_unpadded=unpadded)
        want = otq.encode_rows(vecs)
        quant = qa.TurboQuantizer(dim, qa.Distance.Dot, bits, rotation_unpadded=unpadded)
        assert np.array_equal(quant.encode(vecs), want), (dim, "wave")
        qa.set_option("tq_rotate_block", 1)
        try:
            assert np.array_equal(quant.encode(vecs), want), (dim, "block")
        finally:
            qa.set_option("tq_rotate_block", -1)


def test_tq_oversampled_search_with_rescoring(qa):
    """the quantized stage + postprocess_search_result over a TQ storage (qmx_search_quantized): after rescoring the scores are the exact ones"""
    n, dim = 30000, 128
    rng, vecs, otq, rows, st = _world(qa, O.COSINE, dim, O.TQ_BITS4, n, seed=9)
    vs = qa.VectorStorage(vecs, qa.Distance.Cosine)
    queries = rng.uniform(-1.0, 1.0, (8, dim)).astype(np.float32)
    exact = O.DenseStorage(O.F32, O.COSINE, vecs).peek_top(queries, 10)
    got = qa.search_quantized(qa.new_raw_scorer(queries, st), qa.new_raw_scorer(queries, vs), top=10, oversampling=3.0, rescore=True)
    hits = 0
    for g, e in zip(got, exact):
        hits += len(set(g["idx"].tolist()) & set(e["idx"].tolist()))
        pos = {int(i): k for k, i in enumerate(e["idx"])}
        for i, s_ in zip(g["idx"], g["score"]):
            if int(i) in pos:
                assert np.float32(s_).view(np.uint32) == e["score"][pos[int(i)]].view(np.uint32)
    assert hits >= 0.9 * 80


def test_tq_unpadded_rotation_and_hnsw_walk(qa):
    """TQRotation::Unpadded (the TQ-as-datatype storages) and the HNSW walk with the TQ scorer == the oracle's walk"""
    n, dim = 3000, 100
    rng, vecs, otq, rows, st = _world(qa, O.COSINE, dim, O.TQ_BITS4, n, seed=77, unpadded=True)
    queries = rng.uniform(-1.0, 1.0, (16, dim)).astype(np.float32)
    ids = np.arange(200, dtype=np.uint32)
    scorer = qa.new_raw_scorer(queries, st)
    want = otq.score_points(O.preprocess(O.COSINE, queries), ids)
    assert np.array_equal(_bits(scorer.score_points(ids)), _bits(want))
    dense = O.DenseStorage(O.F32, O.COSINE, vecs)
    g = O.Hnsw(dense, m=8, ef_construct=48, seed=5)
    graph = qa.GraphLayers.from_plain(g.export_plain())
    got = graph.search(10, 64, scorer)
    want = g.search_tq(dense, otq, O.preprocess(O.COSINE, queries), 10, 64)                # the oracle walks the same graph with its TQ scorer
    for gq, wq in zip(got, want):
        assert gq["idx"].tolist() == wq["idx"].tolist() and np.array_equal(_bits(gq["score"]), _bits(wq["score"]))
    full = otq.score_points(O.preprocess(O.COSINE, queries), np.arange(n))
    for qi, r in enumerate(got):
        assert len(r) == 10 and np.all(np.diff(r["score"]) <= 0)
        assert np.array_equal(_bits(full[qi][r["idx"]]), _bits(r["score"]))                # true TQ scores of the returned points
        best = np.sort(full[qi])[::-1][:10]
        assert len(np.intersect1d(_bits(r["score"]), _bits(best))) >= 5                    # a walk, not a scan: most of the true top-10


def test_tq_argument_errors(qa):
    big = qa.TurboQuantizer(100000, qa.Distance.Dot, O.TQ_BITS4)
    with pytest.raises(qa.QmxError):                                                          # the rotation runs in LDS: padded dim <= 8192
        qa.EncodedVectorsTQ(np.zeros((2, big.quantized_vector_size()), dtype=np.uint8), big)
    # the fit and the statistics refuse what they cannot do, loudly
    import ctypes as C
    F = qa._ffi
    buf = np.zeros(64, dtype=np.float32)
    p = qa.TurboQuantizer(64, qa.Distance.Dot, O.TQ_BITS4).params()
    assert F.lib().qmx_tq_fit_plus(0, int(qa.Distance.Dot), 64, None, F.ptr(buf), 1, F.ptr(buf), F.ptr(buf)) == F.ERR_BAD_ARG
    assert F.lib().qmx_tq_fit_plus(0, int(qa.Distance.Dot), 64, C.byref(p), None, 1, F.ptr(buf), F.ptr(buf)) == F.ERR_BAD_ARG
    assert F.lib().qmx_tq_fit_plus(99, int(qa.Distance.Dot), 64, C.byref(p), F.ptr(buf), 1, F.ptr(buf), F.ptr(buf)) != F.OK
    assert F.lib().qmx_vector_stats(0, None, 3, 64, None, None, F.ptr(buf), F.ptr(buf)) == F.ERR_BAD_ARG
    assert F.lib().qmx_vector_stats(0, F.ptr(buf), 1, 64, None, None, None, F.ptr(buf)) == F.ERR_BAD_ARG
    with pytest.raises(qa.QmxError) as e:                                                     # a TQ build needs the original vectors
        rows = qa.TurboQuantizer(64, qa.Distance.Dot, O.TQ_BITS4).encode(np.ones((8, 64), dtype=np.float32))
        qa.GraphLayers.build(qa.EncodedVectorsTQ(rows, qa.TurboQuantizer(64, qa.Distance.Dot, O.TQ_BITS4)), m=4, ef_construct=8)
    assert e.value.status == F.ERR_NOT_SUPPORTED


@pytest.mark.parametrize("bits", BITS)
@pytest.mark.parametrize("distance", [O.DOT, O.COSINE, O.EUCLID])
def test_tq_plus_mode_bit_exact(qa, distance, bits):
    """TQMode::Plus with the storage's error correction (shift / scale per rotated coordinate, fitted here from exact quantiles of the data):
    rows encoded on the device, asymmetric scores (16-bit query planes for 1-bit storages), score_symmetric_ec - all equal to the oracle's bits."""
    for dim in (65, 128, 384):
        n, nq = 260, 4
        rng = np.random.default_rng(dim * 5 + bits + distance)
        raw = rng.uniform(-1.0, 1.0, (n, dim)).astype(np.float32)
        raw[:, : dim // 8] += 1.5                                         # anisotropic: the correction has something to do
        vecs = O.preprocess(distance, raw)
        shift, scale = O.tq_plus_fit(distance, dim, bits, vecs)
        otq = O.TqOracle(distance, dim, bits, shift=shift, scale=scale)
        quant = qa.TurboQuantizer(dim, _dist(qa, distance), bits, shift=shift, scale=scale)
        assert quant.quantized_vector_size() == otq.row_bytes
        rows = otq.encode_rows(vecs)
        assert np.array_equal(quant.encode(vecs), rows)
        st = qa.EncodedVectorsTQ(rows, quant)
        assert np.array_equal(st.get_quantized_vector([0, n - 1]), rows[[0, n - 1]])
        queries = rng.uniform(-1.0, 1.0, (nq, dim)).astype(np.float32)
        scorer = qa.new_raw_scorer(queries, st)
        ids = rng.permutation(n).astype(np.uint32)[:200]
        assert np.array_equal(_bits(scorer.score_points(ids)), _bits(otq.score_points(O.preprocess(distance, queries), ids)))
        a, b = ids[:64], ids[64:128]
        assert np.array_equal(_bits(scorer.score_internal(a, b)), _bits(otq.score_internal(a, b)))
        res = qa.BatchFilteredSearcher(queries, st, 5).peek_top_all()
        full = otq.score_points(O.preprocess(distance, queries), np.arange(n))
        for qi, r in enumerate(res):
            assert np.array_equal(_bits(r["score"]), _bits(np.sort(full[qi])[::-1][:5]))


@pytest.mark.parametrize("bits", BITS)
@pytest.mark.parametrize("distance", [O.DOT, O.COSINE, O.EUCLID])
def test_tq_plus_fit_on_device_equals_the_oracle(qa, distance, bits):
    """qmx_tq_fit_plus (rotation, length rescale, one pair of 7-marker P-square estimators per rotated coordinate, shift / scale) == the oracle's
    restatement of the reference's first pass, bit for bit - f64 arithmetic in the reference's operation order incl. the fused desired-position
    update of its AVX2 path; then the fitted quantizer encodes and scores like the oracle's TQ+ quantizer with the same parameters."""
    for dim, n in ((65, 300), (128, 2048), (384, 4)):
        rng = np.random.default_rng(dim * 7 + bits * 3 + distance)
        raw = rng.standard_normal((n, dim)).astype(np.float32)
        raw[:, : dim // 8] += 1.5
        raw[:, dim // 8] = 0.25 if distance != O.COSINE else raw[:, dim // 8]             # a constant coordinate before the rotation
        if n > 10:
            raw[5] = 0.0                                                                   # zero vector: no rescale, zeros pushed
        vecs = O.preprocess(distance, raw)
        want_shift, want_scale = O.tq_plus_fit_p2(distance, dim, bits, vecs)
        quant = qa.TurboQuantizer.fit_plus(vecs, dim, _dist(qa, distance), bits)
        assert np.array_equal(_bits(quant.shift), _bits(want_shift)) and np.array_equal(_bits(quant.scale), _bits(want_scale))
        assert quant.plus_mode and np.all(np.isfinite(quant.scale)) and np.all(quant.scale > 0)
    # the fitted quantizer is a TQ+ quantizer like any other
    otq = O.TqOracle(distance, dim, bits, shift=want_shift, scale=want_scale)
    data = O.preprocess(distance, rng.standard_normal((50, dim)).astype(np.float32))
    assert np.array_equal(quant.encode(data), otq.encode_rows(data))
    # no sample: identity parameters (quantile.rs:151-153)
    empty = qa.TurboQuantizer.fit_plus(np.zeros((0, 32), dtype=np.float32), 32, _dist(qa, distance), bits)
    assert np.all(empty.shift == 0.0) and np.all(empty.scale == 1.0)


@pytest.mark.parametrize("bits", [O.TQ_BITS4, O.TQ_BITS1])
def test_tq_large_top_many_queries_and_id_lists(qa, bits):
    """top beyond one wave list (multi-pass selection), 40 queries (two matrix-core tiles, the second one ragged), an id list with a deleted
    flag set: scores and order == the oracle's."""
    n, dim, nq, top = 6000, 200, 40, 300
    rng, vecs, otq, rows, st = _world(qa, O.DOT, dim, bits, n, seed=4242 + bits)
    queries = rng.uniform(-1.0, 1.0, (nq, dim)).astype(np.float32)
    deleted = rng.random(n) < 0.1
    st.set_deleted(deleted)
    full = otq.score_points(O.preprocess(O.DOT, queries), np.arange(n))
    res = qa.BatchFilteredSearcher(queries, st, top).peek_top_all()
    for qi, r in enumerate(res):
        sc = full[qi].copy()
        sc[deleted] = -np.inf
        assert len(r) == top and np.array_equal(_bits(r["score"]), _bits(np.sort(sc)[::-1][:top]))
        assert not deleted[r["idx"]].any() and np.array_equal(_bits(full[qi][r["idx"]]), _bits(r["score"]))
    ids = rng.permutation(n).astype(np.uint32)[:1500]
    res = qa.BatchFilteredSearcher(queries, st, 25).peek_top_iter(ids)
    for qi, r in enumerate(res):
        sc = full[qi][ids].copy()
        sc[deleted[ids]] = -np.inf
        assert np.array_equal(_bits(r["score"]), _bits(np.sort(sc)[::-1][:25]))


# ---------------------------------------------------------------------------------------------------------------------------------
# Distance::Manhattan (DistanceType::L1): no integer kernel in the reference - score_precomputed dequantises the row, rotates it back and sums
# |q - v| (turboquant/quantization.rs:596-607), score_symmetric does it with the difference of two rows (:429-440); tq_l1.hip
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("bits,plus,unpadded,dim", [(O.TQ_BITS4, False, False, 96), (O.TQ_BITS4, False, False, 768), (O.TQ_BITS2, False, True, 100),
                                                     (O.TQ_BITS1, False, False, 65), (O.TQ_BITS1_5, False, False, 64), (O.TQ_BITS4, True, False, 128),
                                                     (O.TQ_BITS2, True, False, 70)])
def test_tq_manhattan_scores_bit_exact(qa, bits, plus, unpadded, dim):
    n, nq = 500, 11
    rng = np.random.default_rng(dim * 7 + bits)
    vecs = rng.uniform(-1.0, 1.0, (n, dim)).astype(np.float32)          # ManhattanMetric::preprocess is the identity
    vecs[5] = 0.0
    shift = scale = None
    if plus:
        shift, scale = O.tq_plus_fit(O.MANHATTAN, dim, bits, vecs)
    otq = O.TqOracle(O.MANHATTAN, dim, bits, rotation_unpadded=unpadded, shift=shift, scale=scale)
    rows = otq.encode_rows(vecs)
    quant = qa.TurboQuantizer(dim, qa.Distance.Manhattan, bits, rotation_unpadded=unpadded, shift=shift, scale=scale)
    assert quant.quantized_vector_size() == otq.row_bytes and quant.invert and otq.invert
    assert np.array_equal(quant.encode(vecs), rows)                      # quantize: the scaling factor of L1 rows is the bare l2 length
    st = qa.EncodedVectorsTQ(rows, quant)
    queries = rng.uniform(-1.0, 1.0, (nq, dim)).astype(np.float32)
    queries[3] = 0.0
    scorer = qa.new_raw_scorer(queries, st)
    ids = rng.permutation(n).astype(np.uint32)[:200]
    want = otq.score_points(queries, ids)
    assert np.array_equal(_bits(scorer.score_points(ids)), _bits(want))
    assert (want <= 0).all()                                             # `invert`: the negated distance
    rag = scorer.score_points_ragged([ids[:9], ids[9:40], ids[40:41], ids[:0], ids[41:60]] + [ids[:1]] * (nq - 5))
    assert np.array_equal(_bits(rag[1]), _bits(want[1, 9:40])) and np.array_equal(_bits(rag[4]), _bits(want[4, 41:60]))
    a, b = ids[:64], ids[64:128]
    assert np.array_equal(_bits(scorer.score_internal(a, b)), _bits(otq.score_internal(a, b)))
    # brute force with deleted points: BatchFilteredSearcher over the L1 scorer
    deleted = rng.random(n) < 0.2
    st.set_deleted(deleted)
    res = qa.BatchFilteredSearcher(queries, st, 10).peek_top_all()
    full = otq.score_points(queries, np.arange(n))
    for qi, r in enumerate(res):
        sc = full[qi].copy()
        sc[deleted] = -np.inf
        assert not deleted[r["idx"]].any()
        assert np.array_equal(_bits(r["score"]), _bits(np.sort(sc)[::-1][:10]))
        assert np.array_equal(_bits(full[qi][r["idx"]]), _bits(r["score"]))
    # the walk THROUGH this scorer (what a Manhattan collection with TurboQuant searches its graph with): every hop dequantises and rotates back its
    # candidates (tq_l1_policy.hpp) - the oracle's walk with its L1 scorer, ids and score bits; rotations that are not a multiple of 16 coordinates: refused
    st.set_deleted(None)
    flags = O.DenseStorage(O.F32, O.MANHATTAN, vecs)
    og = O.Hnsw(flags, m=8, ef_construct=32)
    graph = qa.GraphLayers.from_plain(og.export_plain())
    rot_dim = dim if unpadded else otq.padded_dim
    if rot_dim % 16 == 0:
        for top, ef in ((5, 32), (10, 64), (3, 600)):
            want = og.search_tq(flags, otq, queries, top, ef)
            got = graph.search(top, ef, scorer)
            for g_, w_ in zip(got, want):
                assert g_["idx"].tolist() == w_["idx"].tolist() and np.array_equal(_bits(g_["score"]), _bits(w_["score"]))
    else:
        with pytest.raises(qa.QmxError) as e:
            graph.search(5, 32, scorer)
        assert e.value.status == qa._ffi.ERR_NOT_SUPPORTED


def test_tq_manhattan_oversampled_search_with_rescoring(qa):
    """qmx_search_quantized over a TQ-L1 storage: the quantized stage ranks by the dequantised L1, the rescoring gives the exact Manhattan scores"""
    n, dim = 20000, 128
    rng = np.random.default_rng(77)
    vecs = rng.uniform(-1.0, 1.0, (n, dim)).astype(np.float32)
    quant = qa.TurboQuantizer(dim, qa.Distance.Manhattan, O.TQ_BITS4)
    st = qa.EncodedVectorsTQ(quant.encode(vecs), quant)
    vs = qa.VectorStorage(vecs, qa.Distance.Manhattan)
    queries = rng.uniform(-1.0, 1.0, (8, dim)).astype(np.float32)
    exact = O.DenseStorage(O.F32, O.MANHATTAN, vecs).peek_top(queries, 10)
    got = qa.search_quantized(qa.new_raw_scorer(queries, st), qa.new_raw_scorer(queries, vs), top=10, oversampling=3.0, rescore=True)
    hits = 0
    for g, e in zip(got, exact):
        hits += len(set(g["idx"].tolist()) & set(e["idx"].tolist()))
        pos = {int(i): k for k, i in enumerate(e["idx"])}
        for i, s_ in zip(g["idx"], g["score"]):
            if int(i) in pos:
                assert np.float32(s_).view(np.uint32) == e["score"][pos[int(i)]].view(np.uint32)
    assert hits >= 0.85 * 80

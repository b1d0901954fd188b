"""GPU parity: Metric<f16> and Metric<u8> storages through the C-ABI against the CPU oracle.

u8 is integer work: scores must be BIT-EXACT, in both conversion orders the reference has
(AVX2 lane order = the live x86 path, metric_uint/avx2/*.rs; scalar order, metric_uint/simple_*.rs).
f16 is floating point: elements and the cast query are bit-exact (IEEE RNE), scores agree within
1e-5 relative (north_star tolerance), measured against sum(abs(terms)) because a dot product of
near-orthogonal vectors has no meaningful relative error (the reference's own f16 SIMD-vs-scalar
tests allow 5e-4, metric_f16/avx/dot.rs:120-124).
"""
import json
import os

import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu
REL = 1e-5
G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))


@pytest.fixture(scope="module")
def qa():
    import qdrant_amd
    assert qdrant_amd.device_count() >= 1
    return qdrant_amd


def _dist(qa, d):
    return {O.COSINE: qa.Distance.Cosine, O.DOT: qa.Distance.Dot, O.EUCLID: qa.Distance.Euclid,
            O.MANHATTAN: qa.Distance.Manhattan}[d]


def _scale_f16(dist, q16, rows16, ids):
    """sum(abs(terms)) per (query, row): the magnitude the 1e-5 tolerance is relative to."""
    q = O.f16_to_f32(q16).astype(np.float64)
    v = O.f16_to_f32(rows16[ids]).astype(np.float64)
    if dist in (O.DOT, O.COSINE):
        return np.abs(q[:, None, :] * v[None, :, :]).sum(-1)
    if dist == O.EUCLID:
        return ((q[:, None, :] - v[None, :, :]) ** 2).sum(-1)
    return np.abs(q[:, None, :] - v[None, :, :]).sum(-1)


@pytest.mark.parametrize("dist", [O.DOT, O.COSINE, O.EUCLID, O.MANHATTAN])
@pytest.mark.parametrize("dim", [5, 16, 24, 32, 40, 64, 100, 288, 768, 1000])
def test_f16_score_points(qa, dist, dim):
    rng = np.random.default_rng(dim * 11 + dist)
    n, nq = 600, 5
    raw = rng.standard_normal((n, dim)).astype(np.float32)
    rows16 = O.to_f16(O.preprocess(dist, raw))          # normalise in f32, THEN cast (simple_cosine.rs:60-87)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    st = qa.VectorStorage(rows16.view(np.float16), _dist(qa, dist), qa.VectorStorageDatatype.Float16)
    scorer = qa.new_raw_scorer(queries, st)
    ost = O.DenseStorage(O.F16, dist, rows16)
    q16 = ost.encode_queries(queries)
    for i in range(nq):  # cast is integer work: exact
        assert np.array_equal(scorer.encoded_query(i).view(np.uint16), q16[i])
    ids = rng.permutation(n).astype(np.uint32)[:400]
    got = scorer.score_points(ids)
    want = ost.score_points(queries, ids)
    scale = _scale_f16(dist, q16, rows16, ids)
    assert np.all(np.abs(got.astype(np.float64) - want) <= REL * scale + 1e-30)
    if dim < 32:  # SSE / scalar leaves are restated exactly
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_f16_reference_literals(qa):
    # metric_f16/avx/{dot,euclid,manhattan}.rs tests: 288-element literals, SIMD vs scalar rel < 5e-4
    for key, dist in (("f16_avx_dot", O.DOT), ("f16_avx_euclid", O.EUCLID), ("f16_avx_manhattan", O.MANHATTAN)):
        v1, v2 = O.to_f16(G[key]["v1_f32"]), O.to_f16(G[key]["v2_f32"])
        st = qa.VectorStorage(v2[None, :].view(np.float16), _dist(qa, dist), qa.VectorStorageDatatype.Float16)
        got = qa.new_raw_scorer(O.f16_to_f32(v1), st).score_points([0])[0, 0]
        scalar = O.similarity(O.F16, dist, v1, v2, O.ISA_SCALAR)
        avx = O.similarity(O.F16, dist, v1, v2, O.ISA_AVX)
        assert abs(got - scalar) / abs(scalar) < 0.0005
        assert abs(got - avx) <= REL * abs(avx)


@pytest.mark.parametrize("dist", [O.COSINE, O.EUCLID])
def test_f16_peek_top(qa, dist):
    rng = np.random.default_rng(5 + dist)
    n, dim, nq, top = 20000, 128, 6, 10
    rows16 = O.to_f16(O.preprocess(dist, rng.standard_normal((n, dim)).astype(np.float32)))
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    st = qa.VectorStorage(rows16.view(np.float16), _dist(qa, dist), qa.VectorStorageDatatype.Float16)
    got = qa.BatchFilteredSearcher(queries, st, top).peek_top_all()
    ost = O.DenseStorage(O.F16, dist, rows16)
    want = ost.peek_top(queries, top)
    allscores = ost.score_points(queries, np.arange(n, dtype=np.uint32))
    for i, (g, w) in enumerate(zip(got, want)):
        assert len(g) == top
        assert np.allclose(g["score"], w["score"], rtol=REL, atol=0)
        # same ids modulo scores closer than the tolerance
        kth = w["score"][-1]
        assert np.all(allscores[i][g["idx"]] >= kth - REL * abs(kth))
        assert np.allclose(allscores[i][g["idx"]], g["score"], rtol=REL, atol=0)


def _u8_data(rng, n, dim, hot=False):
    rows = rng.integers(0, 256, (n, dim), dtype=np.uint8)
    if hot:
        rows[: n // 4] = rng.integers(200, 256, (n // 4, dim), dtype=np.uint8)  # partial sums beyond 2^24
    rows[0] = 0          # zero vector: cosine must return 0.0 (simple_cosine.rs:79-86)
    rows[1] = 255
    return rows


@pytest.mark.parametrize("scalar_order", [False, True])
@pytest.mark.parametrize("dist", [O.DOT, O.COSINE, O.EUCLID, O.MANHATTAN])
@pytest.mark.parametrize("dim", [1, 7, 15, 16, 20, 31, 32, 33, 64, 96, 100, 128, 160, 768, 2000])
def test_u8_score_points_bit_exact(qa, dist, dim, scalar_order):
    rng = np.random.default_rng(dim * 13 + dist)
    n, nq = 500, 4
    rows = _u8_data(rng, n, dim, hot=dim >= 768)
    queries = rng.uniform(-20.0, 300.0, (nq, dim)).astype(np.float32)   # exercises the saturating cast
    queries[1] = 255.0
    queries[2] = 0.0
    from qdrant_amd import _ffi as F
    st = qa.VectorStorage(rows, _dist(qa, dist), qa.VectorStorageDatatype.Uint8,
                          flags=F.SEG_U8_SCALAR_ORDER if scalar_order else 0)
    scorer = qa.new_raw_scorer(queries, st)
    ost = O.DenseStorage(O.U8, dist, rows, u8_isa=O.ISA_SCALAR if scalar_order else O.ISA_AUTO)
    qenc = ost.encode_queries(queries)
    for i in range(nq):
        assert np.array_equal(scorer.encoded_query(i), qenc[i])
    ids = np.arange(n, dtype=np.uint32)
    got = scorer.score_points(ids)
    want = ost.score_points(queries, ids)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (dim, dist, scalar_order)
    if dist == O.COSINE:
        assert got[0, 0] == 0.0 and got[2, 5] == 0.0


def test_u8_reference_literals(qa):
    # metric_uint/avx2/{dot,cosine,euclid,manhattan}.rs tests: assert_eq!(simd, scalar) on 100-byte literals
    for key, dist in (("u8_avx2_dot", O.DOT), ("u8_avx2_cosine", O.COSINE), ("u8_avx2_euclid", O.EUCLID),
                      ("u8_avx2_manhattan", O.MANHATTAN)):
        v1, v2 = np.array(G[key]["v1"], dtype=np.uint8), np.array(G[key]["v2"], dtype=np.uint8)
        st = qa.VectorStorage(v2[None, :], _dist(qa, dist), qa.VectorStorageDatatype.Uint8)
        got = qa.new_raw_scorer(v1.astype(np.float32), st).score_points([0])[0, 0]
        assert got == O.similarity(O.U8, dist, v1, v2, O.ISA_SCALAR) == O.similarity(O.U8, dist, v1, v2, O.ISA_AVX)


@pytest.mark.parametrize("dist", [O.DOT, O.COSINE, O.EUCLID, O.MANHATTAN])
def test_u8_peek_top_and_internal(qa, dist):
    rng = np.random.default_rng(77 + dist)
    n, dim, nq, top = 30000, 96, 20, 10
    rows = _u8_data(rng, n, dim)
    queries = rng.uniform(0, 255, (nq, dim)).astype(np.float32)
    st = qa.VectorStorage(rows, _dist(qa, dist), qa.VectorStorageDatatype.Uint8)
    got = qa.BatchFilteredSearcher(queries, st, top).peek_top_all()
    ost = O.DenseStorage(O.U8, dist, rows)
    want = ost.peek_top(queries, top)
    for g, w in zip(got, want):
        assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32))
        last = w["score"][-1]  # integer scores tie often: ids equal above the cut
        assert set(g["idx"][g["score"] > last]) == set(w["idx"][w["score"] > last])
    # stored points as queries (FilteredScorer::new_internal): the row IS the encoded query
    from qdrant_amd.scorer import new_raw_scorer_internal
    pts = np.array([5, 17, 0, 1], dtype=np.uint32)
    sc = new_raw_scorer_internal(pts, st)
    ids = np.arange(200, dtype=np.uint32)
    got = sc.score_points(ids)
    want = ost.score_points(rows[pts], ids, encoded=True)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))

"""The reference's own property tests of scalar quantization - lib/quantization/tests/integration/test_simple.rs - run against the oracle's
restatement of `EncodedVectorsU8` (oracle/qdrant_oracle.c): the same sizes (129 vectors, dim 65 / 8 / 70), the same value ranges, the same fits
(min / max, quantile 0.99, quantile 1 - eps), the same tolerances.  The reference draws its vectors from `StdRng::seed_from_u64(42)` (the rand crate: not in
the tree), these tests from numpy - the properties hold for every draw, which is what the reference asserts.  The device is held to the oracle bit for
bit elsewhere (test_gpu_sq.py); these tests pin the oracle to the behaviour the reference tests."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O

L = O._lib


class EncodedU8:
    """`EncodedVectorsU8::encode(vectors, ..., quantile, Int8)` + `encode_query` + `score_point_simple` / `score_internal` on the oracle."""

    def __init__(self, distance, invert, vectors, quantile=None):
        v = np.ascontiguousarray(vectors, dtype=np.float32)
        self.n, self.dim = v.shape
        self.sq = O.Sq()
        interval = O.sq_quantile_interval(v, self.n, quantile) if quantile is not None else None     # (count <= the sample size: every vector is sampled)
        if interval is None:
            L.qo_sq_init(C.byref(self.sq), distance, 1 if invert else 0, self.dim, O._p(v), self.n)     # find_min_max_from_iter
        else:
            mn, mx = interval
            L.qo_sq_init_params(C.byref(self.sq), distance, 1 if invert else 0, self.dim, float((mx - mn) / np.float32(127.0)), float(mn))
        self.ad = self.sq.actual_dim
        self.rows = np.zeros((self.n, 4 + self.ad), dtype=np.uint8)
        for i in range(self.n):
            L.qo_sq_encode_row(C.byref(self.sq), O._p(v[i]), O._p(self.rows[i]))

    def score_query(self, q, i):
        q = np.ascontiguousarray(q, dtype=np.float32)
        codes = np.zeros(self.ad, dtype=np.uint8)
        off = C.c_float()
        L.qo_sq_encode_query(C.byref(self.sq), O._p(q), O._p(codes), C.byref(off))
        return L.qo_sq_score(C.byref(self.sq), O._p(codes), off.value, O._p(self.rows[i]), O.ISA_SCALAR)

    def scores(self, q):
        return np.array([self.score_query(q, i) for i in range(self.n)], dtype=np.float32)

    def score_internal(self, i, j):
        return L.qo_sq_score_internal(C.byref(self.sq), O._p(self.rows[i]), O._p(self.rows[j]), O.ISA_SCALAR)


def dot(a, b):
    return np.float32((a.astype(np.float32) * b).sum(dtype=np.float32))


def l2(a, b):
    return np.float32(((a - b) ** 2).sum(dtype=np.float32))


def l1(a, b):
    return np.float32(np.abs(a - b).sum(dtype=np.float32))


METRIC = {O.DOT: dot, O.EUCLID: l2, O.MANHATTAN: l1}


@pytest.mark.parametrize("distance,invert,lo", [
    (O.DOT, False, 0.0),          # test_dot_simple           (:17-58)
    (O.EUCLID, False, 0.0),       # test_l2_simple            (:61-102)
    (O.MANHATTAN, False, 0.0),    # test_l1_simple            (:105-150)
    (O.DOT, True, -1.0),          # test_dot_inverted_simple  (:153-194: values in -1 ..= 1)
    (O.EUCLID, True, -1.0),       # test_l2_inverted_simple   (:197-236)
    (O.MANHATTAN, True, -1.0),    # test_l1_inverted_simple   (:377-422)
])
@pytest.mark.parametrize("seed", [42, 7])
def test_scores_stay_within_a_tenth_per_coordinate(distance, invert, lo, seed):
    rng = np.random.default_rng(seed)
    n, dim = 129, 65
    vectors = rng.uniform(lo, 1.0, (n, dim)).astype(np.float32)
    query = rng.uniform(lo, 1.0, dim).astype(np.float32)
    enc = EncodedU8(distance, invert, vectors)
    for i in range(n):
        original = METRIC[distance](query, vectors[i])
        assert abs(enc.score_query(query, i) - (-original if invert else original)) < dim * 0.1


@pytest.mark.parametrize("invert", [False, True])       # test_dot_internal_simple (:425-464), test_dot_inverted_internal_simple (:467-506)
def test_internal_scores_stay_within_a_tenth_per_coordinate(invert):
    rng = np.random.default_rng(42)
    n, dim = 129, 65
    vectors = rng.uniform(0.0, 1.0, (n, dim)).astype(np.float32)
    enc = EncodedU8(O.DOT, invert, vectors)
    for i in range(1, n):
        original = dot(vectors[0], vectors[i])
        assert abs(enc.score_internal(0, i) - (-original if invert else original)) < dim * 0.1


def _affine_vectors(seed=1008):                           # deterministic_affine_vectors (:274-292): 129 x 8 in -100 ..= 100
    rng = np.random.default_rng(seed)
    return rng.uniform(-100.0, 100.0, (129, 8)).astype(np.float32), rng.uniform(-100.0, 100.0, 8).astype(np.float32)


@pytest.mark.parametrize("distance", [O.MANHATTAN, O.EUCLID])
@pytest.mark.parametrize("invert", [False, True])
def test_distance_scores_are_translation_invariant(distance, invert):
    """test_scalar_quantized_distance_scores_are_translation_invariant (:329-351): + 1000 on every coordinate, quantile 0.99, tolerance 0.25."""
    vectors, query = _affine_vectors()
    base = EncodedU8(distance, invert, vectors, 0.99).scores(query)
    shifted = EncodedU8(distance, invert, vectors * np.float32(1.0) + np.float32(1000.0), 0.99).scores(query * np.float32(1.0) + np.float32(1000.0))
    assert np.abs(base - shifted).max() < 0.25


@pytest.mark.parametrize("distance,expected_scale", [(O.DOT, 6.25), (O.MANHATTAN, 2.5), (O.EUCLID, 6.25)])
@pytest.mark.parametrize("invert", [False, True])
def test_scores_follow_positive_scaling(distance, invert, expected_scale):
    """test_scalar_quantized_scores_follow_positive_scaling (:353-374): x 2.5 on every coordinate, tolerance 1.0."""
    vectors, query = _affine_vectors()
    base = EncodedU8(distance, invert, vectors, 0.99).scores(query)
    scaled = EncodedU8(distance, invert, vectors * np.float32(2.5), 0.99).scores(query * np.float32(2.5))
    assert np.abs(base * np.float32(expected_scale) - scaled).max() < 1.0


def test_a_quantile_next_to_one():
    """test_u8_large_quantile (:508-551): quantile 1 - f32::EPSILON behaves like no quantile."""
    rng = np.random.default_rng(42)
    n, dim = 129, 65
    vectors = rng.uniform(0.0, 1.0, (n, dim)).astype(np.float32)
    query = rng.uniform(0.0, 1.0, dim).astype(np.float32)
    enc = EncodedU8(O.DOT, False, vectors, float(np.float32(1.0) - np.finfo(np.float32).eps))
    for i in range(n):
        assert abs(enc.score_query(query, i) - dot(query, vectors[i])) < dim * 0.1


@pytest.mark.parametrize("invert", [False, True])
@pytest.mark.parametrize("distance", [O.DOT, O.EUCLID, O.MANHATTAN])
def test_a_stored_vector_scores_like_its_own_query(distance, invert):
    """test_sq_u8_encode_internal (:553-617): a vector encoded as a query and the same vector as the stored row (`encode_internal_vector`: the row itself in
    this restatement, as in the device path) score against row 0 within 1e-3 of each other."""
    rng = np.random.default_rng(42)
    n, dim = 129, 70
    vectors = (2.0 * rng.random((n, dim)) - 1.0).astype(np.float32)
    enc = EncodedU8(distance, invert, vectors, float(np.float32(1.0) - np.finfo(np.float32).eps))
    for i in range(n):
        assert abs(enc.score_query(vectors[i], 0) - enc.score_internal(i, 0)) < 1e-3

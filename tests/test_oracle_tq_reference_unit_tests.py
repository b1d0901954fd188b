"""The unit tests of the reference's TurboQuantizer - lib/quantization/src/turboquant/quantization.rs:885-1232 - run against the oracle's restatement
(oracle/qdrant_oracle_tq.c): the same dims, bit widths, vector models (coordinates uniform in -1 .. 1, pairs of a stated similarity), tolerances and
counts.  The reference draws from `StdRng::seed_from_u64(42)` (the rand crate: not in the tree), these from numpy; the properties are stated for every
draw.  `make_tq` = TQMode::Normal, TQRotation::Padded (:620-629); Cosine takes unit vectors, the quantizer itself never inverts (`invert=False`)."""
import numpy as np
import pytest

import oracle_ffi as O


def _tq(dim, bits, distance):
    return O.TqOracle(distance, dim, bits, invert=False)


def _prep(distance, v):
    return (v / np.float32(np.linalg.norm(v.astype(np.float64)))).astype(np.float32) if distance == O.COSINE else v


def _pair(rng, dim, similarity):                         # generate_random_vector_pair_with_similarity (:644-658)
    a = rng.uniform(-1.0, 1.0, dim).astype(np.float32)
    noise = rng.uniform(-1.0, 1.0, dim).astype(np.float32)
    return a, (np.float32(similarity) * a + np.float32(1.0 - similarity) * noise).astype(np.float32)


def _scores(tq, a, b):
    """(symmetric score of the two quantized vectors, asymmetric score of `a` against quantized `b`)"""
    tq.encode_rows(np.stack([a, b]))
    return float(tq.score_internal([0], [1])[0]), float(tq.score_points(a[None, :], [1])[0, 0])


def _norm(v):
    return float(np.linalg.norm(v.astype(np.float64)))


@pytest.mark.parametrize("dim", [127, 128, 300, 512, 513, 1000, 1024, 1025, 2000, 4000])
@pytest.mark.parametrize("distance", [O.DOT, O.COSINE])
def test_score_approximates_true_similarity(dim, distance):
    """:885-936: 4 bits; |score - <a, b>| < 0.05 |a| |b| (0.05 for cosine), symmetric and asymmetric, at similarities 0.2 / 0.5 / 0.8."""
    rng = np.random.default_rng(42 + dim)
    tq = _tq(dim, O.TQ_BITS4, distance)
    for similarity in (0.2, 0.5, 0.8):
        a, b = (_prep(distance, v) for v in _pair(rng, dim, similarity))
        truth = float(np.dot(a.astype(np.float64), b.astype(np.float64)))
        tol = 0.05 * (1.0 if distance == O.COSINE else _norm(a) * _norm(b))
        sym, asym = _scores(tq, a, b)
        assert abs(sym - truth) < tol and abs(asym - truth) < tol, (similarity, sym, asym, truth, tol)


@pytest.mark.parametrize("dim", [127, 128, 300, 512, 513, 1024, 1025, 2000])
@pytest.mark.parametrize("distance", [O.DOT, O.COSINE])
@pytest.mark.parametrize("sign", [1.0, -1.0])
def test_score_of_a_vector_with_itself_and_with_its_opposite(dim, distance, sign):
    """score_self_similarity (:941-983) and score_antipodal_is_negative (:987-1031): +-|v|^2 (dot), +-1 (cosine), within 5 %."""
    rng = np.random.default_rng(42 + dim)
    tq = _tq(dim, O.TQ_BITS4, distance)
    v = _prep(distance, rng.uniform(-1.0, 1.0, dim).astype(np.float32))
    expected = sign * (1.0 if distance == O.COSINE else _norm(v) ** 2)
    tol = 0.05 * max(abs(expected), 1.0)
    sym, asym = _scores(tq, v, (np.float32(sign) * v).astype(np.float32))
    assert abs(sym - expected) < tol and abs(asym - expected) < tol, (sym, asym, expected)


@pytest.mark.parametrize("dim", [512, 513])
@pytest.mark.parametrize("distance", [O.DOT, O.COSINE])
def test_higher_bits_reduce_error(dim, distance):
    """:1035-1077: mean absolute error of the symmetric score over 32 pairs of similarity 0.5: 4 bits <= 2 bits <= 1 bit."""
    def mae(bits):
        rng = np.random.default_rng(42)
        tq = _tq(dim, bits, distance)
        total = 0.0
        for _ in range(32):
            a, b = (_prep(distance, v) for v in _pair(rng, dim, 0.5))
            total += abs(_scores(tq, a, b)[0] - float(np.dot(a.astype(np.float64), b.astype(np.float64))))
        return total / 32
    m1, m2, m4 = mae(O.TQ_BITS1), mae(O.TQ_BITS2), mae(O.TQ_BITS4)
    assert m4 <= m2 <= m1, (m1, m2, m4)


@pytest.mark.parametrize("dim", [127, 128, 513])
@pytest.mark.parametrize("distance", [O.DOT, O.COSINE])
def test_extreme_magnitudes_score_finite(dim, distance):
    """score_extreme_magnitudes_finite (:1081-1114)"""
    tq = _tq(dim, O.TQ_BITS4, distance)
    for val in (1000.0, -1000.0, 1e6, -1e6):
        v = _prep(distance, np.full(dim, val, dtype=np.float32))
        sym, asym = _scores(tq, v, v)
        assert np.isfinite(sym) and np.isfinite(asym)


@pytest.mark.parametrize("dim", [512, 513])
@pytest.mark.parametrize("distance", [O.DOT, O.COSINE])
def test_rank_preservation(dim, distance):
    """:1118-1189: ten candidates at similarities 0.1 .. 1.0 to the query: fewer than 15 % of the pairs change order under the asymmetric 4-bit score."""
    rng = np.random.default_rng(42)
    tq = _tq(dim, O.TQ_BITS4, distance)
    query_raw = rng.uniform(-1.0, 1.0, dim).astype(np.float32)
    cands = []
    for i in range(1, 11):
        s = np.float32(i / 10.0)
        noise = rng.uniform(-1.0, 1.0, dim).astype(np.float32)
        cands.append(_prep(distance, (s * query_raw + (np.float32(1.0) - s) * noise).astype(np.float32)))
    query = _prep(distance, query_raw)
    true = np.array([np.dot(query.astype(np.float64), c.astype(np.float64)) for c in cands])
    tq.encode_rows(np.stack(cands))
    quant = tq.score_points(query[None, :], list(range(10)))[0].astype(np.float64)
    inversions = sum(1 for i in range(10) for j in range(i + 1, 10)
                     if np.sign(true[i] - true[j]) != 0 and np.sign(true[i] - true[j]) != np.sign(quant[i] - quant[j]))
    assert inversions * 100 < 15 * 45, inversions


@pytest.mark.parametrize("dim", [512, 513])
def test_dot_scores_scale_with_the_vectors(dim):
    """score_linearity_dot (:1193-1232): score(q, k v) ~ k <q, v>, score(k q, k v) ~ k^2 <q, v>, within 5 % of the norms' product."""
    rng = np.random.default_rng(42)
    tq = _tq(dim, O.TQ_BITS4, O.DOT)
    q, v = _pair(rng, dim, 0.5)
    truth = float(np.dot(q.astype(np.float64), v.astype(np.float64)))
    for k in (0.5, 2.0, 5.0):
        qs, vs = (np.float32(k) * q).astype(np.float32), (np.float32(k) * v).astype(np.float32)
        tq.encode_rows(np.stack([qs, vs]))
        sym = float(tq.score_internal([0], [1])[0])
        asym = float(tq.score_points(q[None, :], [1])[0, 0])
        assert abs(asym - k * truth) < 0.05 * _norm(q) * _norm(vs), (k, asym)
        assert abs(sym - k * k * truth) < 0.05 * _norm(qs) * _norm(vs), (k, sym)


VALUE_BITS = {O.TQ_BITS4: 4, O.TQ_BITS2: 2, O.TQ_BITS1: 1}


def _codes(tq, row, bits):
    return row[:tq.padded_dim * VALUE_BITS[bits] // 8]          # split_vector: the centroid codes, then the extras


@pytest.mark.parametrize("bits", [O.TQ_BITS1, O.TQ_BITS2, O.TQ_BITS4])
@pytest.mark.parametrize("dim", [3, 7, 127, 513, 1025])
def test_unpadded_rotation_keeps_the_padding_zero(bits, dim):
    """unpadded_rotation_keeps_padding_zero (:1270-1298): odd dims pad for every bit width; the rotation over the original coordinates leaves the tail at 0.0."""
    rng = np.random.default_rng(42 + dim)
    for distance in (O.DOT, O.COSINE):
        tq = O.TqOracle(distance, dim, bits, rotation_unpadded=True, invert=False)
        assert tq.padded_dim > dim
        out = tq.rotate(rng.uniform(-1.0, 1.0, dim).astype(np.float32))
        assert (out[dim:] == 0.0).all() and np.abs(out[:dim]).max() > 0.0


@pytest.mark.parametrize("bits", [O.TQ_BITS1, O.TQ_BITS2, O.TQ_BITS4])
@pytest.mark.parametrize("dim", [8, 64, 128, 512])
def test_unpadded_rotation_is_the_padded_one_where_nothing_is_padded(bits, dim):
    """unpadded_rotation_matches_padded_for_padding_free_dims (:1302-1338): the same bytes, the same asymmetric score bits."""
    rng = np.random.default_rng(7 + dim)
    for distance in (O.DOT, O.COSINE):
        padded = O.TqOracle(distance, dim, bits, invert=False)
        unpadded = O.TqOracle(distance, dim, bits, rotation_unpadded=True, invert=False)
        assert padded.padded_dim == dim
        v = _prep(distance, rng.uniform(-1.0, 1.0, dim).astype(np.float32))
        a, b = padded.encode_rows(v[None, :]), unpadded.encode_rows(v[None, :])
        assert np.array_equal(a, b)
        assert np.array_equal(padded.score_points(v[None, :], [0]).view(np.uint32), unpadded.score_points(v[None, :], [0]).view(np.uint32))


@pytest.mark.parametrize("dim", [127, 513, 1025])
@pytest.mark.parametrize("distance", [O.DOT, O.COSINE])
def test_unpadded_rotation_scores_as_well_on_padded_dims(dim, distance):
    """unpadded_rotation_score_accuracy_padded_dims (:1342-1392): the tolerance of score_approximates_true_similarity."""
    rng = np.random.default_rng(42 + dim)
    tq = O.TqOracle(distance, dim, O.TQ_BITS4, rotation_unpadded=True, invert=False)
    for similarity in (0.2, 0.5, 0.8):
        a, b = (_prep(distance, v) for v in _pair(rng, dim, similarity))
        truth = float(np.dot(a.astype(np.float64), b.astype(np.float64)))
        tol = 0.05 * (1.0 if distance == O.COSINE else _norm(a) * _norm(b))
        sym, asym = _scores(tq, a, b)
        assert abs(sym - truth) < tol and abs(asym - truth) < tol


@pytest.mark.parametrize("dim", [7, 127, 513, 1025])
@pytest.mark.parametrize("distance", [O.DOT, O.COSINE])
def test_unpadded_rotation_round_trip_keeps_the_codes(dim, distance):
    """unpadded_rotation_roundtrip_preserves_codes_for_padded_dims (:1399-1434): quantize -> dequantize -> rotate back -> drop the padding -> quantize:
    the same centroid codes, eight vectors per case.  (Exercises qo_tq_dequantize / qo_tq_rotate_inverse: what the Manhattan scores of the device rest on.)"""
    rng = np.random.default_rng(42 + dim)
    tq = O.TqOracle(distance, dim, O.TQ_BITS4, rotation_unpadded=True, invert=False)
    for _ in range(8):
        v = _prep(distance, rng.uniform(-1.0, 1.0, dim).astype(np.float32))
        q1 = tq.encode_rows(v[None, :])[0].copy()
        readback = tq.dequantize(q1, rotate_back=True)[:dim].astype(np.float32)
        q2 = tq.encode_rows(readback[None, :])[0]
        assert np.array_equal(_codes(tq, q1, O.TQ_BITS4), _codes(tq, q2, O.TQ_BITS4))


@pytest.mark.parametrize("bits", [O.TQ_BITS1, O.TQ_BITS2, O.TQ_BITS4])
def test_every_bit_width_scores_through_precompute_query(bits):
    """score_precomputed_dispatches_all_bit_widths (:1450-1495): finite, positive against itself, negative against its opposite, a gap wider than either."""
    rng = np.random.default_rng(0xD15DA7C4)
    for distance in (O.DOT, O.COSINE):
        tq = _tq(512, bits, distance)
        v = _prep(distance, rng.uniform(-1.0, 1.0, 512).astype(np.float32))
        tq.encode_rows(np.stack([v, -v]))
        s, a = (float(x) for x in tq.score_points(v[None, :], [0, 1])[0])
        assert np.isfinite(s) and np.isfinite(a) and s > 0.0 and a < 0.0 and s - a > max(abs(s), abs(a))

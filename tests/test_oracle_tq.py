"""The oracle's TurboQuant restatement (oracle/qdrant_oracle_tq.c) against everything the reference's own tests hold for it (CPU only).

  rotation.rs:283-315       test_compute_chunk_sizes: literal decompositions
  permutation.rs:163-190    lcg / permute round trips (the forward maps are permutations)
  rotation.rs (tests)       the rotation is orthonormal: norms and dot products are preserved; Hadamard spreads concentrated energy
  simd/query{4,2}bit        test_codebook_matches_lloyd_max: the integer codebooks are round(c * scale) + 128 of the f32 centroids
  lloyd_max.rs:170-190      the centroid constants are the Lloyd-Max solution for N(0, 1) (checked here by the fixed-point property)
  tests/integration/test_tq.rs:19-60,262-475,770-870   |score - exact| < coef(bits) * signal_std for dot / cosine / l2, asymmetric and
                                                       internal, with the reference's data model (U[-1, 1]^d), dims and bits
  encoding.rs:172-201       quantized sizes and padded dims
Nothing here is a bit-level pin (the reference has none for this quantizer): DESIGN 4."""
import numpy as np
import pytest

import oracle_ffi as O

BITS = [O.TQ_BITS4, O.TQ_BITS2, O.TQ_BITS1_5, O.TQ_BITS1]
COEF = {O.TQ_BITS1: 5.1, O.TQ_BITS1_5: 4.0, O.TQ_BITS2: 3.0, O.TQ_BITS4: 0.9}
MIN_DIM = {O.TQ_BITS1: 64, O.TQ_BITS1_5: 48, O.TQ_BITS2: 32, O.TQ_BITS4: 8}
DIMS = [16, 64, 65, 128, 384, 512]


def test_chunk_sizes_literals():
    def chunks(d):
        out = np.zeros(32, dtype=np.uint32)
        n = O._lib.qo_tq_chunk_sizes(d, O._p(out))
        return out[:n].tolist()
    assert chunks(128) == [128]
    assert chunks(700) == [512, 128, 32, 16, 8, 4]
    assert chunks(1536) == [1024, 512]
    assert chunks(4096) == [4096]
    for d in [5, 128, 129, 300, 700, 712, 1536, 4096]:
        c = chunks(d)
        assert sum(c) == d and all(x & (x - 1) == 0 for x in c) and c == sorted(c, reverse=True)


def test_padded_dims_and_sizes():
    assert O._lib.qo_tq_padded_dim_for(65, O.TQ_BITS1) == 72 and O._lib.qo_tq_padded_dim_for(65, O.TQ_BITS2) == 68
    assert O._lib.qo_tq_padded_dim_for(65, O.TQ_BITS4) == 66 and O._lib.qo_tq_padded_dim_for(64, O.TQ_BITS1_5) == 96
    assert O.TqOracle(O.DOT, 768, O.TQ_BITS4).row_bytes == 384 + 4 and O.TqOracle(O.EUCLID, 768, O.TQ_BITS2).row_bytes == 192 + 8
    assert O.TqOracle(O.COSINE, 128, O.TQ_BITS1_5).row_bytes == 24 + 4


def test_permutation_maps_are_permutations_and_differ_by_seed():
    for count in (2, 5, 64, 300, 1024, 1536):
        maps = []
        for seed in (654605292835415893, 8636605637963351413, 1775280196666917949, 42):
            m = np.zeros(count, dtype=np.uint32)
            O._lib.qo_tq_permutation_map(seed, count, O._p(m))
            assert sorted(m.tolist()) == list(range(count))
            maps.append(m)
        if count >= 64:
            assert not np.array_equal(maps[0], maps[1]) and not np.array_equal(maps[0], np.arange(count))
    # Fisher-Yates replay by hand for a tiny case: LCG state = state * A + C, j = (state >> 32) % (i + 1), swap(i, j) for i = n-1 .. 1
    A, Cc, M = 6364136223846793005, 1442695040888963407, (1 << 64) - 1
    arr, state = list(range(7)), 42
    for i in range(6, 0, -1):
        state = (state * A + Cc) & M
        j = (state >> 32) % (i + 1)
        arr[i], arr[j] = arr[j], arr[i]
    m = np.zeros(7, dtype=np.uint32)
    O._lib.qo_tq_permutation_map(42, 7, O._p(m))
    assert m.tolist() == arr


@pytest.mark.parametrize("dim", [100, 101, 300, 384, 512, 1024, 1025, 1586])
def test_rotation_is_orthonormal_and_spreads_energy(dim):
    rng = np.random.default_rng(dim)
    t = O.TqOracle(O.DOT, dim, O.TQ_BITS4)
    a, b = rng.standard_normal(dim), rng.standard_normal(dim)
    ra, rb = t.rotate(a), t.rotate(b)
    assert abs(np.dot(ra, rb) - np.dot(a, b)) < 1e-9 * dim and abs(np.linalg.norm(ra) - np.linalg.norm(a)) < 1e-9
    spike = np.zeros(dim)
    spike[3] = 1.0
    r = t.rotate(spike)
    assert np.abs(r).max() < 0.5 and abs(np.linalg.norm(r) - 1.0) < 1e-12          # energy no longer sits in one coordinate
    # the WHT is its own inverse up to n
    x = rng.standard_normal(256)
    y = x.copy()
    O._lib.qo_tq_wht(O._p(y), 256)
    O._lib.qo_tq_wht(O._p(y), 256)
    assert np.allclose(y / 256.0, x, atol=1e-12)


def test_integer_codebooks_match_the_lloyd_max_centroids():
    c4 = np.array([-2.733, -2.069, -1.618, -1.256, -0.9424, -0.6568, -0.3881, -0.1284, 0.1284, 0.3881, 0.6568, 0.9424, 1.256, 1.618, 2.069, 2.733])
    want4 = [0, 31, 52, 69, 84, 97, 110, 122, 134, 146, 159, 172, 187, 204, 225, 255]
    assert np.clip(np.round(c4 * (128.0 / 2.733)) + 128, 0, 255).astype(int).tolist() == want4
    c2 = np.array([-1.510, -0.4528, 0.4528, 1.510])
    assert np.clip(np.round(c2 * (128.0 / 1.510)) + 128, 0, 255).astype(int).tolist() == [0, 90, 166, 255]
    # Lloyd-Max fixed point for N(0, 1): each centroid is the conditional mean of its cell (lloyd_max.rs tests, tolerance 1e-3)
    from math import erf, exp, pi, sqrt
    pdf = lambda x: exp(-0.5 * x * x) / sqrt(2 * pi)      # noqa: E731
    cdf = lambda x: 0.5 * (1 + erf(x / sqrt(2)))          # noqa: E731
    for c in (c4, c2, np.array([-0.7978846, 0.7978846])):
        edges = [-np.inf] + ((c[:-1] + c[1:]) / 2).tolist() + [np.inf]
        for i, ci in enumerate(c):
            a, b = edges[i], edges[i + 1]
            pa, pb = (0.0 if a == -np.inf else pdf(a)), (0.0 if b == np.inf else pdf(b))
            ca, cb = (0.0 if a == -np.inf else cdf(a)), (1.0 if b == np.inf else cdf(b))
            assert abs((pa - pb) / (cb - ca) - ci) < 1e-3


def _data(dim, n, seed, normalize=False):
    rng = np.random.default_rng(seed)
    v = rng.uniform(-1.0, 1.0, (n + 1, dim)).astype(np.float32)
    if normalize:
        v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v[:n], v[n]


@pytest.mark.parametrize("bits", BITS)
@pytest.mark.parametrize("dim", DIMS)
def test_scores_within_the_reference_error_model(bits, dim):
    """test_tq_dot / test_tq_cosine / test_tq_l2 and their _internal variants (TQMode::Normal): 513 vectors, one query."""
    if dim < MIN_DIM[bits]:
        pytest.skip("should_test(dim, bits) is false in the reference")
    n = 513
    dot_std, cos_std = (dim / 9.0) ** 0.5, 1.0 / dim ** 0.5
    for distance, std, normalize in ((O.DOT, dot_std, False), (O.COSINE, cos_std, True), (O.EUCLID, 2 * dot_std, False)):
        vecs, q = _data(dim, n, seed=42 + dim, normalize=normalize)
        t = O.TqOracle(distance, dim, bits, invert=False)
        t.encode_rows(vecs)
        err = COEF[bits] * std
        got = t.score_points(q[None, :], np.arange(n))[0]
        if distance == O.EUCLID:
            exact = ((vecs - q) ** 2).sum(axis=1)
        elif distance == O.COSINE:
            exact = (vecs @ q) / (np.linalg.norm(vecs, axis=1) * np.linalg.norm(q))
        else:
            exact = vecs @ q
        assert np.abs(got - exact).max() < err
        gi = t.score_internal(np.zeros(n - 1, dtype=int), np.arange(1, n))
        if distance == O.EUCLID:
            ei = ((vecs[1:] - vecs[0]) ** 2).sum(axis=1)
        elif distance == O.COSINE:
            ei = (vecs[1:] @ vecs[0]) / (np.linalg.norm(vecs[1:], axis=1) * np.linalg.norm(vecs[0]))
        else:
            ei = vecs[1:] @ vecs[0]
        assert np.abs(gi - ei).max() < err


@pytest.mark.parametrize("bits", BITS)
def test_zero_vectors_and_zero_queries(bits):
    """test_tq_zero_vector_* / test_tq_zero_query_*: finite scores, a zero query scores 0 (dot) against everything."""
    dim = 128
    vecs, q = _data(dim, 32, seed=7)
    vecs[5] = 0.0
    for distance in (O.DOT, O.COSINE):
        t = O.TqOracle(distance, dim, bits, invert=False)
        t.encode_rows(vecs)
        s = t.score_points(np.stack([q, np.zeros(dim, dtype=np.float32)]), np.arange(32))
        assert np.isfinite(s).all()
        assert abs(s[0, 5]) < COEF[bits] * (dim / 9.0) ** 0.5 and np.abs(s[1]).max() < 1e-3


def test_query_encoding_fields():
    """Query4bitSimd::new: q_signed = clamp(round(v * 8127 / max|v|)), postprocess_scale = 1 / (q_scale * 128 / 2.733); 1 bit: 127, 0.7978846 / q_scale"""
    dim = 64
    rng = np.random.default_rng(3)
    q = rng.standard_normal(dim).astype(np.float32)
    for bits, qmax in ((O.TQ_BITS4, 8127), (O.TQ_BITS2, 8127), (O.TQ_BITS1, 127)):
        t = O.TqOracle(O.DOT, dim, bits)
        qs, ps, l2, sq = t.query(q)
        rot = t.rotate(q.astype(np.float64)).astype(np.float32)
        scale = np.float32(qmax) / np.abs(rot).max()
        assert np.abs(qs).max() == qmax and int(qs.sum()) == sq
        assert np.array_equal(qs, np.clip(np.sign(rot * scale) * np.floor(np.abs(rot * scale) + np.float32(0.5)), -qmax, qmax).astype(np.int32))
        assert abs(float(l2) - np.linalg.norm(q)) < 1e-4


@pytest.mark.parametrize("bits", BITS)
@pytest.mark.parametrize("dim", [64, 128, 384])
def test_plus_mode_within_the_reference_error_model(bits, dim):
    """TQMode::Plus (test_tq_dot / _cosine / _l2 and the _internal variants, #[case::plus]): the same bars with per-coordinate error correction.
    The data is made anisotropic (a few coordinates with a large offset) so that the correction has something to do."""
    if dim < MIN_DIM[bits]:
        pytest.skip("should_test(dim, bits) is false in the reference")
    n = 513
    dot_std, cos_std = (dim / 9.0) ** 0.5, 1.0 / dim ** 0.5
    for distance, std, normalize in ((O.DOT, dot_std, False), (O.COSINE, cos_std, True), (O.EUCLID, 2 * dot_std, False)):
        vecs, q = _data(dim, n, seed=4242 + dim, normalize=normalize)
        shift, scale = O.tq_plus_fit(distance, dim, bits, vecs)
        t = O.TqOracle(distance, dim, bits, invert=False, shift=shift, scale=scale)
        assert t.row_bytes == O.TqOracle(distance, dim, bits).row_bytes + 4           # the trailing xm of TqVectorExtras
        t.encode_rows(vecs)
        err = COEF[bits] * std
        got = t.score_points(q[None, :], np.arange(n))[0]
        if distance == O.EUCLID:
            exact, ei = ((vecs - q) ** 2).sum(axis=1), ((vecs[1:] - vecs[0]) ** 2).sum(axis=1)
        elif distance == O.COSINE:
            exact = (vecs @ q) / (np.linalg.norm(vecs, axis=1) * np.linalg.norm(q))
            ei = (vecs[1:] @ vecs[0]) / (np.linalg.norm(vecs[1:], axis=1) * np.linalg.norm(vecs[0]))
        else:
            exact, ei = vecs @ q, vecs[1:] @ vecs[0]
        assert np.abs(got - exact).max() < err
        gi = t.score_internal(np.zeros(n - 1, dtype=int), np.arange(1, n))
        assert np.abs(gi - ei).max() < err


def test_plus_mode_with_identity_correction_equals_normal_mode_scores():
    """shift = 0, scale = 1: X+ = X, D' = 1, M = 0 - the asymmetric scores are the Normal-mode scores (1-bit: up to the wider 16-bit query)"""
    dim, n = 128, 64
    vecs, q = _data(dim, n, seed=5)
    for bits in (O.TQ_BITS4, O.TQ_BITS2):
        a = O.TqOracle(O.DOT, dim, bits, invert=False)
        b = O.TqOracle(O.DOT, dim, bits, invert=False, shift=np.zeros(a.padded_dim), scale=np.ones(a.padded_dim))
        ra, rb = a.encode_rows(vecs), b.encode_rows(vecs)
        assert np.array_equal(ra, rb[:, :a.row_bytes]) and np.all(rb[:, a.row_bytes:].view(np.float32) == 0.0)      # same codes, same scaling factor, xm = -0 * x = 0
        assert np.array_equal(a.score_points(q[None, :], np.arange(n)).view(np.uint32), b.score_points(q[None, :], np.arange(n)).view(np.uint32))


def test_p_square_estimator_properties():
    """The reference's own tests of its extended P-square estimator (p_square.rs:631-832), mirrored: uniform / normal / Student / Poisson
    streams of 10 000 values land within its tolerances of the exact sample quantile, all-zero streams and streams shorter than the
    marker count are exact, the 7-marker grid is increasing."""
    rng = np.random.default_rng(42)
    u = rng.random(10000)
    p = O.p2_quantile(0.99, u)
    assert abs(p - np.quantile(u, 0.99)) < 1e-2 and abs(p - 0.99) < 1e-2                    # test_p_square
    g = rng.standard_normal(10000)
    assert abs(O.p2_quantile(0.99, g) - np.quantile(g, 0.99)) < 0.1                          # test_p_square_normal (ERROR = 0.1)
    assert abs(O.p2_quantile(0.01, g) - np.quantile(g, 0.01)) < 0.1                          # test_p_square_normal_low
    t = rng.standard_t(5, 10000)
    assert abs(O.p2_quantile(0.99, t) - np.quantile(t, 0.99)) < 0.5                          # test_p_square_student (heavy tails)
    po = rng.poisson(4.0, 10000).astype(np.float64)
    assert abs(O.p2_quantile(0.99, po) - np.quantile(po, 0.99)) < 1.0                        # test_p_square_poisson (ties)
    assert O.p2_quantile(0.99, np.zeros(10000)) == 0.0                                       # test_p_square_zeros
    assert O.p2_quantile(0.99, np.zeros(3)) == 0.0                                           # test_p_square_linear
    assert O.p2_quantile(0.5, []) == 0.0 and O.p2_quantile(0.5, [3.5]) == 3.5
    assert O.p2_quantile(0.25, [4.0, 0.0, 2.0]) == 1.0                                       # estimate_quantile_from_slice: k = 0.5 between 0 and 2
    assert O.p2_quantile(0.9, [1.0, np.nan, np.inf, 2.0]) == pytest.approx(1.9)              # NaN / inf observations are dropped
    _, grid = O.p2_quantile(0.99, [], with_grid=True)
    assert np.all(np.diff(grid) > 0) and grid[0] == 0.0 and grid[3] == 0.99 and grid[6] == 1.0   # test_p_square_extended_grid
    assert grid[1] == 0.99 * 0.5 and grid[2] == 0.99 * (0.7 + 0.3 * 1.0 / 3.0)


def test_tq_plus_fit_from_p_square_estimates():
    """TQMode::Plus first pass: quantiles at Phi(+-c_outer) of every rotated, length-rescaled coordinate; data that already is N(0, 1) per
    coordinate gives shift ~ 0, scale ~ 1 (encoded_vectors_tq.rs:150-155); an anisotropic block is pulled back onto the codebook grid."""
    lo, hi, c = O.tq_plus_quantiles(O.TQ_BITS4)
    assert c == np.float32(2.733) and abs(hi - 0.99686) < 1e-4 and lo == pytest.approx(1.0 - hi)
    assert O.tq_plus_quantiles(O.TQ_BITS1)[2] == np.float32(0.7978846)
    rng = np.random.default_rng(7)
    dim = 128
    iso = rng.standard_normal((4096, dim)).astype(np.float32)
    shift, scale = O.tq_plus_fit_p2(O.DOT, dim, O.TQ_BITS2, iso)
    assert np.abs(shift).max() < 0.2 and np.abs(scale - 1.0).max() < 0.2
    aniso = iso.copy()
    aniso[:, :16] += 2.0
    s_exact, c_exact = O.tq_plus_fit(O.DOT, dim, O.TQ_BITS2, aniso)
    s_p2, c_p2 = O.tq_plus_fit_p2(O.DOT, dim, O.TQ_BITS2, aniso)
    assert np.abs(s_exact - s_p2).max() < 0.2 and np.abs(c_exact / c_p2 - 1.0).max() < 0.2
    assert np.abs(s_p2).max() > 0.3                                                          # there is something to correct
    # degenerate inputs: no sample -> identity; a constant coordinate keeps scale = 1 (MIN_QUANTILE_WIDTH)
    s0, c0 = O.tq_plus_fit_p2(O.DOT, dim, O.TQ_BITS4, iso[:0])
    assert np.all(s0 == 0.0) and np.all(c0 == 1.0)
    sz, cz = O.tq_plus_fit_p2(O.COSINE, dim, O.TQ_BITS4, np.zeros((100, dim), dtype=np.float32))
    assert np.all(sz == 0.0) and np.all(cz == 1.0)


def test_vector_stats_welford():
    """VectorStats::build (vector_stats.rs): streaming Welford == the two-pass mean / sample stddev up to rounding; min / max exact; the degenerate
    counts of its build() (0 rows: mean 0, stddev 0, min = f32::MAX, max = f32::MIN; 1 row: stddev 0)."""
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((5000, 33)) * 3.0 + 1.5).astype(np.float32)
    mn, mx, mean, sd = O.vector_stats(x)
    assert np.array_equal(mn, x.min(axis=0)) and np.array_equal(mx, x.max(axis=0))
    assert np.allclose(mean, x.astype(np.float64).mean(axis=0), rtol=1e-6) and np.allclose(sd, x.astype(np.float64).std(axis=0, ddof=1), rtol=1e-6)
    mn, mx, mean, sd = O.vector_stats(x[:1])
    assert np.array_equal(mean, x[0]) and np.all(sd == 0.0) and np.array_equal(mn, x[0]) and np.array_equal(mx, x[0])
    mn, mx, mean, sd = O.vector_stats(np.zeros((0, 4), dtype=np.float32))
    assert np.all(mean == 0.0) and np.all(sd == 0.0) and np.all(mn == np.finfo(np.float32).max) and np.all(mx == np.finfo(np.float32).min)


@pytest.mark.parametrize("quantile", [0.95])
def test_quantile_interval_per_coordinate_property(quantile):
    """quantile.rs:330-385 (test_quantile_interval_per_coord) mirrored: 2 048 sampled vectors of uniform [0, 1) coordinates, identity preprocess - each
    coordinate's pair of P-square estimates lands within 0.05 of ((1 - q) / 2, 1 - (1 - q) / 2)."""
    rng = np.random.default_rng(42)
    data = rng.random((2048, 4)).astype(np.float32)
    lo_want = (1.0 - quantile) / 2.0
    for d in range(4):
        col = data[:, d].astype(np.float64)
        assert abs(O.p2_quantile(lo_want, col) - lo_want) < 0.05
        assert abs(O.p2_quantile(1.0 - lo_want, col) - (1.0 - lo_want)) < 0.05


# error_l1 (lib/quantization/tests/integration/test_tq.rs:59-77): per-bits coefficient x sqrt(dim)
L1_COEF = {O.TQ_BITS1: 7.5, O.TQ_BITS1_5: 4.5, O.TQ_BITS2: 3.0, O.TQ_BITS4: 0.7}


@pytest.mark.parametrize("plus", [False, True])
@pytest.mark.parametrize("bits", BITS)
def test_l1_scores_within_the_reference_error_model(bits, plus):
    """test_tq_l1 / test_tq_l1_internal (test_tq.rs:871-1010), TQMode::Normal and Plus: 513 vectors of U[-1, 1]^d, one query; the dequantise + inverse rotation
    fallback (turboquant/quantization.rs:429-440,596-607) within error_l1 of the true L1 distance.  What pins qo_tq_dequantize / qo_tq_rotate_inverse:
    the same bars as the reference's own tests - tolerances, no literals (parity of the L1 path at the bit level: unpinned, as for the rest of TurboQuant)."""
    for dim in DIMS:
        if dim < MIN_DIM[bits]:
            continue
        n = 513 if dim <= 256 else 129
        vecs, q = _data(dim, n, seed=42 + dim)
        shift = scale = None
        if plus:
            shift, scale = O.tq_plus_fit(O.MANHATTAN, dim, bits, vecs)
        t = O.TqOracle(O.MANHATTAN, dim, bits, invert=False, shift=shift, scale=scale)
        t.encode_rows(vecs)
        err = L1_COEF[bits] * dim ** 0.5
        got = t.score_points(q[None, :], np.arange(n))[0]
        assert np.abs(got - np.abs(vecs - q).sum(axis=1)).max() < err
        gi = t.score_internal(np.zeros(n - 1, dtype=int), np.arange(1, n))
        assert np.abs(gi - np.abs(vecs[1:] - vecs[0]).sum(axis=1)).max() < err


def test_inverse_rotation_undoes_the_rotation():
    """HadamardRotation::apply_inverse(apply(x)) == x (rotation.rs test_rotation_roundtrip: 1e-9), padded and unpadded, chunked dims included"""
    rng = np.random.default_rng(3)
    for dim, unpadded in ((64, False), (100, True), (700, False), (768, False), (1000, True)):
        t = O.TqOracle(O.MANHATTAN, dim, O.TQ_BITS4, rotation_unpadded=unpadded)
        x = rng.standard_normal(dim)
        y = t.rotate(x)
        assert np.abs(t.rotate_inverse(y)[:dim] - x).max() < 1e-9
        # the padding tail of an unpadded rotation is not touched
        if unpadded and t.padded_dim > dim:
            y2 = y.copy()
            y2[dim:] = 5.0
            assert (t.rotate_inverse(y2)[dim:] == 5.0).all()

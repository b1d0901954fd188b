"""The bound behind the PQ walk's hop prefilter (pq.hip pq_walk_lut8_kernel + pq_hop_prefilter), restated in numpy with the kernel's f32 / f64 operations
and checked against the oracle's exact scores (score_point_sse order) - no GPU.

  q_cj = rint((LUT[c][j] - lo_c) * (255 / R)) clamped to 0..255        (f32, as the kernel evaluates it)
  A(row) = sum_c q_{c, code_c}                                           (an integer)
  exact f32 score S(row) <= L + step (A + ROUND m) + Es,  L = sum_c lo_c (f64), step = R / 255 (f64), Es = (m + 1) 2^-24 sum_c max_j |LUT[c][j]|
  the walk drops a candidate iff A < floor((T - L - Es) / step - ROUND m) - 1, T = score of the beam's worst entry: every dropped row must have S < T.
"""
import numpy as np
import pytest

import oracle_ffi as O

ROUND = 0.5 + 1.0e-4      # PQ_WALK_ROUND


def lut8(lut):
    """pq_walk_lut8_body: -> (q8 [m][ncent] uint8, L, Es, step, usable)"""
    lut = np.asarray(lut, dtype=np.float32)
    m = lut.shape[0]
    lo, hi, ab = lut.min(axis=1), lut.max(axis=1), np.abs(lut).max(axis=1)
    R = np.float32(0.0)
    E = np.float32(0.0)
    L = 0.0
    for c in range(m):                      # the kernel's loop: f32 max / f32 sum / f64 sum, in chunk order
        R = np.float32(max(R, np.float32(hi[c] - lo[c])))
        E = np.float32(E + ab[c])
        L += float(lo[c])
    flat = not (R > 0.0) or not (R < 3.0e38) or not np.isfinite(lut).all()
    inv_step = np.float32(0.0) if flat else np.float32(255.0) / R
    with np.errstate(invalid="ignore"):
        x = np.rint((lut - lo[:, None]).astype(np.float32) * inv_step).astype(np.float32)
    q8 = np.clip(np.nan_to_num(x, nan=0.0, posinf=255.0, neginf=0.0), 0.0, 255.0).astype(np.uint8)
    return q8, L, float(m + 1) * 2.0 ** -24 * float(E), (1.0 if flat else float(R) / 255.0), not flat


def a_min(T, L, Es, step, m):
    t = np.floor((float(T) - L - Es) / step - ROUND * m) - 1.0
    return None if not (t > 0.0) else int(min(t, 1.0e9))


@pytest.mark.parametrize("distance", [O.DOT, O.COSINE, O.EUCLID, O.MANHATTAN])
@pytest.mark.parametrize("dim,chunk", [(64, 4), (96, 8), (128, 2), (1536, 16)])
def test_upper_bound_holds_and_no_dropped_row_could_have_entered(distance, dim, chunk):
    rng = np.random.default_rng(dim * 7 + chunk + distance)
    n, nq = 1500, 6
    rows = O.preprocess(distance, rng.standard_normal((n, dim)).astype(np.float32) * rng.uniform(0.2, 3.0, size=(1, dim)).astype(np.float32))
    cen = O.PqOracle.train(rows[:1000], dim, chunk, 256, iters=2)
    pq = O.PqOracle(distance, dim, chunk, cen)
    codes = pq.encode(rows)
    queries = O.preprocess(distance, rng.standard_normal((nq, dim)).astype(np.float32))
    queries[-1] *= 1.0e-20                     # a tiny query: steps near the bottom of the f32 range
    ids = np.arange(n, dtype=np.uint32)
    exact = pq.score_points(queries, ids).astype(np.float64)            # the f32 scores the walk would compute, score_point_sse order
    m = pq.m
    dropped_total = 0
    for qi in range(nq):
        q8, L, Es, step, usable = lut8(pq.lut(queries[qi]))
        if not usable:
            continue
        A = q8[np.arange(m)[None, :], codes].astype(np.int64).sum(axis=1)
        upper = L + step * (A + ROUND * m) + Es
        assert np.all(exact[qi] <= upper), (qi, float(np.max(exact[qi] - upper)))
        # the band is what it claims to be: about m / 2 table units each way
        lower = L + step * (A - ROUND * m) - Es
        assert np.all(exact[qi] >= lower)
        # the walk's integer test, for bounds at the 50th, 90th and 99th percentile of the scores and at the maximum
        for T in np.quantile(exact[qi], [0.5, 0.9, 0.99, 1.0]).astype(np.float32):
            t = a_min(T, L, Es, step, m)
            if t is None:
                continue
            dropped = A < t
            assert np.all(exact[qi][dropped] < float(T))
            dropped_total += int(dropped.sum())
    assert dropped_total > 0, "the prefilter never dropped anything: the test did not test it"


def test_flat_and_non_finite_tables_drop_nothing():
    for lut in (np.full((8, 256), 3.0, dtype=np.float32), np.where(np.arange(8 * 256).reshape(8, 256) == 5, np.inf, 1.0).astype(np.float32),
                np.where(np.arange(8 * 256).reshape(8, 256) == 7, np.nan, 1.0).astype(np.float32)):
        assert lut8(lut)[4] is False          # usable = 0: pq_hop_prefilter returns k unchanged
    # a bound below everything the table can make, or a NaN bound: nothing is dropped either
    q8, L, Es, step, usable = lut8(np.random.default_rng(1).standard_normal((16, 256)).astype(np.float32))
    assert usable and a_min(-1.0e30, L, Es, step, 16) is None and a_min(np.float32(np.nan), L, Es, step, 16) is None

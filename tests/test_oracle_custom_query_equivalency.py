"""The reference's `compare_scoring_equivalency` (lib/segment/src/vector_storage/tests/custom_query_scorer_equivalency.rs:118-258) on the oracle: custom
queries (RecommendBestScore / RecommendSumScores / Discover / Context, 1..3 examples or pairs per side as `fixtures/query_fixtures.rs` draws them) scored
over a raw storage and over a quantized copy of it - SQ int8 (quantile 0.5, rows ~ N(0, 8)), PQ x4 (one coordinate per chunk, rows ~ U[0, 1)), binary -
on 100 sampled points of 600 (a tenth deleted), 50 attempts each: at least 70 % of the top 10 % are shared.  Without quantization the scores are equal.
The device is held to these oracle scorers bit for bit (test_gpu_custom_quantized.py); this pins the oracle's `QuantizedCustomQueryScorer` restatement
(qo_scorer kind 6) to the behaviour the reference tests.  The reference feeds its binary case vectors of `f32::from(x as u8)` with x in [-1, 1]: all
zeros (but for x == 1.0), so every score is equal and the two top sets coincide by the stable sort - mirrored as it is ("bq"); on rows that do vary
(uniform in [-1, 1], "bq_uniform") one bit per coordinate at 128 dimensions shares about half of the top tenth, which is stated as what it is."""
import numpy as np
import pytest

import oracle_ffi as O

DIMS, NUM_POINTS, SAMPLE_SIZE, ATTEMPTS, MAX_EXAMPLE_PAIRS = 128, 600, 100, 50, 4
RECO_BEST, RECO_SUM, DISCOVER, CONTEXT = 0, 1, 2, 3


def _sampler(quant):
    if quant == "sq":
        return lambda rng, n: rng.normal(0.0, 8.0, (n, DIMS)).astype(np.float32)
    if quant == "pq":
        return lambda rng, n: rng.random((n, DIMS), dtype=np.float32)
    if quant == "bq":          # rng.sample_iter(Uniform::new_inclusive(-1.0, 1.0)).map(|x| f32::from(x as u8)) (:104-111): a saturating cast
        return lambda rng, n: np.floor(np.clip(rng.uniform(-1.0, 1.0, (n, DIMS)), 0.0, 1.0)).astype(np.float32)
    return lambda rng, n: rng.uniform(-1.0, 1.0, (n, DIMS)).astype(np.float32)


def _random_query(kind, rng, gen):
    """examples in flat_iter() order + (n_a, n_b), drawn like fixtures/query_fixtures.rs:48-99"""
    if kind in (RECO_BEST, RECO_SUM):
        k = int(rng.integers(1, MAX_EXAMPLE_PAIRS))
        return gen(rng, 2 * k), k, k                                  # positives, then negatives
    if kind == DISCOVER:
        k = int(rng.integers(1, MAX_EXAMPLE_PAIRS))
        return gen(rng, 1 + 2 * k), 1, k                              # target, then (positive, negative) pairs
    k = int(rng.integers(0, MAX_EXAMPLE_PAIRS))
    return gen(rng, 2 * k), 0, k


def _quantized_factory(quant, rows, flags):
    if quant == "sq":
        mn, mx = O.sq_quantile_interval(rows, rows.shape[0], 0.5)     # ScalarQuantizationConfig { quantile: Some(0.5) }
        sq = O.SqOracle(O.DOT, DIMS, float((mx - mn) / np.float32(127.0)), float(mn))
        sq.encode_rows(rows)
        return O.ScorerFactory("sq", flags, sq)
    if quant == "pq":
        cen, _ = O.PqOracle.train_ex(rows, DIMS, 1, 256)              # CompressionRatio::X4: one coordinate per chunk, 256 centroids
        pq = O.PqOracle(O.DOT, DIMS, 1, cen)
        pq.encode(rows)
        return O.ScorerFactory("pq", flags, pq)
    bq = O.BqOracle(O.DOT, DIMS)
    bq.encode_rows(rows)
    return O.ScorerFactory("bq", flags, bq)


@pytest.mark.parametrize("quant", [None, "pq", "sq", "bq", "bq_uniform"])
@pytest.mark.parametrize("kind", [RECO_BEST, RECO_SUM, DISCOVER, CONTEXT])
def test_custom_queries_over_a_quantized_storage_rank_like_over_the_raw_one(kind, quant):
    rng = np.random.default_rng(42 + 10 * kind)
    gen = _sampler(quant)
    rows = gen(rng, NUM_POINTS)
    deleted = np.zeros(NUM_POINTS, dtype=bool)
    deleted[rng.choice(NUM_POINTS, NUM_POINTS // 10, replace=False)] = True
    raw = O.DenseStorage(O.F32, O.DOT, rows, point_deleted=deleted)
    other = O.ScorerFactory("dense", raw) if quant is None else _quantized_factory(quant, rows, raw)
    top = SAMPLE_SIZE // 10
    shared = []
    for attempt in range(ATTEMPTS):
        examples, n_a, n_b = _random_query(kind, rng, gen)
        points = np.sort(rng.choice(NUM_POINTS, SAMPLE_SIZE, replace=False))
        points = points[~deleted[points]]                             # score_points drops deleted points
        scores = O.custom_scores(raw, examples, kind, n_a, n_b, points)
        scorer, keep = other.custom(list(examples), kind, n_a, n_b)
        other_scores = O.scorer_score_points(scorer, points)
        if quant is None:
            assert np.array_equal(scores.view(np.uint32), other_scores.view(np.uint32)), attempt
            continue
        # `.sorted().rev().take(top)`: a stable ascending sort by score, reversed
        raw_top = set(points[np.argsort(scores, kind="stable")[::-1][:top]].tolist())
        other_top = set(points[np.argsort(other_scores, kind="stable")[::-1][:top]].tolist())
        shared.append(len(raw_top & other_top) / top)
        if quant != "bq_uniform":
            assert shared[-1] >= 0.7, (attempt, n_a, n_b, sorted(raw_top), sorted(other_top))
    if quant == "bq_uniform":
        assert np.mean(shared) >= 0.4, np.mean(shared)        # (ten random points of ~90 would share 0.11)

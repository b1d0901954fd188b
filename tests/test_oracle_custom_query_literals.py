"""The unit tests of the reference's custom queries - lib/segment/src/vector_storage/query/{reco_query,discover_query,context_query}.rs - on the oracle's
`Query::score_by` restatement (qo_custom_combine; the device is held to it bit for bit in test_gpu_custom_queries.py): every literal case with its
expected value, and the proptest properties over 1 000 seeded draws each.  Similarities are fed directly (the reference's tests use an identity
"similarity" over numbers the same way)."""
import numpy as np
import pytest

import oracle_ffi as O

RECO_BEST, RECO_SUM, DISCOVER, CONTEXT = 0, 1, 2, 3
f32 = np.float32


def combine(kind, n_a, n_b, sims):
    a = np.asarray(sims, dtype=np.float32)
    return f32(O._lib.qo_custom_combine(kind, n_a, n_b, O._p(a) if len(a) else None))


def scaled_fast_sigmoid(x):                                  # lib/common/common/src/math.rs:7-18
    x = f32(x)
    return f32(0.5) * (x / (f32(1.0) + abs(x)) + f32(1.0))


def reco_best(pos, neg):
    return combine(RECO_BEST, len(pos), len(neg), list(pos) + list(neg))


def discover(target, pairs):
    return combine(DISCOVER, 1, len(pairs), [target] + [v for p in pairs for v in p])


@pytest.mark.parametrize("pos,neg,positive,expected", [               # reco_query.rs:150-182 score_query
    ([42], [4], True, 42.0), ([4], [42], False, 42.0), ([-1], [0], False, 0.0), ([0], [-1], True, 0.0), ([-42], [-84], True, -42.0),
    ([-84], [-42], False, -42.0), ([1, 2, 3], [4, 5, 6], False, 6.0), ([10, 2, 3], [4, 5, 6], True, 10.0)])
def test_reco_best_score_literals(pos, neg, positive, expected):
    want = scaled_fast_sigmoid(expected) if positive else -scaled_fast_sigmoid(expected)
    assert reco_best(pos, neg) == want


def _ulps_eq(a, b, ulps=80):                                 # reco_query.rs:186-196
    if np.sign(a) != np.sign(b):
        return False
    return abs(int(f32(a).view(np.uint32)) - int(f32(b).view(np.uint32))) <= ulps


def _cmp(a, b):
    return 0 if _ulps_eq(a, b) else (-1 if a < b else 1)


def test_reco_best_score_orders():
    """correct_negative_order, correct_positive_order, correct_positive_and_negative_order (reco_query.rs:206-262), 1 000 draws"""
    rng = np.random.default_rng(1)
    for a, b in rng.uniform(-100.0, 100.0, (1000, 2)).astype(np.float32):
        before = _cmp(a, b)
        after = _cmp(reco_best([], [a]), reco_best([], [b]))
        assert (after == before) if before == 0 else (after != before)                   # a score chosen from the negatives inverts the order
        if before != 0:
            sa, sb = reco_best([a], []), reco_best([b], [])
            assert (-1 if sa < sb else (1 if sa > sb else 0)) == before                  # ... from the positives preserves it
        assert not (reco_best([a], []) < reco_best([], [b]))                             # and a positive choice never ranks below a negative one


@pytest.mark.parametrize("pairs,rank", [                                                 # discover_query.rs:98-121 context_ranking (target 42)
    ([], 0), ([(10, 4)], 1), ([(4, 10)], -1), ([(11, 11)], 0), ([(10, 4), (4, 10)], 0), ([(10, 4), (4, 2)], 2), ([(4, 10), (2, 4)], -2),
    ([(1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (0, 4)], 4)])
def test_discover_rank_literals(pairs, rank):
    assert discover(42, pairs) == f32(rank) + scaled_fast_sigmoid(42)


@pytest.mark.parametrize("target,pairs,order", [                                         # discover_query.rs:123-148 score_better, against 2.5
    (1, [], -1), (1, [(1, 0), (1, 0)], 1), (-1, [(1, 0), (1, 0)], -1), (-1000, [(1, 0), (1, 0), (1, 0)], 1), (1000, [(1, 0), (0, 1)], -1)])
def test_discover_scores_against_a_fixed_score(target, pairs, order):
    s = discover(target, pairs)
    assert (-1 if s < f32(2.5) else 1) == order


def test_discover_score_is_rank_plus_target_part():
    """same_target_only_changes_rank, same_context_only_changes_target (discover_query.rs:150-195), 1 000 draws each"""
    rng = np.random.default_rng(2)
    for _ in range(1000):
        target = f32(rng.uniform(-1000.0, 1000.0))
        p1 = [tuple(x) for x in rng.uniform(0.0, 1000.0, (int(rng.integers(0, 10)), 2)).astype(np.float32)]
        p2 = [tuple(x) for x in rng.uniform(0.0, 1000.0, (int(rng.integers(0, 10)), 2)).astype(np.float32)]
        s1, s2 = discover(target, p1), discover(target, p2)
        assert abs((s1 - np.floor(s1)) - (s2 - np.floor(s2))) <= 1.0e-6
        t2 = f32(rng.uniform(-1000.0, 1000.0))
        assert np.floor(discover(target, p1)) == np.floor(discover(t2, p1))


def test_context_loss_is_between_minus_one_and_zero_per_pair():
    """loss_is_not_more_than_1_per_pair (context_query.rs:150-163), 1 000 draws"""
    rng = np.random.default_rng(3)
    for p, n in rng.uniform(-100.0, 100.0, (1000, 2)).astype(np.float32):
        s = combine(CONTEXT, 0, 1, [p, n])
        assert -1.0 < s <= 0.0

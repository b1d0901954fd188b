"""What the reference does with equal scores, pinned on the oracle's restatement of `FixedLengthPriorityQueue` (= Rust's `BinaryHeap<Reverse<T>>`,
lib/common/common/src/fixed_length_priority_queue.rs:20-65; `Ord for ScoredPointOffset` compares scores only, types.rs:21-25) - the facts
tests/parity_asserts.py builds on.  CPU only."""
import numpy as np

import oracle_ffi as O


def _ids(pairs, k):
    return [int(p["idx"]) for p in O.topk_push_all(pairs, k)]


def test_a_full_queue_rejects_an_equal_score():
    # push :53-57 replaces the root only on strict root < value: the later of two equal scores stays out once the queue is full
    assert _ids([(0, 3.0), (1, 5.0), (2, 3.0)], 2) == [1, 0]
    assert _ids([(0, 1.0), (1, 1.0), (2, 1.0), (3, 1.0)], 3) == [0, 1, 2] or sorted(_ids([(0, 1.0), (1, 1.0), (2, 1.0), (3, 1.0)], 3)) == [0, 1, 2]


def test_equal_scores_inside_the_queue_are_evicted_in_heap_order_not_in_offset_order():
    # two equal scores are inside, a better one arrives: the ROOT goes - here the EARLIER row, so "the lowest offsets survive" is not the reference's rule
    assert sorted(_ids([(0, 3.0), (1, 3.0), (2, 5.0)], 2)) == [1, 2]


def test_the_heaps_survivors_among_ties_differ_from_the_lowest_offsets_on_scan_data():
    """Small integer rows: many equal dot products.  The oracle's linear scan (the reference's heap, pushed in offset order) and the rule
    'sort by (score desc, offset asc), keep k' agree on the scores at every rank, and disagree on WHICH tied rows fill the boundary for a
    good share of the queries - in both directions of the offset order."""
    rng = np.random.default_rng(1)
    n, dim, nq, top = 20_000, 16, 64, 10
    rows = rng.integers(-2, 3, size=(n, dim)).astype(np.float32)
    queries = rng.integers(-2, 3, size=(nq, dim)).astype(np.float32)
    st = O.DenseStorage(O.F32, O.DOT, rows)
    heap = st.peek_top(queries, top)
    exact = st.score_points(queries, np.arange(n, dtype=np.uint32))
    same_set = other_set = 0
    for j in range(nq):
        rule = np.lexsort((np.arange(n), -exact[j]))[:top]
        assert np.array_equal(heap[j]["score"], exact[j][rule])                     # the scores per rank are the same list
        s_k = heap[j]["score"][-1]
        assert set(heap[j]["idx"][heap[j]["score"] > s_k].tolist()) == set(rule[exact[j][rule] > s_k].tolist())   # above the boundary: the same rows
        if set(heap[j]["idx"].tolist()) == set(rule.tolist()):
            same_set += 1
        else:
            other_set += 1
            assert all(exact[j][i] == s_k for i in set(heap[j]["idx"].tolist()) ^ set(rule.tolist()))            # they differ in boundary ties only
    assert other_set > 0 and same_set > 0


def test_assert_reference_lists_accepts_the_rule_and_rejects_a_wrong_boundary():
    import pytest
    from parity_asserts import assert_reference_lists
    rng = np.random.default_rng(2)
    n, dim, nq, top = 5_000, 16, 16, 10
    rows = rng.integers(-2, 3, size=(n, dim)).astype(np.float32)
    queries = rng.integers(-2, 3, size=(nq, dim)).astype(np.float32)
    st = O.DenseStorage(O.F32, O.DOT, rows)
    exact = st.score_points(queries, np.arange(n, dtype=np.uint32))
    rule = []
    for j in range(nq):
        o = np.lexsort((np.arange(n), -exact[j]))[:top]
        a = np.zeros(top, dtype=O.ScoredPointOffset)
        a["idx"], a["score"] = o, exact[j][o]
        rule.append(a)
    assert assert_reference_lists(rule, st, queries, top, threads=0) > 0             # tied boundaries exist and the rule's lists pass
    heap = st.peek_top(queries, top)
    with pytest.raises(AssertionError):                                              # the heap's own lists break the rule somewhere (order or survivors)
        assert_reference_lists(heap, st, queries, top, threads=0)


def test_first_divergence_is_a_tie_classifies_pop_sequences():
    """tests/parity_asserts.first_divergence_is_a_tie, the rule bench.py and the GPU tests apply to a device walk's pop sequence against the oracle's."""
    import numpy as np
    import oracle_ffi as O
    from parity_asserts import first_divergence_is_a_tie as f

    def seq(*pairs):
        a = np.zeros(len(pairs), dtype=O.ScoredPointOffset)
        for i, (idx, sc) in enumerate(pairs):
            a[i] = (idx, sc)
        return a
    base = seq((7, 3.0), (2, 2.5), (9, 2.5), (4, 1.0))
    assert f(base, base.copy()) == "same"
    assert f(seq((7, 3.0), (9, 2.5), (2, 2.5), (4, 1.0)), base) == "tie"                      # the walks part at two equal scores: what follows may differ freely
    assert f(seq((7, 3.0), (9, 2.5), (5, 0.5)), base) == "tie"
    assert f(seq((7, 3.0), (2, 2.5), (9, 2.5), (5, 1.5)), base).startswith("scores differ at 3")   # different candidates with different scores: a defect
    assert f(seq((7, 3.0), (2, 2.25), (9, 2.5)), base) == "equal ids with different scores" or f(seq((7, 3.0), (2, 2.25), (9, 2.5)), base).startswith("scores differ")
    # one walk ends, the other pops one more candidate: a tie only if that candidate equals the bound (strict `candidate.score < lower_bound` in the reference)
    longer = seq((7, 3.0), (2, 2.5), (9, 2.5), (4, 1.0), (11, 1.0))
    assert f(base, longer, bound_score=np.float32(1.0)) == "tie" and f(longer, base, bound_score=np.float32(1.0)) == "tie"
    assert f(base, longer, bound_score=np.float32(0.5)).startswith("one sequence ends at 4")
    assert f(base, longer).endswith("no bound was given")
    # -0.0 and 0.0 are different bits and EQUAL scores for the reference's OrderedFloat: a tie
    assert f(seq((1, 0.0)), seq((2, -0.0))) == "tie"

"""How brute-force lists are compared with the oracle when scores tie.

The reference's result is `FixedLengthPriorityQueue::into_sorted_vec` (lib/common/common/src/fixed_length_priority_queue.rs:47-65) over
`ScoredPointOffset`s whose `Ord` looks at the score ONLY (types.rs:21-25).  Three things follow, and `assert_reference_lists` checks each:

  1. the multiset of the k best scores is unique, so the device's score at every rank must carry the oracle's bits;
  2. a score strictly above the k-th best one brings ALL its rows into the list: per such score the id sets must be equal;
  3. among the rows that tie with the k-th best score the reference keeps whichever its `BinaryHeap` happens to hold - a function of the whole push
     history, not of (score, offset): a full queue rejects an equal score (strict `<` in `push`), but equal elements already inside can be evicted in
     any order when a better one arrives, and `into_sorted_vec` orders equal scores by heap position (tests/test_oracle_ties.py pins both facts on
     the oracle's restatement of the heap).  The device's rule is the deterministic one a linear scan with a stable queue would give: the LOWEST
     offsets among the tied rows, ascending inside equal scores.  That rule is asserted exactly: the boundary group of the device must be the lowest
     offsets of ALL live rows that carry the boundary score (taken from an oracle list long enough to see the whole group, or from the oracle's
     scores of every row when the group is a mass).

Where no score ties, (1) + (2) make the lists identical element for element.  Test infrastructure (imports the oracle)."""
import numpy as np


def assert_reference_lists(got, storage, queries, top, live=None, extra=64, threads=8):
    """got: the device's lists (structured arrays idx / score); storage: oracle_ffi.DenseStorage (or anything with peek_top / score_points) holding
    the same rows and deleted flags; live: bool mask of the rows a search may return (None: all) - only used when a tie group is a mass."""
    want = storage.peek_top(queries, top + extra, threads=threads) if threads else storage.peek_top(queries, top + extra)
    assert len(got) == len(want)
    n_tied_boundaries = 0
    for qi, (g, w_ext) in enumerate(zip(got, want)):
        w = w_ext[:top]
        assert len(g) == len(w), (qi, len(g), len(w))
        assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32)), qi                      # (1)
        if len(g) == 0:
            continue
        gi, gs = g["idx"].astype(np.int64), g["score"]
        for i in range(len(g) - 1):                                                                            # ascending offsets inside equal scores
            if gs[i] == gs[i + 1]:
                assert gi[i] < gi[i + 1], (qi, i)
        s_k = gs[-1]
        above = gs > s_k if s_k == s_k else np.zeros(len(g), dtype=bool)
        assert sorted(gi[above].tolist()) == sorted(w["idx"][w["score"] > s_k].astype(np.int64).tolist()), qi  # (2)
        m = int((~above).sum())
        if len(g) < top:                                                                                       # fewer live rows than k: everything is in
            assert sorted(gi.tolist()) == sorted(w["idx"].astype(np.int64).tolist()), qi
            continue
        # (3) the boundary group: all rows with score == s_k, as far as the longer oracle list shows them
        tail_same = len(w_ext) > top and (w_ext["score"][-1] == s_k or (s_k != s_k and w_ext["score"][-1] != w_ext["score"][-1]))
        if not tail_same and len(w_ext) >= top:
            group = w_ext["idx"][(w_ext["score"] == s_k) | ((s_k != s_k) & (w_ext["score"] != w_ext["score"]))].astype(np.int64)
        else:                                                                                                  # a mass of ties: score every row
            n = storage.rows.shape[0]
            sc = storage.score_points(queries[qi:qi + 1], np.arange(n, dtype=np.uint32))[0]
            mask = (sc == s_k) if s_k == s_k else (sc != sc)
            if live is not None:
                mask &= live
            group = np.flatnonzero(mask).astype(np.int64)
        if len(group) > m:
            n_tied_boundaries += 1
        assert gi[~above].tolist() == np.sort(group)[:m].tolist(), (qi, gi[~above].tolist(), np.sort(group)[:m + 4].tolist())
    return n_tied_boundaries

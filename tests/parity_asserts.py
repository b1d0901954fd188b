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


def first_divergence_is_a_tie(got_pops, want_pops, bound_score=None):
    """Two pop sequences of `search_on_level` over ONE graph (structured arrays idx / score: qmx_hnsw_search_traced on the device,
    qo_hnsw_search_traced in the oracle).  A walk is a deterministic function of its pop sequence, so:
      "same"  the sequences are equal: the walks are the same walk;
      "tie"   at the first position where they differ both popped candidates carry equal scores (OrderedFloat equality: the same bits, or +0.0 / -0.0) (each walk picked another of several equal
              candidates: the reference by the arrangement of its BinaryHeap, the device by ascending id) - or one sequence ends there and the other pops
              one more candidate whose score is bit-equal to `bound_score` (the worst score of the full `nearest` list at that moment: the reference
              breaks on strict `candidate.score < lower_bound` only, graph_layers.rs:126, so a candidate EQUAL to the bound is still expanded when it
              is still in `candidates`; which of several candidates equal to the bound is still there is again the heap's choice);
      anything else ("scores differ at i: ...") is a defect."""
    n = min(len(got_pops), len(want_pops))
    gi, wi = got_pops["idx"][:n], want_pops["idx"][:n]
    gs, ws = got_pops["score"][:n].view(np.uint32), want_pops["score"][:n].view(np.uint32)
    diff = np.nonzero(gi != wi)[0]
    if len(diff):
        i = int(diff[0])
        if not np.array_equal(gs[:i], ws[:i]):
            return "scores differ before the first id difference (%d)" % i
        a, b = np.float32(got_pops["score"][i]), np.float32(want_pops["score"][i])
        equal = gs[i] == ws[i] or a == b or (a != a and b != b)          # OrderedFloat equality: the same bits, +0.0 / -0.0, NaN with NaN
        return "tie" if equal else "scores differ at %d: %r vs %r" % (i, got_pops[i], want_pops[i])
    if not np.array_equal(gs, ws):
        return "equal ids with different scores"
    if len(got_pops) == len(want_pops):
        return "same"
    longer = got_pops if len(got_pops) > len(want_pops) else want_pops
    if bound_score is None:
        return "one sequence ends at %d and no bound was given" % n
    extra, bound = np.float32(longer["score"][n]), np.float32(bound_score)
    equal = extra.view(np.uint32) == bound.view(np.uint32) or extra == bound or (extra != extra and bound != bound)
    return "tie" if equal else "one sequence ends at %d, the other pops %r (bound %r)" % (n, longer[n], bound_score)

"""GPU parity of the custom-query scorers (recommend best-score / sum-scores, discover, context, naive feedback) against the oracle's
restatement of Query::score_by (vector_storage/query/{reco,discover,context,feedback}_query.rs) over bit-exact similarities:
every score bit-exact, brute-force top-k identical.  Known answers from the reference's own unit tests included."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    import qdrant_amd
    assert qdrant_amd.device_count() >= 1
    return qdrant_amd


def _dist(qa, d):
    return {O.COSINE: qa.Distance.Cosine, O.DOT: qa.Distance.Dot, O.EUCLID: qa.Distance.Euclid, O.MANHATTAN: qa.Distance.Manhattan}[d]


@pytest.mark.filterwarnings("ignore::RuntimeWarning")
def test_combine_known_answers():
    """reco_query.rs / context_query.rs / discover_query.rs unit-test style literals on the oracle (CPU part of this file's oracle use)."""
    def comb(kind, n_a, n_b, sims):
        a = np.asarray(sims, dtype=np.float32)
        return O._lib.qo_custom_combine(kind, n_a, n_b, O._p(a))
    sig = lambda x: np.float32(0.5) * (np.float32(x) / (np.float32(1.0) + abs(np.float32(x))) + np.float32(1.0))   # noqa: E731
    assert comb(0, 2, 1, [0.5, 2.0, 1.0]) == sig(2.0)                      # best positive wins
    assert comb(0, 1, 2, [0.5, 2.0, 1.0]) == -sig(2.0)                     # best negative wins -> negated
    assert np.isnan(comb(0, 0, 0, []))                                     # no examples: -scaled_fast_sigmoid(-inf) = -(0.5 * (-inf / inf + 1)) = NaN, as in Rust
    assert comb(1, 2, 2, [1.0, 2.0, 0.5, 0.25]) == np.float32(2.25)
    assert comb(2, 1, 2, [0.0, 3.0, 1.0, 1.0, 2.0]) == np.float32(0.0) + sig(0.0)          # ranks +1 and -1
    assert comb(2, 1, 2, [1.0, 3.0, 1.0, 5.0, 2.0]) == np.float32(2.0) + sig(1.0)
    d = np.float32(1.0) - np.float32(3.0) - np.float32(1.1920929e-07)
    assert comb(3, 0, 2, [3.0, 1.0, 1.0, 3.0]) == np.float32(0.0) + d / (np.float32(1.0) + abs(d))   # first pair on the right side: loss 0
    # FeedbackQuery::score_by (feedback_query.rs:198-226): a * sim(target) + sum pc * (sim(pos) - sim(neg))
    sims, cf = np.asarray([2.0, 1.0, 0.25, 0.5, 3.0], dtype=np.float32), np.asarray([0.5, 2.0, 4.0], dtype=np.float32)
    assert O._lib.qo_custom_feedback(2, O._p(sims), O._p(cf)) == np.float32(0.5 * 2.0 + 2.0 * 0.75 + 4.0 * -2.5)


@pytest.mark.parametrize("dist", [O.COSINE, O.DOT, O.EUCLID, O.MANHATTAN])
@pytest.mark.parametrize("dim", [24, 96])
def test_custom_scores_and_topk_bit_exact(qa, dist, dim):
    rng = np.random.default_rng(dim + dist)
    n = 3000
    rows = O.preprocess(dist, rng.standard_normal((n, dim)).astype(np.float32))
    vs = qa.VectorStorage(rows, _dist(qa, dist))
    deleted = rng.random(n) < 0.15
    vs.set_deleted(deleted, None)
    st = O.DenseStorage(O.F32, dist, rows, point_deleted=deleted)
    V = lambda k: [rng.standard_normal(dim).astype(np.float32) for _ in range(k)]       # noqa: E731
    queries = [
        qa.CustomQuery.recommend_best_score(V(3), V(2)),
        qa.CustomQuery.recommend_best_score(V(1), []),
        qa.CustomQuery.recommend_best_score([], V(2)),
        qa.CustomQuery.recommend_sum_scores(V(4), V(3)),
        qa.CustomQuery.discover(V(1)[0], [tuple(V(2)) for _ in range(3)]),
        qa.CustomQuery.context([tuple(V(2)) for _ in range(4)]),
        qa.CustomQuery.context([]),
        # FeedbackNaive: 4 scored feedback vectors -> 6 ordered pairs above the margin, coefficients (a, b, c)
        qa.CustomQuery.feedback_naive(V(1)[0], list(zip(V(4), [0.9, 0.1, 0.5, 0.7])), a=0.8, b=1.5, c=0.3),
        qa.CustomQuery.feedback_naive(V(1)[0], [(V(1)[0], 0.4)], a=1.25, b=2.0, c=1.0),        # one item: no pairs, a * sim(target)
    ]
    assert queries[-2].n_b == 6 and len(queries[-2].coefs) == 7 and queries[-1].n_b == 0
    scorer = qa.CustomRawScorer(queries, vs)
    ids = rng.permutation(n)[:500].astype(np.uint32)
    got = scorer.score_points(ids)
    for qi, q in enumerate(queries):
        ex = np.stack(q.examples) if q.examples else np.zeros((0, dim), dtype=np.float32)
        if len(q.examples):
            want = O.custom_scores(st, ex, q.kind, q.n_a, q.n_b, ids, q.coefs)
        else:
            want = np.full(len(ids), O._lib.qo_custom_combine(q.kind, q.n_a, q.n_b, None), dtype=np.float32)
        assert np.array_equal(got[qi].view(np.uint32), want.view(np.uint32)), qi
    # brute force over every live point: BatchFilteredSearcher over custom scorers
    top = 100                                                                          # > 64: two bounded passes
    res = scorer.peek_top(top)
    all_ids = np.arange(n, dtype=np.uint32)
    for qi, q in enumerate(queries):
        if not q.examples:
            continue
        sc = O.custom_scores(st, np.stack(q.examples), q.kind, q.n_a, q.n_b, all_ids, q.coefs)
        live = ~deleted
        order = np.lexsort((all_ids[live], -sc[live].astype(np.float64)))[:top]       # descending score, ties -> lower id
        assert np.array_equal(res[qi]["score"].view(np.uint32), sc[live][order].view(np.uint32)), qi
        uniq = np.array([(res[qi]["score"] == x).sum() == 1 for x in res[qi]["score"]])
        assert np.array_equal(res[qi]["idx"][uniq], all_ids[live][order][uniq])
        assert not deleted[res[qi]["idx"]].any()
    # candidate list + payload filter bitmap
    allowed = rng.random(n) < 0.5
    scorer.examples.set_filter(allowed)
    sub = scorer.peek_top(10, points=ids)
    for qi, q in enumerate(queries):
        if not q.examples:
            continue
        ok = allowed[ids] & ~deleted[ids]
        sc = O.custom_scores(st, np.stack(q.examples), q.kind, q.n_a, q.n_b, ids, q.coefs)
        want = np.sort(sc[ok])[::-1][:10]
        assert np.array_equal(sub[qi]["score"].view(np.uint32), want.view(np.uint32))
        assert allowed[sub[qi]["idx"]].all()


def test_custom_query_argument_errors(qa):
    from qdrant_amd import _ffi as F
    rows = np.random.default_rng(0).standard_normal((50, 32)).astype(np.float32)
    vs = qa.VectorStorage(rows, qa.Distance.Dot)
    ex = qa.new_raw_scorer(rows[:3], vs)
    d = (F.CustomQuery * 1)()
    d[0].kind, d[0].first, d[0].n_a, d[0].n_b = F.CUSTOM_RECO_SUM_SCORES, 1, 2, 1                 # needs examples 1..3, only 0..2 exist
    out = np.zeros((1, 4), dtype=np.float32)
    ids = np.arange(4, dtype=np.uint32)
    assert F.lib().qmx_custom_score_points(ex._h, d, 1, F.ptr(ids), 4, F.ptr(out)) == F.ERR_OUT_OF_BOUNDS
    d[0].kind, d[0].first, d[0].n_a, d[0].n_b = F.CUSTOM_DISCOVER, 0, 2, 0                        # two targets
    assert F.lib().qmx_custom_score_points(ex._h, d, 1, F.ptr(ids), 4, F.ptr(out)) == F.ERR_BAD_ARG


@pytest.mark.parametrize("n,top", [(700, 10), (8192, 10), (9765, 64), (16384, 100), (16385, 10), (5000, 700), (300, 1), (40, 64)])
def test_topk_of_short_score_rows_both_kernels(qa, n, top):
    """launch_custom_topk: rows of up to 16 384 scores are pruned by a bound from the lane maxima and rank-sorted (custom_topk_small_kernel), longer ones and
    option no_topk_small = 1 go through the insertion kernel: the same lists - descending score, ties by ascending id, deleted points skipped - as numpy's."""
    dim = 16
    rng = np.random.default_rng(n + top)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    rows[n // 3] = rows[n // 5]                                   # equal scores: ordered by id
    vs = qa.VectorStorage(rows, qa.Distance.Dot)
    deleted = rng.random(n) < 0.3
    vs.set_deleted(deleted, None)
    queries = [qa.CustomQuery.recommend_sum_scores([rng.standard_normal(dim).astype(np.float32)], []) for _ in range(5)]
    scorer = qa.CustomRawScorer(queries, vs)
    sc = scorer.score_points(np.arange(n, dtype=np.uint32))
    lists = []
    for flag in (0, 1):
        qa.set_option("no_topk_small", flag)
        try:
            lists.append(scorer.peek_top(top))
        finally:
            qa.set_option("no_topk_small", -1)
    ids = np.arange(n)
    for qi in range(len(queries)):
        live = ~deleted
        order = np.lexsort((ids[live], -sc[qi][live].astype(np.float64)))[:top]
        for res in lists:
            assert res[qi]["idx"].tolist() == ids[live][order].tolist()
            assert np.array_equal(res[qi]["score"].view(np.uint32), sc[qi][live][order].view(np.uint32))

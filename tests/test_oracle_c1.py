"""BASELINE.json configs[0] ("C1") at its stated size, on the CPU: one segment of 100 000 x 128 f32, cosine, brute-force exact top-10 through the
oracle's restatement of `BatchFilteredSearcher::peek_top_iter` (index/hnsw_index/point_scorer.rs:423-472) over the AVX2+FMA `CosineMetric`
(spaces/simple_avx.rs:32-213), against a float64 numpy ground truth.  The reference's own shapes of this experiment:
lib/segment/benches/vector_search.rs:21,34-104 (100 k x 65 536 there; NUM_VECTORS / DIM constants), and
lib/segment/tests/integration/exact_search_test.rs:165-236 (exact search == plain search, top-k id sets compared).
SURVEY 8(d): rows ~ N(0,1) normalised through cosine_preprocess, seed 0x5EED0000 + config id, 1 024 distinct queries, top = 10."""
import numpy as np

import oracle_ffi as O

N, DIM, NQ, TOP = 100_000, 128, 1024, 10
SEED = 0x5EED0001


def c1_inputs():
    rows = O.preprocess(O.COSINE, O.synth(SEED, 0, N, DIM))
    queries = O.synth(SEED + 1, 0, NQ, DIM)
    return rows, queries


def test_c1_oracle_peek_top_iter_against_float64_ground_truth():
    rows, queries = c1_inputs()
    assert rows.shape == (N, DIM) and queries.shape == (NQ, DIM)
    got = O.DenseStorage(O.F32, O.COSINE, rows).peek_top(queries, TOP, threads=8)
    qn = O.preprocess(O.COSINE, queries).astype(np.float64)           # Metric::preprocess of the query (cosine: normalise), then exact arithmetic
    r64 = rows.astype(np.float64)
    swapped = 0
    for q0 in range(0, NQ, 128):
        exact = qn[q0:q0 + 128] @ r64.T                                # [128, N] float64
        order = np.argsort(-exact, axis=1, kind="stable")[:, :TOP + 1]
        for j in range(exact.shape[0]):
            g = got[q0 + j]
            assert len(g) == TOP
            want_ids = order[j, :TOP]
            # scores: f32 AVX2+FMA chain vs float64, <= 1e-5 relative (north_star's tolerance for f32 distances)
            ex = exact[j, g["idx"].astype(np.int64)]
            assert np.all(np.abs(g["score"].astype(np.float64) - ex) <= 1e-5 * np.maximum(np.abs(ex), 1e-30)), (q0 + j)
            assert np.all(np.diff(g["score"]) <= 0)                                        # into_sorted_vec: descending
            if set(g["idx"].tolist()) != set(want_ids.tolist()):
                # the only legitimate difference: a swap across the k-th place between scores float32 cannot tell apart
                kth = exact[j, order[j, TOP - 1]]
                extra = [i for i in g["idx"].tolist() if i not in set(want_ids.tolist())]
                assert all(abs(exact[j, i] - kth) <= 2e-7 for i in extra), (q0 + j, extra)
                swapped += 1
    assert swapped <= 2      # (iid rows: the 10th and 11th best are ~1e-3 apart; a swap is a rarity, not a mode)


def test_c1_batched_equals_one_query_at_a_time_and_one_thread():
    """peek_top_iter scores a batch against 64-id chunks (point_scorer.rs:433-451): per query the result is that of the single-query search, whatever
    the batch or the number of threads the oracle splits the rows over."""
    rows, queries = c1_inputs()
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    a = st.peek_top(queries[:64], TOP, threads=8)
    b = st.peek_top(queries[:64], TOP, threads=0)
    for j in range(64):
        assert a[j].tobytes() == b[j].tobytes()
    for j in (0, 17, 63):
        c = st.peek_top(queries[j:j + 1], TOP, threads=0)[0]
        assert c.tobytes() == a[j].tobytes()

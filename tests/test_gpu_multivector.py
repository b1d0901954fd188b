"""Multi-dense vectors with the MaxSim comparator (score_max_similarity, lib/segment/src/vector_storage/query_scorer/mod.rs:70-97)
through the C-ABI (qmx_multi_score_points / qmx_multi_search_topk) against the oracle: the similarities are the dense leaves'
bits and the two loops are the reference's, so scores are BIT-EXACT.  The CPU test pins the oracle to the reference's own literal
(`test_score_multi_euclidean`, query_scorer/mod.rs:168-184)."""
import numpy as np
import pytest

import oracle_ffi as O


def test_oracle_maxsim_matches_the_reference_literal():
    a = np.array([[1, 2, 3], [3, 3, 3], [4, 5, 6]], dtype=np.float32)
    b = np.array([[3, 3, 3], [4, 2, 1]], dtype=np.float32)

    def table(x, y):
        return np.array([[O.similarity(O.F32, O.EUCLID, u, v) for v in y] for u in x], dtype=np.float32)
    assert O.max_similarity(table(a, a)) == -0.0            # distance to itself
    assert O.max_similarity(table(a, b)) == np.float32(-19.0)


@pytest.fixture(scope="module")
def qa():
    import qdrant_amd
    assert qdrant_amd.device_count() >= 1
    return qdrant_amd


def _dist(qa, d):
    return {O.COSINE: qa.Distance.Cosine, O.DOT: qa.Distance.Dot, O.EUCLID: qa.Distance.Euclid, O.MANHATTAN: qa.Distance.Manhattan}[d]


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.gpu
def test_reference_literal_on_the_device(qa):
    a = np.array([[1, 2, 3], [3, 3, 3], [4, 5, 6]], dtype=np.float32)
    b = np.array([[3, 3, 3], [4, 2, 1]], dtype=np.float32)
    st = qa.MultiDenseVectorStorage(np.concatenate([a, b]), [0, 3, 5], qa.Distance.Euclid)
    got = st.score_points([a], [0, 1])
    assert got[0, 0] == 0.0 and got[0, 1] == np.float32(-19.0)


@pytest.mark.gpu
@pytest.mark.parametrize("distance,dim", [(O.COSINE, 128), (O.DOT, 96), (O.EUCLID, 33), (O.MANHATTAN, 768)])
def test_maxsim_scores_and_topk_bit_exact(qa, distance, dim):
    rng = np.random.default_rng(dim)
    n_points = 400
    lens = rng.integers(1, 12, n_points)
    offsets = np.zeros(n_points + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    inner = rng.standard_normal((int(offsets[-1]), dim)).astype(np.float32)
    stored = O.preprocess(distance, inner)                      # cosine rows are normalised at insert
    ost = O.DenseStorage(O.F32, distance, stored)
    st = qa.MultiDenseVectorStorage(stored, offsets, _dist(qa, distance))
    queries = [rng.standard_normal((k, dim)).astype(np.float32) for k in (1, 4, 9, 32)]
    first = np.concatenate([[0], np.cumsum([len(q) for q in queries])]).astype(np.uint32)
    flat = np.concatenate(queries)
    ids = rng.permutation(n_points).astype(np.uint32)[:150]
    want = O.multi_scores(ost, flat, first, offsets, ids)
    assert np.array_equal(_bits(st.score_points(queries, ids)), _bits(want))
    full = O.multi_scores(ost, flat, first, offsets, np.arange(n_points))
    for res, sc in zip(st.peek_top_all(queries, 10), full):
        assert np.array_equal(_bits(res["score"]), _bits(np.sort(sc)[::-1][:10]))
        assert np.array_equal(_bits(sc[res["idx"]]), _bits(res["score"]))
    # deleted points (per point, not per inner vector) and a candidate id list
    deleted = rng.random(n_points) < 0.3
    st.set_deleted(deleted)
    for res, sc in zip(st.peek_top_all(queries, 7, ids=ids), full):
        live = ids[~deleted[ids]]
        assert not deleted[res["idx"]].any() and set(res["idx"].tolist()) <= set(ids.tolist())
        assert np.array_equal(_bits(res["score"]), _bits(np.sort(sc[live])[::-1][:7]))


@pytest.mark.gpu
def test_maxsim_argument_errors(qa):
    inner = np.random.default_rng(0).standard_normal((20, 16)).astype(np.float32)
    st = qa.MultiDenseVectorStorage(inner, [0, 5, 20], qa.Distance.Dot)
    with pytest.raises(qa.QmxError):
        st.score_points([inner[:2]], [2])                       # point id past the storage
    bad = qa.MultiDenseVectorStorage(inner, [0, 25], qa.Distance.Dot)
    with pytest.raises(qa.QmxError):
        bad.score_points([inner[:2]], [0])                      # offsets past the inner rows

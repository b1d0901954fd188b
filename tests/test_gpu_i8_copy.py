"""The int8 copy of an f32 block (QMX_SEG_I8_COPY; qdrant_amd/csrc/scan_split.hip, "The INT8 copy"): the prefilter streams one byte per element,
multiplies on the int8 matrix cores and keeps every row whose approximate score lies within a worst-case band of an exact lower bound of the
k-th best score; the survivors are re-scored with the exact gather kernel.  As with the f16 copies the approximate scores never leave the
library: the result must be the exact scan's - the oracle's - ids and score bits, ties included, whatever the data does to the band."""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    import qdrant_amd
    assert qdrant_amd.device_count() >= 1
    return qdrant_amd


def _same(got, want):
    """Score bits identical, ids identical; where scores tie (the synthetic rows are coarse: equal sums happen) the ids agree as sets and come lower id
    first, as the linear scan keeps them (the threaded oracle merges its parts' equal scores in another order)."""
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32))
        if g["idx"].tolist() != w["idx"].tolist():
            assert sorted(g["idx"].tolist()) == sorted(w["idx"].tolist())
            for i in range(len(g) - 1):
                if g["score"][i] == g["score"][i + 1]:
                    assert g["idx"][i] < g["idx"][i + 1]
            assert all(g["score"][i] == w["score"][i] for i in range(len(g)) if g["idx"][i] != w["idx"][i])


def _kernel(qa, searcher):
    return qa._ffi.last_kernel(searcher.scorer._h)


N = 300_000          # >= 2^18: the prefilter applies


@pytest.mark.parametrize("distance,dim,nq,top", [(O.COSINE, 128, 128, 10), (O.DOT, 256, 65, 1), (O.COSINE, 768, 200, 10), (O.DOT, 128, 130, 64),
                                                 (O.COSINE, 512, 70, 10), (O.COSINE, 128, 1, 10), (O.DOT, 256, 300, 5), (O.COSINE, 128, 7, 3)])
def test_int8_copy_returns_the_exact_scan(qa, distance, dim, nq, top):
    n = N if dim < 512 else 270_000
    rows = O.preprocess(distance, O.synth(0x5EED0700 + dim, 0, n, dim))
    queries = O.synth(0x5EED0701 + nq, 0, nq, dim)
    st = O.DenseStorage(O.F32, distance, rows)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine if distance == O.COSINE else qa.Distance.Dot, flags=qa._ffi.SEG_I8_COPY)
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    assert "scan_i8copy_kernel" in _kernel(qa, s), _kernel(qa, s)
    _same(got, st.peek_top(queries, top, threads=8))
    assert s.counters.prefilter_queries == nq and s.counters.fallback_queries == 0
    assert s.counters.verified_rows <= 2048 * nq


def test_int8_copy_with_deleted_rows_and_filter(qa):
    n, dim, nq, top = N, 128, 100, 10
    rng = np.random.default_rng(3)
    rows = O.preprocess(O.COSINE, O.synth(0x5EED0710, 0, n, dim))
    queries = O.synth(0x5EED0711, 0, nq, dim)
    deleted = rng.random(n) < 0.4
    vdel = rng.random(n) < 0.05
    st = O.DenseStorage(O.F32, O.COSINE, rows, point_deleted=deleted, vec_deleted=vdel)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine, flags=qa._ffi.SEG_I8_COPY)
    vs.set_deleted(deleted, vdel)
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    assert "scan_i8copy_kernel" in _kernel(qa, s)
    _same(got, st.peek_top(queries, top, threads=8))
    allowed = rng.random(n) < 0.3
    s.scorer.set_filter(allowed)
    st_f = O.DenseStorage(O.F32, O.COSINE, rows, point_deleted=deleted | ~allowed, vec_deleted=vdel)
    _same(s.peek_top_all(), st_f.peek_top(queries, top, threads=8))


@pytest.mark.parametrize("kind", ["outlier_columns", "sparse", "scaled_rows"])
def test_int8_copy_on_rows_that_stretch_the_band(qa, kind):
    """Columns forty times the others (their own scale: nothing lost), rows that are mostly zeros, rows of very different lengths (dot): the band
    grows, more rows are verified or a query takes the exact scan - the lists stay the exact scan's."""
    n, dim, nq, top = N, 128, 96, 10
    rng = np.random.default_rng(11)
    rows = O.synth(0x5EED0720, 0, n, dim)
    if kind == "outlier_columns":
        rows[:, :3] *= 40.0
    elif kind == "sparse":
        rows *= rng.random((n, dim)) < 0.05
    else:
        rows *= np.exp(rng.standard_normal((n, 1))).astype(np.float32)
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    queries = O.synth(0x5EED0721, 0, nq, dim)
    st = O.DenseStorage(O.F32, O.DOT, rows)
    vs = qa.VectorStorage(rows, qa.Distance.Dot, flags=qa._ffi.SEG_I8_COPY)
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    assert "scan_i8copy_kernel" in _kernel(qa, s)
    _same(got, st.peek_top(queries, top, threads=8))
    assert s.counters.prefilter_queries == nq


def test_int8_copy_falls_back_when_scores_tie_in_masses(qa):
    """Every row exists 3000 times: more rows inside the band than the verification list takes -> the queries take the exact scan behind the prefilter."""
    dim, nq, top, rep = 128, 70, 10, 3000
    base = O.preprocess(O.COSINE, O.synth(0x5EED0730, 0, N // rep, dim))
    rows = np.tile(base, (rep, 1))
    queries = O.synth(0x5EED0731, 0, nq, dim)
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine, flags=qa._ffi.SEG_I8_COPY)
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    assert "scan_i8copy_kernel" in _kernel(qa, s)
    assert s.counters.prefilter_queries == nq and s.counters.fallback_queries == nq
    want = st.peek_top(queries, top)
    for g, w in zip(got, want):
        assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32))
        assert sorted((g["idx"] % len(base)).tolist()) == sorted((w["idx"] % len(base)).tolist())


def test_a_block_with_an_infinite_element_gets_no_int8_copy(qa):
    n, dim, nq, top = N, 128, 80, 5
    rows = O.synth(0x5EED0740, 0, n, dim)
    rows[12345, 7] = np.inf
    queries = O.synth(0x5EED0741, 0, nq, dim)
    vs = qa.VectorStorage(rows, qa.Distance.Dot, flags=qa._ffi.SEG_I8_COPY)
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    assert "scan_i8copy_kernel" not in _kernel(qa, s)
    _same(got, O.DenseStorage(O.F32, O.DOT, rows).peek_top(queries, top, threads=8))

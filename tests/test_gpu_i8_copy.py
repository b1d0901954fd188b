"""The int8 copy of an f32 block (QMX_SEG_I8_COPY; qdrant_amd/csrc/scan_split.hip, "The INT8 copy"): the prefilter streams one byte per element,
multiplies on the int8 matrix cores and keeps every row whose approximate score lies within a worst-case band of an exact lower bound of the
k-th best score; the survivors are re-scored with the exact gather kernel.  As with the f16 copies the approximate scores never leave the
library: the result must be the exact scan's - the oracle's - ids and score bits, ties included, whatever the data does to the band."""
import numpy as np
import pytest

import oracle_ffi as O
from parity_asserts import assert_reference_lists

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    import qdrant_amd
    assert qdrant_amd.device_count() >= 1
    return qdrant_amd


def _same(got, storage, queries, top, live=None):
    """tests/parity_asserts.py: the oracle's score bits at every rank, the same rows for every score above the k-th best, and among the rows that tie
    with the k-th best score the lowest offsets, ascending (the reference's heap keeps whichever ties its push history left inside: test_oracle_ties.py)."""
    return assert_reference_lists(got, storage, queries, top, live=live)


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
    _same(got, st, queries, top)
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
    _same(got, st, queries, top, live=~(deleted | vdel))
    allowed = rng.random(n) < 0.3
    s.scorer.set_filter(allowed)
    st_f = O.DenseStorage(O.F32, O.COSINE, rows, point_deleted=deleted | ~allowed, vec_deleted=vdel)
    _same(s.peek_top_all(), st_f, queries, top, live=~(deleted | ~allowed | vdel))


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
    _same(got, st, queries, top)
    assert s.counters.prefilter_queries == nq


def test_int8_copy_falls_back_when_scores_tie_in_masses(qa):
    """Every row exists 30 000 times: 70 queries with 30 000 tied rows each at the top want more rows than the batch's verification pool holds -> the
    queries that come too late take the exact scan behind the prefilter, the others verify their ties."""
    dim, nq, top, rep = 128, 70, 10, 30_000
    base = O.preprocess(O.COSINE, O.synth(0x5EED0730, 0, N // rep, dim))
    rows = np.tile(base, (rep, 1))
    queries = O.synth(0x5EED0731, 0, nq, dim)
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine, flags=qa._ffi.SEG_I8_COPY)
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    assert "scan_i8copy_kernel" in _kernel(qa, s)
    assert s.counters.prefilter_queries == nq and 0 < s.counters.fallback_queries <= nq
    # every result score is a 30 000-fold tie: the device keeps the LOWEST offsets of each tied group (parity_asserts.py scores every row for these) ...
    assert _same(got, st, queries, top) == nq
    # ... which here are the reference's own survivors: the copies of a row arrive in offset order and each enters while the queue's smallest score is
    # below theirs, a better row never evicts them (the root is always a smaller score), and once k of them fill the queue the strict `<` of `push`
    # rejects the rest - the unthreaded oracle (the reference's heap, pushed in offset order) returns the same ids (its order inside equal scores is the heap's)
    want = st.peek_top(queries[:8], top)
    for g, w in zip(got[:8], want):
        assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32))
        assert sorted(g["idx"].tolist()) == sorted(w["idx"].tolist())


def test_a_block_with_an_infinite_element_gets_no_int8_copy(qa):
    n, dim, nq, top = N, 128, 80, 5
    rows = O.synth(0x5EED0740, 0, n, dim)
    rows[12345, 7] = np.inf
    queries = O.synth(0x5EED0741, 0, nq, dim)
    vs = qa.VectorStorage(rows, qa.Distance.Dot, flags=qa._ffi.SEG_I8_COPY)
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    assert "scan_i8copy_kernel" not in _kernel(qa, s)
    _same(got, O.DenseStorage(O.F32, O.DOT, rows), queries, top)


# ---- rows that stretch the worst-case band (DESIGN 3.1e's table), at the headline's row length -------------------------------------------------------
def _family(kind, n, dim, seed):
    """Unit rows of the families of DESIGN 3.1e, generated on the device (torch) and downloaded for the oracle."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn((n, dim), generator=g, device="cuda", dtype=torch.float32)
    if kind == "laplace":
        u = torch.rand((n, dim), generator=g, device="cuda", dtype=torch.float32) - 0.5
        x = -torch.sign(u) * torch.log1p(-2.0 * u.abs().clamp(max=0.4999999))
    elif kind.startswith("student"):
        nu = int(kind[len("student"):])
        chi = torch.zeros((n, dim), device="cuda", dtype=torch.float32)
        for _ in range(nu):
            chi += torch.randn((n, dim), generator=g, device="cuda", dtype=torch.float32) ** 2
        x = x / torch.sqrt(chi / nu)
    elif kind == "dominant8":
        x[:, :8] *= 12.0
    x = x / x.norm(dim=1, keepdim=True)
    return np.ascontiguousarray(x.cpu().numpy())


@pytest.mark.parametrize("kind", ["laplace", "student5", "student3", "dominant8"])
def test_int8_copy_on_heavy_tailed_and_dominant_coordinate_rows(qa, kind):
    """The families on which the worst-case band is widest (heavy-tailed elements: the column maximum is far above the column's spread; a few dominant
    coordinates: the query's one scale is set by them).  Whatever that does to the candidate and verification lists - more rows re-scored, queries that
    take the exact scan - the lists are the exact scan's, and the counters account for every query."""
    n, dim, nq, top = 270_000, 768, 128, 10
    rows = _family(kind, n, dim, 0x5EED0750)
    queries = _family(kind, nq, dim, 0x5EED0751)
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine, flags=qa._ffi.SEG_I8_COPY)
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    assert "scan_i8copy_kernel" in _kernel(qa, s), _kernel(qa, s)
    _same(got, st, queries, top)
    c = s.counters
    assert c.prefilter_queries == nq and c.fallback_queries <= nq
    assert c.verified_rows <= 16384 * (nq - c.fallback_queries)
    info = vs.info()
    assert info["derived_copy"] == "i8" and not info["chosen_by_trial"]
    if kind == "dominant8":
        # the column scales are balanced for blocks like this one (split_i8_choose_scales): the band stays narrow, nobody falls back
        assert info["i8_scale_balance"] > 0 and c.fallback_queries == 0 and c.verified_rows <= 2048 * nq, (info, c.verified_rows)
    else:
        assert info["i8_scale_balance"] == 0.0        # columns of one size keep max |x| / 127


def test_int8_copy_with_queries_that_are_not_finite(qa):
    """A NaN / infinite query has no usable threshold or band: it takes the exact scan alone (NaN scores order as the greatest, types.rs:21-25 over
    ordered-float), the other queries of the batch keep the prefilter."""
    n, dim, nq, top = N, 128, 96, 10
    rows = O.preprocess(O.COSINE, O.synth(0x5EED0760, 0, n, dim))
    queries = O.synth(0x5EED0761, 0, nq, dim)
    queries[5, 17] = np.nan
    queries[40, 3] = np.inf
    st = O.DenseStorage(O.F32, O.DOT, rows)
    vs = qa.VectorStorage(rows, qa.Distance.Dot, flags=qa._ffi.SEG_I8_COPY)
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    assert "scan_i8copy_kernel" in _kernel(qa, s)
    want = st.peek_top(queries, top)
    for qi, (g, w) in enumerate(zip(got, want)):
        if qi == 5:        # every score is NaN: a 300 000-fold tie of the greatest value - the lowest offsets (and the heap's: it fills with rows 0..9 and rejects the rest)
            assert np.isnan(g["score"]).all() and g["idx"].tolist() == list(range(top)) and sorted(w["idx"].tolist()) == list(range(top))
        elif qi == 40:     # NaN where row[3] == 0 (0 x inf; the generator's rows hold exact zeros), +inf where row[3] > 0: NaN orders as the greatest
            nan_rows, inf_rows = np.flatnonzero(rows[:, 3] == 0), np.flatnonzero(rows[:, 3] > 0)
            want_ids = np.concatenate([nan_rows, inf_rows])[:top]
            assert g["idx"].tolist() == want_ids.tolist()
            assert np.isnan(g["score"][:min(top, len(nan_rows))]).all() and np.isposinf(g["score"][len(nan_rows):]).all()
            assert np.array_equal(np.isnan(g["score"]), np.isnan(w["score"]))
        else:
            assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32)), qi
            assert g["idx"].tolist() == w["idx"].tolist(), qi
    assert 2 <= s.counters.fallback_queries <= 4


@pytest.mark.parametrize("kind,want_copy", [("gauss", "i8"), ("dominant8", "i8"), ("student3", None)])
def test_auto_copy_is_chosen_by_the_trial_and_reported(qa, kind, want_copy):
    """QMX_SEG_AUTO_COPY: the library builds the int8 copy, searches 128 stored rows through it and keeps it when they verify few rows each; otherwise
    the half copy is built beside it, the same batch is timed through both and the faster one stays.  Whatever it chose, qmx_segment_get_info says
    so, and the lists are the exact scan's."""
    n, dim, nq, top = 270_000, 768, 64, 10
    rows = _family(kind, n, dim, 0x5EED0770)
    queries = _family(kind, nq, dim, 0x5EED0771)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine, flags=qa._ffi.SEG_AUTO_COPY)
    info = vs.info()
    assert info["chosen_by_trial"] and info["derived_copy"] in ("i8", "half") and info["trial_i8_ms"] > 0
    if want_copy:
        assert info["derived_copy"] == want_copy, info
        assert info["trial_half_ms"] == 0.0 and info["trial_i8_fallback_queries"] == 0       # an easy block: the half copy was never built
    else:
        assert info["trial_half_ms"] > 0.0                                                  # a hard one: both were timed, the faster stayed
        assert (info["derived_copy"] == "half") == (info["trial_half_ms"] < info["trial_i8_ms"]), info
    assert info["derived_copy_bytes"] == (n + 255) // 256 * 256 * dim * (1 if info["derived_copy"] == "i8" else 2)
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    assert ("scan_i8copy_kernel" if info["derived_copy"] == "i8" else "scan_f16pair_kernel<true>") in _kernel(qa, s), _kernel(qa, s)
    _same(got, O.DenseStorage(O.F32, O.COSINE, rows), queries, top)


@pytest.mark.parametrize("dim,flag,kernel", [(1024, "i8", "scan_i8copy_kernel"), (1536, "i8", "scan_i8copy_kernel"), (1536, "half", "scan_f16pair_kernel<true>"),
                                             (2048, "auto", "scan_i8copy_kernel")])
def test_derived_copies_serve_rows_of_up_to_2048_floats(qa, dim, flag, kernel):
    """Rows longer than 768 floats (1 024, 1 536: common embedding sizes; the rescoring rows of C4) take the prefilters too since round 4: the copies and
    their scans were never limited by the row length, only the per-query exact fallback was - its 64-query shape keeps the queries in registers and
    stops at 768 floats; beyond, the conditional passes take 32 queries each.  Lists: the oracle's; a hot query (3 000 copies of one row, a per-query
    limit of 2 048 verified rows) exercises that fallback."""
    n, nq, top = 262_400, 70, 10
    rng = np.random.default_rng(dim)
    raw = O.synth(0x5EED0790 + dim, 0, n, dim)
    v = O.synth(0x5EED0791, 0, 1, dim)[0]
    dup_at = rng.choice(n, 3000, replace=False)
    raw[dup_at] = v
    rows = O.preprocess(O.COSINE, raw)
    queries = O.synth(0x5EED0792, 0, nq, dim)
    hot = [3, 41]
    queries[hot] = v + 0.05 * O.synth(0x5EED0793, 0, len(hot), dim)
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine, flags={"i8": qa._ffi.SEG_I8_COPY, "half": qa._ffi.SEG_HALF_COPY, "auto": qa._ffi.SEG_AUTO_COPY}[flag])
    assert vs.info()["derived_copy"] == ("half" if flag == "half" else "i8")
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    assert kernel in _kernel(qa, s), _kernel(qa, s)
    assert s.counters.prefilter_queries == nq and s.counters.fallback_queries == 0
    _same(got, st, queries, top)
    qa.set_option("verify_max_per_query", 2048)
    try:
        s2 = qa.BatchFilteredSearcher(queries, vs, top)
        got2 = s2.peek_top_all()
    finally:
        qa.set_option("verify_max_per_query", -1)
    # (the hot queries for sure; at 2 048 floats a cold query's int8 band may hold more than 2 048 rows too)
    assert len(hot) <= s2.counters.fallback_queries <= len(hot) + 4, s2.counters.fallback_queries
    for a, b in zip(got, got2):
        assert np.array_equal(a, b)


# ---- the two structures of the scan: queries resident in LDS (scan_i8copy_kernel_res<AHEAD>, rows up to 1 024 coordinates) / staged behind a barrier -------
@pytest.mark.parametrize("ahead", [0, 1])
@pytest.mark.parametrize("distance,dim,nq,top", [(O.COSINE, 128, 128, 10), (O.COSINE, 768, 200, 10), (O.DOT, 1024, 70, 5)])
def test_int8_scan_with_resident_and_with_staged_queries_returns_the_exact_scan(qa, ahead, distance, dim, nq, top):
    """Option i8_resident: 0 = the staged kernel, 1 = the resident kernel.  Same integers, same thresholds, same candidates:
    the exact scan's lists either way (with deleted rows, a partial last tile, blocks without a tile in the first phase)."""
    n = 270_011
    rng = np.random.default_rng(5)
    rows = O.preprocess(distance, O.synth(0x5EED0780 + dim, 0, n, dim))
    queries = O.synth(0x5EED0781 + nq, 0, nq, dim)
    deleted = rng.random(n) < 0.2
    st = O.DenseStorage(O.F32, distance, rows, point_deleted=deleted)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine if distance == O.COSINE else qa.Distance.Dot, flags=qa._ffi.SEG_I8_COPY)
    vs.set_deleted(deleted, None)
    qa.set_option("i8_resident", ahead)
    try:
        s = qa.BatchFilteredSearcher(queries, vs, top)
        got = s.peek_top_all()
        want_kernel = "scan_i8copy_kernel_res<1>" if ahead else "scan_i8copy_kernel("
        assert want_kernel in _kernel(qa, s), _kernel(qa, s)
    finally:
        qa.set_option("i8_resident", -1)
    _same(got, st, queries, top, live=~deleted)
    assert s.counters.prefilter_queries == nq and s.counters.fallback_queries == 0


def test_int8_scan_of_long_rows_keeps_the_staged_queries(qa):
    n, dim, nq, top = 262_400, 1152, 96, 10
    rows = O.preprocess(O.COSINE, O.synth(0x5EED0790, 0, n, dim))
    queries = O.synth(0x5EED0791, 0, nq, dim)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine, flags=qa._ffi.SEG_I8_COPY)
    s = qa.BatchFilteredSearcher(queries, vs, top)
    got = s.peek_top_all()
    assert "scan_i8copy_kernel(" in _kernel(qa, s), _kernel(qa, s)
    _same(got, O.DenseStorage(O.F32, O.COSINE, rows), queries, top)

"""The PQ walk with one workgroup per search (qdrant_amd/csrc/hnsw_pq_block.hip; opt-in: option no_hnsw_pq_block = 0 - it measured slower than the
one-wave kernel): the query's LUT in LDS, a controller wave that owns the beam and the visited set, worker waves that fetch and score the
links of the beam's best unexpanded entries ahead of the walk.  It serves walks whose LUT is too large to stage per wave (more than 16 KiB: m > 16
at 256 centroids - C4's m = 96 is 96 KiB).  Speculation must not show: ids, score bits AND the
number of scored points equal the oracle's restatement of GraphLayers::search (graph_layers.rs:108-149,247-317,530-562) with the EncodedVectorsPQ
scorer (encoded_vectors_pq.rs:409-443), and the one-wave-per-search kernel (hnsw_search_kernel<HopPQ>, option no_hnsw_pq_block) returns the same."""
import numpy as np
import pytest

import oracle_ffi as O
from test_gpu_hnsw import _dist, _graph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    import qdrant_amd
    assert qdrant_amd.device_count() >= 1
    qdrant_amd.set_option("no_hnsw_pq_block", 0)        # the walks of this module take the block kernel ...
    yield qdrant_amd
    qdrant_amd.set_option("no_hnsw_pq_block", -1)       # ... the library's default (the one-wave kernel) comes back


def _kernel(qa, scorer):
    return qa._ffi.last_kernel(scorer._h)


def _pq(qa, distance, rows, dim, chunk, seed_rows=2000):
    cen = O.PqOracle.train(rows[:seed_rows], dim, chunk, 256, iters=2)
    opq = O.PqOracle(distance, dim, chunk, cen)
    codes = opq.encode(rows)
    quant = qa.ProductQuantizer(dim, _dist(qa, distance), chunk, cen)
    return opq, qa.EncodedVectorsPQ(codes, quant)


def _check(got, want, min_exact):
    """score bits at every rank; ids wherever the oracle's list has distinct scores (PQ scores - sums of few LUT entries - tie now and then, and
    among equals the reference's order is its heap's)"""
    n_cmp = 0
    for gq, wq in zip(got, want):
        assert np.array_equal(gq["score"].view(np.uint32), wq["score"].view(np.uint32))
        if len(np.unique(wq["score"])) == len(wq):
            assert gq["idx"].tolist() == wq["idx"].tolist()
            n_cmp += 1
    assert n_cmp >= min_exact


@pytest.mark.parametrize("distance,dim,chunk", [(O.DOT, 128, 4), (O.COSINE, 384, 4), (O.EUCLID, 96, 2), (O.DOT, 70, 2)])
@pytest.mark.parametrize("waves", [0, 3, 5])
def test_block_walk_is_the_reference_walk(qa, distance, dim, chunk, waves):
    """m = 32 / 96 / 48 / 35 code bytes (LUTs of 32 .. 96 KiB; 35: a tail behind the last group of four, rows of 48 bytes), default and minimal worker
    counts, narrow and wide beams (ef up to 300: the 512-entry register beam)."""
    n, m, nq = 3000, 8, 24
    rows, st, g, plain = _graph(distance, n, dim, m, 0x5EED0A40 + dim)
    queries = O.synth(0x5EED0A41 + dim, 0, nq, dim)
    qpre = O.preprocess(distance, queries)
    opq, enc = _pq(qa, distance, rows, dim, chunk)
    graph = qa.GraphLayers.from_plain(plain)
    scorer = qa.new_raw_scorer(queries, enc)
    qa.set_option("hnsw_pq_block_waves", waves)
    try:
        for top, ef in [(10, 64), (10, 16), (5, 128), (10, 300), (64, 8), (1, 1)]:
            want, stats = g.search_pq(st, opq, qpre, top, ef, with_stats=True)
            got, scored = graph.search(top, ef, scorer, with_scored=True)
            assert "hnsw_pq_block_kernel" in _kernel(qa, scorer), _kernel(qa, scorer)
            _check(got, want, nq // 2)
            qa.set_option("no_hnsw_pq_block", 1)
            try:
                old, scored_old = graph.search(top, ef, scorer, with_scored=True)
                assert "hnsw_search_kernel" in _kernel(qa, scorer)
            finally:
                qa.set_option("no_hnsw_pq_block", 0)
            for a, b in zip(got, old):
                assert np.array_equal(a, b)
            assert scored == scored_old
            # scored points: equal to the oracle's whenever no search met a tie (a tie may send the two walks down different branches)
            if all(len(np.unique(w["score"])) == len(w) for w in want):
                assert scored == sum(stats)
    finally:
        qa.set_option("hnsw_pq_block_waves", -1)


def test_block_walk_with_deleted_points_and_a_filter(qa):
    distance, dim, chunk, n, m, nq = O.DOT, 128, 4, 3000, 8, 20
    rows, _, g, plain = _graph(distance, n, dim, m, 0x5EED0A40 + dim)
    rng = np.random.default_rng(4)
    deleted = rng.random(n) < 0.2
    st = O.DenseStorage(O.F32, distance, rows, point_deleted=deleted)
    queries = O.synth(0x5EED0A51, 0, nq, dim)
    qpre = O.preprocess(distance, queries)
    opq, enc = _pq(qa, distance, rows, dim, chunk)
    enc.set_deleted(deleted)
    graph = qa.GraphLayers.from_plain(plain)
    scorer = qa.new_raw_scorer(queries, enc)
    want = g.search_pq(st, opq, qpre, 10, 96)
    got = graph.search(10, 96, scorer)
    assert "hnsw_pq_block_kernel" in _kernel(qa, scorer)
    _check(got, want, nq // 2)
    assert not any(deleted[r["idx"]].any() for r in got)
    allowed = rng.random(n) < 0.5
    scorer.set_filter(allowed)
    st_f = O.DenseStorage(O.F32, distance, rows, point_deleted=deleted | ~allowed)
    _check(graph.search(10, 96, scorer), g.search_pq(st_f, opq, qpre, 10, 96), nq // 2)


def test_block_walk_restarts_on_the_bitmap_when_its_visited_set_runs_full(qa):
    """A wide search over a larger graph visits more points than the LDS hash set takes at five eighths of its entries: the search starts over with the
    per-slot bitmap in HBM (the one-wave kernel's visited set).  Graph built on the device through the PQ scorer (m0 = 48); the one-wave kernel and the
    oracle's walk of the exported graph are the references."""
    distance, dim, chunk, n, nq = O.DOT, 128, 4, 40_000, 12
    rng = np.random.default_rng(8)
    centers = rng.standard_normal((64, dim)).astype(np.float32)
    rows = (centers[rng.integers(0, 64, n)] + 0.7 * rng.standard_normal((n, dim))).astype(np.float32)
    queries = (centers[rng.integers(0, 64, nq)] + 0.7 * rng.standard_normal((nq, dim))).astype(np.float32)
    opq, enc = _pq(qa, distance, rows, dim, chunk, seed_rows=4000)
    vs = qa.VectorStorage(rows, _dist(qa, distance))
    graph = qa.GraphLayers.build(enc, m=24, ef_construct=64, seed=5, original=vs)
    scorer = qa.new_raw_scorer(queries, enc)
    qa.set_option("hnsw_pq_block_set", 512)      # a visited set of 512 entries: a search that visits more than 320 points starts over on the bitmap
    for top, ef in [(10, 100), (20, 512)]:
        got, scored = graph.search(top, ef, scorer, with_scored=True)
        assert "hnsw_pq_block_kernel" in _kernel(qa, scorer)
        qa.set_option("no_hnsw_pq_block", 1)
        try:
            old, scored_old = graph.search(top, ef, scorer, with_scored=True)
        finally:
            qa.set_option("no_hnsw_pq_block", 0)
        for a, b in zip(got, old):
            assert np.array_equal(a, b)
        assert scored == scored_old
        assert scored > 320 * nq               # (more visited points per search than 5/8 of the 512-entry set: the restart ran)
    walker = O.Hnsw.from_plain(graph.export_plain(), n)
    st = O.DenseStorage(O.F32, distance, rows)
    want = walker.search_pq(st, opq, O.preprocess(distance, queries), 10, 100)
    _check(graph.search(10, 100, scorer), want, nq // 2)
    # the searches are re-entrant: the same batch again, and a second handle on the same graph
    again = graph.search(10, 100, scorer)
    other = graph.search(10, 100, qa.new_raw_scorer(queries, enc))
    for a, b, c in zip(again, other, graph.search(10, 100, scorer)):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    qa.set_option("hnsw_pq_block_set", -1)

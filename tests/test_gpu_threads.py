"""Threading contract of the boundary: a segment / a graph is shared by any number of host threads
(the reference shares storages `&` across its blocking search pool, segments_searcher.rs:255), a query handle belongs
to one thread.  8 threads, each with its own scorers, hammer one f32 segment, one SQ segment and one graph at once;
every result must equal the serial one."""
import threading

import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


def test_concurrent_searches_on_shared_handles():
    import qdrant_amd as qa
    rng = np.random.default_rng(21)
    n, dim, top = 20000, 96, 10
    centers = rng.standard_normal((64, dim)).astype(np.float32) * 2
    rows = O.preprocess(O.COSINE, (centers[rng.integers(0, 64, n)] + 0.5 * rng.standard_normal((n, dim))).astype(np.float32))
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    quant = qa.ScalarQuantizer.fit(rows, dim, qa.Distance.Dot)
    enc = qa.EncodedVectorsU8(quant.encode(rows), quant)
    graph = qa.GraphLayers.build(vs, m=8, ef_construct=64, seed=1)
    n_threads, rounds = 8, 6
    queries = [O.preprocess(O.COSINE, rng.standard_normal((5 + 3 * t, dim)).astype(np.float32)) for t in range(n_threads)]

    def work(t):
        out = []
        for r in range(rounds):
            q = queries[t]
            out.append(qa.BatchFilteredSearcher(q, vs, top).peek_top_all())            # f32 scan (VALU or matrix-core by batch size)
            out.append(qa.BatchFilteredSearcher(q, enc, top).peek_top_all())           # SQ scan
            out.append(graph.search(top, 64, qa.new_raw_scorer(q, vs)))                 # HNSW walk, f32 scorer
            out.append(graph.search(top, 64, qa.new_raw_scorer(q, enc)))                # HNSW walk, SQ scorer
        return out

    serial = [work(t) for t in range(n_threads)]
    results, errors = [None] * n_threads, []

    def run(t):
        try:
            results[t] = work(t)
        except Exception as e:   # pragma: no cover
            errors.append(e)
    threads = [threading.Thread(target=run, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(n_threads):
        for a, b in zip(results[t], serial[t]):
            for x, y in zip(a, b):
                assert np.array_equal(x, y)


def _plain_equal(a, b):
    return (np.array_equal(a.reindex, b.reindex) and np.array_equal(a.level_offsets, b.level_offsets) and np.array_equal(a.offsets, b.offsets) and
            np.array_equal(a.neighbors, b.neighbors) and np.array_equal(a.ep_ids, b.ep_ids) and np.array_equal(a.ep_levels, b.ep_levels) and
            np.array_equal(a.xp_ids, b.xp_ids) and np.array_equal(a.xp_levels, b.xp_levels))


def test_sharded_build_of_independent_segments_equals_the_single_builds():
    """`qmx_sharded_hnsw_build` (north_star: "index build over independent segments shards across the 8 GPUs"; the reference locks one GPU of its
    pool per segment build, gpu_devices_manager.rs:120-143, hnsw/build.rs:53): N host threads inside the call build N graphs at once - here all on
    one device, where build scratch, per-thread launch-attribute guards and the deleted-flag downloads of the builds interleave.  With one insertion
    per launch a build is deterministic (the sequential graph), so every graph must equal the graph of its own single call link for link; the
    default batched build is checked by walking it (recall against the exact search).  Dense f32, SQ and PQ (through its original) segments."""
    import qdrant_amd as qa
    rng = np.random.default_rng(5)
    dim, top = 64, 10
    sizes = [1500, 2100, 900, 1800, 1200, 2400]
    centers = rng.standard_normal((32, dim)).astype(np.float32) * 2
    rows = [O.preprocess(O.COSINE, (centers[rng.integers(0, 32, n)] + 0.6 * rng.standard_normal((n, dim))).astype(np.float32)) for n in sizes]
    dense = [qa.VectorStorage(r, qa.Distance.Cosine) for r in rows[:4]]
    quant = qa.ScalarQuantizer.fit(rows[4], dim, qa.Distance.Dot)
    sq = qa.EncodedVectorsU8(quant.encode(rows[4]), quant)
    deleted = rng.random(sizes[1]) < 0.1
    dense[1].set_deleted(deleted)
    storages = dense + [sq]
    kw = dict(m=8, ef_construct=40, seed=11, max_batch=1)
    single = [qa.GraphLayers.build(s, **kw).export_plain() for s in storages]
    together = qa.GraphLayers.build_sharded(storages, **kw)
    for i, g in enumerate(together):
        assert _plain_equal(g.export_plain(), single[i]), i
    # twice more, to let the threads meet in other orders
    for _ in range(2):
        for i, g in enumerate(qa.GraphLayers.build_sharded(storages, **kw)):
            assert _plain_equal(g.export_plain(), single[i]), i
    # the batched build (the production setting): every graph is searchable and finds its segment's nearest rows
    batched = qa.GraphLayers.build_sharded(storages, m=8, ef_construct=64, seed=11)
    for i, g in enumerate(batched[:4]):
        q = O.preprocess(O.COSINE, rng.standard_normal((16, dim)).astype(np.float32))
        got = g.search(top, 96, qa.new_raw_scorer(q, storages[i]))
        exact = qa.BatchFilteredSearcher(q, storages[i], top).peek_top_all()
        hit = sum(len(set(a["idx"].tolist()) & set(b["idx"].tolist())) for a, b in zip(got, exact))
        assert hit >= 0.9 * 16 * top, (i, hit)
        if i == 1:
            assert not any(deleted[a["idx"]].any() for a in got)


def test_sharded_build_reports_each_segments_status():
    """A segment the build refuses (here: a PQ segment without its original) fails alone: its status and the call's return value say so, the other
    graphs are complete."""
    import ctypes as C
    import qdrant_amd as qa
    from qdrant_amd import _ffi as F
    rng = np.random.default_rng(6)
    dim = 32
    rows = O.preprocess(O.COSINE, rng.standard_normal((1200, dim)).astype(np.float32))
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    centroids = rows[rng.choice(len(rows), 256, replace=False)].copy()            # any 256 centroids make a valid PQ storage
    pq = qa.ProductQuantizer(dim, qa.Distance.Cosine, 4, centroids)
    enc = qa.EncodedVectorsPQ(pq.encode(rows), pq)
    p = F.HnswBuildParams()
    p.m, p.m0, p.ef_construct, p.entry_points_num, p.seed, p.max_batch = 8, 16, 32, 10, 3, 0
    segs = (C.c_void_p * 2)(vs._h, enc._h)
    outs, status = (C.c_void_p * 2)(), (C.c_int32 * 2)()
    rc = F.lib().qmx_sharded_hnsw_build(segs, None, 2, C.byref(p), outs, status)
    assert rc == F.ERR_NOT_SUPPORTED
    assert status[0] == 0 and status[1] == rc and outs[0] and not outs[1]
    assert "segment 1" in F.last_error()
    F.check(F.lib().qmx_hnsw_destroy(C.c_void_p(outs[0])))
    # ... and with the original it builds
    orig = (C.c_void_p * 2)(None, vs._h)
    rc = F.lib().qmx_sharded_hnsw_build(segs, orig, 2, C.byref(p), outs, status)
    assert rc == 0 and outs[0] and outs[1]
    for o in outs:
        F.check(F.lib().qmx_hnsw_destroy(C.c_void_p(o)))

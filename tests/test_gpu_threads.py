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

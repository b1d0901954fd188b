"""GPU parity of the cross-segment merge (`BatchResultAggregator`, lib/shard/src/search_result_aggregator.rs:50-121)
and of the sharded searcher's device path (one process, world size 1 and "segments on one GPU")."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


def _lists(rng, n_lists, nq, k, n_ids=100000):
    lists = np.zeros((n_lists, nq, k), dtype=O.ScoredPointOffset)
    for l in range(n_lists):
        for q in range(nq):
            lists[l, q]["score"] = np.sort(rng.standard_normal(k).astype(np.float32))[::-1]
            lists[l, q]["idx"] = rng.permutation(n_ids)[:k]
    return lists


@pytest.mark.parametrize("n_lists,nq,k", [(1, 1, 1), (2, 3, 10), (8, 16, 10), (5, 7, 64), (9, 4, 33)])
def test_merge_topk_matches_aggregator(n_lists, nq, k):
    from qdrant_amd import _ffi as F
    rng = np.random.default_rng(n_lists * 100 + k)
    lists = _lists(rng, n_lists, nq, k)
    counts = rng.integers(0, k + 1, size=(n_lists, nq)).astype(np.uint32)
    out = np.zeros((nq, k), dtype=O.ScoredPointOffset)
    oc = np.zeros(nq, dtype=np.uint32)
    F.check(F.lib().qmx_merge_topk(0, F.ptr(lists), F.ptr(counts), n_lists, nq, k, F.ptr(out), F.ptr(oc)))
    want = O.merge_topk(lists, counts, k)
    for q in range(nq):
        assert oc[q] == len(want[q])
        assert out[q, :oc[q]]["idx"].tolist() == want[q]["idx"].tolist()
        assert out[q, :oc[q]]["score"].tolist() == want[q]["score"].tolist()


@pytest.mark.parametrize("n_lists,nq,k", [(1, 1, 1), (2, 3, 10), (8, 128, 10), (5, 7, 64), (3, 5, 100)])
def test_merge_of_packed_records_equals_the_merge_of_the_arrays(n_lists, nq, k):
    """qmx_merge_topk_packed_async: one record per list = [nq][k] points, [nq] counts, padding to 8 bytes (the ONE buffer a rank all-gathers);
    same lists as qmx_merge_topk_async over separate arrays and as the aggregator (search_result_aggregator.rs:50-121), id bases included."""
    import torch
    from qdrant_amd import _ffi as F, sharded
    lib = F.lib()
    rng = np.random.default_rng(n_lists * 1000 + nq * 10 + k)
    lists = _lists(rng, n_lists, nq, k, n_ids=5000)
    counts = rng.integers(0, k + 1, size=(n_lists, nq)).astype(np.uint32)
    bases = (np.arange(n_lists, dtype=np.uint32) * 5000).astype(np.uint32)
    words = sharded.record_words(nq, k)
    assert words * 4 == lib.qmx_topk_record_bytes(nq, k)
    rec = np.full((n_lists, words), 0x7FFFFFFF, dtype=np.int32)            # padding words hold rubbish
    for l in range(n_lists):
        rec[l, :nq * k * 2] = lists[l].view(np.int32).reshape(-1)
        rec[l, nq * k * 2:nq * k * 2 + nq] = counts[l].view(np.int32)
    dev = torch.device("cuda", 0)
    d_rec, d_base = torch.from_numpy(rec).to(dev), torch.from_numpy(bases.view(np.int32)).to(dev)
    out, oc = torch.zeros((nq, k, 2), dtype=torch.int32, device=dev), torch.zeros((nq,), dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(dev)
    F.check(lib.qmx_merge_topk_packed_async(0, C.c_void_p(st.cuda_stream), F.ptr(d_rec), F.ptr(d_base), n_lists, nq, k, F.ptr(out), F.ptr(oc)))
    st.synchronize()
    want = O.merge_topk(lists, counts, k, bases)
    got, gc = out.cpu().numpy(), oc.cpu().numpy()
    for q in range(nq):
        assert gc[q] == len(want[q])
        assert got[q, :gc[q], 0].view(np.uint32).tolist() == want[q]["idx"].tolist()
        assert got[q, :gc[q], 1].copy().view(np.float32).tolist() == want[q]["score"].tolist()
    # and through ShardedSearcher at world 1: the local search writes lists and counts straight into the record
    d_l, d_c = torch.from_numpy(lists.view(np.int32).reshape(n_lists, nq, k, 2)).to(dev), torch.from_numpy(counts.view(np.int32)).to(dev)
    out2, oc2 = torch.zeros_like(out), torch.zeros_like(oc)
    F.check(lib.qmx_merge_topk_async(0, C.c_void_p(st.cuda_stream), F.ptr(d_l), F.ptr(d_c), F.ptr(d_base), n_lists, nq, k, F.ptr(out2), F.ptr(oc2)))
    st.synchronize()
    assert torch.equal(oc, oc2)
    for q in range(nq):
        assert torch.equal(out[q, :gc[q]], out2[q, :gc[q]])


def test_merge_ties_prefer_the_lower_id_and_nan_sorts_first():
    from qdrant_amd import _ffi as F
    lists = np.zeros((2, 1, 4), dtype=O.ScoredPointOffset)
    lists[0, 0] = [(9, 1.0), (7, 0.5), (3, 0.5), (1, -1.0)]
    lists[1, 0] = [(4, np.nan), (8, 1.0), (2, 0.5), (0, -2.0)]
    out = np.zeros((1, 4), dtype=O.ScoredPointOffset)
    oc = np.zeros(1, dtype=np.uint32)
    F.check(F.lib().qmx_merge_topk(0, F.ptr(lists), None, 2, 1, 4, F.ptr(out), F.ptr(oc)))
    assert oc[0] == 4
    assert out[0]["idx"].tolist() == [4, 8, 9, 2]      # NaN greatest (OrderedFloat), then 1.0 (lower id first), then 0.5 -> id 2
    assert np.isnan(out[0]["score"][0])


def test_two_segments_on_one_gpu_equal_one_big_segment():
    """segments_searcher.rs:250-285: search each segment, merge; ids globalised by the segment base."""
    import torch
    import qdrant_amd as qa
    from qdrant_amd import _ffi as F, sharded
    dim, nq, top = 64, 6, 10
    sizes = [3000, 1700, 2300]
    rows = [O.preprocess(O.COSINE, O.synth(0x5EED0005 + 16 * r, 0, n, dim)) for r, n in enumerate(sizes)]
    queries = O.synth(0x5EED0006, 0, nq, dim)
    dev = torch.device("cuda", 0)
    qd = torch.from_numpy(queries).to(dev)
    gathered = torch.zeros((len(sizes), nq, top, 2), dtype=torch.int32, device=dev)
    gcounts = torch.zeros((len(sizes), nq), dtype=torch.int32, device=dev)
    backends = []
    for r, seg in enumerate(rows):
        st = qa.VectorStorage(seg, qa.Distance.Cosine)
        b = sharded.HipBackend(st, nq, 0)
        b.local_topk(qd, top, gathered[r], gcounts[r])
        backends.append((st, b))
    base = torch.tensor(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32), device=dev)
    merged = torch.zeros((nq, top, 2), dtype=torch.int32, device=dev)
    mcounts = torch.zeros((nq,), dtype=torch.int32, device=dev)
    backends[0][1].merge(gathered, gcounts, base, top, merged, mcounts)
    torch.cuda.synchronize()
    m, c = merged.cpu().numpy(), mcounts.cpu().numpy()
    want = O.DenseStorage(O.F32, O.COSINE, np.concatenate(rows)).peek_top(queries, top)
    for i in range(nq):
        assert m[i, :c[i], 0].view(np.uint32).tolist() == want[i]["idx"].tolist()
        assert m[i, :c[i], 1].copy().view(np.float32).tolist() == want[i]["score"].tolist()
    for _, b in backends:
        b.close()


def test_sharded_searcher_world_size_one():
    import torch
    import qdrant_amd as qa
    from qdrant_amd import sharded
    n, dim, nq, top = 5000, 96, 4, 10
    rows = O.preprocess(O.COSINE, O.synth(77, 0, n, dim))
    st = qa.VectorStorage(rows, qa.Distance.Cosine)
    dev = torch.device("cuda", 0)
    b = sharded.HipBackend(st, nq, 0)
    s = sharded.ShardedSearcher(b, n, nq, top, device=dev)
    truth = O.DenseStorage(O.F32, O.COSINE, rows)
    for batch in range(2):
        q = O.synth(78, batch * nq, nq, dim)
        s.search(torch.from_numpy(q).to(dev))
        for (gi, gs), w in zip(s.results(), truth.peek_top(q, top)):
            assert gi.tolist() == w["idx"].tolist() and gs.tolist() == w["score"].tolist()
    b.close()


def test_sharded_hnsw_backend_world_size_one_and_two_segments():
    """HNSW variant of the per-rank search: device-built graph per segment, SQ walk + f32 rescoring, same gather/merge."""
    import torch
    import qdrant_amd as qa
    from qdrant_amd import sharded
    rng = np.random.default_rng(4)
    dim, nq, top = 64, 8, 10
    centers = rng.standard_normal((32, dim)).astype(np.float32) * 2
    segs = [O.preprocess(O.COSINE, (centers[rng.integers(0, 32, n)] + 0.5 * rng.standard_normal((n, dim))).astype(np.float32)) for n in (4000, 2500)]
    queries = O.preprocess(O.COSINE, (centers[rng.integers(0, 32, nq)] + 0.5 * rng.standard_normal((nq, dim))).astype(np.float32))
    dev = torch.device("cuda", 0)
    qd = torch.from_numpy(queries).to(dev)
    gathered = torch.zeros((2, nq, top, 2), dtype=torch.int32, device=dev)
    gcounts = torch.zeros((2, nq), dtype=torch.int32, device=dev)
    keep = []
    for r, rows in enumerate(segs):
        vs = qa.VectorStorage(rows, qa.Distance.Cosine)
        graph = qa.GraphLayers.build(vs, m=8, ef_construct=64, seed=r)
        quant = qa.ScalarQuantizer.from_min_max(rows, dim, qa.Distance.Dot)
        enc = qa.EncodedVectorsU8(quant.encode(rows), quant)
        b = sharded.HipHnswBackend(enc, graph, nq, 0, ef=96, rescore_storage=vs, oversampling=3)
        b.local_topk(qd, top, gathered[r], gcounts[r])
        keep.append((vs, graph, enc, b))
    base = torch.tensor([0, len(segs[0])], dtype=torch.int32, device=dev)
    merged = torch.zeros((nq, top, 2), dtype=torch.int32, device=dev)
    mcounts = torch.zeros((nq,), dtype=torch.int32, device=dev)
    keep[0][3].merge(gathered, gcounts, base, top, merged, mcounts)
    torch.cuda.synchronize()
    m, c = merged.cpu().numpy(), mcounts.cpu().numpy()
    truth = O.DenseStorage(O.F32, O.COSINE, np.concatenate(segs))
    exact = truth.peek_top(queries, top)
    hits = 0
    for i in range(nq):
        assert c[i] == top
        ids = m[i, :, 0].view(np.uint32)
        sc = m[i, :, 1].copy().view(np.float32)
        hits += len(set(ids.tolist()) & set(exact[i]["idx"].tolist()))
        w = truth.score_points(queries[i:i + 1], ids)[0]            # rescored: exact f32 scores of the globalised ids
        assert np.array_equal(sc.view(np.uint32), w.view(np.uint32)) and np.all(np.diff(sc) <= 0)
    assert hits / (nq * top) > 0.8
    # a walk without rescoring, through ShardedSearcher at world size 1
    vs, graph, enc, _ = keep[0]
    b1 = sharded.HipHnswBackend(vs, graph, nq, 0, ef=96)
    s = sharded.ShardedSearcher(b1, len(segs[0]), nq, top, device=dev)
    s.search(qd)
    want = graph.search(top, 96, qa.new_raw_scorer(queries, vs))
    for (gi, gs), w in zip(s.results(), want):
        assert gi.tolist() == w["idx"].tolist() and gs.tolist() == w["score"].tolist()
    for _, _, _, b in keep:
        b.close()
    b1.close()

"""qmx_sharded_search_topk / qmx_sharded_hnsw_search: ONE host process, N segments (on one device here; one per device on a multi-GPU
node) = `SegmentsSearcher::search` (lib/collection/src/collection_manager/segments_searcher.rs:250-285) + `BatchResultAggregator`
(lib/shard/src/search_result_aggregator.rs:50-121) behind the C-ABI.  Parity: the merged lists equal the oracle's search of the
concatenated rows, the torch.distributed harness's result (qdrant_amd/sharded.py, world size 1) and - for the row-split form - the
single-segment search of the same block, bit for bit."""
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


def _same(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g["idx"].tolist() == w["idx"].tolist()
        assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32))


@pytest.mark.parametrize("sizes,nq,top", [([3000, 1700, 2300], 6, 10), ([5000], 3, 7), ([40, 9000, 1, 700, 350], 20, 64), ([2000, 2000], 1, 1)])
def test_segments_on_one_device_equal_the_concatenated_block(qa, sizes, nq, top):
    from qdrant_amd import sharded
    dim = 64
    rows = [O.preprocess(O.COSINE, O.synth(0x5EED0705 + 16 * r, 0, n, dim)) for r, n in enumerate(sizes)]
    queries = O.synth(0x5EED0706, 0, nq, dim)
    sts = [qa.VectorStorage(r, qa.Distance.Cosine) for r in rows]
    s = sharded.SegmentsSearcher(sts, nq)
    got = s.search(queries, top)
    want = O.DenseStorage(O.F32, O.COSINE, np.concatenate(rows)).peek_top(queries, top)
    _same(got, want)
    assert s.counters.vectors_scored == nq * sum(sizes)
    # a second batch through the same handles
    q2 = O.synth(0x5EED0707, 0, nq, dim)
    _same(s.search(q2, top), O.DenseStorage(O.F32, O.COSINE, np.concatenate(rows)).peek_top(q2, top))
    s.close()


def test_sharded_call_equals_the_torch_distributed_harness(qa):
    """The same three segments through qdrant_amd/sharded.py's per-rank path (HipBackend + merge), world size 1 per segment."""
    import torch
    from qdrant_amd import sharded
    dim, nq, top = 128, 8, 10
    sizes = [3000, 1700, 2300]
    rows = [O.preprocess(O.COSINE, O.synth(0x5EED0715 + 16 * r, 0, n, dim)) for r, n in enumerate(sizes)]
    queries = O.synth(0x5EED0716, 0, nq, dim)
    dev = torch.device("cuda", 0)
    qd = torch.from_numpy(queries).to(dev)
    sts = [qa.VectorStorage(r, qa.Distance.Cosine) for r in rows]
    gathered = torch.zeros((len(sizes), nq, top, 2), dtype=torch.int32, device=dev)
    gcounts = torch.zeros((len(sizes), nq), dtype=torch.int32, device=dev)
    backends = [sharded.HipBackend(st, nq, 0) for st in sts]
    for r, b in enumerate(backends):
        b.local_topk(qd, top, gathered[r], gcounts[r])
    base = torch.tensor(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32), device=dev)
    merged = torch.zeros((nq, top, 2), dtype=torch.int32, device=dev)
    mcounts = torch.zeros((nq,), dtype=torch.int32, device=dev)
    backends[0].merge(gathered, gcounts, base, top, merged, mcounts)
    torch.cuda.synchronize()
    m, c = merged.cpu().numpy(), mcounts.cpu().numpy()
    s = sharded.SegmentsSearcher(sts, nq)
    # device queries + device outputs, enqueue only
    out = torch.zeros((nq, top, 2), dtype=torch.int32, device=dev)
    cnt = torch.zeros((nq,), dtype=torch.int32, device=dev)
    s.search_async(qd, top, out, cnt)
    s.synchronize()
    assert torch.equal(out, merged) and torch.equal(cnt, mcounts)
    got = s.search(queries, top)
    for i in range(nq):
        assert got[i]["idx"].tolist() == m[i, :c[i], 0].view(np.uint32).tolist()
        assert np.array_equal(got[i]["score"].view(np.uint32), m[i, :c[i], 1].copy().view(np.uint32))
    for b in backends:
        b.close()
    s.close()


@pytest.mark.parametrize("parts", [2, 3, 8])
def test_row_split_of_one_block_equals_the_single_segment_search(qa, parts):
    """Strong scaling: one block split by contiguous row range, id_bases = first row of each range."""
    import torch
    from qdrant_amd import sharded, _ffi as F
    n, dim, nq, top = 50_000, 128, 33, 10
    rows = O.preprocess(O.COSINE, O.synth(0x5EED0720, 0, n, dim))
    queries = O.synth(0x5EED0721, 0, nq, dim)
    dev = torch.device("cuda", 0)
    block = torch.from_numpy(rows).to(dev)
    whole = qa.VectorStorage(block, qa.Distance.Cosine)
    want = qa.BatchFilteredSearcher(queries, whole, top).peek_top_all()
    _same(want, O.DenseStorage(O.F32, O.COSINE, rows).peek_top(queries, top, threads=8))
    bounds = [n * r // parts for r in range(parts + 1)]
    sts = [qa.VectorStorage(block[bounds[r]:bounds[r + 1]], qa.Distance.Cosine) for r in range(parts)]      # device rows adopted in place
    s = sharded.SegmentsSearcher(sts, nq, id_bases=bounds[:-1])
    _same(s.search(queries, top), want)
    s.close()


def test_segments_with_deleted_rows_and_the_prefilter_track(qa):
    """Two 300k-row segments with a derived f16 copy, 128 queries: each local stage is the prefilter + exact verification; counters add up."""
    from qdrant_amd import sharded
    n, dim, nq, top = 300_000, 128, 128, 10
    rng = np.random.default_rng(5)
    rows = [O.preprocess(O.COSINE, O.synth(0x5EED0730 + 16 * r, 0, n, dim)) for r in range(2)]
    queries = O.synth(0x5EED0731, 0, nq, dim)
    deleted = [rng.random(n) < 0.2 for _ in range(2)]
    sts = []
    for r in range(2):
        st = qa.VectorStorage(rows[r], qa.Distance.Cosine, flags=qa._ffi.SEG_HALF_COPY)
        st.set_deleted(deleted[r], None)
        sts.append(st)
    s = sharded.SegmentsSearcher(sts, nq)
    got = s.search(queries, top)
    want = O.DenseStorage(O.F32, O.COSINE, np.concatenate(rows), point_deleted=np.concatenate(deleted)).peek_top(queries, top, threads=8)
    _same(got, want)
    assert s.counters.prefilter_queries == 2 * nq and s.counters.fallback_queries == 0
    assert s.counters.verified_rows >= 2 * nq * top
    s.close()


def test_sharded_hnsw_search_merges_the_per_segment_walks(qa):
    from qdrant_amd import sharded
    dim, nq, top, ef = 64, 16, 10, 64
    sizes = [4000, 2500]
    rows = [O.preprocess(O.COSINE, O.synth(0x5EED0740 + 16 * r, 0, n, dim)) for r, n in enumerate(sizes)]
    queries = O.synth(0x5EED0741, 0, nq, dim)
    sts = [qa.VectorStorage(r, qa.Distance.Cosine) for r in rows]
    graphs = [qa.GraphLayers.build(st, m=8, ef_construct=48, seed=3 + i) for i, st in enumerate(sts)]
    # reference: each segment's own walk, merged by the oracle's aggregator
    lists = np.zeros((len(sizes), nq, top), dtype=O.ScoredPointOffset)
    counts = np.zeros((len(sizes), nq), dtype=np.uint32)
    base = [0, sizes[0]]
    for r, (st, g) in enumerate(zip(sts, graphs)):
        res = g.search(top, ef, qa.new_raw_scorer(queries, st))
        for i, x in enumerate(res):
            lists[r, i, :len(x)]["idx"] = x["idx"] + base[r]
            lists[r, i, :len(x)]["score"] = x["score"]
            counts[r, i] = len(x)
    want = O.merge_topk(lists, counts, top)
    s = sharded.SegmentsSearcher(sts, nq, graphs=graphs)
    _same(s.search(queries, top, ef=ef), want)
    s.close()


def test_segments_on_two_devices(qa):
    """One segment per device, gathered by hipMemcpyPeerAsync: needs a multi-GPU node (skipped on the 1-GPU boxes)."""
    if qa.device_count() < 2:
        pytest.skip("one device")
    from qdrant_amd import sharded
    dim, nq, top = 128, 40, 10
    sizes = [30_000, 20_000]
    rows = [O.preprocess(O.COSINE, O.synth(0x5EED0750 + 16 * r, 0, n, dim)) for r, n in enumerate(sizes)]
    queries = O.synth(0x5EED0751, 0, nq, dim)
    sts = [qa.VectorStorage(r, qa.Distance.Cosine, device_id=i) for i, r in enumerate(rows)]
    s = sharded.SegmentsSearcher(sts, nq)
    _same(s.search(queries, top), O.DenseStorage(O.F32, O.COSINE, np.concatenate(rows)).peek_top(queries, top, threads=8))
    s.close()


def test_bad_arguments_are_refused(qa):
    from qdrant_amd import sharded, _ffi as F
    rows = O.preprocess(O.COSINE, O.synth(1, 0, 500, 32))
    st = qa.VectorStorage(rows, qa.Distance.Cosine)
    a, b = qa.new_raw_scorer(O.synth(2, 0, 4, 32), st), qa.new_raw_scorer(O.synth(2, 0, 5, 32), st)
    out = np.zeros((4, 3), dtype=O.ScoredPointOffset)
    cnt = np.zeros(4, dtype=np.uint32)
    hs = (C.c_void_p * 2)(a._h.value, b._h.value)
    assert F.lib().qmx_sharded_search_topk(hs, 2, 3, None, F.ptr(out), F.ptr(cnt), None, None) == F.ERR_BAD_ARG      # different batch sizes
    hs = (C.c_void_p * 2)(a._h.value, a._h.value)
    assert F.lib().qmx_sharded_search_topk(hs, 2, 3, None, F.ptr(out), F.ptr(cnt), None, None) == F.ERR_BAD_ARG      # one batch twice
    assert F.lib().qmx_sharded_search_topk(hs, 0, 3, None, F.ptr(out), F.ptr(cnt), None, None) == F.ERR_BAD_ARG

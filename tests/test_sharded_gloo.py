"""N>1 path on CPU: world_size-2 (and 3) gloo process groups drive qdrant_amd.sharded with the oracle
injected as the compute backend (the checker stands in for the two HIP calls; everything else — id
bases, the all-gather layout, buffer reuse across batches, the merge contract — is the product code).
The merged result must equal the oracle's exact search over the concatenation of all segments."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleBackend:
    """local_topk / merge with the CPU oracle (test only)."""

    def __init__(self, O, rows, distance):
        self.O = O
        self.st = O.DenseStorage(O.F32, distance, rows)
        self.device = torch.device("cpu")

    def local_topk(self, queries, top, out, counts):
        res = self.st.peek_top(queries.numpy(), top)
        o = out.numpy()
        o[:] = 0
        for i, r in enumerate(res):
            o[i, :len(r), 0] = r["idx"].view(np.int32)
            o[i, :len(r), 1] = r["score"].view(np.int32)
            counts[i] = len(r)

    def merge(self, gathered, gcounts, idx_base, top, merged, mcounts):
        O = self.O
        g = gathered.numpy()
        lists = np.zeros(g.shape[:3], dtype=O.ScoredPointOffset)
        lists["idx"] = g[..., 0].view(np.uint32)
        lists["score"] = g[..., 1].copy().view(np.float32)
        res = O.merge_topk(lists, gcounts.numpy().astype(np.uint32), top, idx_base.numpy().view(np.uint32))
        m = merged.numpy()
        m[:] = 0
        for i, r in enumerate(res):
            m[i, :len(r), 0] = r["idx"].view(np.int32)
            m[i, :len(r), 1] = r["score"].view(np.int32)
            mcounts[i] = len(r)


def _worker(rank, world, port, sizes, dim, nq, top, distance):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_ffi as O
        from qdrant_amd import sharded
        seed = 0x5EED0005
        rows = O.preprocess(distance, O.synth(seed + 16 * rank, 0, sizes[rank], dim))
        backend = OracleBackend(O, rows, distance)
        s = sharded.ShardedSearcher(backend, sizes[rank], nq, top)
        base = s.idx_base.numpy().view(np.uint32)
        assert base.tolist() == np.concatenate([[0], np.cumsum(sizes)[:-1]]).tolist()
        # single-process truth over the union of all segments
        all_rows = np.concatenate([O.preprocess(distance, O.synth(seed + 16 * r, 0, sizes[r], dim)) for r in range(world)])
        truth = O.DenseStorage(O.F32, distance, all_rows)
        for batch in range(3):     # buffers are reused across batches
            queries = O.synth(seed + 1, batch * nq, nq, dim)
            s.search(torch.from_numpy(queries))
            got = s.results()
            want = truth.peek_top(queries, top)
            for (gi, gs), w in zip(got, want):
                assert gi.tolist() == w["idx"].tolist(), (rank, batch)
                assert gs.tolist() == w["score"].tolist()
        # the exchange step was ONE collective per search: lists and counts travel in one packed record (qmx_topk_record_bytes' layout)
        assert s.collectives == 3 and sharded.COLLECTIVE_CALLS == 3
        words = sharded.record_words(nq, top)
        assert words % 2 == 0 and words >= nq * top * 2 + nq and s.record.shape == (words,) and s.records.shape == (world, words)
        assert s.out.data_ptr() == s.record.data_ptr() and s.counts.data_ptr() == s.record.data_ptr() + nq * top * 8
        assert torch.equal(s.records[rank], s.record) and torch.equal(s.gathered[rank], s.out) and torch.equal(s.gcounts[rank], s.counts)
        # every rank holds the same merged lists
        mine = s.merged.clone()
        allm = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allm, mine)
        for m in allm:
            assert torch.equal(m, mine)
    finally:
        dist.destroy_process_group()


def _worker_strong(rank, world, port, n_total, dim, nq, top, distance):
    """--scaling strong of bench.py: ONE segment row-split over the ranks (sharded.row_split); the merged result must be the
    single-process search of the whole segment, ids included (base = first row of the slice)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_ffi as O
        from qdrant_amd import sharded
        seed = 0x5EED0006
        row0, n_local = sharded.row_split(n_total)
        assert (row0, n_local) == (n_total * rank // world, n_total * (rank + 1) // world - n_total * rank // world)
        whole = O.preprocess(distance, O.synth(seed, 0, n_total, dim))
        mine = O.preprocess(distance, O.synth(seed, row0, n_local, dim))      # the generator is counter-based: a slice is the slice
        assert np.array_equal(mine.view(np.uint32), whole[row0:row0 + n_local].view(np.uint32))
        s = sharded.ShardedSearcher(OracleBackend(O, mine, distance), n_local, nq, top)
        assert int(s.idx_base.numpy().view(np.uint32)[rank]) == row0
        truth = O.DenseStorage(O.F32, distance, whole)
        for batch in range(2):
            queries = O.synth(seed + 1, batch * nq, nq, dim)
            s.search(torch.from_numpy(queries))
            for (gi, gs), w in zip(s.results(), truth.peek_top(queries, top)):
                assert gi.tolist() == w["idx"].tolist() and gs.tolist() == w["score"].tolist()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 1001), (3, 700)])
def test_row_split_of_one_segment_matches_the_single_segment_search(world, n_total):
    import oracle_ffi  # noqa: F401
    mp.spawn(_worker_strong, args=(world, _free_port(), n_total, 40, 4, 10, 0), nprocs=world, join=True)


@pytest.mark.parametrize("world,sizes", [(2, [700, 500]), (3, [300, 5, 450])])
def test_sharded_search_matches_single_process(world, sizes):
    import oracle_ffi  # noqa: F401  (builds the oracle before the workers start)
    mp.spawn(_worker, args=(world, _free_port(), sizes, 48, 5, 10, 0), nprocs=world, join=True)


def test_record_words_is_the_header_formula():
    sys.path.insert(0, ROOT)
    from qdrant_amd import sharded, _ffi as F
    lib = F.lib()
    for nq, top in [(1, 1), (5, 10), (128, 10), (3, 7), (4096, 64)]:
        assert sharded.record_words(nq, top) * 4 == lib.qmx_topk_record_bytes(nq, top) == (nq * top * 8 + nq * 4 + 7) // 8 * 8


def test_gather_topk_single_process_is_a_copy():
    sys.path.insert(0, ROOT)
    from qdrant_amd import sharded
    out = torch.arange(2 * 3 * 2, dtype=torch.int32).reshape(2, 3, 2)
    cnt = torch.tensor([3, 1], dtype=torch.int32)
    g, gc = sharded.gather_topk(out, cnt)
    assert g.shape == (1, 2, 3, 2) and torch.equal(g[0], out) and torch.equal(gc[0], cnt)
    assert sharded.segment_id_bases(123).tolist() == [0]

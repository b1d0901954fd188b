"""Segment-sharded search: one process per GPU, one (or more) segments per process.

The reference searches every segment of a shard independently and merges the per-segment top-k lists
(`SegmentsSearcher::search`, lib/collection/src/collection_manager/segments_searcher.rs:212-285 ->
`BatchResultAggregator`, lib/shard/src/search_result_aggregator.rs:50-121).  Here a segment lives on one
GPU, every rank scores the same query batch against its own segment and the only exchange step is an
all-gather of ONE packed record per rank (`Q x top x 8` bytes of lists + `Q x 4` of counts; RCCL over xGMI
under `torch.distributed`, backend "nccl"), followed by the k-way merge with segment-local offsets globalised by a per-segment id base.

Host logic only.  The two compute steps are delegated to a backend object:
  * `HipBackend`   — the product: qmx_search_topk_async + qmx_merge_topk_packed_async of libqdrant_amd.so
                     (fails loudly without a gfx950 device; there is no CPU fallback here);
  * tests inject an oracle-based backend to exercise the collective / id-globalisation logic on CPU
    with the gloo backend (tests/test_sharded_gloo.py).
"""
import ctypes as C
from typing import Optional

import torch
import torch.distributed as dist

from . import _ffi as F


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def segment_id_bases(n_local: int, group=None, device=None) -> torch.Tensor:
    """Exclusive prefix sum of the segment sizes over ranks: segment-local offset + base[rank] is the
    collection-wide id (the reference maps (segment, offset) to an external point id through the
    id tracker, lib/segment/src/id_tracker; disjoint ranges are the synthetic stand-in for it)."""
    rank, world = _world(group)
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    if world > 1:
        mine = torch.tensor([n_local], dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(sizes, mine, group=group)
    else:
        sizes[0] = n_local
    base = torch.cumsum(sizes, 0) - sizes
    if int(base[-1] + sizes[-1]) > 0xFFFFFFFF:
        raise ValueError("globalised ids exceed PointOffsetType (u32)")
    return base.to(torch.int32)  # bit pattern of u32


def row_split(n_total: int, group=None):
    """Strong scaling of ONE big segment ("if a collection is one big segment, split by contiguous row range"): rank r
    holds rows [row0, row0 + n_local) of the segment; with `segment_id_bases(n_local)` as the id base the merged result is the
    single-segment result (segments_searcher.rs:250-285 merges per-segment lists the same way)."""
    rank, world = _world(group)
    row0 = n_total * rank // world
    return row0, n_total * (rank + 1) // world - row0


def gather_topk(local_out: torch.Tensor, local_counts: torch.Tensor, gathered: Optional[torch.Tensor] = None,
                gcounts: Optional[torch.Tensor] = None, group=None):
    """All-gather of the per-rank result lists.
    local_out [Q, top, 2] int32 (ScoredPointOffset rows: idx bits, f32 score bits), local_counts [Q] int32
    -> gathered [world, Q, top, 2], gcounts [world, Q]  (list l = rank l's segment)."""
    rank, world = _world(group)
    if gathered is None:
        gathered = torch.empty((world,) + tuple(local_out.shape), dtype=local_out.dtype, device=local_out.device)
    if gcounts is None:
        gcounts = torch.empty((world,) + tuple(local_counts.shape), dtype=local_counts.dtype, device=local_counts.device)
    if world == 1:
        gathered[0].copy_(local_out)
        gcounts[0].copy_(local_counts)
    else:
        # output viewed as the concatenation along dim 0 (the layout both RCCL and gloo accept)
        global COLLECTIVE_CALLS
        dist.all_gather_into_tensor(gathered.view((-1,) + tuple(local_out.shape[1:])), local_out.contiguous(), group=group)
        dist.all_gather_into_tensor(gcounts.view(-1), local_counts.contiguous(), group=group)
        COLLECTIVE_CALLS += 2
    return gathered, gcounts


class HipBackend:
    """Local brute-force top-k and the merge, both on this rank's GPU through the C-ABI."""

    def __init__(self, storage, nq: int, device_id: int, stream: Optional[torch.cuda.Stream] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("HipBackend needs a gfx950 device; qdrant_amd has no CPU fallback")
        self.lib = F.lib()
        self.storage = storage
        self.nq = nq
        self.device_id = device_id
        self.device = torch.device("cuda", device_id)
        # A NULL hipStream_t means "the query's own stream" in the C-ABI, so torch's legacy default stream
        # (handle 0) cannot be handed over: use a dedicated stream then and order it against the caller's
        # current stream around every call (_enter / _leave).
        self.stream = stream or torch.cuda.current_stream(self.device)
        if self.stream.cuda_stream == 0:
            self.stream = torch.cuda.Stream(self.device)
        self.qh = C.c_void_p()
        zeros = torch.zeros((nq, storage.dim), dtype=torch.float32, device=self.device)
        F.check(self.lib.qmx_query_create(storage._h, F.ptr(zeros), nq, C.byref(self.qh)))
        F.check(self.lib.qmx_query_set_stream(self.qh, C.c_void_p(self.stream.cuda_stream)))

    def local_topk(self, queries: torch.Tensor, top: int, out: torch.Tensor, counts: torch.Tensor):
        """queries [nq, dim] f32 on this device (original, un-preprocessed); enqueues only."""
        assert queries.is_cuda and queries.shape[0] == self.nq
        cur = self._enter(queries, out, counts)
        F.check(self.lib.qmx_query_update(self.qh, F.ptr(queries)))
        F.check(self.lib.qmx_search_topk_async(self.qh, top, None, 0, F.ptr(out), F.ptr(counts)))
        self._leave(cur)

    def merge(self, gathered, gcounts, idx_base, top: int, merged, mcounts):
        n_lists, nq = gathered.shape[0], gathered.shape[1]
        cur = self._enter(gathered, gcounts, idx_base, merged, mcounts)
        F.check(self.lib.qmx_merge_topk_async(self.device_id, C.c_void_p(self.stream.cuda_stream), F.ptr(gathered),
                                              F.ptr(gcounts), F.ptr(idx_base), n_lists, nq, top, F.ptr(merged),
                                              F.ptr(mcounts)))
        self._leave(cur)

    def merge_packed(self, records, idx_base, nq: int, top: int, merged, mcounts):
        """records [n_lists, record_words(nq, top)] int32: the all-gathered packed records (lists + counts of every rank)."""
        assert records.is_contiguous() and records.shape[1] == record_words(nq, top)
        cur = self._enter(records, idx_base, merged, mcounts)
        F.check(self.lib.qmx_merge_topk_packed_async(self.device_id, C.c_void_p(self.stream.cuda_stream), F.ptr(records), F.ptr(idx_base),
                                                     records.shape[0], nq, top, F.ptr(merged), F.ptr(mcounts)))
        self._leave(cur)

    def _enter(self, *tensors):
        """Work enqueued on self.stream must see what the caller's current stream produced."""
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream != self.stream.cuda_stream:
            self.stream.wait_stream(cur)
            for t in tensors:
                t.record_stream(self.stream)   # the caching allocator must not recycle them under our kernels
        return cur

    def _leave(self, cur):
        if cur.cuda_stream != self.stream.cuda_stream:
            cur.wait_stream(self.stream)

    def close(self):
        if self.qh:
            self.lib.qmx_query_destroy(self.qh)
            self.qh = C.c_void_p()


class HipHnswBackend(HipBackend):
    """Local HNSW search instead of the brute-force scan: the graph of this rank's segment (built on this GPU with
    `GraphLayers.build`, segments are independent: no communication at build time), walked with the scorer of
    `storage` (dense, SQ or PQ); when `rescore_storage` is given the walk returns `oversampling * top` candidates and
    they are re-scored with the original vectors (`postprocess_search_result`).  Same output contract as HipBackend,
    so ShardedSearcher's all-gather + merge is unchanged."""

    def __init__(self, storage, graph, nq: int, device_id: int, ef: int = 128, rescore_storage=None, oversampling: int = 2,
                 stream: Optional[torch.cuda.Stream] = None):
        super().__init__(storage, nq, device_id, stream)
        self.graph, self.ef = graph, ef
        self.rescore_storage, self.oversampling = rescore_storage, (oversampling if rescore_storage is not None else 1)
        self.rqh = C.c_void_p()
        if rescore_storage is not None:
            zeros = torch.zeros((nq, rescore_storage.dim), dtype=torch.float32, device=self.device)
            F.check(self.lib.qmx_query_create(rescore_storage._h, F.ptr(zeros), nq, C.byref(self.rqh)))
            F.check(self.lib.qmx_query_set_stream(self.rqh, C.c_void_p(self.stream.cuda_stream)))
        self._cand = None

    def local_topk(self, queries: torch.Tensor, top: int, out: torch.Tensor, counts: torch.Tensor):
        assert queries.is_cuda and queries.shape[0] == self.nq
        stop = top * self.oversampling
        if self.rescore_storage is not None and (self._cand is None or self._cand[0].shape[1] != stop):
            self._cand = (torch.zeros((self.nq, stop, 2), dtype=torch.int32, device=self.device),
                          torch.zeros((self.nq,), dtype=torch.int32, device=self.device),
                          torch.zeros((self.nq, stop), dtype=torch.int32, device=self.device))
        cur = self._enter(queries, out, counts)
        F.check(self.lib.qmx_query_update(self.qh, F.ptr(queries)))
        if self.rescore_storage is None:
            F.check(self.lib.qmx_hnsw_search_async(self.graph._h, self.qh, top, self.ef, F.ptr(out), F.ptr(counts), None))
        else:
            cand, ccnt, cids = self._cand
            F.check(self.lib.qmx_hnsw_search_async(self.graph._h, self.qh, stop, self.ef, F.ptr(cand), F.ptr(ccnt), None))
            with torch.cuda.stream(self.stream):
                cids.copy_(cand[:, :, 0])                      # ScoredPointOffset.idx column
            F.check(self.lib.qmx_query_update(self.rqh, F.ptr(queries)))
            F.check(self.lib.qmx_rescore(self.rqh, F.ptr(cids), F.ptr(ccnt), stop, top, F.ptr(out), F.ptr(counts)))
        self._leave(cur)

    def close(self):
        if self.rqh:
            self.lib.qmx_query_destroy(self.rqh)
            self.rqh = C.c_void_p()
        super().close()


COLLECTIVE_CALLS = 0      # data-path collectives this module issued in this process (bench.py reports the number per step)


def record_words(nq: int, top: int) -> int:
    """32-bit words of one packed record = qmx_topk_record_bytes(nq, top) / 4: [nq][top] ScoredPointOffset, [nq] counts, padded to 8 bytes
    (include/qdrant_amd.h, qmx_merge_topk_packed_async)."""
    return (nq * top * 2 + nq + 1) // 2 * 2


class ShardedSearcher:
    """search(queries) on every rank returns the merged top-k over all ranks' segments.

    The exchange step is ONE collective per batch: a rank's answer - its `Q x top` lists AND their counts - is one packed record
    (`record_words`), written in place by the local search (`out` / `counts` are views of it), all-gathered as one buffer and merged
    from the gathered records (`qmx_merge_topk_packed_async`).  The reference's aggregator takes a batch's per-segment lists in one
    pass as well (`BatchResultAggregator::update_batch_results`, lib/shard/src/search_result_aggregator.rs:91-106).

    All buffers are allocated once; `search` only enqueues work on the backend's stream (scan -> all-gather -> merge are ordered on
    that stream), so consecutive batches pipeline, and several searchers on several streams keep several batches in flight - every
    rank must then call them in the same order (a search holds a collective).  `timing = True` brackets the three stages with stream
    events (`stage_us()`), for measurement runs outside a timed region."""

    def __init__(self, backend, n_local: int, nq: int, top: int, device=None, group=None):
        self.backend, self.nq, self.top, self.group = backend, nq, top, group
        self.rank, self.world = _world(group)
        self.device = device if device is not None else getattr(backend, "device", torch.device("cpu"))
        dev = self.device
        self.idx_base = segment_id_bases(n_local, group, dev)
        words, lw = record_words(nq, top), nq * top * 2
        self.record = torch.zeros((words,), dtype=torch.int32, device=dev)
        self.out = self.record[:lw].view(nq, top, 2)                 # [Q, top, 2]: ScoredPointOffset rows (idx bits, f32 score bits)
        self.counts = self.record[lw:lw + nq]                        # [Q]
        self.records = torch.zeros((self.world, words), dtype=torch.int32, device=dev)      # the all-gather's output: list l = rank l's segment
        self.gathered = self.records[:, :lw].view(self.world, nq, top, 2)                    # views for backends without a packed merge
        self.gcounts = self.records[:, lw:lw + nq]
        self.merged = torch.zeros((nq, top, 2), dtype=torch.int32, device=dev)
        self.mcounts = torch.zeros((nq,), dtype=torch.int32, device=dev)
        self.collectives = 0
        self.timing = False
        self._events = []

    def _mark(self):
        if self.timing and self.device.type == "cuda":
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(self.device))
            return e
        return None

    def search(self, queries):
        global COLLECTIVE_CALLS
        t0 = self._mark()
        self.backend.local_topk(queries, self.top, self.out, self.counts)
        t1 = self._mark()
        if self.world == 1:
            self.records[0].copy_(self.record)
        else:
            dist.all_gather_into_tensor(self.records.view(-1), self.record, group=self.group)
            self.collectives += 1
            COLLECTIVE_CALLS += 1
        t2 = self._mark()
        if hasattr(self.backend, "merge_packed"):
            self.backend.merge_packed(self.records, self.idx_base, self.nq, self.top, self.merged, self.mcounts)
        else:
            self.backend.merge(self.gathered, self.gcounts, self.idx_base, self.top, self.merged, self.mcounts)
        t3 = self._mark()
        if t0 is not None:
            self._events.append((t0, t1, t2, t3))
        return self.merged, self.mcounts

    def stage_us(self):
        """Mean device time of the three stages over the searches issued with `timing` on (synchronise first): microseconds between the
        stream events around local search / all-gather (incl. the hand-over to and from the collective's stream) / merge."""
        if not self._events:
            return None
        n = float(len(self._events))
        out = {"searches": len(self._events),
               "local_search_us": round(sum(a.elapsed_time(b) for a, b, _, _ in self._events) * 1e3 / n, 2),
               "allgather_us": round(sum(b.elapsed_time(c) for _, b, c, _ in self._events) * 1e3 / n, 2),
               "merge_us": round(sum(c.elapsed_time(d) for _, _, c, d in self._events) * 1e3 / n, 2)}
        self._events = []
        return out

    def results(self):
        """Host copy of the last search: list of (idx u32 [c], score f32 [c]) per query (synchronises)."""
        m = self.merged.cpu().numpy()
        c = self.mcounts.cpu().numpy()
        import numpy as np
        return [(m[i, :c[i], 0].view(np.uint32).copy(), m[i, :c[i], 1].copy().view(np.float32)) for i in range(self.nq)]


class SegmentsSearcher:
    """ONE host process that owns every segment (the reference's shape: `SegmentsSearcher::search`, segments_searcher.rs:250-285):
    `qmx_sharded_search_topk` enqueues the local stage of every segment on its own device, gathers the lists by peer copies and
    merges them on the first segment's device.  `storages[i]` may live on any device; `id_bases[i]` globalises segment-local offsets
    (default: the exclusive prefix sum of the segment sizes).  With `graphs` the local stage is each segment's HNSW walk."""

    def __init__(self, storages, nq: int, id_bases=None, graphs=None):
        import numpy as np
        self.lib = F.lib()
        self.storages, self.nq, self.graphs = list(storages), int(nq), (list(graphs) if graphs is not None else None)
        n = len(self.storages)
        if id_bases is None:
            sizes = [int(s.total_vector_count()) for s in self.storages]
            id_bases = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        self.id_bases = np.ascontiguousarray(id_bases, dtype=np.uint32)
        assert len(self.id_bases) == n
        self._handles = (C.c_void_p * n)()
        zeros = np.zeros((nq, self.storages[0].dim), dtype=np.float32)
        for i, st in enumerate(self.storages):
            h = C.c_void_p()
            F.check(self.lib.qmx_query_create(st._h, F.ptr(zeros), nq, C.byref(h)))
            self._handles[i] = h.value
        self._graph_handles = None
        if self.graphs is not None:
            self._graph_handles = (C.c_void_p * n)(*[g._h.value if hasattr(g._h, "value") else g._h for g in self.graphs])
        self.counters = F.Counters()

    def search(self, queries, top: int, ef: int = 0):
        """queries [nq, dim] f32 (numpy, or a torch tensor every segment's device can read) -> list of ScoredPointOffset arrays."""
        import numpy as np
        n = len(self.storages)
        q = queries if hasattr(queries, "data_ptr") else np.ascontiguousarray(queries, dtype=np.float32)
        assert tuple(q.shape) == (self.nq, self.storages[0].dim)
        F.check(self.lib.qmx_sharded_query_update(self._handles, n, F.ptr(q)))
        out = np.zeros((self.nq, top), dtype=np.dtype([("idx", np.uint32), ("score", np.float32)]))
        counts = np.zeros(self.nq, dtype=np.uint32)
        if self._graph_handles is not None:
            F.check(self.lib.qmx_sharded_hnsw_search(self._graph_handles, self._handles, n, top, ef, F.ptr(self.id_bases), F.ptr(out), F.ptr(counts),
                                                      None, C.byref(self.counters)))
        else:
            F.check(self.lib.qmx_sharded_search_topk(self._handles, n, top, F.ptr(self.id_bases), F.ptr(out), F.ptr(counts), None,
                                                      C.byref(self.counters)))
        return [out[i, :counts[i]].copy() for i in range(self.nq)]

    def search_async(self, queries, top: int, out, counts):
        """Enqueue only: `queries`, `out` [nq, top, 2] int32 and `counts` [nq] int32 are torch tensors on the first segment's device."""
        F.check(self.lib.qmx_sharded_query_update(self._handles, len(self.storages), F.ptr(queries)))
        F.check(self.lib.qmx_sharded_search_topk_async(self._handles, len(self.storages), top, F.ptr(self.id_bases), F.ptr(out), F.ptr(counts)))

    def synchronize(self):
        F.check(self.lib.qmx_query_synchronize(C.c_void_p(self._handles[0])))

    def close(self):
        for i in range(len(self.storages)):
            if self._handles[i]:
                self.lib.qmx_query_destroy(C.c_void_p(self._handles[i]))
                self._handles[i] = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

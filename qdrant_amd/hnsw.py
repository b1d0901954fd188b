"""Host-side mirror of `GraphLayers` (lib/segment/src/index/hnsw_index/graph_layers.rs:58-72) for
the device-resident search: the graph is given as the plain `GraphLinks` arrays
(graph_links/serializer.rs:52-176) + `EntryPoints`, uploaded once, and `search` runs
`GraphLayers::search` (:530-562) for a whole batch of queries in one kernel launch
(`qmx_hnsw_search`).  No CPU fallback: without the HIP library / a gfx950 device this raises."""
import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _ffi as F
from .scorer import RawScorer, ScoredPointOffset


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class GraphLayers:
    """`GraphLayers {hnsw_m, links, entry_points}` on one GPU."""

    def __init__(self, m: int, m0: int, reindex, level_offsets, offsets, neighbors, entry_point_ids, entry_point_levels,
                 extra_entry_point_ids=(), extra_entry_point_levels=(), device_id: int = 0):
        self.m, self.m0 = int(m), int(m0)
        self._keep = [_u32(reindex), _u64(level_offsets), _u64(offsets), _u32(neighbors), _u32(entry_point_ids),
                      _u32(entry_point_levels), _u32(extra_entry_point_ids), _u32(extra_entry_point_levels)]
        re, lo, off, nb, ep, epl, xp, xpl = self._keep
        d = F.HnswDesc()
        d.m, d.m0 = self.m, self.m0
        d.n_points = len(re)
        d.n_levels = max(len(lo) - 1, 0)
        d.reindex, d.level_offsets, d.offsets, d.neighbors = (F.ptr(re).value, F.ptr(lo).value, F.ptr(off).value,
                                                              F.ptr(nb).value if len(nb) else None)
        d.n_offsets, d.n_neighbors = len(off), len(nb)
        d.entry_point_ids, d.entry_point_levels, d.n_entry_points = F.ptr(ep).value, F.ptr(epl).value, len(ep)
        d.extra_entry_point_ids, d.extra_entry_point_levels, d.n_extra_entry_points = (F.ptr(xp).value, F.ptr(xpl).value,
                                                                                        len(xp))
        d.device_id = device_id
        self.n_points = d.n_points
        self._h = C.c_void_p()
        F.check(F.lib().qmx_hnsw_create(C.byref(d), C.byref(self._h)))
        self.counters = F.Counters()

    @classmethod
    def from_plain(cls, links, device_id: int = 0):
        """`links`: any object with m, m0, reindex, level_offsets, offsets, neighbors, ep_ids, ep_levels."""
        return cls(links.m, links.m0, links.reindex, links.level_offsets, links.offsets, links.neighbors, links.ep_ids,
                   links.ep_levels, getattr(links, "xp_ids", ()), getattr(links, "xp_levels", ()), device_id)

    @classmethod
    def from_plain_file(cls, data: bytes, m: int, m0: int, entry_point_ids, entry_point_levels, extra_entry_point_ids=(),
                        extra_entry_point_levels=(), device_id: int = 0):
        """The bytes of a plain graph-links file (graph_links/serializer.rs, GraphLinksFormatParam::Plain)."""
        self = cls.__new__(cls)
        self.m, self.m0 = int(m), int(m0)
        ep, epl, xp, xpl = _u32(entry_point_ids), _u32(entry_point_levels), _u32(extra_entry_point_ids), _u32(extra_entry_point_levels)
        buf = np.frombuffer(data, dtype=np.uint8)
        self._keep = [ep, epl, xp, xpl, buf]
        d = F.HnswDesc()
        d.m, d.m0 = self.m, self.m0
        d.entry_point_ids, d.entry_point_levels, d.n_entry_points = F.ptr(ep).value, F.ptr(epl).value, len(ep)
        d.extra_entry_point_ids, d.extra_entry_point_levels, d.n_extra_entry_points = F.ptr(xp).value, F.ptr(xpl).value, len(xp)
        d.device_id = device_id
        self._h = C.c_void_p()
        F.check(F.lib().qmx_hnsw_create_from_plain_file(F.ptr(buf), len(buf), C.byref(d), C.byref(self._h)))
        self.n_points = int(np.frombuffer(data[:8], dtype=np.uint64)[0]) if len(data) >= 8 else 0
        self.counters = F.Counters()
        return self

    @classmethod
    def from_file(cls, data: bytes, entry_point_ids, entry_point_levels, extra_entry_point_ids=(), extra_entry_point_levels=(),
                  m: int = 0, m0: int = 0, device_id: int = 0):
        """The bytes of a graph-links file in any `GraphLinksFormat` (Plain needs m / m0; the compressed headers carry
        them): `GraphLinksView::load` (graph_links/view.rs:110-208) done once on the host, then the upload."""
        self = cls.__new__(cls)
        ep, epl, xp, xpl = _u32(entry_point_ids), _u32(entry_point_levels), _u32(extra_entry_point_ids), _u32(extra_entry_point_levels)
        buf = np.frombuffer(data, dtype=np.uint8)
        self._keep = [ep, epl, xp, xpl, buf]
        d = F.HnswDesc()
        d.m, d.m0 = int(m), int(m0)
        d.entry_point_ids, d.entry_point_levels, d.n_entry_points = F.ptr(ep).value, F.ptr(epl).value, len(ep)
        d.extra_entry_point_ids, d.extra_entry_point_levels, d.n_extra_entry_points = F.ptr(xp).value, F.ptr(xpl).value, len(xp)
        d.device_id = device_id
        self._h = C.c_void_p()
        F.check(F.lib().qmx_hnsw_create_from_file(F.ptr(buf), len(buf), C.byref(d), C.byref(self._h)))
        info = F.HnswInfo()
        F.check(F.lib().qmx_hnsw_get_info(self._h, C.byref(info)))
        self.m, self.m0, self.n_points = info.m, info.m0, info.n_points
        self.counters = F.Counters()
        return self

    @classmethod
    def build(cls, storage, m: int = 16, m0: Optional[int] = None, ef_construct: int = 100, seed: int = 42,
              entry_points_num: int = 10, max_batch: int = 0, original=None):
        """`GraphLayersBuilder` on the storage's GPU (qmx_hnsw_build): over a dense f32 / f16 / u8 (not cosine) VectorStorage, or over an
        EncodedVectorsU8 / EncodedVectorsBin — the reference builds through the quantized scorer when the segment has one
        (hnsw/build.rs:334-341)."""
        self = cls.__new__(cls)
        self.m, self.m0 = int(m), int(2 * m if m0 is None else m0)
        p = F.HnswBuildParams()
        p.m, p.m0, p.ef_construct, p.entry_points_num, p.seed, p.max_batch = self.m, self.m0, ef_construct, entry_points_num, seed, max_batch
        self._keep = []
        self._h = C.c_void_p()
        # `original`: the f32 VectorStorage a PQ storage was encoded from (its insertion searches score through the LUT of the original
        # vector, point_scorer.rs:197-212); ignored by storages that can use a stored row as the query
        F.check(F.lib().qmx_hnsw_build_quantized(storage._h, original._h if original is not None else None, C.byref(p), C.byref(self._h)))
        self.n_points = storage.count
        self.counters = F.Counters()
        return self

    @classmethod
    def build_sharded(cls, storages, m: int = 16, m0: Optional[int] = None, ef_construct: int = 100, seed: int = 42, entry_points_num: int = 10,
                      max_batch: int = 0, originals=None):
        """One graph per segment, built side by side (`qmx_sharded_hnsw_build`: one host thread per segment inside the library, each on its
        segment's device - the reference's one-locked-GPU-per-segment-build, gpu_devices_manager.rs:120-143).  Returns the graphs in segment order."""
        n = len(storages)
        p = F.HnswBuildParams()
        p.m, p.m0, p.ef_construct, p.entry_points_num, p.seed, p.max_batch = int(m), int(2 * m if m0 is None else m0), ef_construct, entry_points_num, seed, max_batch
        segs = (C.c_void_p * n)(*[s._h for s in storages])
        orig = None if originals is None else (C.c_void_p * n)(*[None if o is None else o._h for o in originals])
        outs = (C.c_void_p * n)()
        status = (C.c_int32 * n)()
        rc = F.lib().qmx_sharded_hnsw_build(segs, orig, n, C.byref(p), outs, status)
        graphs = []
        for i in range(n):
            g = cls.__new__(cls)
            g.m, g.m0, g._keep, g._h = p.m, p.m0, [], C.c_void_p(outs[i])
            g.n_points, g.counters = storages[i].count, F.Counters()
            graphs.append(g)
        if rc != 0:
            for g in graphs:
                if g._h:
                    g.close()
            F.check(rc)
        return graphs

    @classmethod
    def build_multi(cls, multi_storage, m: int = 16, m0: Optional[int] = None, ef_construct: int = 100, seed: int = 42,
                    entry_points_num: int = 10, max_batch: int = 0, original=None):
        """`GraphLayersBuilder` over the POINTS of a MultiDenseVectorStorage / QuantizedMultivectorStorage on its GPU (qmx_multi_hnsw_build): every
        score of the build is MaxSim between two stored multi-vectors (`score_internal`, multi_metric_query_scorer.rs; quantized inner rows:
        `score_internal_max_similarity`, quantized_multivector_storage/mod.rs:366-393).  PQ inner rows cannot be turned back into queries: pass
        `original` = the f32 VectorStorage of the inner rows they were encoded from (qmx_multi_hnsw_build_quantized; the searches of an insertion then
        score through the LUTs of the point's original inner vectors, point_scorer.rs:183-218)."""
        self = cls.__new__(cls)
        self.m, self.m0 = int(m), int(2 * m if m0 is None else m0)
        p = F.HnswBuildParams()
        p.m, p.m0, p.ef_construct, p.entry_points_num, p.seed, p.max_batch = self.m, self.m0, ef_construct, entry_points_num, seed, max_batch
        self._keep = []
        self._h = C.c_void_p()
        deleted = multi_storage.point_deleted
        words = None if deleted is None else np.packbits(np.asarray(deleted, dtype=bool), bitorder="little")
        if words is not None:
            words = np.ascontiguousarray(np.pad(words, (0, (-len(words)) % 8))).view(np.uint64)
        F.check(F.lib().qmx_multi_hnsw_build_quantized(multi_storage.inner._h, None if original is None else original._h, F.ptr(multi_storage.offsets),
                                                       multi_storage.count, F.ptr(words), 0 if deleted is None else len(deleted), C.byref(p), C.byref(self._h)))
        self.n_points = multi_storage.count
        self.counters = F.Counters()
        return self

    def export_plain(self):
        """The plain GraphLinks arrays + entry points of a graph built on the device (an object with the fields
        `from_plain` takes)."""
        info = F.HnswInfo()
        F.check(F.lib().qmx_hnsw_get_info(self._h, C.byref(info)))

        class Plain:
            pass
        o = Plain()
        o.m, o.m0 = info.m, info.m0
        o.reindex = np.zeros(info.n_points, dtype=np.uint32)
        o.level_offsets = np.zeros(info.n_levels + 1, dtype=np.uint64)
        o.offsets = np.zeros(info.n_offsets, dtype=np.uint64)
        o.neighbors = np.zeros(max(info.n_neighbors, 1), dtype=np.uint32)
        o.ep_ids, o.ep_levels = np.zeros(info.n_entry_points, dtype=np.uint32), np.zeros(info.n_entry_points, dtype=np.uint32)
        o.xp_ids = np.zeros(info.n_extra_entry_points, dtype=np.uint32)
        o.xp_levels = np.zeros(info.n_extra_entry_points, dtype=np.uint32)
        F.check(F.lib().qmx_hnsw_export_plain(self._h, F.ptr(o.reindex), F.ptr(o.level_offsets), F.ptr(o.offsets), F.ptr(o.neighbors),
                                              F.ptr(o.ep_ids), F.ptr(o.ep_levels), F.ptr(o.xp_ids), F.ptr(o.xp_levels)))
        o.neighbors = o.neighbors[:info.n_neighbors]
        return o

    def search(self, top: int, ef: int, points_scorer: RawScorer, is_stopped=None, with_scored: bool = False, acorn: bool = False):
        """`GraphLayers::search(top, ef, Hnsw | Acorn, points_scorer, None, is_stopped)` for every query of the
        scorer batch -> list of ScoredPointOffset arrays (descending score, at most `top`).  `acorn`: SearchAlgorithm::Acorn
        (filter-aware 2-hop expansion on level 0, graph_layers.rs:154-243)."""
        nq = points_scorer.nq
        out = np.zeros((nq, max(top, 1)), dtype=ScoredPointOffset)
        counts = np.zeros(nq, dtype=np.uint32)
        stop = None
        if is_stopped is not None:
            stop = is_stopped if isinstance(is_stopped, np.ndarray) else np.array([1 if is_stopped else 0], dtype=np.uint8)
        fn = F.lib().qmx_hnsw_search_acorn if acorn else F.lib().qmx_hnsw_search
        F.check(fn(self._h, points_scorer._h, top, ef, F.ptr(out), F.ptr(counts), F.ptr(stop), C.byref(self.counters)))
        res = [out[i, :counts[i]].copy() for i in range(nq)]
        return (res, int(self.counters.vectors_scored)) if with_scored else res

    def search_traced(self, top: int, ef: int, points_scorer: RawScorer, pop_cap: int = 0):
        """`search` that also returns, per query, the candidates `search_on_level` popped and expanded on level 0, in order, with their scores
        (qmx_hnsw_search_traced) -> (lists, pops).  Two walks of one graph agree as long as their pop sequences agree; where they first differ,
        the two popped candidates show why (equal scores = the reference's heap order among ties).  With
        `set_option("hnsw_reference_heap_order", 1)` the walk keeps the reference's two binary heaps and returns ITS lists among equal scores."""
        nq = points_scorer.nq
        cap = int(pop_cap) if pop_cap else 32 * max(top, ef) + 256
        out = np.zeros((nq, max(top, 1)), dtype=ScoredPointOffset)
        counts = np.zeros(nq, dtype=np.uint32)
        while True:
            pops = np.zeros((nq, cap), dtype=ScoredPointOffset)
            pcnt = np.zeros(nq, dtype=np.uint32)
            F.check(F.lib().qmx_hnsw_search_traced(self._h, points_scorer._h, top, ef, F.ptr(out), F.ptr(counts), F.ptr(pops), cap, F.ptr(pcnt)))
            if int(pcnt.max(initial=0)) <= cap:
                break
            cap = int(pcnt.max())          # (the walk is deterministic: the second run pops the same candidates, now all listed)
        return [out[i, :counts[i]].copy() for i in range(nq)], [pops[i, :pcnt[i]].copy() for i in range(nq)]

    def search_with_vectors(self, top: int, ef: int, links_scorer: RawScorer, base_scorer: RawScorer, is_stopped=None, with_scored: bool = False,
                            raw_output: bool = False):
        """`GraphLayers::search_with_vectors(top, ef, links_scorer, links_scorer_bytes, base_scorer, None, is_stopped)` (graph_layers.rs:564-596):
        the walk over a graph with inline storage - steered by the quantized link vectors (`links_scorer`, over the quantized storage), every
        popped candidate scored on its full base vector (`base_scorer`, over the original storage); the best `top` base scores are returned."""
        nq = links_scorer.nq
        out = np.zeros((nq, max(top, 1)), dtype=ScoredPointOffset)
        counts = np.zeros(nq, dtype=np.uint32)
        stop = None
        if is_stopped is not None:
            stop = is_stopped if isinstance(is_stopped, np.ndarray) else np.array([1 if is_stopped else 0], dtype=np.uint8)
        F.check(F.lib().qmx_hnsw_search_with_vectors(self._h, links_scorer._h, base_scorer._h, top, ef, F.ptr(out), F.ptr(counts), F.ptr(stop),
                                                     C.byref(self.counters)))
        res = (out, counts) if raw_output else [out[i, :counts[i]].copy() for i in range(nq)]
        return (res, int(self.counters.vectors_scored)) if with_scored else res

    def close(self):
        if self._h:
            F.lib().qmx_hnsw_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode_links_file(data: bytes):
    """qmx_graph_links_decode: a graph-links file (Plain / Compressed / CompressedWithVectors) unpacked on the host into the
    plain arrays.  Needs no device.  Returns an object with format, m, m0, reindex, level_offsets, offsets, neighbors."""
    buf = np.frombuffer(data, dtype=np.uint8)
    g = F.GraphLinks()
    F.check(F.lib().qmx_graph_links_decode(F.ptr(buf), len(buf), C.byref(g)))
    try:
        class Links:
            pass
        o = Links()
        o.format, o.m, o.m0 = g.format, g.m, g.m0
        o.reindex = np.ctypeslib.as_array(g.reindex, (g.n_points,)).copy() if g.n_points else np.zeros(0, np.uint32)
        o.level_offsets = np.ctypeslib.as_array(g.level_offsets, (g.n_levels + 1,)).copy()
        o.offsets = np.ctypeslib.as_array(g.offsets, (g.n_offsets,)).copy()
        o.neighbors = np.ctypeslib.as_array(g.neighbors, (g.n_neighbors,)).copy() if g.n_neighbors else np.zeros(0, np.uint32)
        return o
    finally:
        F.lib().qmx_graph_links_free(C.byref(g))

"""Graph-links file formats (f4 of SURVEY §8): the oracle's restatement of the reference's writer / iterator
(oracle/qdrant_oracle_links.c) against the reference's own known-answer test, and the library's host-side reader
(qmx_graph_links_decode, qdrant_amd/csrc/links_file.hip — no device needed) against files the oracle wrote.

Reference tests mirrored: lib/common/common/src/bitpacking.rs `test_simple` (:181-209),
bitpacking_links.rs `test_random` cases (:230-312: only-unsorted / only-sorted / exact / empty / both),
bitpacking_ordered.rs `test_compress_decompress` sequences (:331-406),
graph_links/tests.rs `test_save_load` shapes (random links of random levels, every format)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_ffi as O
from qdrant_amd import _ffi as F
from qdrant_amd.hnsw import decode_links_file


def test_bitwriter_known_answer_of_the_reference():
    # bitpacking.rs:181-209
    values = [0b01010, 0b10110, 0b10100, 0b010110010, 0b101100001, 0b001001101, 0x12345678]
    bits = [5, 5, 5, 9, 9, 9, 32]
    packed = O.bitpack_write(values, bits)
    assert len(packed) == 10
    assert O.bitpack_read(packed, bits).tolist() == values
    # the layout itself: LSB-first, value i starts at bit sum(bits[:i])
    acc = 0
    for i, (v, b) in enumerate(zip(values, bits)):
        acc |= v << sum(bits[:i])
    assert packed == acc.to_bytes(10, "little")


@pytest.mark.parametrize("bits", [1, 7, 8, 13, 31, 32, 33, 56, 63, 64])
def test_bitwriter_reader_round_trip_every_width(bits):
    rng = np.random.default_rng(bits)
    n = 300
    hi = (1 << bits) - 1
    values = [int(x) & hi for x in rng.integers(0, 1 << 63, n, dtype=np.uint64) * 2 + rng.integers(0, 2, n, dtype=np.uint64)]
    packed = O.bitpack_write(values, [bits] * n)
    assert len(packed) == (bits * n + 7) // 8
    assert O.bitpack_read(packed, [bits] * n).tolist() == values


@pytest.mark.parametrize("case", ["only_unsorted", "only_sorted", "only_sorted_exact", "empty", "both"])
def test_pack_links_round_trip_cases_of_the_reference(case):
    # bitpacking_links.rs:230-312
    rng = np.random.default_rng(42)
    for _ in range(300):
        bits_per_unsorted = int(rng.integers(8, 33))
        sorted_count = int(rng.integers(0, 100))
        if case == "only_unsorted":
            sorted_count, total = 0, int(rng.integers(1, 100))
        elif case == "only_sorted":
            sorted_count = max(sorted_count, 2)
            total = int(rng.integers(1, sorted_count))
        elif case == "only_sorted_exact":
            sorted_count = max(sorted_count, 1)
            total = sorted_count
        elif case == "empty":
            total = 0
        else:
            total = sorted_count + int(rng.integers(1, 100))
        raw = rng.integers(0, 1 << bits_per_unsorted, total, dtype=np.uint64).astype(np.uint32)
        packed, left = O.pack_links(raw, bits_per_unsorted, sorted_count)
        expect = raw.copy()
        k = min(sorted_count, total)
        expect[:k] = np.sort(expect[:k])
        assert left.tolist() == expect.tolist()                       # what the reference leaves in raw_links
        assert O.iterate_packed_links(packed, bits_per_unsorted, sorted_count).tolist() == expect.tolist()
        assert O.packed_links_size(packed + b"\xAA" * 5, bits_per_unsorted, sorted_count, total) == len(packed)


def _sequences():
    rng = np.random.default_rng(42)
    yield [0]
    yield [1]
    yield [2 ** 64 - 1]
    yield [2 ** 64 - 1, 2 ** 64 - 1]
    yield [0, 2 ** 64 - 1]                                     # the "incomplete chunk" case: only chunk_len_log2 = 0 is admissible
    for max_delta, n in [(10, 1000), (20, 10_000), (10_000_000, 10_000), (0x123456789AB, 1000)]:
        yield np.cumsum(rng.integers(0, max_delta + 1, n, dtype=np.uint64), dtype=np.uint64).tolist()


def test_ordered_compress_round_trip():
    # bitpacking_ordered.rs:331-406
    for values in _sequences():
        data, params = O.ordered_compress(values)
        base_bits, delta_bits, cl = params
        chunk = (base_bits + delta_bits * ((1 << cl) - 1) + 7) // 8
        assert len(data) == -(-len(values) // (1 << cl)) * chunk + 7 and data[-7:] == b"\xff" * 7
        idx = range(len(values)) if len(values) <= 2000 else np.random.default_rng(1).integers(0, len(values), 2000)
        for i in idx:
            assert O.ordered_get(data, len(values), params, int(i)) == int(values[int(i)])


def _random_plain(n, m, m0, seed, max_level=4, full=False, consistent=False):
    """graph_links/tests.rs random_links: every point gets a random level and random link lists of random length.
    `consistent`: links on level l only name points of level >= l, as every built graph does (the reference's format tests
    draw them from all points: fine for the codecs, but such a graph cannot be walked and qmx_hnsw_create refuses it)."""
    rng = np.random.default_rng(seed)
    levels = np.minimum((-np.log(rng.random(n)) * (1.0 / np.log(max(m, 2)))).round().astype(np.int64), max_level)
    if n:
        levels[rng.integers(0, n)] = max_level
    order = np.argsort(-levels, kind="stable").astype(np.uint32)          # back_index
    reindex = np.zeros(n, dtype=np.uint32)
    reindex[order] = np.arange(n, dtype=np.uint32)
    n_levels = int(levels.max()) + 1 if n else 0
    level_offsets, offsets, neighbors = [], [0], []
    slot = 0
    for l in range(n_levels):
        level_offsets.append(slot)
        ids = np.arange(n) if l == 0 else order[: int((levels >= l).sum())]
        lm = m0 if l == 0 else m
        for _ in ids:
            # up to 2 x level_m links: lists longer than level_m exercise the unsorted tail (the reference allows it, tests.rs)
            k = lm if full else int(rng.integers(0, 2 * lm + 1))
            k = min(k, len(ids) if consistent else n)
            neighbors.extend(rng.choice(ids if consistent else n, size=k, replace=False).tolist() if k else [])
            offsets.append(len(neighbors))
            slot += 1
    level_offsets.append(slot)
    return O.PlainLinks(m, m0, reindex, np.array(level_offsets, dtype=np.uint64), np.array(offsets, dtype=np.uint64),
                        np.array(neighbors, dtype=np.uint32), [int(order[0])] if n else [], [int(levels[order[0]])] if n else [])


def _expected_lists(p):
    """links() of the compressed view: the first level_m links of every list ascending, the rest in place."""
    out = []
    n_levels = len(p.level_offsets) - 1
    for l in range(n_levels):
        lm = p.m0 if l == 0 else p.m
        for idx in range(int(p.level_offsets[l]), int(p.level_offsets[l + 1])):
            run = np.array(p.neighbors[int(p.offsets[idx]):int(p.offsets[idx + 1])], dtype=np.uint32)
            k = min(lm, len(run))
            run[:k] = np.sort(run[:k])
            out.append(run.tolist())
    return out


def _lists(d):
    return [d.neighbors[int(d.offsets[i]):int(d.offsets[i + 1])].tolist() for i in range(len(d.offsets) - 1)]


@pytest.mark.parametrize("n,m,m0", [(1, 2, 4), (2, 2, 4), (200, 4, 8), (255, 4, 8), (257, 8, 16), (3000, 16, 32), (70_000, 4, 8)])
def test_library_reads_compressed_links_written_by_the_oracle(n, m, m0):
    p = _random_plain(n, m, m0, seed=n)
    data = O.compressed_links_file(p)
    assert np.frombuffer(data[8:16], dtype="<u8")[0] == 0xFFFFFFFFFFFFFF01 and np.frombuffer(data[:8], dtype="<u8")[0] == n
    d = decode_links_file(data)
    assert (d.format, d.m, d.m0) == (1, m, m0)
    assert d.reindex.tolist() == np.asarray(p.reindex).tolist()
    assert d.level_offsets.tolist() == np.asarray(p.level_offsets).tolist()
    assert _lists(d) == _expected_lists(p)
    if n >= 200:   # smaller than the plain file it came from
        assert len(data) < len(O.plain_links_file(p))


@pytest.mark.parametrize("base_size,base_align,link_size,link_align", [(16, 8, 8, 8), (12, 4, 6, 2), (3, 1, 1, 1), (64, 16, 4, 4)])
def test_library_reads_compressed_links_with_inline_vectors(base_size, base_align, link_size, link_align):
    n, m, m0 = 500, 4, 8
    p = _random_plain(n, m, m0, seed=base_size)
    rng = np.random.default_rng(7)
    base = rng.integers(0, 256, (n, base_size), dtype=np.uint8)
    link = rng.integers(0, 256, (n, link_size), dtype=np.uint8)
    data = O.compressed_links_file(p, base, link, base_align, link_align)
    assert np.frombuffer(data[8:16], dtype="<u8")[0] == 0xFFFFFFFFFFFFFF02
    d = decode_links_file(data)
    assert (d.format, d.m, d.m0) == (2, m, m0)
    assert d.reindex.tolist() == np.asarray(p.reindex).tolist()
    assert _lists(d) == _expected_lists(p)


def test_library_reads_the_plain_file_too():
    p = _random_plain(300, 4, 8, seed=3)
    d = decode_links_file(O.plain_links_file(p))
    assert (d.format, d.m, d.m0) == (0, 0, 0)
    assert d.offsets.tolist() == np.asarray(p.offsets).tolist() and d.neighbors.tolist() == np.asarray(p.neighbors).tolist()
    assert d.level_offsets.tolist() == np.asarray(p.level_offsets).tolist()


def test_oracle_graph_survives_the_compressed_file():
    """A graph the oracle BUILT (real HNSW lists, <= level_m links each), compressed, read back by the library, and walked
    again by the oracle: the searches see the same lists up to the ascending order inside a list."""
    rows = O.preprocess(O.COSINE, O.synth(11, 0, 2000, 32))
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    g = O.Hnsw(st, m=8, ef_construct=64, seed=5)
    p = g.export_plain()
    d = decode_links_file(O.compressed_links_file(p))
    assert _lists(d) == [sorted(x) for x in _lists(p)]     # built lists never exceed level_m: fully sorted
    p2 = O.PlainLinks(d.m, d.m0, d.reindex, d.level_offsets, d.offsets, d.neighbors, p.ep_ids, p.ep_levels, p.xp_ids, p.xp_levels)
    g2 = O.Hnsw.from_plain(p2, 2000)
    q = O.synth(12, 0, 16, 32)
    hits = 0
    for a, b in zip(g.search_dense(st, q, 10, 64), g2.search_dense(st, q, 10, 64)):
        hits += len(set(a["idx"].tolist()) & set(b["idx"].tolist()))
    assert hits >= 150   # link order changes tie-breaks and hop batches, not the neighbourhoods


def _rc(data):
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    g = F.GraphLinks()
    rc = F.lib().qmx_graph_links_decode(F.ptr(buf), len(buf), C.byref(g))
    if rc == F.OK:
        F.lib().qmx_graph_links_free(C.byref(g))
    return rc


def test_malformed_compressed_files_are_refused():
    p = _random_plain(400, 4, 8, seed=9)
    data = O.compressed_links_file(p)
    assert _rc(data) == F.OK
    assert _rc(data[:40]) == F.ERR_BAD_ARG                                   # shorter than a header
    assert _rc(data[:-3]) == F.ERR_BAD_ARG                                   # tail of the offsets cut
    assert _rc(data[:len(data) // 2]) == F.ERR_BAD_ARG
    bad = bytearray(data)
    bad[41] = 0                                                              # delta_bits = 0 (Parameters::validate)
    assert _rc(bad) == F.ERR_BAD_ARG
    bad = bytearray(data)
    bad[42] = 9                                                              # chunk_len_log2 > 7
    assert _rc(bad) == F.ERR_BAD_ARG
    bad = bytearray(data)
    bad[24:32] = np.array([len(data) * 2], dtype="<u8").tobytes()            # total_neighbors_bytes past the file
    assert _rc(bad) == F.ERR_BAD_ARG
    bad = bytearray(data)
    bad[0:8] = np.array([3], dtype="<u8").tobytes()                          # point_count 3: links out of range / sections shift
    assert _rc(bad) in (F.ERR_BAD_ARG, F.ERR_OUT_OF_BOUNDS)
    # every single-byte corruption is either decoded or refused: never a crash, never an out-of-range link
    rng = np.random.default_rng(0)
    for _ in range(300):
        bad = bytearray(data)
        bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        buf = np.frombuffer(bytes(bad), dtype=np.uint8)
        g = F.GraphLinks()
        rc = F.lib().qmx_graph_links_decode(F.ptr(buf), len(buf), C.byref(g))
        if rc == F.OK:
            nb = np.ctypeslib.as_array(g.neighbors, (g.n_neighbors,)) if g.n_neighbors else np.zeros(0, np.uint32)
            assert (nb < g.n_points).all()
            F.lib().qmx_graph_links_free(C.byref(g))
        else:
            assert rc in (F.ERR_BAD_ARG, F.ERR_OUT_OF_BOUNDS)


def test_create_from_file_needs_the_device_only_after_the_file_is_valid():
    import torch
    p = _random_plain(300, 4, 8, seed=4, consistent=True)
    data = O.compressed_links_file(p)

    def create(buf, m=0, m0=0):
        d = F.HnswDesc()
        d.m, d.m0 = m, m0
        ep, epl = np.ascontiguousarray(p.ep_ids, dtype=np.uint32), np.ascontiguousarray(p.ep_levels, dtype=np.uint32)
        d.entry_point_ids, d.entry_point_levels, d.n_entry_points = ep.ctypes.data, epl.ctypes.data, len(ep)
        h = C.c_void_p()
        arr = np.frombuffer(bytes(buf), dtype=np.uint8)
        rc = F.lib().qmx_hnsw_create_from_file(F.ptr(arr), len(arr), C.byref(d), C.byref(h))
        if rc == F.OK:
            F.lib().qmx_hnsw_destroy(h)
        return rc
    assert create(data[:50]) == F.ERR_BAD_ARG
    assert create(data, m=5) == F.ERR_BAD_ARG                                # header says m = 4
    assert create(data) == (F.OK if torch.cuda.is_available() else F.ERR_NO_DEVICE)
    assert create(data, m=4, m0=8) == (F.OK if torch.cuda.is_available() else F.ERR_NO_DEVICE)


@pytest.mark.parametrize("kind", ["plain", "with_vectors"])
def test_corrupted_plain_and_inline_vector_files_never_crash(kind):
    """Single-bit corruptions and truncations of the other two formats: decoded with in-range links, or refused."""
    p = _random_plain(300, 4, 8, seed=12)
    if kind == "plain":
        data = O.plain_links_file(p)
    else:
        rng = np.random.default_rng(3)
        data = O.compressed_links_file(p, rng.integers(0, 256, (300, 12), dtype=np.uint8), rng.integers(0, 256, (300, 6), dtype=np.uint8), 4, 2)
    assert _rc(data) == F.OK
    rng = np.random.default_rng(1)
    variants = [bytes(data[:k]) for k in rng.integers(0, len(data), 40)]
    for _ in range(400):
        bad = bytearray(data)
        bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        variants.append(bytes(bad))
    for v in variants:
        buf = np.frombuffer(v, dtype=np.uint8) if len(v) else np.zeros(1, dtype=np.uint8)
        g = F.GraphLinks()
        rc = F.lib().qmx_graph_links_decode(F.ptr(buf), len(v), C.byref(g))
        if rc == F.OK:
            nb = np.ctypeslib.as_array(g.neighbors, (g.n_neighbors,)) if g.n_neighbors else np.zeros(0, np.uint32)
            off = np.ctypeslib.as_array(g.offsets, (g.n_offsets,))
            assert (nb < max(g.n_points, 1)).all() and (np.diff(off.astype(np.int64)) >= 0).all() and int(off[-1]) <= g.n_neighbors
            F.lib().qmx_graph_links_free(C.byref(g))
        else:
            assert rc in (F.ERR_BAD_ARG, F.ERR_OUT_OF_BOUNDS)


def test_links_to_nodes_missing_on_their_level_are_refused_on_any_host():
    """A decodable links file may still name, on level L >= 1, a node that has no slot on level L (or an entry point with a level it
    does not have): the walk would index offsets[] out of range.  qmx_hnsw_create refuses it before it needs a device."""
    import torch
    p = _random_plain(400, 4, 8, seed=21, consistent=True)
    assert len(p.level_offsets) - 1 >= 2
    expect_ok = F.OK if torch.cuda.is_available() else F.ERR_NO_DEVICE

    def create(neighbors, ep_ids, ep_levels):
        d = F.HnswDesc()
        re, lo, off = (np.ascontiguousarray(p.reindex, dtype=np.uint32), np.ascontiguousarray(p.level_offsets, dtype=np.uint64),
                       np.ascontiguousarray(p.offsets, dtype=np.uint64))
        nb, ep, epl = (np.ascontiguousarray(neighbors, dtype=np.uint32), np.ascontiguousarray(ep_ids, dtype=np.uint32),
                       np.ascontiguousarray(ep_levels, dtype=np.uint32))
        d.m, d.m0, d.n_points, d.n_levels = p.m, p.m0, len(re), len(lo) - 1
        d.reindex, d.level_offsets, d.offsets, d.n_offsets = re.ctypes.data, lo.ctypes.data, off.ctypes.data, len(off)
        d.neighbors, d.n_neighbors = nb.ctypes.data, len(nb)
        d.entry_point_ids, d.entry_point_levels, d.n_entry_points = ep.ctypes.data, epl.ctypes.data, len(ep)
        h = C.c_void_p()
        rc = F.lib().qmx_hnsw_create(C.byref(d), C.byref(h))
        if rc == F.OK:
            F.lib().qmx_hnsw_destroy(h)
        return rc
    assert create(p.neighbors, p.ep_ids, p.ep_levels) == expect_ok
    # a level-1 list that names a level-0-only node
    lo, off = np.asarray(p.level_offsets), np.asarray(p.offsets)
    size1 = int(lo[2] - lo[1])
    low = int(np.flatnonzero(np.asarray(p.reindex) >= size1)[0])          # a point that is not on level 1
    first = next(s for s in range(int(lo[1]), int(lo[2])) if off[s + 1] > off[s])
    bad = np.array(p.neighbors, dtype=np.uint32)
    bad[int(off[first])] = low
    assert create(bad, p.ep_ids, p.ep_levels) == F.ERR_OUT_OF_BOUNDS
    # an entry point that claims a level it is not on
    assert create(p.neighbors, [low], [1]) == F.ERR_OUT_OF_BOUNDS
    assert create(p.neighbors, [low], [0]) == expect_ok


def test_kernel_path_options_are_read_once_and_switchable():
    lib = F.lib()
    v = C.c_int64(-7)
    assert lib.qmx_get_option(b"prescan_shift", C.byref(v)) == F.OK and v.value == int(os.environ.get("QMX_PRESCAN_SHIFT", "10"))
    assert lib.qmx_set_option(b"no_mfma16", 1) == F.OK
    assert lib.qmx_get_option(b"no_mfma16", C.byref(v)) == F.OK and v.value == 1
    os.environ["QMX_NO_MFMA16"] = "0"                     # the environment is NOT consulted again
    assert lib.qmx_set_option(b"no_mfma16", -1) == F.OK   # back to the load-time value
    del os.environ["QMX_NO_MFMA16"]
    assert lib.qmx_get_option(b"no_mfma16", C.byref(v)) == F.OK and v.value == int(os.environ.get("QMX_NO_MFMA16", "0") != "0")
    assert lib.qmx_set_option(b"m16_dbg", 4) == F.ERR_BAD_ARG         # result-breaking debug knobs do not exist in the shipped library
    assert lib.qmx_set_option(None, 1) == F.ERR_BAD_ARG

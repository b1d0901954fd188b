"""GPU parity: Metric<f32> scoring, RawScorer::score_points and BatchFilteredSearcher::peek_top_iter
through the C-ABI (libqdrant_amd.so) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): f32 scores within 1e-5 relative of the reference CPU scorer; the
kernel is built to reproduce the AVX accumulation order, so most cases are checked bit-exact too.
Top-k: same ids modulo exact score ties, same scores, descending order.
Mirrors lib/segment/benches/vector_search.rs (peek_top_all), lib/segment/tests/integration/
exact_search_test.rs (exact == plain) and batch_search_test.rs (batched == single).
"""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu

REL = 1e-5  # tolerance stated by north_star for f32 distances


@pytest.fixture(scope="module")
def qa():
    import qdrant_amd
    assert qdrant_amd.device_count() >= 1
    return qdrant_amd


def _dist(qa, d):
    return {O.COSINE: qa.Distance.Cosine, O.DOT: qa.Distance.Dot, O.EUCLID: qa.Distance.Euclid,
            O.MANHATTAN: qa.Distance.Manhattan}[d]


def _rows(rng, n, dim, dist):
    raw = rng.standard_normal((n, dim)).astype(np.float32)
    return O.preprocess(dist, raw)  # stored rows are preprocessed at insert (named_vectors.rs:350-368)


def _close(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= REL * np.maximum(np.abs(b), 1e-30) + 1e-30)


@pytest.mark.parametrize("dist", [O.DOT, O.COSINE, O.EUCLID, O.MANHATTAN])
@pytest.mark.parametrize("dim", [32, 64, 100, 128, 768, 1536])
def test_score_points_matches_oracle(qa, dist, dim):
    rng = np.random.default_rng(dim * 7 + dist)
    n, nq = 777, 5
    rows = _rows(rng, n, dim, dist)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    st = qa.VectorStorage(rows, _dist(qa, dist))
    scorer = qa.new_raw_scorer(queries, st)
    ids = rng.permutation(n).astype(np.uint32)[:500]
    got = scorer.score_points(ids)
    ost = O.DenseStorage(O.F32, dist, rows)
    want = ost.score_points(queries, ids)
    assert got.shape == want.shape
    assert _close(got, want)
    # the lane map reproduces the AVX register order: expect bit equality, not just 1e-5
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # the query the device holds is Metric::preprocess(query) bit for bit
    enc = scorer.encoded_query(2)
    assert np.array_equal(enc.view(np.uint32), ost.encode_queries(queries)[2].view(np.uint32))


@pytest.mark.parametrize("dim", [1, 3, 7, 16, 17, 31, 33, 50, 70])
def test_score_points_odd_dims(qa, dim):
    # dims below the AVX/SSE thresholds and not multiples of 4 take the reference's SSE / scalar
    # paths (spaces/simple.rs:15,22); 70 = the reference's own test vector length (simple_avx.rs:223)
    rng = np.random.default_rng(dim)
    n, nq = 300, 3
    for dist in (O.DOT, O.COSINE, O.EUCLID, O.MANHATTAN):
        rows = _rows(rng, n, dim, dist)
        queries = rng.standard_normal((nq, dim)).astype(np.float32)
        st = qa.VectorStorage(rows, _dist(qa, dist))
        got = qa.new_raw_scorer(queries, st).score_points(np.arange(n, dtype=np.uint32))
        want = O.DenseStorage(O.F32, dist, rows).score_points(queries, np.arange(n, dtype=np.uint32))
        assert _close(got, want), (dim, dist)


def test_reference_literal_vectors(qa):
    # lib/segment/src/spaces/simple_avx.rs:223-252: 70-element literals, exact integer answers
    import json, os
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))
    v1, v2 = O.f32(G["f32_avx"]["v1"]), O.f32(G["f32_avx"]["v2"])
    for dist, exact in ((O.DOT, float(np.dot(v1, v2))), (O.EUCLID, -float(((v1 - v2) ** 2).sum())),
                        (O.MANHATTAN, -float(np.abs(v1 - v2).sum()))):
        st = qa.VectorStorage(v2[None, :], _dist(qa, dist))
        got = qa.new_raw_scorer(v1, st).score_points([0])
        assert got[0, 0] == exact


def _check_topk(got, want, rows_scores=None):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert len(g) == len(w)
        assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32))
        assert np.all(np.diff(g["score"]) <= 0)
        if len(g) == 0:
            continue
        # ids equal modulo exact ties at the cut: everything strictly above the last score must match
        last = w["score"][-1]
        assert set(g["idx"][g["score"] > last]) == set(w["idx"][w["score"] > last])


@pytest.mark.parametrize("dist", [O.COSINE, O.DOT, O.EUCLID, O.MANHATTAN])
@pytest.mark.parametrize("nq,top", [(1, 10), (3, 1), (16, 10), (20, 64), (37, 5)])
def test_peek_top_all_matches_oracle(qa, dist, nq, top):
    rng = np.random.default_rng(nq * 100 + top + dist)
    n, dim = 20011, 128
    rows = _rows(rng, n, dim, dist)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    st = qa.VectorStorage(rows, _dist(qa, dist))
    got = qa.BatchFilteredSearcher(queries, st, top).peek_top_all()
    want = O.DenseStorage(O.F32, dist, rows).peek_top(queries, top)
    _check_topk(got, want)
    for g, w in zip(got, want):  # continuous data: no ties, ids identical
        assert g["idx"].tolist() == w["idx"].tolist()


def test_peek_top_with_deleted_flags_and_id_list(qa):
    rng = np.random.default_rng(99)
    n, dim, nq, top = 5000, 64, 4, 10
    rows = _rows(rng, n, dim, O.COSINE)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    pdel = rng.random(n) < 0.3
    vdel = rng.random(n) < 0.1
    st = qa.VectorStorage(rows, qa.Distance.Cosine)
    st.set_deleted(pdel, vdel)
    ost = O.DenseStorage(O.F32, O.COSINE, rows, pdel, vdel)
    s = qa.BatchFilteredSearcher(queries, st, top)
    got = s.peek_top_all()
    want = ost.peek_top(queries, top)
    for g, w in zip(got, want):
        assert g["idx"].tolist() == w["idx"].tolist()
        assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32))
        assert not (pdel[g["idx"]] | vdel[g["idx"]]).any()
    # payload-filtered candidate list (plain_vector_index/read_view/search.rs:104-108)
    ids = rng.permutation(n).astype(np.uint32)[:700]
    got = s.peek_top_iter(ids)
    want = ost.peek_top(queries, top, ids=ids)
    for g, w in zip(got, want):
        assert g["idx"].tolist() == w["idx"].tolist()
    # shorter point-deleted bitslice: points past its end count as deleted (raw_scorer.rs:596-603)
    st.set_deleted(pdel[:1000], None)
    got = qa.BatchFilteredSearcher(queries, st, top).peek_top_all()
    want = O.DenseStorage(O.F32, O.COSINE, rows, pdel[:1000], None).peek_top(queries, top)
    for g, w in zip(got, want):
        assert g["idx"].tolist() == w["idx"].tolist() and g["idx"].max() < 1000


def test_ties_keep_lowest_offsets(qa):
    # FixedLengthPriorityQueue replaces the root only on strict `<`: among equal scores the
    # earlier-pushed (lower offset) survives (fixed_length_priority_queue.rs:53-57)
    rng = np.random.default_rng(1)
    dim = 32
    base = rng.standard_normal((1, dim)).astype(np.float32)
    rows = np.repeat(base, 1000, axis=0)
    rows[500:] *= 0.5
    st = qa.VectorStorage(rows, qa.Distance.Dot)
    got = qa.BatchFilteredSearcher(base, st, 10).peek_top_all()[0]
    want = O.DenseStorage(O.F32, O.DOT, rows).peek_top(base, 10)[0]
    assert sorted(got["idx"].tolist()) == sorted(want["idx"].tolist()) == list(range(10))
    assert np.array_equal(got["score"], want["score"])


def test_edge_cases(qa):
    rng = np.random.default_rng(3)
    dim = 64
    rows = _rows(rng, 7, dim, O.DOT)
    q = rng.standard_normal((2, dim)).astype(np.float32)
    st = qa.VectorStorage(rows, qa.Distance.Dot)
    # fewer rows than `top`
    got = qa.BatchFilteredSearcher(q, st, 10).peek_top_all()
    want = O.DenseStorage(O.F32, O.DOT, rows).peek_top(q, 10)
    for g, w in zip(got, want):
        assert len(g) == 7 and g["idx"].tolist() == w["idx"].tolist()
    # empty candidate list
    got = qa.BatchFilteredSearcher(q, st, 10).peek_top_iter(np.zeros(0, dtype=np.uint32))
    assert all(len(g) == 0 for g in got)
    # everything deleted
    st.set_deleted(np.ones(7, dtype=bool), None)
    assert all(len(g) == 0 for g in qa.BatchFilteredSearcher(q, st, 10).peek_top_all())
    st.set_deleted(None, None)
    # top == 0 panics in the reference (FixedLengthPriorityQueue::new)
    with pytest.raises(ValueError):
        qa.BatchFilteredSearcher(q, st, 0)
    # out-of-range offset: the reference panics, the ABI reports OUT_OF_BOUNDS
    with pytest.raises(qa.QmxError) as e:
        qa.new_raw_scorer(q, st).score_points([0, 7])
    assert e.value.status == qa._ffi.ERR_OUT_OF_BOUNDS
    # cancellation (check_process_stopped, point_scorer.rs:433,437)
    with pytest.raises(qa.QmxError) as e:
        qa.BatchFilteredSearcher(q, st, 3).peek_top_all(is_stopped=True)
    assert e.value.status == qa._ffi.ERR_CANCELLED
    # NaN scores sort greatest (OrderedFloat, types.rs:21-25)
    rows2 = rows.copy()
    rows2[3, 0] = np.nan
    st2 = qa.VectorStorage(rows2, qa.Distance.Dot)
    got = qa.BatchFilteredSearcher(q, st2, 3).peek_top_all()
    assert all(g["idx"][0] == 3 and np.isnan(g["score"][0]) for g in got)


def test_batched_equals_single(qa):
    # lib/segment/tests/integration/batch_search_test.rs: a batch returns what single searches return
    rng = np.random.default_rng(17)
    n, dim, nq = 3000, 96, 9
    rows = _rows(rng, n, dim, O.EUCLID)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    st = qa.VectorStorage(rows, qa.Distance.Euclid)
    batch = qa.BatchFilteredSearcher(queries, st, 7).peek_top_all()
    for i in range(nq):
        single = qa.BatchFilteredSearcher(queries[i], st, 7).peek_top_all()[0]
        assert single["idx"].tolist() == batch[i]["idx"].tolist()
        assert np.array_equal(single["score"], batch[i]["score"])


def test_device_resident_block_and_synth(qa):
    # rows generated on device (bench path) equal the oracle's generator bit for bit; adopting a
    # device block without copying gives the same search results as an uploaded copy
    import ctypes as C
    import torch
    from qdrant_amd import _ffi as F
    n, dim, seed = 4096, 128, 0x5EED0002
    t = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    F.check(F.lib().qmx_synth_fill_f32(0, seed, 0, n, dim, F.ptr(t)))
    host = O.synth(seed, 0, n, dim)
    assert np.array_equal(t.cpu().numpy().view(np.uint32), host.view(np.uint32))
    F.check(F.lib().qmx_preprocess_f32(0, int(qa.Distance.Cosine), F.ptr(t), n, dim, F.ptr(t)))
    want_rows = O.preprocess(O.COSINE, host)
    assert np.array_equal(t.cpu().numpy().view(np.uint32), want_rows.view(np.uint32))
    st = qa.VectorStorage(t, qa.Distance.Cosine)
    q = O.synth(seed + 1, 0, 3, dim)
    got = qa.BatchFilteredSearcher(q, st, 10).peek_top_all()
    want = O.DenseStorage(O.F32, O.COSINE, want_rows).peek_top(q, 10)
    for g, w in zip(got, want):
        assert g["idx"].tolist() == w["idx"].tolist()
        assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32))


def test_chunked_storage_ingest_equals_contiguous(qa):
    """`ChunkedVectors` (32 MiB chunks, chunked_vectors.rs): rows gathered from a chunk list score like the contiguous block."""
    import ctypes as C
    from qdrant_amd import _ffi as F
    rng = np.random.default_rng(12)
    n, dim, per = 1000, 40, 300                       # 4 chunks, the last one partial; dim * 4 = 160 B rows
    rows = O.preprocess(O.COSINE, rng.standard_normal((n, dim)).astype(np.float32))
    chunks = [np.ascontiguousarray(rows[i:i + per]) for i in range(0, n, per)]
    ptrs = (C.c_void_p * len(chunks))(*[c.ctypes.data for c in chunks])
    d = F.SegmentDesc()
    d.dtype, d.distance, d.dim, d.n, d.device_id = F.DTYPE_F32, int(qa.Distance.Cosine), dim, n, 0
    st = qa.VectorStorage.__new__(qa.VectorStorage)
    st._h, st.distance, st.datatype, st.dim, st.count, st._keep = C.c_void_p(), qa.Distance.Cosine, qa.VectorStorageDatatype.Float32, dim, n, None
    F.check(F.lib().qmx_segment_create_chunked(C.byref(d), ptrs, per, len(chunks), C.byref(st._h)))
    assert np.array_equal(st.get_dense([0, 299, 300, 999]), rows[[0, 299, 300, 999]])
    queries = rng.standard_normal((9, dim)).astype(np.float32)
    got = qa.BatchFilteredSearcher(queries, st, 10).peek_top_all()
    want = O.DenseStorage(O.F32, O.COSINE, rows).peek_top(queries, 10)
    for g, w in zip(got, want):
        assert g["idx"].tolist() == w["idx"].tolist()
        assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32))
    bad = F.lib().qmx_segment_create_chunked(C.byref(d), ptrs, 200, len(chunks), C.byref(C.c_void_p()))   # 4 x 200 < 1000 rows
    assert bad == F.ERR_BAD_ARG


def test_open_the_reference_storage_files(qa, tmp_path):
    """qmx_segment_create_from_files: the immutable dense vector file ("data" + rows) and the "drop" flags file
    (dense/immutable_dense_vectors.rs:25-27, 90, 364-378), and a flat quantized storage file (quantized_storage.rs:25-70)."""
    rng = np.random.default_rng(12)
    n, dim = 1237, 100                                           # 400-byte rows: re-packed to the 16-byte pitch on the way in
    rows = O.preprocess(O.COSINE, rng.standard_normal((n, dim)).astype(np.float32))
    deleted = rng.random(n) < 0.25
    vec_file, del_file = tmp_path / "matrix.dat", tmp_path / "deleted.dat"
    vec_file.write_bytes(b"data" + rows.tobytes())
    words = np.packbits(np.concatenate([deleted, np.zeros((-n) % 64, dtype=bool)]).reshape(-1, 8), axis=1, bitorder="little").tobytes()
    del_file.write_bytes(b"drop" + b"\0" * 4 + words)
    st = qa.VectorStorage.from_files(str(vec_file), dim, qa.Distance.Cosine, deleted_path=str(del_file))
    assert st.total_vector_count() == n
    assert np.array_equal(st.get_dense([0, 5, n - 1]), rows[[0, 5, n - 1]])
    queries = rng.standard_normal((5, dim)).astype(np.float32)
    got = qa.BatchFilteredSearcher(queries, st, 10).peek_top_all()
    want = O.DenseStorage(O.F32, O.COSINE, rows, vec_deleted=deleted).peek_top(queries, 10)
    for g, w in zip(got, want):
        assert np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32)) and g["idx"].tolist() == w["idx"].tolist()
    # wrong magic / short deleted file
    bad = tmp_path / "bad.dat"
    bad.write_bytes(b"dat0" + rows.tobytes())
    with pytest.raises(qa.QmxError):
        qa.VectorStorage.from_files(str(bad), dim, qa.Distance.Cosine)
    short = tmp_path / "short.dat"
    short.write_bytes(b"drop" + b"\0" * 4 + words[:8])
    with pytest.raises(qa.QmxError):
        qa.VectorStorage.from_files(str(vec_file), dim, qa.Distance.Cosine, deleted_path=str(short))
    # a flat SQ storage file: [f32 offset][codes] rows, no header
    import ctypes as C
    from qdrant_amd import _ffi as F
    quant = qa.ScalarQuantizer.from_min_max(rows, dim, qa.Distance.Dot)
    sq_rows = quant.encode(rows)
    qfile = tmp_path / "quantized.data"
    qfile.write_bytes(sq_rows.tobytes())
    p = quant.params()
    d = F.SegmentDesc()
    d.dtype, d.distance, d.dim, d.n, d.device_id, d.sq = F.DTYPE_SQ_U8, int(qa.Distance.Dot), dim, 0, 0, C.pointer(p)
    h = C.c_void_p()
    F.check(F.lib().qmx_segment_create_from_files(C.byref(d), str(qfile).encode(), None, C.byref(h)))
    back = np.empty((3, quant.quantized_vector_size()), dtype=np.uint8)
    ids = np.array([0, 7, n - 1], dtype=np.uint32)
    F.check(F.lib().qmx_segment_read_rows(h, F.ptr(ids), 3, F.ptr(back)))
    assert np.array_equal(back, sq_rows[ids])
    F.check(F.lib().qmx_segment_destroy(h))


@pytest.mark.parametrize("kind", ["sq", "pq", "bq", "tq"])
def test_chunked_quantized_storage_equals_contiguous(qa, kind):
    """`QuantizedChunkedMmapStorage` (quantized_chunked_mmap_storage/read_only.rs:20, read_write.rs:18): the appendable form of the quantized
    storages.  Rows gathered from a chunk list (the quantizer's own row layout) score bit for bit like the contiguous storage."""
    import ctypes as C
    from qdrant_amd import _ffi as F
    rng = np.random.default_rng(21)
    n, dim, per = 1000, 64, 300
    vecs = O.preprocess(O.DOT, rng.standard_normal((n, dim)).astype(np.float32))
    d = F.SegmentDesc()
    d.distance, d.dim, d.n, d.device_id = int(qa.Distance.Dot), dim, n, 0
    if kind == "sq":
        quant = qa.ScalarQuantizer.from_min_max(vecs, dim, qa.Distance.Dot)
        rows, whole = quant.encode(vecs), None
        whole = qa.EncodedVectorsU8(rows, quant)
        p = quant.params()
        d.dtype, d.sq = F.DTYPE_SQ_U8, C.pointer(p)
    elif kind == "pq":
        cen = O.PqOracle.train(vecs, dim, 8, 256, iters=2)
        quant = qa.ProductQuantizer(dim, qa.Distance.Dot, 8, cen)
        rows = quant.encode(vecs)
        whole = qa.EncodedVectorsPQ(rows, quant)
        p = quant.params()
        d.dtype, d.pq = F.DTYPE_PQ, C.pointer(p)
    elif kind == "bq":
        quant = qa.BinaryQuantizer(dim, qa.Distance.Dot)
        rows = quant.encode(vecs)
        whole = qa.EncodedVectorsBin(rows, quant)
        p = quant.params()
        d.dtype, d.bq = F.DTYPE_BQ, C.pointer(p)
    else:
        quant = qa.TurboQuantizer(dim, qa.Distance.Dot, O.TQ_BITS4)
        rows = quant.encode(vecs)
        whole = qa.EncodedVectorsTQ(rows, quant)
        p = quant.params()
        d.dtype, d.tq = F.DTYPE_TQ, C.pointer(p)
    chunks = [np.ascontiguousarray(rows[i:i + per]) for i in range(0, n, per)]
    ptrs = (C.c_void_p * len(chunks))(*[c.ctypes.data for c in chunks])
    st = type(whole).__new__(type(whole))
    st.__dict__.update({k: v for k, v in whole.__dict__.items() if k != "_h"})
    st._h = C.c_void_p()
    F.check(F.lib().qmx_segment_create_chunked(C.byref(d), ptrs, per, len(chunks), C.byref(st._h)))
    queries = rng.standard_normal((7, dim)).astype(np.float32)
    ids = rng.permutation(n)[:333].astype(np.uint32)
    a, b = qa.new_raw_scorer(queries, st), qa.new_raw_scorer(queries, whole)
    assert np.array_equal(a.score_points(ids).view(np.uint32), b.score_points(ids).view(np.uint32))
    got, want = qa.BatchFilteredSearcher(queries, st, 10).peek_top_all(), qa.BatchFilteredSearcher(queries, whole, 10).peek_top_all()
    for g, w in zip(got, want):
        assert g["idx"].tolist() == w["idx"].tolist() and np.array_equal(g["score"].view(np.uint32), w["score"].view(np.uint32))
    assert np.array_equal(st.get_quantized_vector([0, 299, 300, 999]), rows[[0, 299, 300, 999]])

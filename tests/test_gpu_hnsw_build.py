"""Device HNSW build (`qmx_hnsw_build`, hnsw_build.hpp) against the CPU oracle's GraphLayersBuilder restatement.

A concurrently built graph is not link-for-link the sequential one (the reference's own rayon / Vulkan builders are not
either), so the bars are the reference's own for its builders (graph_layers_builder.rs tests, hnsw/tests): structural
invariants, the same level draw, and search quality equal to the sequential CPU build within noise."""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    import qdrant_amd
    assert qdrant_amd.device_count() >= 1
    return qdrant_amd


def _dist(qa, d):
    return {O.COSINE: qa.Distance.Cosine, O.DOT: qa.Distance.Dot, O.EUCLID: qa.Distance.Euclid,
            O.MANHATTAN: qa.Distance.Manhattan}[d]


def _clustered(n, dim, seed, k=64):
    """Low intrinsic dimension (mixture of gaussians): HNSW recall is meaningful here, unlike on iid noise in d >> 10."""
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((k, dim)).astype(np.float32) * 2.0
    return (centers[rng.integers(0, k, n)] + rng.standard_normal((n, dim)).astype(np.float32) * 0.6).astype(np.float32)


def _recall(got, want):
    return sum(len(set(g["idx"].tolist()) & set(w["idx"].tolist())) for g, w in zip(got, want)) / float(sum(len(w) for w in want))


def _levels_of(plain, n):
    lv = np.zeros(n, dtype=np.int64)
    L = len(plain.level_offsets) - 1
    order = np.argsort(plain.reindex)
    for l in range(1, L):
        cnt = int(plain.level_offsets[l + 1] - plain.level_offsets[l])
        lv[order[:cnt]] = l
    return lv


@pytest.mark.parametrize("distance,dim", [(O.COSINE, 48), (O.EUCLID, 100), (O.DOT, 24)])
def test_build_invariants_levels_and_recall(qa, distance, dim):
    n, m, efc, seed = 6000, 8, 64, 7
    rows = O.preprocess(distance, _clustered(n, dim, seed))
    st = O.DenseStorage(O.F32, distance, rows)
    cpu = O.Hnsw(st, m=m, ef_construct=efc, seed=seed)
    vs = qa.VectorStorage(rows, _dist(qa, distance))
    gpu = qa.GraphLayers.build(vs, m=m, ef_construct=efc, seed=seed)
    p = gpu.export_plain()
    # same level draw as the oracle (graph_layers_builder.rs:388-396)
    lv = _levels_of(p, n)
    assert lv.tolist() == [cpu.point_level(i) for i in range(n)]
    assert int(p.level_offsets[1]) == n and sorted(p.reindex.tolist()) == list(range(n))
    # structural invariants
    L = len(p.level_offsets) - 1
    order = np.argsort(p.reindex)
    empty0 = 0
    for l in range(L):
        cnt = int(p.level_offsets[l + 1] - p.level_offsets[l])
        for j in range(0, cnt, 7 if l == 0 else 1):
            pid = j if l == 0 else int(order[j])
            slot = int(p.level_offsets[l]) + j
            ln = p.neighbors[int(p.offsets[slot]):int(p.offsets[slot + 1])]
            assert len(ln) <= (2 * m if l == 0 else m)
            assert len(set(ln.tolist())) == len(ln) and pid not in ln
            assert np.all(lv[ln] >= l)
            empty0 += int(l == 0 and len(ln) == 0)
    assert empty0 == 0
    ep_l = int(p.ep_levels[0])
    assert ep_l == lv.max() and lv[int(p.ep_ids[0])] == ep_l
    # search quality: device-built graph vs the sequential CPU build, same ef, both searched on the device
    queries = O.preprocess(distance, _clustered(200, dim, seed + 1)) if distance == O.COSINE else _clustered(200, dim, seed + 1)
    scorer = qa.new_raw_scorer(queries, vs)
    exact = st.peek_top(queries, 10)
    r_gpu = _recall(gpu.search(10, 64, scorer), exact)
    r_cpu = _recall(qa.GraphLayers.from_plain(cpu.export_plain()).search(10, 64, scorer), exact)
    assert r_cpu > 0.6 and r_gpu > r_cpu - 0.03, (r_gpu, r_cpu)
    # the CPU oracle walks the DEVICE-built graph exactly like the device does (ids, score bits, scored points)
    walk = O.Hnsw.from_plain(p, n)
    want, stats = walk.search_dense(st, queries[:60], 10, 64, with_stats=True)
    scorer60 = qa.new_raw_scorer(queries[:60], vs)
    got, scored = gpu.search(10, 64, scorer60, with_scored=True)
    for gq, wq in zip(got, want):
        assert gq["idx"].tolist() == wq["idx"].tolist()
        assert np.array_equal(gq["score"].view(np.uint32), wq["score"].view(np.uint32))
    assert scored == sum(stats)
    # the exported arrays round-trip through qmx_hnsw_create and the oracle-side plain file writer
    again = qa.GraphLayers.from_plain(p)
    a, b = gpu.search(10, 64, scorer), again.search(10, 64, scorer)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_build_skips_deleted_points_and_handles_tiny_inputs(qa):
    dim = 32
    rows = O.preprocess(O.COSINE, _clustered(3000, dim, 3))
    deleted = np.zeros(3000, dtype=bool)
    deleted[::5] = True
    deleted[0] = True
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    vs.set_deleted(deleted, None)
    g = qa.GraphLayers.build(vs, m=8, ef_construct=48, seed=1)
    p = g.export_plain()
    for i in range(0, 3000, 5):                       # deleted points: no links in or out
        assert int(p.offsets[i + 1]) == int(p.offsets[i])
    assert not deleted[p.neighbors].any() and not deleted[p.ep_ids].any()
    queries = O.preprocess(O.COSINE, _clustered(50, dim, 4))
    st = O.DenseStorage(O.F32, O.COSINE, rows, point_deleted=deleted)
    scorer = qa.new_raw_scorer(queries, vs)
    assert _recall(g.search(10, 128, scorer), st.peek_top(queries, 10)) > 0.7
    for n in (1, 2, 5):                              # tiny graphs
        small = qa.VectorStorage(rows[:n], qa.Distance.Cosine)
        gs = qa.GraphLayers.build(small, m=4, ef_construct=8, seed=2)
        res = gs.search(3, 8, qa.new_raw_scorer(queries[:4], small))
        assert all(len(r) == min(3, n) for r in res)
    # (u8 cosine rows are built too since round 2: test_u8_cosine_build)


def test_f16_build_and_large_batches(qa):
    n, dim = 20000, 64
    rows = O.preprocess(O.COSINE, _clustered(n, dim, 11))
    stored = O.to_f16(rows)
    vs = qa.VectorStorage(stored.view(np.float16), qa.Distance.Cosine, qa.VectorStorageDatatype.Float16)
    g = qa.GraphLayers.build(vs, m=16, ef_construct=100, seed=5, max_batch=4096)
    queries = _clustered(100, dim, 12)
    scorer = qa.new_raw_scorer(queries, vs)
    exact = O.DenseStorage(O.F16, O.COSINE, stored).peek_top(queries, 10)
    assert _recall(g.search(10, 256, scorer), exact) > 0.8


@pytest.mark.parametrize("distance,dim", [(O.DOT, 96), (O.EUCLID, 64), (O.MANHATTAN, 40)])
def test_sq_build_through_the_quantized_scorer(qa, distance, dim):
    """The reference builds the graph with the QUANTIZED scorer when the segment has one (hnsw/build.rs:334-341:
    FilteredScorer::new_internal over QuantizedVectors; SQ scores stored <-> stored through encode_internal_vector,
    encoded_vectors_u8.rs:715-728).  qmx_hnsw_build over an SQ-int8 segment: structural invariants, the oracle walks the
    device-built graph with the SQ scorer exactly like the device does, and search quality (SQ walk + f32 rescoring) is
    that of the graph built over the original vectors."""
    n, m, efc, seed = 6000, 8, 64, 17
    rows = O.preprocess(distance, _clustered(n, dim, seed))
    st = O.DenseStorage(O.F32, distance, rows)
    quant = qa.ScalarQuantizer.from_min_max(rows, dim, _dist(qa, distance))
    osq = O.SqOracle(distance, dim, quant.alpha, quant.offset)
    osq.encode_rows(rows)
    enc = qa.EncodedVectorsU8(quant.encode(rows), quant)
    vs = qa.VectorStorage(rows, _dist(qa, distance))
    g_sq = qa.GraphLayers.build(enc, m=m, ef_construct=efc, seed=seed)
    g_f32 = qa.GraphLayers.build(vs, m=m, ef_construct=efc, seed=seed)
    p = g_sq.export_plain()
    lv = _levels_of(p, n)
    assert lv.tolist() == _levels_of(g_f32.export_plain(), n).tolist()        # the level draw does not depend on the storage
    assert int(p.level_offsets[1]) == n and sorted(p.reindex.tolist()) == list(range(n))
    L = len(p.level_offsets) - 1
    order = np.argsort(p.reindex)
    empty0 = 0
    for l in range(L):
        cnt = int(p.level_offsets[l + 1] - p.level_offsets[l])
        for j in range(0, cnt, 5 if l == 0 else 1):
            pid = j if l == 0 else int(order[j])
            slot = int(p.level_offsets[l]) + j
            ln = p.neighbors[int(p.offsets[slot]):int(p.offsets[slot + 1])]
            assert len(ln) <= (2 * m if l == 0 else m)
            assert len(set(ln.tolist())) == len(ln) and pid not in ln
            assert np.all(lv[ln] >= l)
            empty0 += int(l == 0 and len(ln) == 0)
    assert empty0 == 0
    queries = _clustered(100, dim, seed + 1)
    qpre = O.preprocess(distance, queries)
    sq_scorer = qa.new_raw_scorer(queries, enc)
    # the oracle walks the device-built graph with its SQ scorer: same lists, same score bits (L1 scores tie: same quality)
    walk = O.Hnsw.from_plain(p, n)
    want = walk.search_sq(st, osq, qpre[:40], 10, 64)
    got = g_sq.search(10, 64, qa.new_raw_scorer(queries[:40], enc))
    if distance != O.MANHATTAN:
        for gq, wq in zip(got, want):
            assert gq["idx"].tolist() == wq["idx"].tolist()
            assert np.array_equal(gq["score"].view(np.uint32), wq["score"].view(np.uint32))
    # quality after rescoring with the original vectors: SQ-built graph vs f32-built graph, same SQ walk
    raw = qa.new_raw_scorer(queries, vs)
    exact = st.peek_top(queries, 10)
    r_sq = _recall(qa.search_quantized(sq_scorer, raw, 10, oversampling=2.0, rescore=True, graph=g_sq, hnsw_ef=64), exact)
    r_f32 = _recall(qa.search_quantized(sq_scorer, raw, 10, oversampling=2.0, rescore=True, graph=g_f32, hnsw_ef=64), exact)
    assert r_f32 > 0.4 and r_sq > r_f32 - 0.05, (r_sq, r_f32)


@pytest.mark.parametrize("distance", [O.DOT, O.EUCLID, O.MANHATTAN])
def test_u8_build(qa, distance):
    """Metric<u8> storages (dot / euclid / manhattan: the stored row is a complete query): invariants by construction of the same kernels,
    the oracle walks the device-built graph like the device, recall against exact u8 search."""
    n, dim, m, efc = 5000, 64, 8, 64
    rows = np.clip(_clustered(n, dim, 31) * 20.0 + 128.0, 0, 255).astype(np.uint8)
    st = O.DenseStorage(O.U8, distance, rows)
    vs = qa.VectorStorage(rows, _dist(qa, distance), qa.VectorStorageDatatype.Uint8)
    g = qa.GraphLayers.build(vs, m=m, ef_construct=efc, seed=9)
    p = g.export_plain()
    assert int(p.level_offsets[1]) == n and all(int(p.offsets[i + 1]) > int(p.offsets[i]) for i in range(0, n, 11))
    queries = np.clip(_clustered(80, dim, 32) * 20.0 + 128.0, 0, 255).astype(np.float32)
    scorer = qa.new_raw_scorer(queries, vs)
    exact = st.peek_top(queries, 10)
    got = g.search(10, 64, scorer)
    assert _recall(got, exact) > 0.6
    walk = O.Hnsw.from_plain(p, n)
    want = walk.search_dense(st, queries[:30], 10, 64)
    for gq, wq in zip(g.search(10, 64, qa.new_raw_scorer(queries[:30], vs)), want):
        assert np.array_equal(gq["score"].view(np.uint32), wq["score"].view(np.uint32))   # integer scores tie: ids may differ among equals


@pytest.mark.parametrize("query_encoding", [0, 2])
def test_bq_build_through_the_quantized_scorer(qa, query_encoding):
    """Binary-quantized segments: the build scores stored <-> stored rows with the one-bit xor-popcount (score_internal,
    encoded_vectors_binary.rs:892-917), whatever the segment's QueryEncoding.  Invariants + search quality after rescoring
    (binary quantization rescores by default, quantized_vectors/accessors.rs:16-38) against the graph built over the originals."""
    n, dim, m, efc, seed = 6000, 256, 8, 64, 23
    rng = np.random.default_rng(seed)                       # the data of test_gpu_bq's walk test: sign bits must tell neighbours apart
    centers = rng.standard_normal((32, dim)).astype(np.float32)

    def draw(count):
        return O.preprocess(O.COSINE, (centers[rng.integers(0, 32, count)] + 0.6 * rng.standard_normal((count, dim))).astype(np.float32))
    rows = draw(n)
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    quant = qa.BinaryQuantizer(dim, qa.Distance.Cosine, query_encoding=query_encoding)
    enc = qa.EncodedVectorsBin(quant.encode(rows), quant)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    g_bq = qa.GraphLayers.build(enc, m=m, ef_construct=efc, seed=seed)
    g_f32 = qa.GraphLayers.build(vs, m=m, ef_construct=efc, seed=seed)
    p = g_bq.export_plain()
    lv = _levels_of(p, n)
    assert lv.tolist() == _levels_of(g_f32.export_plain(), n).tolist()
    order = np.argsort(p.reindex)
    for l in range(len(p.level_offsets) - 1):
        cnt = int(p.level_offsets[l + 1] - p.level_offsets[l])
        for j in range(0, cnt, 5 if l == 0 else 1):
            pid = j if l == 0 else int(order[j])
            slot = int(p.level_offsets[l]) + j
            ln = p.neighbors[int(p.offsets[slot]):int(p.offsets[slot + 1])]
            assert 0 < len(ln) <= (2 * m if l == 0 else m) or l > 0
            assert len(set(ln.tolist())) == len(ln) and pid not in ln and np.all(lv[ln] >= l)
    queries = draw(100)
    scorer = qa.new_raw_scorer(queries, enc)
    raw = qa.new_raw_scorer(queries, vs)
    exact = st.peek_top(queries, 10)
    r_bq = _recall(qa.search_quantized(scorer, raw, 10, oversampling=3.0, rescore=True, graph=g_bq, hnsw_ef=64), exact)
    r_f32 = _recall(qa.search_quantized(scorer, raw, 10, oversampling=3.0, rescore=True, graph=g_f32, hnsw_ef=64), exact)
    assert r_f32 > 0.4 and r_bq > r_f32 - 0.1, (r_bq, r_f32)


def _check_invariants(p, n, m):
    lv = _levels_of(p, n)
    assert int(p.level_offsets[1]) == n and sorted(p.reindex.tolist()) == list(range(n))
    order = np.argsort(p.reindex)
    for l in range(len(p.level_offsets) - 1):
        cnt = int(p.level_offsets[l + 1] - p.level_offsets[l])
        for j in range(0, cnt, 5 if l == 0 else 1):
            pid = j if l == 0 else int(order[j])
            slot = int(p.level_offsets[l]) + j
            ln = p.neighbors[int(p.offsets[slot]):int(p.offsets[slot + 1])]
            assert len(ln) <= (2 * m if l == 0 else m) and (len(ln) > 0 or l > 0)
            assert len(set(ln.tolist())) == len(ln) and pid not in ln and np.all(lv[ln] >= l)
    return lv


@pytest.mark.parametrize("distance,lut_mfma", [(O.DOT, False), (O.EUCLID, False), (O.DOT, True)])
def test_pq_build_through_the_quantized_scorer(qa, distance, lut_mfma):
    """A PQ segment builds its graph like the reference (hnsw/build.rs:334-341, point_scorer.rs:183-218): insertion searches score
    through the LUT of the point's ORIGINAL vector, the heuristic and the back links through EncodedVectorsPQ::score_internal.
    Bars: the structural invariants; the oracle's PQ walk of the device-built graph == the device's (ids, score bits: exact-order LUT);
    recall after rescoring within 0.05 of (a) the graph the ORACLE builds the same way on the CPU and (b) the f32-built graph."""
    n, dim, chunk, m, efc, seed = 5000, 64, 4, 8, 64, 11
    rows = O.preprocess(distance, _clustered(n, dim, seed, k=48))
    st = O.DenseStorage(O.F32, distance, rows)
    cen = O.PqOracle.train(rows[:2000], dim, chunk, 256, iters=4)
    opq = O.PqOracle(distance, dim, chunk, cen)
    codes = opq.encode(rows)
    quant = qa.ProductQuantizer(dim, _dist(qa, distance), chunk, cen, lut_mfma=lut_mfma)
    enc = qa.EncodedVectorsPQ(codes, quant)
    vs = qa.VectorStorage(rows, _dist(qa, distance))
    with pytest.raises(qa.QmxError) as e:                     # without the original vectors there is no query for a stored PQ row
        qa.GraphLayers.build(enc, m=m, ef_construct=efc, seed=seed)
    assert e.value.status == qa._ffi.ERR_NOT_SUPPORTED
    g_pq = qa.GraphLayers.build(enc, m=m, ef_construct=efc, seed=seed, original=vs)
    g_f32 = qa.GraphLayers.build(vs, m=m, ef_construct=efc, seed=seed)
    p = g_pq.export_plain()
    lv = _check_invariants(p, n, m)
    assert lv.tolist() == _levels_of(g_f32.export_plain(), n).tolist()
    queries = _clustered(100, dim, seed + 1, k=48)
    qpre = O.preprocess(distance, queries)
    pq_scorer = qa.new_raw_scorer(queries, enc)
    if not lut_mfma:
        walk = O.Hnsw.from_plain(p, n)
        want = walk.search_pq(st, opq, qpre[:40], 10, 64)
        got = g_pq.search(10, 64, qa.new_raw_scorer(queries[:40], enc))
        for gq, wq in zip(got, want):
            assert gq["idx"].tolist() == wq["idx"].tolist()
            assert np.array_equal(gq["score"].view(np.uint32), wq["score"].view(np.uint32))
    raw = qa.new_raw_scorer(queries, vs)
    exact = st.peek_top(queries, 10)
    r_pq = _recall(qa.search_quantized(pq_scorer, raw, 10, oversampling=4.0, rescore=True, graph=g_pq, hnsw_ef=64), exact)
    r_f32 = _recall(qa.search_quantized(pq_scorer, raw, 10, oversampling=4.0, rescore=True, graph=g_f32, hnsw_ef=64), exact)
    # the oracle's sequential build through the same scorers (asymmetric searches, symmetric heuristic), walked by the device
    cpu = O.Hnsw.build_pq(st, opq, m=m, ef_construct=efc, seed=seed)
    g_cpu = qa.GraphLayers.from_plain(cpu.export_plain())
    r_cpu = _recall(qa.search_quantized(pq_scorer, raw, 10, oversampling=4.0, rescore=True, graph=g_cpu, hnsw_ef=64), exact)
    assert r_f32 > 0.4 and r_pq > r_f32 - 0.05 and r_pq > r_cpu - 0.05, (r_pq, r_cpu, r_f32)


@pytest.mark.parametrize("distance,dim", [(O.COSINE, 48), (O.EUCLID, 33)])
def test_one_point_per_launch_builds_the_sequential_graph(qa, distance, dim):
    """max_batch = 1: every insertion sees the graph of all earlier points, as the oracle's sequential GraphLayersBuilder does - the device
    graph then equals the oracle's link for link (same scores bit for bit, same tie rules in the queues and the heuristic)."""
    n, m, efc, seed = 1500, 8, 40, 21
    rows = O.preprocess(distance, _clustered(n, dim, seed, k=32))
    st = O.DenseStorage(O.F32, distance, rows)
    vs = qa.VectorStorage(rows, _dist(qa, distance))
    seq = qa.GraphLayers.build(vs, m=m, ef_construct=efc, seed=seed, max_batch=1).export_plain()
    ref = O.Hnsw(st, m=m, ef_construct=efc, seed=seed).export_plain()
    _same_graph(seq, ref)


@pytest.mark.parametrize("kind,efc", [("f32", 600), ("f32", 1000), ("sq", 700), ("pq", 520)])
def test_wide_construction_beams_build_the_sequential_graph(qa, kind, efc):
    """ef_construct above 512 (the register beam of the insertion searches): the list lives in LDS (hnsw.hpp Beam<0>, as for walks wider than 512) - one
    insertion per launch the graph is still the oracle's sequential graph link for link, through the dense, the SQ and the PQ scorer.  The reference has
    no limit on ef_construct (HnswConfig.ef_construct: usize); here 4 096 (HNSW_MAX_EF), refused beyond."""
    n, dim, m, seed = 1300, 48, 8, 33
    distance = O.COSINE if kind == "f32" else O.DOT
    rows = O.preprocess(O.COSINE, _clustered(n, dim, seed, k=16))
    st = O.DenseStorage(O.F32, distance, rows)
    vs = qa.VectorStorage(rows, _dist(qa, distance))
    if kind == "f32":
        seq = qa.GraphLayers.build(vs, m=m, ef_construct=efc, seed=seed, max_batch=1)
        ref = O.Hnsw(st, m=m, ef_construct=efc, seed=seed)
    elif kind == "sq":
        quant = qa.ScalarQuantizer.fit(rows, dim, qa.Distance.Dot)
        osq = O.SqOracle(O.DOT, dim, quant.alpha, quant.offset)
        osq.rows = osq.encode_rows(rows)
        seq = qa.GraphLayers.build(qa.EncodedVectorsU8(quant.encode(rows), quant), m=m, ef_construct=efc, seed=seed, max_batch=1)
        ref = O.Hnsw.build_sq(st, osq, m=m, ef_construct=efc, seed=seed)
    else:
        cen = O.PqOracle.train(rows[:1000], dim, 4, 256, iters=2)
        opq = O.PqOracle(O.DOT, dim, 4, cen)
        codes = opq.encode(rows)
        quant = qa.ProductQuantizer(dim, qa.Distance.Dot, 4, cen)
        seq = qa.GraphLayers.build(qa.EncodedVectorsPQ(codes, quant), m=m, ef_construct=efc, seed=seed, max_batch=1, original=vs)
        ref = O.Hnsw.build_pq(st, opq, m=m, ef_construct=efc, seed=seed)
    a, b = seq.export_plain(), ref.export_plain()
    if kind == "sq":
        # SQ scores are a multiplier times an INTEGER dot: with a list of 700 of the 1300 points nearly every insertion meets equal scores, and among
        # equals the reference's order is its heap's (unpinned) - the graphs agree list for list as SETS on all but a few points
        assert np.array_equal(a.reindex, b.reindex) and a.ep_ids.tolist() == b.ep_ids.tolist()
        same = sum(int(set(a.neighbors[int(a.offsets[i]):int(a.offsets[i + 1])].tolist()) == set(b.neighbors[int(b.offsets[i]):int(b.offsets[i + 1])].tolist()))
                   for i in range(len(a.offsets) - 1))
        assert same >= 0.97 * (len(a.offsets) - 1), same
    else:
        _same_graph(a, b)
    with pytest.raises(qa.QmxError) as e:
        qa.GraphLayers.build(vs, m=m, ef_construct=5000, seed=seed)
    assert e.value.status == qa._ffi.ERR_NOT_SUPPORTED


@pytest.mark.parametrize("m,m0", [(40, 80), (64, 128), (24, 100)])
def test_wide_link_lists_build_the_sequential_graph(qa, m, m0):
    """m0 up to 128 (m = 64 collections): the same bar as above - one point per launch, the oracle's sequential graph link for link - and the plain walk of
    the result equals the oracle's walk (lists of more than 63 links: the CSR arrays, 64 links per trip)."""
    # rows whose similarities factor, sim(i, j) = s_i s_j with s ascending in insertion order: the heuristic drops a candidate only for a kept link with
    # a larger s than the TARGET's, and the point being inserted always has the largest s so far - its lists fill to the brim (natural data rarely gets
    # past ~60 links under the heuristic); distinct s keep the scores distinct
    n, efc, seed = 400, 200, 5
    dim = n + 1
    w = np.linspace(0.5, 2.0, n) + np.random.default_rng(seed).uniform(0, 1e-3, n)
    rows = np.zeros((n, dim), dtype=np.float32)
    rows[np.arange(n), np.arange(n)] = 1.0
    rows[:, n] = w
    rows = O.preprocess(O.COSINE, rows)
    st = O.DenseStorage(O.F32, O.COSINE, rows)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    g = qa.GraphLayers.build(vs, m=m, m0=m0, ef_construct=efc, seed=seed, max_batch=1)
    ref = O.Hnsw(st, m=m, m0=m0, ef_construct=efc, seed=seed)
    seq = g.export_plain()
    _same_graph(seq, ref.export_plain())
    assert np.diff(seq.offsets[:n + 1].astype(np.int64)).max() > 64          # the case is what it says
    queries = O.synth(77, 0, 10, dim)
    scorer = qa.new_raw_scorer(queries, vs)
    want = ref.search_dense(st, queries, 10, 64)
    got = g.search(10, 64, scorer)
    for a, b in zip(got, want):
        assert a["idx"].tolist() == b["idx"].tolist() and np.array_equal(a["score"].view(np.uint32), b["score"].view(np.uint32))
    with pytest.raises(qa.QmxError):
        qa.GraphLayers.build(vs, m=64, m0=129, ef_construct=efc, seed=seed)


def _same_graph(seq, ref):
    assert np.array_equal(seq.reindex, ref.reindex) and np.array_equal(seq.offsets, ref.offsets)
    assert np.array_equal(seq.neighbors, ref.neighbors)
    assert seq.ep_ids.tolist() == ref.ep_ids.tolist()


@pytest.mark.parametrize("kind", ["sq", "pq"])
def test_one_point_per_launch_builds_the_sequential_graph_other_storages(qa, kind):
    """The same bar for the SQ and PQ quantized scorers (PQ: searches through the LUT of the original vector, heuristic through score_internal;
    TurboQuant: test_tq_build_through_the_quantized_scorer).  Not a bar for u8 / BQ rows - integer scores tie all the time and the order among
    equal scores inside the reference's binary heaps is not restated on the device (ties unpinned) - nor for f16 rows, whose device
    scores equal the oracle's to 1e-5, not to the bit; their branches below document what was tried."""
    n, dim, m, efc, seed = 1200, 64, 8, 40, 29
    raw = _clustered(n, dim, seed, k=32)
    if kind == "f16":
        rows = O.preprocess(O.DOT, raw).astype(np.float16)
        st = O.DenseStorage(O.F16, O.DOT, rows)
        seq = qa.GraphLayers.build(qa.VectorStorage(rows, qa.Distance.Dot, qa.VectorStorageDatatype.Float16), m=m, ef_construct=efc, seed=seed, max_batch=1)
        ref = O.Hnsw(st, m=m, ef_construct=efc, seed=seed)
    elif kind == "u8":
        rows = np.clip(raw * 20.0 + 128.0, 0, 255).astype(np.uint8)
        st = O.DenseStorage(O.U8, O.EUCLID, rows)
        seq = qa.GraphLayers.build(qa.VectorStorage(rows, qa.Distance.Euclid, qa.VectorStorageDatatype.Uint8), m=m, ef_construct=efc, seed=seed, max_batch=1)
        ref = O.Hnsw(st, m=m, ef_construct=efc, seed=seed)
    else:
        rows = O.preprocess(O.COSINE, raw)
        st = O.DenseStorage(O.F32, O.COSINE, rows)
        vs = qa.VectorStorage(rows, qa.Distance.Cosine)
        if kind == "sq":
            quant = qa.ScalarQuantizer.from_min_max(rows, dim, qa.Distance.Cosine)
            osq = O.SqOracle(O.COSINE, dim, quant.alpha, quant.offset)
            osq.encode_rows(rows)
            seq = qa.GraphLayers.build(qa.EncodedVectorsU8(quant.encode(rows), quant), m=m, ef_construct=efc, seed=seed, max_batch=1)
            ref = O.Hnsw.build_sq(st, osq, m=m, ef_construct=efc, seed=seed)
        elif kind == "bq":
            quant = qa.BinaryQuantizer(dim, qa.Distance.Cosine)
            obq = O.BqOracle(O.COSINE, dim)
            obq.encode_rows(rows)
            seq = qa.GraphLayers.build(qa.EncodedVectorsBin(quant.encode(rows), quant), m=m, ef_construct=efc, seed=seed, max_batch=1)
            ref = O.Hnsw.build_bq(st, obq, m=m, ef_construct=efc, seed=seed)
        else:
            cen = O.PqOracle.train(rows[:1000], dim, 4, 256, iters=3)
            opq = O.PqOracle(O.COSINE, dim, 4, cen)
            codes = opq.encode(rows)
            quant = qa.ProductQuantizer(dim, qa.Distance.Cosine, 4, cen, lut_mfma=False)
            seq = qa.GraphLayers.build(qa.EncodedVectorsPQ(codes, quant), m=m, ef_construct=efc, seed=seed, max_batch=1, original=vs)
            ref = O.Hnsw.build_pq(st, opq, m=m, ef_construct=efc, seed=seed)
    _same_graph(seq.export_plain(), ref.export_plain())


@pytest.mark.parametrize("distance,bits,plus", [(O.COSINE, O.TQ_BITS4, False), (O.EUCLID, O.TQ_BITS2, False), (O.DOT, O.TQ_BITS1, False),
                                                (O.DOT, O.TQ_BITS4, True), (O.EUCLID, O.TQ_BITS1_5, True),
                                                # over Manhattan (round 4): the searches score the original vector against the dequantised, back-rotated
                                                # candidates (score_precomputed's L1 arm), stored <-> stored pairs through score_symmetric's (hnsw_build_tq_l1.hip)
                                                (O.MANHATTAN, O.TQ_BITS4, False), (O.MANHATTAN, O.TQ_BITS2, True)])
def test_tq_build_through_the_quantized_scorer(qa, distance, bits, plus):
    """A TurboQuant segment builds like a PQ one (EncodedVectorsTQ::encode_internal_vector -> None, point_scorer.rs:183-218): insertion
    searches score through precompute_query of the point's ORIGINAL vector, the heuristic and the back links through score_symmetric (TQ+:
    score_symmetric_ec).  Bars: inserted one point per launch (max_batch = 1) the device builds the ORACLE's sequential graph link for link;
    batched: the structural invariants, the oracle's TQ walk of the device-built graph == the device's (ids, score bits), recall after
    rescoring within 0.05 of the oracle-built and of the f32-built graph."""
    n, dim, m, efc, seed = 3000, 64, 8, 48, 13
    rows = O.preprocess(distance, _clustered(n, dim, seed, k=48))
    st = O.DenseStorage(O.F32, distance, rows)
    shift = scale = None
    if plus:
        shift, scale = O.tq_plus_fit(distance, dim, bits, rows)
    otq = O.TqOracle(distance, dim, bits, shift=shift, scale=scale)
    codes = otq.encode_rows(rows)
    quant = qa.TurboQuantizer(dim, _dist(qa, distance), bits, shift=shift, scale=scale)
    enc = qa.EncodedVectorsTQ(codes, quant)
    vs = qa.VectorStorage(rows, _dist(qa, distance))
    with pytest.raises(qa.QmxError) as e:                     # without the original vectors there is no query for a stored TQ row
        qa.GraphLayers.build(enc, m=m, ef_construct=efc, seed=seed)
    assert e.value.status == qa._ffi.ERR_NOT_SUPPORTED
    cpu = O.Hnsw.build_tq(st, otq, m=m, ef_construct=efc, seed=seed)
    cpu_plain = cpu.export_plain()
    # sequential on the device == sequential on the CPU, link for link (scores are the oracle's bits, ties broken alike)
    n_seq = 700
    enc_s = qa.EncodedVectorsTQ(codes[:n_seq], quant)
    vs_s = qa.VectorStorage(rows[:n_seq], _dist(qa, distance))
    st_s = O.DenseStorage(O.F32, distance, rows[:n_seq])
    otq_s = O.TqOracle(distance, dim, bits, shift=shift, scale=scale)
    otq_s.rows = codes[:n_seq]
    seq = qa.GraphLayers.build(enc_s, m=m, ef_construct=efc, seed=seed, original=vs_s, max_batch=1).export_plain()
    ref = O.Hnsw.build_tq(st_s, otq_s, m=m, ef_construct=efc, seed=seed).export_plain()
    assert np.array_equal(seq.reindex, ref.reindex) and np.array_equal(seq.offsets, ref.offsets)
    assert np.array_equal(seq.neighbors, ref.neighbors)
    assert seq.ep_ids.tolist() == ref.ep_ids.tolist()
    # batched
    g_tq = qa.GraphLayers.build(enc, m=m, ef_construct=efc, seed=seed, original=vs)
    g_f32 = qa.GraphLayers.build(vs, m=m, ef_construct=efc, seed=seed)
    p = g_tq.export_plain()
    lv = _check_invariants(p, n, m)
    assert lv.tolist() == _levels_of(g_f32.export_plain(), n).tolist()
    queries = _clustered(100, dim, seed + 1, k=48)
    qpre = O.preprocess(distance, queries)
    tq_scorer = qa.new_raw_scorer(queries, enc)
    walk = O.Hnsw.from_plain(p, n)
    want = walk.search_tq(st, otq, qpre[:40], 10, 64)
    got = g_tq.search(10, 64, qa.new_raw_scorer(queries[:40], enc))
    for gq, wq in zip(got, want):
        assert np.array_equal(gq["score"].view(np.uint32), wq["score"].view(np.uint32))
        if bits not in (O.TQ_BITS1, O.TQ_BITS1_5):            # 1-bit scores tie: the order among equal scores is not pinned
            assert gq["idx"].tolist() == wq["idx"].tolist()
    raw = qa.new_raw_scorer(queries, vs)
    exact = st.peek_top(queries, 10)
    r_tq = _recall(qa.search_quantized(tq_scorer, raw, 10, oversampling=4.0, rescore=True, graph=g_tq, hnsw_ef=64), exact)
    r_f32 = _recall(qa.search_quantized(tq_scorer, raw, 10, oversampling=4.0, rescore=True, graph=g_f32, hnsw_ef=64), exact)
    g_cpu = qa.GraphLayers.from_plain(cpu_plain)
    r_cpu = _recall(qa.search_quantized(tq_scorer, raw, 10, oversampling=4.0, rescore=True, graph=g_cpu, hnsw_ef=64), exact)
    assert r_f32 > (0.3 if bits in (O.TQ_BITS1, O.TQ_BITS1_5) else 0.4) and r_tq > r_f32 - 0.05 and r_tq > r_cpu - 0.05, (r_tq, r_cpu, r_f32)


def test_tq_manhattan_build_with_an_unpadded_rotation(qa):
    """TQRotation::Unpadded over 96 coordinates (chunks of 64 + 32), 2-bit codes with the TQ+ correction.  One point per launch: the oracle's sequential
    graph, link for link; the walk: the oracle's.  (A rotation shorter than the code - the 1.5-bit layout - is refused by the reference itself with an
    unpadded rotation: `Bits1_5 requires TQRotation::Padded`.)"""
    n, dim, m, efc, seed = 600, 96, 8, 40, 5
    rows = _clustered(n, dim, seed, k=24)
    st = O.DenseStorage(O.F32, O.MANHATTAN, rows)
    shift, scale = O.tq_plus_fit(O.MANHATTAN, dim, O.TQ_BITS2, rows, rotation_unpadded=True)
    otq = O.TqOracle(O.MANHATTAN, dim, O.TQ_BITS2, rotation_unpadded=True, shift=shift, scale=scale)
    assert otq.padded_dim == 96
    codes = otq.encode_rows(rows)
    otq.rows = codes
    quant = qa.TurboQuantizer(dim, qa.Distance.Manhattan, O.TQ_BITS2, rotation_unpadded=True, shift=shift, scale=scale)
    enc = qa.EncodedVectorsTQ(codes, quant)
    vs = qa.VectorStorage(rows, qa.Distance.Manhattan)
    seq = qa.GraphLayers.build(enc, m=m, ef_construct=efc, seed=seed, original=vs, max_batch=1)
    p = seq.export_plain()
    ref = O.Hnsw.build_tq(st, otq, m=m, ef_construct=efc, seed=seed).export_plain()
    assert np.array_equal(p.reindex, ref.reindex) and np.array_equal(p.offsets, ref.offsets)
    assert np.array_equal(p.neighbors, ref.neighbors)
    queries = _clustered(20, dim, seed + 1, k=24)
    want = O.Hnsw.from_plain(p, n).search_tq(st, otq, queries, 10, 48)
    got = seq.search(10, 48, qa.new_raw_scorer(queries, enc))
    for gq, wq in zip(got, want):
        assert np.array_equal(gq["score"].view(np.uint32), wq["score"].view(np.uint32))


@pytest.mark.parametrize("tables", [False, True])
@pytest.mark.parametrize("distance,dim,chunk", [(O.COSINE, 64, 4), (O.EUCLID, 96, 16), (O.MANHATTAN, 80, 8), (O.DOT, 320, 16), (O.DOT, 70, 8)])
def test_pq_build_with_and_without_tables_is_the_sequential_graph(qa, distance, dim, chunk, tables):
    """The PQ build recomputes what it needs from the codebook by default (round 4, pq.hip HopPQDirectBuild + HopPQInternalDirect: the LUT entries of the
    insertion searches in pq_lut_kernel's order, the centroid-pair terms of score_internal in pq_pair_table_kernel's - the same bits as the tables hold);
    option hnsw_pq_table_build = round 2's build through per-insertion LUTs and the 25 MB pair table.  One point per launch both are the oracle's sequential
    graph link for link.  (70 = 8 x 8 + 6: a ragged last chunk keeps the tables whatever the option.)"""
    n, m, efc, seed = 900, 8, 40, 29
    rows = O.preprocess(distance, _clustered(n, dim, seed, k=24))
    st = O.DenseStorage(O.F32, distance, rows)
    vs = qa.VectorStorage(rows, _dist(qa, distance))
    cen = O.PqOracle.train(rows[:800], dim, chunk, 256, iters=3)
    opq = O.PqOracle(distance, dim, chunk, cen)
    codes = opq.encode(rows)
    quant = qa.ProductQuantizer(dim, _dist(qa, distance), chunk, cen, lut_mfma=False)
    qa.set_option("hnsw_pq_table_build", 1 if tables else 0)
    try:
        seq = qa.GraphLayers.build(qa.EncodedVectorsPQ(codes, quant), m=m, ef_construct=efc, seed=seed, max_batch=1, original=vs)
    finally:
        qa.set_option("hnsw_pq_table_build", -1)
    ref = O.Hnsw.build_pq(st, opq, m=m, ef_construct=efc, seed=seed)
    _same_graph(seq.export_plain(), ref.export_plain())


def test_pq_pair_table_is_score_internal(qa):
    """The tabulated chunk distances give EncodedVectorsPQ::score_internal bit for bit (the build's stored <-> stored score)."""
    n, dim, chunk = 600, 40, 8
    rows = _clustered(n, dim, 5, k=16)
    cen = O.PqOracle.train(rows, dim, chunk, 64, iters=3)
    for distance in (O.DOT, O.EUCLID, O.MANHATTAN):
        opq = O.PqOracle(distance, dim, chunk, cen)
        codes = opq.encode(rows)
        enc = qa.EncodedVectorsPQ(codes, qa.ProductQuantizer(dim, _dist(qa, distance), chunk, cen))
        # qmx_score_internal reads the same table the build's HopPQInternal reads (and, with the option, recomputes the centroid distances)
        a = np.arange(0, 500, dtype=np.uint32)
        b = ((a * 7 + 3) % n).astype(np.uint32)
        from qdrant_amd import _ffi as F
        want = opq.score_internal(a.tolist(), b.tolist())
        for no_table in (0, 1):
            qa.set_option("no_pq_pair", no_table)
            try:
                out = np.zeros(len(a), dtype=np.float32)
                F.check(F.lib().qmx_score_internal(enc._h, F.ptr(a), F.ptr(b), len(a), F.ptr(out)))
            finally:
                qa.set_option("no_pq_pair", -1)
            assert np.array_equal(out.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("scalar_order", [False, True])
@pytest.mark.parametrize("dim", [24, 100])        # 24: SSE leaf (one lane per row, both norms computed per pair); 100: AVX2 leaf + remainder
def test_u8_cosine_build(qa, scalar_order, dim):
    """Metric<u8> with the per-pair cosine (metric_uint/avx2/cosine.rs, simple_cosine.rs): a stored row as the query takes its norm from
    the per-row norm column.  The oracle walks the device-built graph with identical score bits; recall against exact u8 search."""
    n, m, efc = 5000, 8, 64
    rows = np.clip(_clustered(n, dim, 41) * 20.0 + 128.0, 0, 255).astype(np.uint8)
    st = O.DenseStorage(O.U8, O.COSINE, rows, u8_isa=O.ISA_SCALAR if scalar_order else O.ISA_AUTO)
    vs = qa.VectorStorage(rows, qa.Distance.Cosine, qa.VectorStorageDatatype.Uint8, flags=qa._ffi.SEG_U8_SCALAR_ORDER if scalar_order else 0)
    g = qa.GraphLayers.build(vs, m=m, ef_construct=efc, seed=9)
    p = g.export_plain()
    _check_invariants(p, n, m)
    queries = np.clip(_clustered(80, dim, 42) * 20.0 + 128.0, 0, 255).astype(np.float32)
    exact = st.peek_top(queries, 10)
    got = g.search(10, 64, qa.new_raw_scorer(queries, vs))
    assert _recall(got, exact) > 0.6
    walk = O.Hnsw.from_plain(p, n)
    want = walk.search_dense(st, queries[:30], 10, 64)
    for gq, wq in zip(got[:30], want):
        assert np.array_equal(gq["score"].view(np.uint32), wq["score"].view(np.uint32))
    # the graph is as good as the one the oracle builds sequentially with the same per-pair cosine
    cpu = O.Hnsw(st, m=m, ef_construct=efc, seed=9)
    r_cpu = _recall(qa.GraphLayers.from_plain(cpu.export_plain()).search(10, 64, qa.new_raw_scorer(queries, vs)), exact)
    assert _recall(got, exact) > r_cpu - 0.03


def test_device_build_quality_at_1m_matches_the_cpu_build(qa):
    """at 1 M rows a build batch holds 16 384 concurrent insertions (6 000-point tests never get past 190), so the
    device-built graph is compared with the graph the ORACLE's parallel builder (GraphLayersBuilder restated, 8 threads, 489 s) built
    over the same rows: tests/golden/hnsw_quality_cpu_1m_d768.json, written by tools/hnsw_quality_cpu.py in the build container.  The rows
    and queries are regenerated here bit for bit (qmx_synth_fill_latent_f32 == qo_synth_fill_latent_f32, checked below on a sample), the
    walk is the walk the oracle reproduces exactly (test_gpu_hnsw.py), so recall@10 vs ef is comparable number by number."""
    import json
    import os
    import torch
    from qdrant_amd import _ffi as F
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hnsw_quality_cpu_1m_d768.json")))
    n, dim, nq, seed, K, noise = gold["rows"], gold["dim"], gold["nq"], gold["seed"], gold["latent_dim"], gold["noise"]
    lib, dev = F.lib(), torch.device("cuda", 0)
    rows = torch.empty((n, dim), dtype=torch.float32, device=dev)
    F.check(lib.qmx_synth_fill_latent_f32(0, seed, 0, n, dim, K, noise, F.ptr(rows)))
    assert np.array_equal(rows[123456:123460].cpu().numpy().view(np.uint32), O.synth_latent(seed, 123456, 4, dim, K, noise).view(np.uint32))
    F.check(lib.qmx_preprocess_f32(0, int(qa.Distance.Cosine), F.ptr(rows), n, dim, F.ptr(rows)))
    queries = torch.empty((nq, dim), dtype=torch.float32, device=dev)
    F.check(lib.qmx_synth_fill_latent_f32(0, seed, 1 << 40, nq, dim, K, noise, F.ptr(queries)))
    F.check(lib.qmx_preprocess_f32(0, int(qa.Distance.Cosine), F.ptr(queries), nq, dim, F.ptr(queries)))
    vs = qa.VectorStorage(rows, qa.Distance.Cosine)
    exact = qa.BatchFilteredSearcher(queries.cpu().numpy(), vs, 10).peek_top_all()
    g = qa.GraphLayers.build(vs, m=gold["m"], ef_construct=gold["ef_construct"], seed=42)
    scorer = qa.new_raw_scorer(queries, vs)
    for ef in (64, 128, 256):
        res, scored = g.search(10, ef, scorer, with_scored=True)
        rec = _recall(res, exact)
        cpu = gold["recall_vs_ef"][str(ef)]
        assert rec >= cpu["recall_at_10"] - 0.02, (ef, rec, cpu)
        # the walks do the same amount of work on both graphs (same degree distribution): within 3 %
        assert abs(scored / nq - cpu["points_scored_per_query"]) <= 0.03 * cpu["points_scored_per_query"], (ef, scored / nq, cpu)

"""Pins oracle/ (the CPU restatement of the reference) against every known-answer vector the
reference's own unit tests hold for the scoring path.  CPU only."""
import json
import math
import os

import numpy as np
import pytest

import oracle_ffi as O

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))


def _bits(x):
    return np.float32(x).view(np.uint32)


# lib/segment/src/spaces/simple_avx.rs:215-257 `test_spaces_avx`: assert_eq!(simd, scalar), exact
@pytest.mark.parametrize("key,isa", [("f32_avx", O.ISA_AVX), ("f32_sse", O.ISA_SSE)])
def test_f32_simd_equals_scalar(key, isa):
    v1, v2 = O.f32(G[key]["v1"]), O.f32(G[key]["v2"])
    for dist in (O.EUCLID, O.MANHATTAN, O.DOT):
        simd = O.similarity(O.F32, dist, v1, v2, isa)
        scalar = O.similarity(O.F32, dist, v1, v2, O.ISA_SCALAR)
        assert _bits(simd) == _bits(scalar), (key, dist)
    a = O.preprocess(O.COSINE, v1, isa)
    b = O.preprocess(O.COSINE, v1, O.ISA_SCALAR)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_f32_values_against_integer_arithmetic():
    # the literals are small integers, so every partial sum is exact: scores are known exactly
    v1, v2 = np.array(G["f32_avx"]["v1"]), np.array(G["f32_avx"]["v2"])
    assert O.similarity(O.F32, O.DOT, v1, v2) == float(np.dot(v1, v2))
    assert O.similarity(O.F32, O.EUCLID, v1, v2) == -float(((v1 - v2) ** 2).sum())
    assert O.similarity(O.F32, O.MANHATTAN, v1, v2) == -float(np.abs(v1 - v2).sum())


# lib/segment/src/spaces/simple.rs:247-251 and :255-277
def test_cosine_preprocessing_zero_and_stable():
    ka = G["known_answers"]["cosine_zero"]
    assert O.preprocess(O.COSINE, ka["input"]).tolist() == ka["expected"]
    rng = np.random.default_rng(7)
    for _ in range(100):
        lo, hi = rng.uniform(-2.5, 0.0), rng.uniform(0.0, 2.5)
        v = rng.uniform(lo, hi, 1500).astype(np.float32)
        p1 = O.preprocess(O.COSINE, v)
        p2 = O.preprocess(O.COSINE, p1)
        assert np.array_equal(p1.view(np.uint32), p2.view(np.uint32)), "renormalization is not stable"


# lib/segment/src/spaces/metric_f16/avx/{dot,euclid,manhattan}.rs tests: rel < 5e-4 vs scalar
@pytest.mark.parametrize("key,dist", [("f16_avx_dot", O.DOT), ("f16_avx_euclid", O.EUCLID), ("f16_avx_manhattan", O.MANHATTAN)])
def test_f16_simd_close_to_scalar(key, dist):
    v1, v2 = O.to_f16(G[key]["v1_f32"]), O.to_f16(G[key]["v2_f32"])
    scalar = O.similarity(O.F16, dist, v1, v2, O.ISA_SCALAR)
    for isa in (O.ISA_AVX, O.ISA_SSE):
        simd = O.similarity(O.F16, dist, v1, v2, isa)
        assert abs(simd - scalar) / abs(scalar) < 0.0005


def test_f16_conversions_match_numpy_ieee():
    # half 2.7.1 f16::from_f32 / to_f32 are IEEE binary16 RNE (unpinned by the reference's tests:
    # pinned here exhaustively against numpy's IEEE implementation)
    allh = np.arange(65536, dtype=np.uint16)
    f = O.f16_to_f32(allh)
    ref = allh.view(np.float16).astype(np.float32)
    same = (f.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(f) & np.isnan(ref))
    assert same.all()
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * 10,
                        rng.uniform(-7e4, 7e4, 50000).astype(np.float32),
                        np.float32([0.0, -0.0, 65504.0, 65520.0, 1e-8, 6e-8, 5.96e-8, np.inf, -np.inf])])
    h = O.to_f16(x)
    with np.errstate(over="ignore"):
        assert np.array_equal(h, x.astype(np.float16).view(np.uint16))


# lib/segment/src/spaces/metric_uint/avx2/*.rs tests: assert_eq!(simd, scalar)
@pytest.mark.parametrize("key,dist", [("u8_avx2_dot", O.DOT), ("u8_avx2_cosine", O.COSINE),
                                      ("u8_avx2_euclid", O.EUCLID), ("u8_avx2_manhattan", O.MANHATTAN)])
def test_u8_simd_equals_scalar(key, dist):
    v1, v2 = np.array(G[key]["v1"], dtype=np.uint8), np.array(G[key]["v2"], dtype=np.uint8)
    scalar = O.similarity(O.U8, dist, v1, v2, O.ISA_SCALAR)
    for isa in (O.ISA_AVX, O.ISA_SSE):
        assert _bits(O.similarity(O.U8, dist, v1, v2, isa)) == _bits(scalar)
    a, b = v1.astype(np.int64), v2.astype(np.int64)
    exact = {O.DOT: float((a * b).sum()), O.EUCLID: -float(((a - b) ** 2).sum()), O.MANHATTAN: -float(np.abs(a - b).sum())}
    if dist in exact:
        assert scalar == exact[dist]


# metric_uint/simple_cosine.rs:79-86 `test_zero`, avx2/cosine.rs:143-160
def test_u8_cosine_zero_vector():
    ka = G["known_answers"]["u8_cosine_zero"]
    v1, v2 = np.array(ka["v1"], dtype=np.uint8), np.array(ka["v2"], dtype=np.uint8)
    for isa in (O.ISA_SCALAR, O.ISA_SSE, O.ISA_AVX):
        for a, b in ((v1, v2), (v2, v1), (v1, v1)):
            assert O.similarity(O.U8, O.COSINE, a, b, isa) == ka["expected"]
    # >= 32 elements so the AVX2 body runs
    z, w = np.zeros(64, dtype=np.uint8), np.arange(64, dtype=np.uint8)
    assert O.similarity(O.U8, O.COSINE, z, w, O.ISA_AVX) == 0.0


# metric_uint/simple_euclid.rs `test_conversion_to_bytes`
def test_conversion_to_bytes():
    ka = G["known_answers"]["f32_to_u8"]
    assert O.to_u8(ka["input"]).tolist() == ka["expected"]
    assert O.to_u8([np.nan, 254.9, 255.5, -0.5, 1e9]).tolist() == [0, 254, 255, 0, 255]


# lib/segment/src/spaces/tools.rs:59-76
def test_peek_top():
    ka = G["known_answers"]["peek_top_scores"]
    data = ka["data"]
    res = O.topk_push_all([(i, v) for i, v in enumerate(data)], ka["top"])
    assert res["score"].tolist() == ka["largest"]
    res = O.topk_push_all([(i, -v) for i, v in enumerate(data)], ka["top"])  # Reverse<E>
    assert (-res["score"]).tolist() == ka["smallest"]


def test_fixed_length_priority_queue_semantics():
    # fixed_length_priority_queue.rs:47-59: replaces the root only on strict `<`; NaN is greatest
    # (OrderedFloat, common/src/types.rs:21-25); into_sorted_vec is descending (:63-65)
    rng = np.random.default_rng(11)
    scores = rng.standard_normal(1000).astype(np.float32)
    res = O.topk_push_all(enumerate(scores), 10)
    assert np.array_equal(res["score"], np.sort(scores)[::-1][:10])
    res = O.topk_push_all([(0, 1.0), (1, float("nan")), (2, 5.0), (3, 2.0)], 2)
    assert math.isnan(res["score"][0]) and res["score"][1] == 5.0
    # ties at the boundary: a later equal score never evicts an earlier one
    res = O.topk_push_all([(0, 1.0), (1, 1.0), (2, 1.0), (3, 0.5)], 2)
    assert sorted(res["idx"].tolist()) == [0, 1]
    # fewer elements than `length`
    res = O.topk_push_all([(5, 3.0), (6, 4.0)], 10)
    assert res["idx"].tolist() == [6, 5]


def test_peek_top_iter_chunking_and_deleted():
    rng = np.random.default_rng(5)
    n, dim = 1000, 48
    rows = O.preprocess(O.COSINE, rng.standard_normal((n, dim)).astype(np.float32))
    q = rng.standard_normal((3, dim)).astype(np.float32)
    pdel = np.zeros(n, dtype=bool)
    pdel[::7] = True
    vdel = np.zeros(n, dtype=bool)
    vdel[5::11] = True
    st = O.DenseStorage(O.F32, O.COSINE, rows, pdel, vdel)
    got = st.peek_top(q, 10)
    qe = st.encode_queries(q)
    live = ~(pdel | vdel)
    for i in range(3):
        s = np.array([O.similarity(O.F32, O.COSINE, qe[i], rows[j]) for j in range(n)], dtype=np.float32)
        s[~live] = -np.inf
        order = np.argsort(-s, kind="stable")[:10]
        assert got[i]["idx"].tolist() == order.tolist()
        assert np.array_equal(got[i]["score"], s[order])
    # explicit candidate list, fewer than `top` survivors
    got = st.peek_top(q, 10, ids=[0, 1, 2, 3, 5])
    assert all(set(g["idx"].tolist()) == {1, 2, 3} for g in got)  # 0 point-deleted, 5 vector-deleted
    # parallel scan (CPU baseline helper) returns the same sets
    par = st.peek_top(q, 10, threads=4)
    full = st.peek_top(q, 10)
    for a, b in zip(par, full):
        assert np.array_equal(a["score"], b["score"]) and set(a["idx"]) == set(b["idx"])


# the reference's own C kernels (lib/quantization/cpp/avx2.c, sse.c) compiled into oracle/_ref
def test_sq_leaves_match_reference_c_kernels():
    ref = O.load_ref_quant()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(42)
    import ctypes as C
    for dim in (16, 32, 48, 64, 128, 768, 784, 1024, 1536, 2064):
        for _ in range(20):
            q = rng.integers(0, 128, dim, dtype=np.uint8)
            v = rng.integers(0, 128, dim, dtype=np.uint8)
            pq, pv = C.c_void_p(q.ctypes.data), C.c_void_p(v.ctypes.data)
            assert _bits(O.lib.qo_sq_dot_avx(pq, pv, dim)) == _bits(ref.impl_score_dot_avx(pq, pv, dim))
            assert _bits(O.lib.qo_sq_l1_avx(pq, pv, dim)) == _bits(ref.impl_score_l1_avx(pq, pv, dim))
            assert _bits(O.lib.qo_sq_dot_sse(pq, pv, dim)) == _bits(ref.impl_score_dot_sse(pq, pv, dim))
            if dim * 127 < 65536:  # sse.c's 16-bit horizontal add wraps above this (restated as-is)
                assert _bits(O.lib.qo_sq_l1_sse(pq, pv, dim)) == _bits(ref.impl_score_l1_sse(pq, pv, dim))
            if dim <= 1040:  # exact-integer regime: every leaf equals the i32 sum
                exact = float((q.astype(np.int64) * v.astype(np.int64)).sum())
                assert O.lib.qo_sq_dot_avx(pq, pv, dim) == exact
    # worst-case codes
    q = np.full(2064, 127, dtype=np.uint8)
    pq = C.c_void_p(q.ctypes.data)
    assert _bits(O.lib.qo_sq_dot_avx(pq, pq, 2064)) == _bits(ref.impl_score_dot_avx(pq, pq, 2064))


def test_synth_generator_is_counter_based():
    a = O.synth(0x5EED0002, 0, 64, 96)
    b = O.synth(0x5EED0002, 32, 32, 96)
    assert np.array_equal(a[32:], b)
    assert abs(float(a.mean())) < 0.05 and 0.9 < float(a.std()) < 1.1


def test_pq_lut_entry_is_the_sequential_chunk_sum_the_table_free_kernels_recompute():
    """pq.hip's HopPQDirect / HopPQInternalDirect recompute a LUT entry (and a centroid-pair term) instead of gathering it: entry(c, k) = the chunk's terms
    added one by one in f32 from -0.0, a multiply and an add per coordinate (no fma) - `encode_query` (encoded_vectors_pq.rs:519-541).  The restatement of
    that claim against the oracle's LUT and its score_internal, bit for bit, for the three term kinds."""
    import oracle_ffi as O
    rng = np.random.default_rng(5)
    dim, chunk, n = 48, 8, 300
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    cen = O.PqOracle.train(rows, dim, chunk, 256, iters=2)
    m = dim // chunk

    def chain(a, b, kind):
        s = np.float32(-0.0)
        for x, y in zip(a, b):
            t = np.float32(x) * np.float32(y) if kind == 0 else (np.float32(abs(np.float32(x) - np.float32(y))) if kind == 1 else
                                                               np.float32((np.float32(x) - np.float32(y)) * (np.float32(x) - np.float32(y))))
            s = np.float32(s + t)
        return s
    for dist, kind in ((O.DOT, 0), (O.MANHATTAN, 1), (O.EUCLID, 2)):
        opq = O.PqOracle(dist, dim, chunk, cen)
        codes = opq.encode(rows)
        opq.codes = codes
        q = O.preprocess(dist, rng.standard_normal((1, dim)).astype(np.float32))[0]
        lut = np.asarray(opq.lut(q), dtype=np.float32).reshape(m, 256)
        sign = np.float32(-1.0) if dist in (O.EUCLID, O.MANHATTAN) else np.float32(1.0)      # `invert`: the segment's choice for distances (quantized_vectors.rs:232)
        for c in (0, 3, m - 1):
            for k in (0, 17, 255):
                want = chain(q[c * chunk:(c + 1) * chunk], cen[k, c * chunk:(c + 1) * chunk], kind)
                assert np.float32(sign * want).view(np.uint32) == lut[c, k].view(np.uint32), (dist, c, k)
        # score_internal(a, b) = (+/-) the chain over chunks (from -0.0) of the chains over the two rows' centroids
        a, b = 7, 123
        s = np.float32(-0.0)
        for c in range(m):
            s = np.float32(s + chain(cen[codes[a, c], c * chunk:(c + 1) * chunk], cen[codes[b, c], c * chunk:(c + 1) * chunk], kind))
        got = opq.score_internal(np.array([a], dtype=np.uint32), np.array([b], dtype=np.uint32))[0]
        assert np.float32(sign * s).view(np.uint32) == np.float32(got).view(np.uint32), dist

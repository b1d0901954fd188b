"""CPU: pins the oracle's EncodedVectorsBin<u128> restatement (oracle/qdrant_oracle.c, BQ block).

  * xor-popcount against the reference's own C kernel `impl_xor_popcnt_sse_uint128` (lib/quantization/cpp/sse.c:54-75,
    compiled into oracle/_ref by oracle/Makefile) and against a numpy bit count
  * the reference's test vectors (lib/quantization/tests/integration/test_binary.rs:15-22: every coordinate +-1): there
    the BQ score IS the dot product (test_binary_dot_impl :36-74 allows dim * 0.01; it is exact), inverted pairs negate it
    (:88-127), and sorting by the L1 pairings reproduces the order of the true L1 distances (:238-292, :305-359)
  * storage sizes of get_quantized_vector_size_from_params::<u128> and the bit layout of a little-endian u128
"""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O

DIMS = [1, 8, 33, 65, 127, 128, 129, 3 * 129, 768, 1536]     # test_binary.rs uses 1, 8, 33, 65, 3 * 129


def _pm1(rng, n, dim):
    v = np.sign(rng.uniform(-1.0, 1.0, (n, dim))).astype(np.float32)
    v[v == 0] = 1.0
    return v


def test_row_bytes_and_bit_layout():
    assert [O.lib.qo_bq_row_bytes(d) for d in (0, 1, 127, 128, 129, 768, 1536, 3 * 129)] == [16, 16, 16, 16, 32, 96, 192, 64]
    v = np.zeros(200, dtype=np.float32)
    v[[0, 7, 8, 127, 128, 199]] = 0.5
    v[3] = -0.5
    v[5] = 0.0            # > 0.0 only
    v[9] = np.nan         # NaN > 0 is false
    row = O.BqOracle(O.DOT, 200).encode(v)[0]
    want = np.zeros(32, dtype=np.uint8)
    for i in (0, 7, 8, 127, 128, 199):
        want[i // 8] |= 1 << (i % 8)
    assert np.array_equal(row, want)
    as_u128 = [int.from_bytes(row[16 * w:16 * w + 16].tobytes(), "little") for w in range(2)]
    assert as_u128[0] == (1 << 0) | (1 << 7) | (1 << 8) | (1 << 127) and as_u128[1] == (1 << 0) | (1 << 71)


def test_xor_popcnt_matches_the_reference_c_kernel():
    ref = O.load_ref_quant()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    fn = ref.impl_xor_popcnt_sse_uint128
    fn.restype, fn.argtypes = C.c_uint32, [C.c_void_p, C.c_void_p, C.c_uint32]
    rng = np.random.default_rng(5)
    for words in (1, 2, 3, 6, 12, 13, 64):
        for _ in range(20):
            q = rng.integers(0, 256, words * 16, dtype=np.uint8)
            v = rng.integers(0, 256, words * 16, dtype=np.uint8)
            want = int(fn(q.ctypes.data, v.ctypes.data, words))
            assert int(O.lib.qo_bq_xor_popcnt(q.ctypes.data, v.ctypes.data, words)) == want
            assert int(np.unpackbits(q ^ v).sum()) == want


@pytest.mark.parametrize("dim", DIMS)
def test_plus_minus_one_vectors_like_the_reference_tests(dim):
    rng = np.random.default_rng(42 + dim)
    vecs, query = _pm1(rng, 128, dim), _pm1(rng, 1, dim)
    dot = (vecs.astype(np.float64) @ query[0].astype(np.float64)).astype(np.float32)
    l1 = np.abs(vecs - query[0]).sum(axis=1)
    for distance, invert, want in [(O.DOT, 0, dot), (O.DOT, 1, -dot), (O.COSINE, 0, dot), (O.EUCLID, 1, dot), (O.MANHATTAN, 1, dot), (O.MANHATTAN, 0, -dot)]:
        bq = O.BqOracle(distance, dim, invert=invert)
        bq.encode_rows(vecs)
        got = bq.score_points(query, np.arange(128))[0]
        assert np.array_equal(got, want)
        if distance == O.MANHATTAN and not invert:       # test_binary_l1_impl: ascending BQ score == ascending true L1 (2 * xor)
            assert np.array_equal(got, l1 - dim)
        # score_internal (:892-917) == scoring the stored row as the query
        a, b = np.arange(0, 64), np.arange(64, 128)
        assert np.array_equal(bq.score_internal(a, b), bq.score_points(vecs[:64], b)[np.arange(64), np.arange(64)])


def test_default_invert_is_the_segments_choice():
    assert [O.BqOracle(d, 8).invert for d in (O.COSINE, O.DOT, O.EUCLID, O.MANHATTAN)] == [0, 0, 1, 1]   # quantized_vectors.rs:232


# ---- QueryEncoding::Scalar4bits / Scalar8bits (encoded_vectors_binary.rs:49-54, 692-756, 337-409, 783-810) ---------------------------
# The reference's integration tests only run SameAsStorage; what pins the scalar encodings is the reference's own C kernels
# (impl_xor_popcnt_scalar{4,8}_{avx,sse}_uint128, lib/quantization/cpp/avx2.c / sse.c, compiled into oracle/_ref) for the
# plane-weighted popcount, and identities of the encoder that follow from its definition.  Encoder bytes: parity unpinned.

@pytest.mark.parametrize("bits", [4, 8])
def test_scalar_xor_popcnt_matches_the_reference_c_kernels(bits):
    ref = O.load_ref_quant()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    fns = [getattr(ref, "impl_xor_popcnt_scalar%d_%s_uint128" % (bits, isa)) for isa in ("avx", "sse")]
    for fn in fns:
        fn.restype, fn.argtypes = C.c_uint32, [C.c_void_p, C.c_void_p, C.c_uint32]
    rng = np.random.default_rng(bits)
    for words in (1, 2, 3, 6, 7, 12, 13, 64):
        for _ in range(20):
            v = rng.integers(0, 256, words * 16, dtype=np.uint8)
            q = rng.integers(0, 256, words * 16 * bits, dtype=np.uint8)
            got = int(O.lib.qo_bq_xor_popcnt_scalar(v.ctypes.data, q.ctypes.data, words, bits))
            for fn in fns:
                assert int(fn(q.ctypes.data, v.ctypes.data, words)) == got
            planes = q.reshape(words, bits, 16)
            want = sum(int(np.unpackbits(planes[w, b] ^ v[16 * w:16 * w + 16]).sum()) << b for w in range(words) for b in range(bits))
            assert got == want


@pytest.mark.parametrize("encoding", [0, 1, 2])
@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("dim", [1, 8, 33, 127, 128, 129, 3 * 129])
def test_scalar_query_encoder_identities(dim, bits, encoding):
    rng = np.random.default_rng(dim + bits + encoding)
    q = rng.standard_normal(dim).astype(np.float32)
    bq = O.BqOracle(O.DOT, dim, encoding=encoding, mean=np.zeros(dim, np.float32), stddev=np.ones(dim, np.float32))
    enc = bq.encode_scalar_queries(q, bits)[0]
    ext = {0: dim, 1: 2 * dim, 2: dim + (dim + 1) // 2}[encoding]
    assert len(enc) == -(-ext // 128) * bits * 16 == bq.row_bytes * bits
    # de-interleave the planes: value i = sum_b plane_b[i] << b must be round((x + max_abs) / (2 max_abs / (2^bits - 1)))
    planes = np.unpackbits(enc.reshape(-1, bits, 16), axis=2, bitorder="little")           # [chunk, b, 128]
    vals = (planes.astype(np.int64) << np.arange(bits)[None, :, None]).sum(axis=1).reshape(-1)
    x = {0: q, 1: np.concatenate([q, q]),
         2: np.concatenate([q, np.array([max(q[2 * k], q[2 * k + 1]) if 2 * k + 1 < dim else q[2 * k] for k in range((dim + 1) // 2)], np.float32)])}[encoding]
    m = np.float32(np.abs(x).max())
    delta = (m - (-m)) / np.float32(2 ** bits - 1)
    want = np.floor((x - (-m)) / delta + np.float32(0.5)).astype(np.int64) % (2 ** bits)    # round half away (values >= 0)
    assert vals[:ext].tolist() == want.tolist() and not vals[ext:].any()
    assert vals[int(np.argmax(x))] == 2 ** bits - 1 or vals[int(np.argmin(x))] == 0             # an extreme value sits on the range's end


@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("dim", [1, 33, 3 * 129])
def test_scalar_encoded_plus_minus_one_query_scores_like_the_one_bit_query(dim, bits):
    """+-1 queries quantise to 0 / 2^bits - 1: every plane is the sign plane, xor_scalar = (2^bits - 1) * xor, same score."""
    rng = np.random.default_rng(dim)
    vecs, query = _pm1(rng, 128, dim), _pm1(rng, 1, dim)
    for distance, invert in [(O.DOT, 0), (O.DOT, 1), (O.EUCLID, 1), (O.MANHATTAN, 0)]:
        bq = O.BqOracle(distance, dim, invert=invert)
        bq.encode_rows(vecs)
        assert np.array_equal(bq.score_points_scalar(query, np.arange(128), bits), bq.score_points(query, np.arange(128)))
    zero = O.BqOracle(O.DOT, dim)
    zero.encode_rows(vecs)
    z = zero.encode_scalar_queries(np.zeros((1, dim), np.float32), bits)[0]                 # delta = 0: every value quantises to 0
    assert not z.any()

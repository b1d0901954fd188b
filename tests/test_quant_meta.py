"""quantized.meta.json (lib/segment/src/vector_storage/quantized/quantized_vectors/config.rs:13): the library's host-side
reader qmx_quant_meta_parse against documents shaped like serde_json's output for `MetadataInt8`
(lib/quantization/src/encoded_vectors_u8.rs:84-91), PQ `Metadata` (encoded_vectors_pq.rs:46-51), BQ `Metadata`
(encoded_vectors_binary.rs:112-125) and `VectorParameters` (encoded_vectors.rs:28-39).  Needs no device.

The reference holds no literal metadata document; what pins the format are the struct definitions above (field names,
`#[serde(untagged)]`, `#[serde(default)]` / `skip_serializing_if`, `rename = "count"`), restated by the writer below, and
serde_json's number rule (shortest round-trip text; parsed through f64 and narrowed) which the value checks cover bit for bit."""
import ctypes as C
import json

import numpy as np
import pytest

from qdrant_amd import _ffi as F


def _f32_text(x, rng=None):
    """Shortest text that round-trips the f32 (what ryu prints), in positional or exponent form."""
    x = np.float32(x)
    if x == 0:
        return "-0.0" if np.signbit(x) else "0.0"
    a = abs(float(x))
    if 1e-5 <= a < 1e16:
        s = np.format_float_positional(x, unique=True, trim="0")
    else:
        s = np.format_float_scientific(x, unique=True, trim="0").replace("e+", "e").replace("e-0", "e-").replace("e0", "e")
        if s.endswith(".0"):
            s = s[:-2]
        s = s.replace(".0e", "e")
    assert np.float32(float(s)) == x
    return s


def _vp(dim, distance_type, invert, count=None):
    s = '{"dim":%d,"distance_type":"%s","invert":%s' % (dim, distance_type, "true" if invert else "false")
    if count is not None:
        s += ',"count":%d' % count
    return s + "}"


def _parse(dtype, text):
    data = text.encode() if isinstance(text, str) else text
    m = F.QuantMeta()
    rc = F.lib().qmx_quant_meta_parse(dtype, data, len(data), C.byref(m))
    return rc, m


def _free(m):
    F.lib().qmx_quant_meta_free(C.byref(m))


def _bits(x):
    return np.float32(x).view(np.uint32)


@pytest.mark.parametrize("distance_type,distance,invert", [("Dot", F.DOT, False), ("Cosine", F.COSINE, False), ("L2", F.EUCLID, True),
                                                             ("L1", F.MANHATTAN, True)])
def test_sq_metadata_round_trips_bit_for_bit(distance_type, distance, invert):
    rng = np.random.default_rng(1)
    for trial in range(200):
        dim = int(rng.integers(1, 2000))
        actual = (dim + 15) // 16 * 16
        vals = rng.standard_normal(3).astype(np.float32) * np.float32(10.0) ** rng.integers(-12, 12, 3).astype(np.float32)
        alpha, offset, mult = vals
        doc = '{"actual_dim":%d,"alpha":%s,"offset":%s,"multiplier":%s,"vector_parameters":%s}' % (
            actual, _f32_text(alpha), _f32_text(offset), _f32_text(mult), _vp(dim, distance_type, invert, 12345 if trial % 2 else None))
        rc, m = _parse(F.DTYPE_SQ_U8, doc)
        assert rc == F.OK, F.last_error()
        assert (m.dtype, m.dim, m.distance, m.invert) == (F.DTYPE_SQ_U8, dim, distance, int(invert))
        assert (m.has_deprecated_count, m.deprecated_count) == ((1, 12345) if trial % 2 else (0, 0))
        assert m.sq.actual_dim == actual and m.sq.invert == int(invert)
        assert (_bits(m.sq.alpha), _bits(m.sq.offset), _bits(m.sq.multiplier)) == (_bits(alpha), _bits(offset), _bits(mult))
        _free(m)


def test_numbers_go_through_f64_like_serde_json():
    # a decimal with more digits than f32 holds: serde_json -> f64 -> `as f32`
    for text in ["0.1", "1e-45", "3.4028235e38", "1.00000001", "16777217", "0.30000001192092896", "-2.5E+3", "7e0", "123456789012345678901234567890"]:
        doc = '{"actual_dim":16,"alpha":%s,"offset":0,"multiplier":-1,"vector_parameters":%s}' % (text, _vp(16, "Dot", False))
        rc, m = _parse(F.DTYPE_SQ_U8, doc)
        assert rc == F.OK, (text, F.last_error())
        assert _bits(m.sq.alpha) == _bits(np.float32(np.float64(text))) and m.sq.offset == 0.0 and m.sq.multiplier == -1.0
        _free(m)


def test_pq_metadata():
    rng = np.random.default_rng(2)
    for dim, chunk, k in [(32, 4, 256), (30, 4, 256), (7, 2, 5), (1536, 16, 256)]:
        cent = (rng.standard_normal((k, dim)) * 3).astype(np.float32)
        division = [(s, min(s + chunk, dim)) for s in range(0, dim, chunk)]
        doc = ('{"centroids":[' + ",".join("[" + ",".join(_f32_text(v) for v in row) + "]" for row in cent) + '],"vector_division":['
               + ",".join('{"start":%d,"end":%d}' % d for d in division) + '],"vector_parameters":' + _vp(dim, "L2", True) + "}")
        rc, m = _parse(F.DTYPE_PQ, doc)
        assert rc == F.OK, F.last_error()
        assert (m.dim, m.distance, m.invert, m.pq.chunk_size, m.pq.n_centroids, m.pq.invert, m.pq.lut_mfma) == (dim, F.EUCLID, 1, chunk, k, 1, 0)
        got = np.ctypeslib.as_array(C.cast(m.pq.centroids, C.POINTER(C.c_float)), (k, dim))
        assert np.array_equal(got.view(np.uint32), cent.view(np.uint32))
        _free(m)
    # a division the reference never writes
    doc = '{"centroids":[[0,0,0,0]],"vector_division":[{"start":0,"end":1},{"start":1,"end":4}],"vector_parameters":%s}' % _vp(4, "Dot", False)
    assert _parse(F.DTYPE_PQ, doc)[0] == F.ERR_NOT_SUPPORTED
    doc = '{"centroids":[[0,0,0]],"vector_division":[{"start":0,"end":4}],"vector_parameters":%s}' % _vp(4, "Dot", False)
    assert _parse(F.DTYPE_PQ, doc)[0] == F.ERR_BAD_ARG      # centroid of the wrong length


def test_bq_metadata_defaults_and_stats():
    rc, m = _parse(F.DTYPE_BQ, '{"vector_parameters":%s}' % _vp(100, "Dot", False))      # OneBit / SameAsStorage are skipped when default
    assert rc == F.OK and (m.bq.encoding, m.bq_query_encoding, m.bq.mean, m.bq.stddev) == (F.BQ_ONE_BIT, 0, None, None)
    _free(m)
    rng = np.random.default_rng(3)
    dim = 77
    mean, std = rng.standard_normal(dim).astype(np.float32), rng.random(dim).astype(np.float32)
    stats = ",".join('{"min":%s,"max":%s,"mean":%s,"stddev":%s}' % (_f32_text(a - 3), _f32_text(a + 3), _f32_text(a), _f32_text(s))
                     for a, s in zip(mean, std))
    for enc, code in [("TwoBits", F.BQ_TWO_BITS), ("OneAndHalfBits", F.BQ_ONE_AND_HALF_BITS)]:
        for qe, qcode in [("Scalar4bits", 1), ("Scalar8bits", 2)]:
            doc = ('{"vector_parameters":%s,"encoding":"%s","query_encoding":"%s","vector_stats":{"elements_stats":[%s]}}'
                   % (_vp(dim, "L1", True), enc, qe, stats))
            rc, m = _parse(F.DTYPE_BQ, doc)
            assert rc == F.OK, F.last_error()
            assert (m.bq.encoding, m.bq_query_encoding, m.distance, m.invert) == (code, qcode, F.MANHATTAN, 1)
            gm = np.ctypeslib.as_array(C.cast(m.bq.mean, C.POINTER(C.c_float)), (dim,))
            gs = np.ctypeslib.as_array(C.cast(m.bq.stddev, C.POINTER(C.c_float)), (dim,))
            assert np.array_equal(gm.view(np.uint32), mean.view(np.uint32)) and np.array_equal(gs.view(np.uint32), std.view(np.uint32))
            _free(m)


def test_whitespace_key_order_and_unknown_fields_are_tolerated_like_serde():
    doc = json.dumps({"vector_parameters": {"invert": False, "dim": 20, "distance_type": "Dot", "future": [1, {"a": None}]},
                      "multiplier": 0.5, "offset": -1.25, "alpha": 0.0078125, "actual_dim": 32, "note": "x\\u00e9\\n"}, indent=2)
    rc, m = _parse(F.DTYPE_SQ_U8, doc)
    assert rc == F.OK, F.last_error()
    assert (m.dim, m.sq.actual_dim, m.sq.alpha, m.sq.offset, m.sq.multiplier) == (20, 32, 0.0078125, -1.25, 0.5)
    _free(m)


@pytest.mark.parametrize("doc", [
    "", "{", "[]", "nul", '{"actual_dim":16}', '{"actual_dim":16,"alpha":1,"offset":0,"multiplier":1}',
    '{"actual_dim":16,"alpha":1,"offset":0,"multiplier":1,"vector_parameters":{"dim":16,"distance_type":"Hamming","invert":false}}',
    '{"actual_dim":16,"alpha":"1","offset":0,"multiplier":1,"vector_parameters":{"dim":16,"distance_type":"Dot","invert":false}}',
    '{"actual_dim":-16,"alpha":1,"offset":0,"multiplier":1,"vector_parameters":{"dim":16,"distance_type":"Dot","invert":false}}',
    '{"actual_dim":48,"alpha":1,"offset":0,"multiplier":1,"vector_parameters":{"dim":16,"distance_type":"Dot","invert":false}}',
    '{"actual_dim":16,"alpha":1,"offset":0,"multiplier":1,"vector_parameters":{"dim":16,"distance_type":"Dot","invert":0}}',
    '{"actual_dim":16,"alpha":1,"offset":0,"multiplier":1,"vector_parameters":{"dim":16,"distance_type":"Dot","invert":false}} x',
    '{"actual_dim":16,"alpha":01,"offset":0,"multiplier":1,"vector_parameters":{"dim":16,"distance_type":"Dot","invert":false}}',
    '{"actual_dim":16,"alpha":1.,"offset":0,"multiplier":1,"vector_parameters":{"dim":16,"distance_type":"Dot","invert":false}}',
    "[" * 100 + "]" * 100,
])
def test_malformed_metadata_is_refused(doc):
    rc, _ = _parse(F.DTYPE_SQ_U8, doc)
    assert rc == F.ERR_BAD_ARG


def test_only_quantized_dtypes_have_metadata():
    assert _parse(F.DTYPE_F32, "{}")[0] == F.ERR_BAD_ARG
    # truncations of a valid document never crash
    doc = '{"actual_dim":16,"alpha":1.5e-3,"offset":-0.25,"multiplier":1,"vector_parameters":{"dim":16,"distance_type":"Dot","invert":false}}'
    for i in range(len(doc)):
        assert _parse(F.DTYPE_SQ_U8, doc[:i])[0] == F.ERR_BAD_ARG
    assert _parse(F.DTYPE_SQ_U8, doc)[0] == F.OK


def test_load_quantizer_builds_the_parameter_structs_from_the_file():
    """qdrant_amd.load_quantizer: the file's values end up verbatim in what qmx_segment_create takes."""
    import qdrant_amd as qa
    doc = '{"actual_dim":32,"alpha":0.015625,"offset":-1.0,"multiplier":0.00048828125,"vector_parameters":%s}' % _vp(20, "L2", True)
    q = qa.load_quantizer(doc, F.DTYPE_SQ_U8)
    p = q.params()
    assert (q.dim, q.distance, p.actual_dim, p.alpha, p.offset, p.multiplier, p.invert) == (20, qa.Distance.Euclid, 32, 0.015625, -1.0,
                                                                                            0.00048828125, 1)
    assert q.multiplier == qa.ScalarQuantizer(20, qa.Distance.Euclid, 0.015625, -1.0).multiplier     # what encode() would have derived
    cent = np.arange(12, dtype=np.float32).reshape(2, 6)
    doc = ('{"centroids":%s,"vector_division":[{"start":0,"end":4},{"start":4,"end":6}],"vector_parameters":%s}'
           % (json.dumps(cent.tolist()), _vp(6, "Dot", False)))
    q = qa.load_quantizer(doc, F.DTYPE_PQ)
    assert (q.dim, q.chunk_size, q.m, q.n_centroids, q.invert) == (6, 4, 2, 2, False) and np.array_equal(q.centroids, cent)
    q = qa.load_quantizer('{"vector_parameters":%s,"encoding":"TwoBits"}' % _vp(9, "Dot", False), F.DTYPE_BQ)
    assert (q.dim, q.encoding, q.invert, q.mean) == (9, F.BQ_TWO_BITS, False, None)
    q = qa.load_quantizer('{"vector_parameters":%s,"query_encoding":"Scalar8bits"}' % _vp(9, "Dot", False), F.DTYPE_BQ)
    assert (q.query_encoding, q.params().query_encoding, q.encoding) == (F.BQ_QUERY_SCALAR_8BITS, F.BQ_QUERY_SCALAR_8BITS, F.BQ_ONE_BIT)


@pytest.mark.parametrize("bits_text,bits", [("bits4", 0), ("bits2", 1), ("bits1_5", 2), ("bits1", 3)])
def test_tq_metadata(bits_text, bits):
    """EncodedVectorsTQ `Metadata` (lib/quantization/src/encoded_vectors_tq.rs:33-46; TQBits / TQMode / TQRotation are
    `#[serde(rename_all = "snake_case")]`, turboquant/mod.rs:13-100; `rotation` has a serde default: Padded)."""
    for rotation, unp in ((None, 0), ("padded", 0), ("unpadded", 1)):
        text = '{"vector_parameters":%s,"bits":"%s","mode":"normal","error_correction":null%s}' % (
            _vp(768, "L2", True), bits_text, "" if rotation is None else ',"rotation":"%s"' % rotation)
        rc, m = _parse(F.DTYPE_TQ, text)
        assert rc == 0, F.last_error()
        assert (m.dim, m.distance, m.invert) == (768, F.EUCLID, 1)
        assert (m.tq.bits, m.tq.rotation_unpadded, m.tq.invert, m.tq.plus_mode) == (bits, unp, 1, 0)
        _free(m)
    # TQ+ documents parse (the flag says so); creating a segment from them is refused
    pd = {0: 6, 1: 8, 2: 8, 3: 8}[bits]                      # padded dim of dim = 5
    vals = [0.5 + 0.25 * i for i in range(pd)]
    text = '{"vector_parameters":%s,"bits":"%s","mode":"plus","error_correction":{"shift":%s,"scale":%s}}' % (
        _vp(5, "Dot", False), bits_text, json.dumps(vals), json.dumps([v + 1.0 for v in vals]))
    rc, m = _parse(F.DTYPE_TQ, text)
    assert rc == 0 and m.tq.plus_mode == 1, F.last_error()
    got = np.ctypeslib.as_array(C.cast(m.tq.ec_shift, C.POINTER(C.c_float)), (pd,)).copy()
    got2 = np.ctypeslib.as_array(C.cast(m.tq.ec_scale, C.POINTER(C.c_float)), (pd,)).copy()
    assert got.tolist() == vals and got2.tolist() == [v + 1.0 for v in vals]
    _free(m)
    rc, m = _parse(F.DTYPE_TQ, text.replace(json.dumps(vals), json.dumps(vals[:-1])))     # wrong length (new_error_correction_from_metadata :70-91)
    assert rc == F.ERR_BAD_ARG
    rc, m = _parse(F.DTYPE_TQ, '{"vector_parameters":%s,"bits":"%s","mode":"plus","error_correction":null}' % (_vp(5, "Dot", False), bits_text))
    assert rc == F.ERR_BAD_ARG
    for bad in ('{"vector_parameters":%s,"bits":"bits3","mode":"normal"}' % _vp(4, "Dot", False),
                '{"vector_parameters":%s,"bits":"%s"}' % (_vp(4, "Dot", False), bits_text),
                '{"vector_parameters":%s,"bits":"%s","mode":"normal","rotation":"sideways"}' % (_vp(4, "Dot", False), bits_text)):
        rc, m = _parse(F.DTYPE_TQ, bad)
        assert rc == F.ERR_BAD_ARG

"""ctypes wrapper of oracle/libqdrant_oracle.so — TEST INFRASTRUCTURE ONLY (checker, never product)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = C.CDLL(os.path.join(ROOT, "oracle", "libqdrant_oracle.so"))
REF_QUANT_PATH = os.path.join(ROOT, "oracle", "_ref", "libqdrant_ref_quant.so")

COSINE, EUCLID, DOT, MANHATTAN = range(4)
F32, F16, U8 = range(3)
ISA_AUTO, ISA_AVX, ISA_SSE, ISA_SCALAR = range(4)
ISA_REF = 100  # SQ leaves from the reference's own C kernels (oracle/_ref)

ScoredPointOffset = np.dtype([("idx", np.uint32), ("score", np.float32)])
_P = C.c_void_p
_f = C.c_float


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _sig(name, res, args):
    fn = getattr(_lib, name)
    fn.restype = res
    fn.argtypes = args
    return fn


for _n in ("dot", "euclid", "manhattan"):
    _sig(f"qo_{_n}_f32", _f, [_P, _P, C.c_size_t, C.c_int])
    _sig(f"qo_{_n}_f16", _f, [_P, _P, C.c_size_t, C.c_int])
_sig("qo_similarity_f32", _f, [C.c_int, _P, _P, C.c_size_t])
_sig("qo_similarity_f16", _f, [C.c_int, _P, _P, C.c_size_t])
_sig("qo_similarity_u8", _f, [C.c_int, _P, _P, C.c_size_t, C.c_int])
_sig("qo_cosine_preprocess_f32", None, [_P, _P, C.c_size_t, C.c_int])
_sig("qo_preprocess_f32", None, [C.c_int, _P, _P, C.c_size_t])
_sig("qo_postprocess", _f, [C.c_int, _f])
_sig("qo_f32_to_f16", None, [_P, _P, C.c_size_t])
_sig("qo_f16_to_f32", None, [_P, _P, C.c_size_t])
_sig("qo_f32_to_u8", None, [_P, _P, C.c_size_t])
_sig("qo_topk_new", _P, [C.c_size_t])
_sig("qo_topk_free", None, [_P])
_sig("qo_topk_push", None, [_P, C.c_uint32, _f])
_sig("qo_topk_into_sorted", C.c_size_t, [_P, _P])
_sig("qo_synth_value", _f, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32])
_sig("qo_synth_fill_f32", None, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, _P])
_sig("qo_synth_fill_latent_f32", None, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_float, _P])


class Storage(C.Structure):
    _fields_ = [("dtype", C.c_int), ("distance", C.c_int), ("u8_isa", C.c_int), ("rows", _P),
                ("n", C.c_size_t), ("dim", C.c_size_t),
                ("point_deleted", _P), ("n_point_bits", C.c_size_t),
                ("vec_deleted", _P), ("n_vec_bits", C.c_size_t)]


_sig("qo_peek_top_iter", C.c_int, [C.POINTER(Storage), _P, C.c_size_t, C.c_size_t, _P, C.c_size_t, _P, _P, _P])
_sig("qo_peek_top_parallel", C.c_int, [C.POINTER(Storage), _P, C.c_size_t, C.c_size_t, _P, _P, C.c_int])
_sig("qo_score_points", None, [C.POINTER(Storage), _P, _P, C.c_size_t, _P])


class Sq(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("actual_dim", C.c_uint32), ("distance", C.c_int), ("invert", C.c_int),
                ("alpha", _f), ("offset", _f), ("multiplier", _f)]


_sig("qo_sq_init", None, [C.POINTER(Sq), C.c_int, C.c_int, C.c_uint32, _P, C.c_size_t])
_sig("qo_sq_init_params", None, [C.POINTER(Sq), C.c_int, C.c_int, C.c_uint32, _f, _f])
_sig("qo_sq_encode_value", C.c_uint8, [C.POINTER(Sq), _f])
_sig("qo_sq_get_shift", _f, [C.POINTER(Sq)])
_sig("qo_sq_encode_row", None, [C.POINTER(Sq), _P, _P])
_sig("qo_sq_encode_query", None, [C.POINTER(Sq), _P, _P, C.POINTER(_f)])
_sig("qo_sq_score", _f, [C.POINTER(Sq), _P, _f, _P, C.c_int])
_sig("qo_sq_score_internal", _f, [C.POINTER(Sq), _P, _P, C.c_int])
for _n in ("qo_sq_dot_avx", "qo_sq_l1_avx", "qo_sq_dot_sse", "qo_sq_l1_sse"):
    _sig(_n, _f, [_P, _P, C.c_uint32])
_sig("qo_sq_set_ref_kernels", None, [_P, _P])


class Pq(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("chunk_size", C.c_uint32), ("m", C.c_uint32), ("n_centroids", C.c_uint32),
                ("distance", C.c_int), ("invert", C.c_int), ("centroids", _P)]


_sig("qo_pq_init", None, [C.POINTER(Pq), C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, _P])
_sig("qo_pq_encode_vector", None, [C.POINTER(Pq), _P, _P])
_sig("qo_pq_encode_query", None, [C.POINTER(Pq), _P, _P])
_sig("qo_pq_score", _f, [C.POINTER(Pq), _P, _P, C.c_int])
_sig("qo_pq_score_internal", _f, [C.POINTER(Pq), _P, _P])
_sig("qo_pq_train", None, [C.c_uint32, C.c_uint32, C.c_uint32, _P, C.c_size_t, C.c_int, _P])
_sig("qo_custom_combine", _f, [C.c_int, C.c_uint32, C.c_uint32, _P])
_sig("qo_custom_feedback", _f, [C.c_uint32, _P, _P])
_sig("qo_sq_quantile_interval", C.c_int, [_P, C.c_size_t, C.c_uint32, C.c_size_t, _f, C.POINTER(_f), C.POINTER(_f)])
_sig("qo_bq_row_bytes", C.c_size_t, [C.c_uint32])
_sig("qo_bq_encode_row", None, [C.c_uint32, _P, _P])
_sig("qo_bq_xor_popcnt", C.c_uint32, [_P, _P, C.c_uint32])
_sig("qo_bq_score", _f, [C.c_int, C.c_int, C.c_uint32, _P, _P])
_sig("qo_bq_row_bytes_ex", C.c_size_t, [C.c_uint32, C.c_int])
_sig("qo_bq_encode_row_ex", None, [C.c_uint32, C.c_int, _P, _P, _P, _P])
_sig("qo_bq_score_ex", _f, [C.c_int, C.c_int, C.c_uint32, C.c_int, _P, _P])
_sig("qo_pq_train_ex", None, [C.c_uint32, C.c_uint32, C.c_uint32, _P, C.c_size_t, C.c_uint32, C.c_float, C.c_uint32, _P, _P])

lib = _lib
_NP = {F32: np.float32, F16: np.uint16, U8: np.uint8}


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def similarity(dtype, distance, q, v, isa=ISA_AUTO):
    q = np.ascontiguousarray(q, dtype=_NP[dtype])
    v = np.ascontiguousarray(v, dtype=_NP[dtype])
    if dtype == F32:
        if isa == ISA_AUTO:
            return _lib.qo_similarity_f32(distance, _p(q), _p(v), len(q))
        fn = {DOT: _lib.qo_dot_f32, COSINE: _lib.qo_dot_f32, EUCLID: _lib.qo_euclid_f32, MANHATTAN: _lib.qo_manhattan_f32}[distance]
        return fn(_p(q), _p(v), len(q), isa)
    if dtype == F16:
        if isa == ISA_AUTO:
            return _lib.qo_similarity_f16(distance, _p(q), _p(v), len(q))
        fn = {DOT: _lib.qo_dot_f16, COSINE: _lib.qo_dot_f16, EUCLID: _lib.qo_euclid_f16, MANHATTAN: _lib.qo_manhattan_f16}[distance]
        return fn(_p(q), _p(v), len(q), isa)
    return _lib.qo_similarity_u8(distance, _p(q), _p(v), len(q), isa)


def preprocess(distance, v, isa=ISA_AUTO):
    """Metric::preprocess for f32 / f16 storages (u8 never normalises)."""
    v = f32(v)
    out = np.empty_like(v)
    flat_in, flat_out = v.reshape(-1, v.shape[-1]), out.reshape(-1, v.shape[-1])
    for i in range(flat_in.shape[0]):
        if distance == COSINE:
            _lib.qo_cosine_preprocess_f32(_p(flat_in[i]), _p(flat_out[i]), v.shape[-1], isa)
        else:
            flat_out[i] = flat_in[i]
    return out


def to_f16(v):
    v = f32(v)
    out = np.empty(v.shape, dtype=np.uint16)
    _lib.qo_f32_to_f16(_p(v), _p(out), v.size)
    return out


def f16_to_f32(h):
    h = np.ascontiguousarray(h, dtype=np.uint16)
    out = np.empty(h.shape, dtype=np.float32)
    _lib.qo_f16_to_f32(_p(h), _p(out), h.size)
    return out


def to_u8(v):
    v = f32(v)
    out = np.empty(v.shape, dtype=np.uint8)
    _lib.qo_f32_to_u8(_p(v), _p(out), v.size)
    return out


def cast(dtype, v):
    return f32(v) if dtype == F32 else to_f16(v) if dtype == F16 else to_u8(v)


def bits_to_words(bits):
    if bits is None:
        return None
    bits = np.asarray(bits, dtype=bool)
    pad = (-len(bits)) % 64
    b = np.concatenate([bits, np.zeros(pad, dtype=bool)]) if pad else bits
    return np.packbits(b.reshape(-1, 8), axis=1, bitorder="little").reshape(-1).view(np.uint64).copy()


class DenseStorage:
    """rows already preprocessed + cast (what the reference keeps in its VectorStorage)."""

    def __init__(self, dtype, distance, rows, point_deleted=None, vec_deleted=None, u8_isa=ISA_AUTO):
        self.dtype, self.distance = dtype, distance
        self.rows = np.ascontiguousarray(rows, dtype=_NP[dtype])
        self.pw, self.vw = bits_to_words(point_deleted), bits_to_words(vec_deleted)
        self.st = Storage(dtype, distance, u8_isa, _p(self.rows), self.rows.shape[0], self.rows.shape[1],
                          _p(self.pw), 0 if point_deleted is None else len(point_deleted),
                          _p(self.vw), 0 if vec_deleted is None else len(vec_deleted))

    def encode_queries(self, queries):
        """MetricQueryScorer::new: preprocess then cast (metric_query_scorer.rs:35-58)."""
        q = f32(np.atleast_2d(queries))
        if self.dtype != U8:
            q = preprocess(self.distance, q)
        return np.ascontiguousarray(cast(self.dtype, q))

    def peek_top(self, queries, top, ids=None, encoded=False, threads=0):
        q = np.ascontiguousarray(queries, dtype=_NP[self.dtype]) if encoded else self.encode_queries(queries)
        nq = q.shape[0]
        out = np.zeros((nq, top), dtype=ScoredPointOffset)
        counts = np.zeros(nq, dtype=np.uint32)
        if threads:
            rc = _lib.qo_peek_top_parallel(C.byref(self.st), _p(q), nq, top, _p(out), _p(counts), threads)
        else:
            ids_a = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
            rc = _lib.qo_peek_top_iter(C.byref(self.st), _p(q), nq, top, _p(ids_a), 0 if ids_a is None else len(ids_a),
                                       _p(out), _p(counts), None)
        assert rc == 0
        return [out[i, :counts[i]].copy() for i in range(nq)]

    def score_points(self, queries, ids, encoded=False):
        q = np.ascontiguousarray(queries, dtype=_NP[self.dtype]) if encoded else self.encode_queries(queries)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.empty((q.shape[0], len(ids)), dtype=np.float32)
        for i in range(q.shape[0]):
            _lib.qo_score_points(C.byref(self.st), _p(q[i]), _p(ids), len(ids), _p(out[i]))
        return out


def topk_push_all(pairs, length):
    """FixedLengthPriorityQueue: push (idx, score) pairs, return into_sorted_vec()."""
    t = _lib.qo_topk_new(length)
    for idx, score in pairs:
        _lib.qo_topk_push(t, int(idx), float(score))
    out = np.zeros(length, dtype=ScoredPointOffset)
    n = _lib.qo_topk_into_sorted(t, _p(out))
    _lib.qo_topk_free(t)
    return out[:n]


def synth(seed, row0, n, dim):
    out = np.empty((n, dim), dtype=np.float32)
    _lib.qo_synth_fill_f32(seed, row0, n, dim, _p(out))
    return out


def synth_latent(seed, row0, n, dim, latent_dim=32, noise=0.0):
    """Rows of low intrinsic dimension, bit-identical to qmx_synth_fill_latent_f32 on the device."""
    out = np.empty((n, dim), dtype=np.float32)
    _lib.qo_synth_fill_latent_f32(seed, row0, n, dim, latent_dim, noise, _p(out))
    return out


def load_ref_quant():
    """The reference's own C SQ kernels (oracle/_ref, built from /root/reference by oracle/Makefile)."""
    if not os.path.exists(REF_QUANT_PATH):
        return None
    ref = C.CDLL(REF_QUANT_PATH)
    for n in ("impl_score_dot_avx", "impl_score_l1_avx", "impl_score_dot_sse", "impl_score_l1_sse"):
        fn = getattr(ref, n)
        fn.restype = _f
        fn.argtypes = [_P, _P, C.c_uint32]
    _lib.qo_sq_set_ref_kernels(C.cast(ref.impl_score_dot_avx, _P), C.cast(ref.impl_score_l1_avx, _P))
    return ref


class SqOracle:
    """EncodedVectorsU8 on the CPU (oracle): given (alpha, offset)."""

    def __init__(self, distance, dim, alpha, offset, isa=ISA_AUTO):
        self.sq = Sq()
        invert = 1 if distance in (EUCLID, MANHATTAN) else 0
        _lib.qo_sq_init_params(C.byref(self.sq), distance, invert, dim, float(alpha), float(offset))
        self.dim, self.ad, self.isa = dim, self.sq.actual_dim, isa
        self.rows = None

    def encode_rows(self, vectors):
        v = f32(vectors)
        out = np.zeros((v.shape[0], 4 + self.ad), dtype=np.uint8)
        for i in range(v.shape[0]):
            _lib.qo_sq_encode_row(C.byref(self.sq), _p(v[i]), _p(out[i]))
        self.rows = out
        return out

    def encode_query(self, q):
        q = f32(q)
        codes = np.zeros(self.ad, dtype=np.uint8)
        off = _f()
        _lib.qo_sq_encode_query(C.byref(self.sq), _p(q), _p(codes), C.byref(off))
        return codes, off.value

    def score_points(self, queries, ids):
        """queries: already metric-preprocessed f32 [nq, dim]"""
        queries = f32(np.atleast_2d(queries))
        out = np.empty((queries.shape[0], len(ids)), dtype=np.float32)
        for qi in range(queries.shape[0]):
            codes, off = self.encode_query(queries[qi])
            for j, i in enumerate(ids):
                out[qi, j] = _lib.qo_sq_score(C.byref(self.sq), _p(codes), off, _p(self.rows[i]), self.isa)
        return out

    def score_internal(self, a, b):
        return np.array([_lib.qo_sq_score_internal(C.byref(self.sq), _p(self.rows[i]), _p(self.rows[j]), self.isa)
                         for i, j in zip(a, b)], dtype=np.float32)


def sq_quantile_interval(sample, count, quantile):
    """find_quantile_interval on a given sample -> (min, max) or None."""
    v = f32(sample)
    mn, mx = _f(), _f()
    ok = _lib.qo_sq_quantile_interval(_p(v), v.shape[0], v.shape[1], count, float(quantile), C.byref(mn), C.byref(mx))
    return (np.float32(mn.value), np.float32(mx.value)) if ok else None


_sig("qo_bq_encode_scalar_query", C.c_size_t, [C.c_uint32, C.c_int, C.c_uint32, _P, _P])
_sig("qo_bq_xor_popcnt_scalar", C.c_uint64, [_P, _P, C.c_uint32, C.c_uint32])
_sig("qo_bq_score_scalar", _f, [C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_uint32, _P, _P])


class BqOracle:
    """EncodedVectorsBin<u128>, OneBit, SameAsStorage on the CPU (oracle).  invert defaults to the segment's choice
    (quantized_vectors.rs:232: Euclid | Manhattan)."""

    def __init__(self, distance, dim, invert=None, encoding=0, mean=None, stddev=None):
        self.distance, self.dim, self.encoding = distance, dim, int(encoding)
        self.invert = int(distance in (EUCLID, MANHATTAN)) if invert is None else int(invert)
        self.row_bytes = int(_lib.qo_bq_row_bytes_ex(dim, self.encoding))
        self.mean = None if mean is None else f32(mean)
        self.stddev = None if stddev is None else f32(stddev)
        self.rows = None

    def encode(self, vectors):
        v = f32(np.atleast_2d(vectors))
        out = np.zeros((v.shape[0], self.row_bytes), dtype=np.uint8)
        for i in range(v.shape[0]):
            _lib.qo_bq_encode_row_ex(self.dim, self.encoding, _p(self.mean), _p(self.stddev), _p(v[i]), _p(out[i]))
        return out

    def encode_rows(self, vectors):
        self.rows = self.encode(vectors)
        return self.rows

    def score_points(self, queries_preprocessed, ids):
        qs = self.encode(queries_preprocessed)
        out = np.empty((qs.shape[0], len(ids)), dtype=np.float32)
        for qi in range(qs.shape[0]):
            for j, i in enumerate(ids):
                out[qi, j] = _lib.qo_bq_score_ex(self.distance, self.invert, self.dim, self.encoding, _p(qs[qi]), _p(self.rows[i]))
        return out

    def encode_scalar_queries(self, queries_preprocessed, bits):
        """encode_query_vector with QueryEncoding::Scalar4bits / Scalar8bits (encoded_vectors_binary.rs:683-756)."""
        v = f32(np.atleast_2d(queries_preprocessed))
        nb = int(_lib.qo_bq_encode_scalar_query(self.dim, self.encoding, bits, _p(v[0]), None))
        out = np.zeros((v.shape[0], nb), dtype=np.uint8)
        for i in range(v.shape[0]):
            _lib.qo_bq_encode_scalar_query(self.dim, self.encoding, bits, _p(v[i]), _p(out[i]))
        return out

    def score_points_scalar(self, queries_preprocessed, ids, bits):
        qs = self.encode_scalar_queries(queries_preprocessed, bits)
        out = np.empty((qs.shape[0], len(ids)), dtype=np.float32)
        for qi in range(qs.shape[0]):
            for j, i in enumerate(ids):
                out[qi, j] = _lib.qo_bq_score_scalar(self.distance, self.invert, self.dim, self.encoding, bits, _p(qs[qi]), _p(self.rows[i]))
        return out

    def score_internal(self, a, b):
        return np.array([_lib.qo_bq_score_ex(self.distance, self.invert, self.dim, self.encoding, _p(self.rows[i]), _p(self.rows[j]))
                         for i, j in zip(a, b)], dtype=np.float32)


class PqOracle:
    """EncodedVectorsPQ on the CPU (oracle): given centroids [n_centroids, dim]."""

    def __init__(self, distance, dim, chunk_size, centroids, isa=ISA_AUTO, invert=None):
        self.centroids = f32(centroids)
        self.pq = Pq()
        if invert is None:
            invert = 1 if distance in (EUCLID, MANHATTAN) else 0      # the segment's choice (quantized_vectors.rs:232)
        invert = int(invert)
        _lib.qo_pq_init(C.byref(self.pq), distance, invert, dim, chunk_size, self.centroids.shape[0], _p(self.centroids))
        self.m, self.nc, self.isa, self.dim = self.pq.m, self.centroids.shape[0], isa, dim
        self.codes = None

    @staticmethod
    def train(data, dim, chunk_size, n_centroids, iters=5):
        data = f32(data)
        cen = np.zeros((n_centroids, dim), dtype=np.float32)
        _lib.qo_pq_train(dim, chunk_size, n_centroids, _p(data), data.shape[0], iters, _p(cen))
        return cen

    @staticmethod
    def train_ex(data, dim, chunk_size, n_centroids, max_iters=100, accuracy=1e-5, threads=1):
        """kmeans.rs on a given sample: returns (centroids [n_centroids, dim], update steps per chunk)."""
        data = f32(data)
        cen = np.zeros((n_centroids, dim), dtype=np.float32)
        m = (dim + chunk_size - 1) // chunk_size
        iters = np.zeros(m, dtype=np.uint32)
        _lib.qo_pq_train_ex(dim, chunk_size, n_centroids, _p(data), data.shape[0], max_iters, float(accuracy), threads, _p(cen), _p(iters))
        return cen, iters

    def encode(self, vectors):
        v = f32(vectors)
        out = np.zeros((v.shape[0], self.m), dtype=np.uint8)
        for i in range(v.shape[0]):
            _lib.qo_pq_encode_vector(C.byref(self.pq), _p(v[i]), _p(out[i]))
        self.codes = out
        return out

    def lut(self, q):
        q = f32(q)
        lut = np.zeros((self.m, self.nc), dtype=np.float32)
        _lib.qo_pq_encode_query(C.byref(self.pq), _p(q), _p(lut))
        return lut

    def score_points(self, queries, ids):
        queries = f32(np.atleast_2d(queries))
        out = np.empty((queries.shape[0], len(ids)), dtype=np.float32)
        for qi in range(queries.shape[0]):
            lut = self.lut(queries[qi])
            for j, i in enumerate(ids):
                out[qi, j] = _lib.qo_pq_score(C.byref(self.pq), _p(lut), _p(self.codes[i]), self.isa)
        return out

    def score_internal(self, a, b):
        return np.array([_lib.qo_pq_score_internal(C.byref(self.pq), _p(self.codes[i]), _p(self.codes[j])) for i, j in zip(a, b)],
                        dtype=np.float32)


# ---- cross-segment merge, scorer façade, HNSW (oracle/qdrant_oracle_hnsw.c) --------------------------
class Scorer(C.Structure):
    _fields_ = [("kind", C.c_int), ("st", C.POINTER(Storage)), ("query", _P),
                ("sq", C.POINTER(Sq)), ("sq_rows", _P), ("sq_query", _P), ("sq_query_offset", _f),
                ("pq", C.POINTER(Pq)), ("pq_codes", _P), ("pq_lut", _P), ("isa", C.c_int),
                ("bq_rows", _P), ("bq_query", _P), ("bq_dim", C.c_uint32), ("bq_distance", C.c_int), ("bq_invert", C.c_int),
                ("mv_tokens", _P), ("mv_n_tokens", C.c_uint32), ("mv_offsets", _P),
                ("tq", _P), ("tq_rows", _P), ("tq_query", _P), ("tq_invert", C.c_int),
                ("cq_examples", _P), ("cq_kind", C.c_uint32), ("cq_n_a", C.c_uint32), ("cq_n_b", C.c_uint32), ("cq_coefs", _P)]


_sig("qo_merge_topk", None, [_P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P])
_sig("qo_scorer_score_point", _f, [C.POINTER(Scorer), C.c_uint32])
_sig("qo_scorer_score_internal", _f, [C.POINTER(Scorer), C.c_uint32, C.c_uint32])
_sig("qo_hnsw_build", _P, [C.POINTER(Storage), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64])
_sig("qo_hnsw_build_parallel", _P, [C.POINTER(Storage), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64, C.c_int])
_sig("qo_hnsw_build_with", _P, [C.POINTER(Scorer), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64])
_sig("qo_hnsw_free", None, [_P])
_sig("qo_hnsw_point_level", C.c_uint32, [_P, C.c_uint32])
_sig("qo_hnsw_max_level", C.c_uint32, [_P])
_sig("qo_hnsw_links", C.c_uint32, [_P, C.c_uint32, C.c_uint32, _P])
_sig("qo_hnsw_entry_points", C.c_uint32, [_P, _P, _P, C.c_uint32])
_sig("qo_hnsw_extra_entry_points", C.c_uint32, [_P, _P, _P, C.c_uint32])
_sig("qo_hnsw_import_plain", _P, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P, _P, _P, _P, _P, C.c_uint32, _P, _P, C.c_uint32])
_sig("qo_hnsw_export_plain", None, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _P, _P, _P, _P])
_sig("qo_hnsw_search", C.c_uint32, [_P, C.POINTER(Scorer), C.c_uint32, C.c_uint32, _P, C.POINTER(C.c_uint64)])
_sig("qo_hnsw_search_algo", C.c_uint32, [_P, C.POINTER(Scorer), C.c_uint32, C.c_uint32, C.c_int, _P, C.POINTER(C.c_uint64)])
_sig("qo_hnsw_search_traced", C.c_uint32, [_P, C.POINTER(Scorer), C.c_uint32, C.c_uint32, _P, C.POINTER(C.c_uint64), _P, C.c_uint32, C.POINTER(C.c_uint32)])
_sig("qo_hnsw_search_with_vectors", C.c_uint32, [_P, C.POINTER(Scorer), C.POINTER(Scorer), C.c_uint32, C.c_uint32, _P, C.POINTER(C.c_uint64),
                                                 C.POINTER(C.c_uint64)])
_sig("qo_links_heuristic", C.c_uint32, [_P, C.c_uint32, C.c_uint32, _P, C.c_uint32, _P])
_sig("qo_links_connect", C.c_uint32, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P, C.c_uint32])
_sig("qo_links_connect_heuristic", C.c_uint32, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P, C.c_uint32])


def merge_topk(lists, counts, k, idx_base=None):
    """BatchResultAggregator: lists [n_lists, nq, k] ScoredPointOffset, counts [n_lists, nq]."""
    lists = np.ascontiguousarray(lists, dtype=ScoredPointOffset)
    n_lists, nq = lists.shape[0], lists.shape[1]
    counts = None if counts is None else np.ascontiguousarray(counts, dtype=np.uint32)
    base = None if idx_base is None else np.ascontiguousarray(idx_base, dtype=np.uint32)
    out = np.zeros((nq, k), dtype=ScoredPointOffset)
    oc = np.zeros(nq, dtype=np.uint32)
    _lib.qo_merge_topk(_p(lists), _p(counts), _p(base), n_lists, nq, k, _p(out), _p(oc))
    return [out[i, :oc[i]].copy() for i in range(nq)]


class PlainLinks:
    """The plain GraphLinks arrays (graph_links/view.rs) + entry points: what qmx_hnsw_create ingests."""

    def __init__(self, m, m0, reindex, level_offsets, offsets, neighbors, ep_ids, ep_levels, xp_ids=(), xp_levels=()):
        self.m, self.m0 = m, m0
        self.reindex, self.level_offsets, self.offsets, self.neighbors = reindex, level_offsets, offsets, neighbors
        self.ep_ids, self.ep_levels = ep_ids, ep_levels
        self.xp_ids, self.xp_levels = xp_ids, xp_levels          # EntryPoints::extra_entry_points (iter_unsorted order)

    def links(self, point, level):
        idx = point if level == 0 else int(self.level_offsets[level]) + int(self.reindex[point])
        return self.neighbors[int(self.offsets[idx]):int(self.offsets[idx + 1])]


def custom_scores(storage: "DenseStorage", examples, kind, n_a, n_b, ids, coefs=None):
    """CustomQueryScorer over the oracle's similarities: examples [ne, dim] original vectors in flat_iter() order
    (kind 4 = FeedbackQuery: target, then (positive, negative) pairs; coefs = [a, partial_computation...])."""
    sims = storage.score_points(examples, ids)                 # [ne, n] bit-exact leaves
    out = np.empty(len(ids), dtype=np.float32)
    col = np.empty(sims.shape[0], dtype=np.float32)
    cf = None if coefs is None else f32(coefs)
    for j in range(len(ids)):
        col[:] = sims[:, j]
        out[j] = _lib.qo_custom_feedback(n_b, _p(col), _p(cf)) if kind == 4 else _lib.qo_custom_combine(kind, n_a, n_b, _p(col))
    return out


def plain_links_file(p: "PlainLinks") -> bytes:
    """The PLAIN graph-links file of lib/segment/src/index/hnsw_index/graph_links/serializer.rs:52-209:
    HeaderPlain (header.rs:9-20, 64 bytes), level_offsets [levels], reindex, neighbors, padding to 8, offsets."""
    levels = len(p.level_offsets) - 1
    body = (np.asarray(p.level_offsets[:levels], dtype="<u8").tobytes() + np.asarray(p.reindex, dtype="<u4").tobytes()
            + np.asarray(p.neighbors, dtype="<u4").tobytes())
    pad = (-(64 + len(body))) % 8
    header = np.array([len(p.reindex), levels, len(p.neighbors), len(p.offsets), pad, 0, 0, 0], dtype="<u8").tobytes()
    return header + body + b"\0" * pad + np.asarray(p.offsets, dtype="<u8").tobytes()


class Hnsw:
    """GraphLayersBuilder -> GraphLayers on the CPU oracle, built over a DenseStorage."""

    def __init__(self, storage: DenseStorage, m=16, m0=None, ef_construct=100, entry_points_num=10, use_heuristic=True,
                 seed=42, threads=0):
        self.storage = storage
        self.m, self.m0 = m, (2 * m if m0 is None else m0)
        if threads and threads > 1:
            self.h = _lib.qo_hnsw_build_parallel(C.byref(storage.st), self.m, self.m0, ef_construct, entry_points_num,
                                                 1 if use_heuristic else 0, seed, threads)
        else:
            self.h = _lib.qo_hnsw_build(C.byref(storage.st), self.m, self.m0, ef_construct, entry_points_num,
                                        1 if use_heuristic else 0, seed)
        self.n = storage.rows.shape[0]

    @classmethod
    def build_pq(cls, storage: DenseStorage, pq: "PqOracle", m=16, m0=None, ef_construct=100, entry_points_num=10, seed=42):
        """The build of a PQ-quantized segment (hnsw/build.rs:334-341 + point_scorer.rs:183-218): searches of an insertion score
        through the LUT of the point's ORIGINAL vector (storage rows), heuristic / back links through EncodedVectorsPQ::score_internal."""
        self = cls.__new__(cls)
        self.storage, self.m, self.m0, self.n = storage, m, (2 * m if m0 is None else m0), storage.rows.shape[0]
        s = Scorer()
        s.kind, s.st, s.pq, s.pq_codes, s.isa = 2, C.pointer(storage.st), C.pointer(pq.pq), pq.codes.ctypes.data, pq.isa
        self._keep = (s, pq)
        self.h = _lib.qo_hnsw_build_with(C.byref(s), self.m, self.m0, ef_construct, entry_points_num, 1, seed)
        return self

    @classmethod
    def build_sq(cls, storage: DenseStorage, sq: "SqOracle", m=16, m0=None, ef_construct=100, entry_points_num=10, seed=42):
        """The build of an SQ segment: every score through EncodedVectorsU8 (a stored row is its own query, encode_internal_vector)."""
        self = cls.__new__(cls)
        self.storage, self.m, self.m0, self.n = storage, m, (2 * m if m0 is None else m0), storage.rows.shape[0]
        s = Scorer()
        s.kind, s.st, s.sq, s.sq_rows, s.isa = 1, C.pointer(storage.st), C.pointer(sq.sq), sq.rows.ctypes.data, sq.isa
        self._keep = (s, sq)
        self.h = _lib.qo_hnsw_build_with(C.byref(s), self.m, self.m0, ef_construct, entry_points_num, 1, seed)
        return self

    @classmethod
    def build_bq(cls, storage: DenseStorage, bq: "BqOracle", m=16, m0=None, ef_construct=100, entry_points_num=10, seed=42):
        """The build of a BQ segment: one-bit score_internal everywhere."""
        self = cls.__new__(cls)
        self.storage, self.m, self.m0, self.n = storage, m, (2 * m if m0 is None else m0), storage.rows.shape[0]
        s = Scorer()
        s.kind, s.st, s.bq_rows = 3, C.pointer(storage.st), bq.rows.ctypes.data
        s.bq_dim, s.bq_distance, s.bq_invert = bq.dim, bq.distance, bq.invert
        self._keep = (s, bq)
        self.h = _lib.qo_hnsw_build_with(C.byref(s), self.m, self.m0, ef_construct, entry_points_num, 1, seed)
        return self

    @classmethod
    def build_tq(cls, storage: DenseStorage, tq: "TqOracle", m=16, m0=None, ef_construct=100, entry_points_num=10, seed=42):
        """The build of a TurboQuant segment: like PQ, EncodedVectorsTQ cannot turn a stored row into a query, so an insertion's searches score
        through precompute_query(ORIGINAL vector) and the heuristic / back links through score_symmetric (point_scorer.rs:183-218)."""
        self = cls.__new__(cls)
        self.storage, self.m, self.m0, self.n = storage, m, (2 * m if m0 is None else m0), storage.rows.shape[0]
        s = Scorer()
        s.kind, s.st, s.tq, s.tq_rows, s.tq_invert = 5, C.pointer(storage.st), tq.h, tq.rows.ctypes.data, 1 if tq.invert else 0
        self._keep = (s, tq)
        self.h = _lib.qo_hnsw_build_with(C.byref(s), self.m, self.m0, ef_construct, entry_points_num, 1, seed)
        return self

    @classmethod
    def from_plain(cls, p, n):
        """GraphLayers::load of plain GraphLinks arrays: the oracle walks a graph it did not build."""
        self = cls.__new__(cls)
        self.storage, self.m, self.m0, self.n = None, p.m, p.m0, n
        arrs = [np.ascontiguousarray(p.reindex, dtype=np.uint32), np.ascontiguousarray(p.level_offsets, dtype=np.uint64),
                np.ascontiguousarray(p.offsets, dtype=np.uint64), np.ascontiguousarray(p.neighbors, dtype=np.uint32),
                np.ascontiguousarray(p.ep_ids, dtype=np.uint32), np.ascontiguousarray(p.ep_levels, dtype=np.uint32),
                np.ascontiguousarray(p.xp_ids, dtype=np.uint32), np.ascontiguousarray(p.xp_levels, dtype=np.uint32)]
        self.h = _lib.qo_hnsw_import_plain(n, p.m, p.m0, len(arrs[1]) - 1, _p(arrs[0]), _p(arrs[1]), _p(arrs[2]),
                                           _p(arrs[3]) if len(arrs[3]) else None, _p(arrs[4]), _p(arrs[5]), len(arrs[4]),
                                           _p(arrs[6]) if len(arrs[6]) else None, _p(arrs[7]) if len(arrs[7]) else None, len(arrs[6]))
        return self

    def __del__(self):
        try:
            _lib.qo_hnsw_free(self.h)
        except Exception:
            pass

    def point_level(self, i):
        return _lib.qo_hnsw_point_level(self.h, i)

    def links(self, i, level):
        buf = np.zeros(self.m0 + self.m, dtype=np.uint32)
        n = _lib.qo_hnsw_links(self.h, i, level, _p(buf))
        return buf[:n].copy()

    def entry_points(self):
        n = _lib.qo_hnsw_entry_points(self.h, None, None, 0)
        ids, lv = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)
        _lib.qo_hnsw_entry_points(self.h, _p(ids), _p(lv), n)
        return ids, lv

    def extra_entry_points(self):
        n = _lib.qo_hnsw_extra_entry_points(self.h, None, None, 0)
        ids, lv = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)
        _lib.qo_hnsw_extra_entry_points(self.h, _p(ids), _p(lv), n)
        return ids, lv

    def export_plain(self) -> PlainLinks:
        nl, no, nn = C.c_uint32(), C.c_uint64(), C.c_uint64()
        _lib.qo_hnsw_export_plain(self.h, C.byref(nl), C.byref(no), C.byref(nn), None, None, None, None)
        reindex = np.zeros(self.n, dtype=np.uint32)
        level_offsets = np.zeros(nl.value + 1, dtype=np.uint64)
        offsets = np.zeros(no.value, dtype=np.uint64)
        neighbors = np.zeros(max(1, nn.value), dtype=np.uint32)
        _lib.qo_hnsw_export_plain(self.h, C.byref(nl), C.byref(no), C.byref(nn), _p(reindex), _p(level_offsets), _p(offsets),
                                  _p(neighbors))
        ids, lv = self.entry_points()
        xids, xlv = self.extra_entry_points()
        return PlainLinks(self.m, self.m0, reindex, level_offsets, offsets, neighbors[:nn.value], ids, lv, xids, xlv)

    # -- searches: one qo_scorer per query ------------------------------------------------------------
    algorithm = 0      # SearchAlgorithm: 0 = Hnsw, 1 = Acorn (set on the instance for a run)

    pops = None        # a list: every search appends the candidates its level-0 loop popped and expanded (qo_hnsw_search_traced; plain walk only)

    def _run(self, scorer, top, ef):
        out = np.zeros(max(top, 1), dtype=ScoredPointOffset)
        ns = C.c_uint64()
        if self.pops is not None and self.algorithm == 0:
            cap, npop = 32 * max(top, ef) + 256, C.c_uint32()
            while True:
                tr = np.zeros(cap, dtype=ScoredPointOffset)
                n = _lib.qo_hnsw_search_traced(self.h, C.byref(scorer), top, ef, _p(out), C.byref(ns), _p(tr), cap, C.byref(npop))
                if npop.value <= cap:
                    break
                cap = npop.value
            self.pops.append(tr[:npop.value].copy())
            return out[:n].copy(), ns.value
        n = _lib.qo_hnsw_search_algo(self.h, C.byref(scorer), top, ef, self.algorithm, _p(out), C.byref(ns))
        return out[:n].copy(), ns.value

    def search_dense(self, storage: DenseStorage, queries, top, ef, encoded=False, with_stats=False):
        q = np.ascontiguousarray(queries, dtype=_NP[storage.dtype]) if encoded else storage.encode_queries(queries)
        res, stats = [], []
        for i in range(q.shape[0]):
            s = Scorer()
            s.kind, s.st, s.query = 0, C.pointer(storage.st), q[i].ctypes.data
            r, ns = self._run(s, top, ef)
            res.append(r)
            stats.append(ns)
        return (res, stats) if with_stats else res

    def search_sq(self, flags_storage: DenseStorage, sq: "SqOracle", queries_preprocessed, top, ef):
        res = []
        for qv in f32(np.atleast_2d(queries_preprocessed)):
            codes, off = sq.encode_query(qv)
            s = Scorer()
            s.kind, s.st, s.sq, s.sq_rows = 1, C.pointer(flags_storage.st), C.pointer(sq.sq), sq.rows.ctypes.data
            s.sq_query, s.sq_query_offset, s.isa = codes.ctypes.data, off, sq.isa
            res.append(self._run(s, top, ef)[0])
        return res

    def search_bq(self, flags_storage: DenseStorage, bq: "BqOracle", queries_preprocessed, top, ef):
        res = []
        for qb in bq.encode(queries_preprocessed):
            s = Scorer()
            s.kind, s.st, s.bq_rows, s.bq_query = 3, C.pointer(flags_storage.st), bq.rows.ctypes.data, qb.ctypes.data
            s.bq_dim, s.bq_distance, s.bq_invert = bq.dim, bq.distance, bq.invert
            res.append(self._run(s, top, ef)[0])
        return res

    def search_with_vectors(self, base_storage: DenseStorage, links, queries, top, ef):
        """GraphLayers::search_with_vectors: links = ("sq", SqOracle) | ("bq", BqOracle) | ("pq", PqOracle) over the same points as
        `base_storage` (preprocessed original rows).  Returns (results, link vectors scored, base vectors scored) per query."""
        qpre = preprocess(base_storage.distance, f32(np.atleast_2d(queries)))
        qbase = base_storage.encode_queries(queries)
        res, n_links, n_base = [], [], []
        for i in range(qpre.shape[0]):
            b = Scorer()
            b.kind, b.st, b.query = 0, C.pointer(base_storage.st), qbase[i].ctypes.data
            s = Scorer()
            kind, quant = links
            if kind == "sq":
                codes, off = quant.encode_query(qpre[i])
                s.kind, s.st, s.sq, s.sq_rows = 1, C.pointer(base_storage.st), C.pointer(quant.sq), quant.rows.ctypes.data
                s.sq_query, s.sq_query_offset, s.isa = codes.ctypes.data, off, quant.isa
                keep = codes
            elif kind == "bq":
                keep = quant.encode(qpre[i][None, :])[0]
                s.kind, s.st, s.bq_rows, s.bq_query = 3, C.pointer(base_storage.st), quant.rows.ctypes.data, keep.ctypes.data
                s.bq_dim, s.bq_distance, s.bq_invert = quant.dim, quant.distance, quant.invert
            else:
                keep = quant.lut(qpre[i])
                s.kind, s.st, s.pq, s.pq_codes = 2, C.pointer(base_storage.st), C.pointer(quant.pq), quant.codes.ctypes.data
                s.pq_lut, s.isa = keep.ctypes.data, quant.isa
            out = np.zeros(max(top, 1), dtype=ScoredPointOffset)
            nl, nb = C.c_uint64(), C.c_uint64()
            n = _lib.qo_hnsw_search_with_vectors(self.h, C.byref(s), C.byref(b), top, ef, _p(out), C.byref(nl), C.byref(nb))
            res.append(out[:n].copy())
            n_links.append(nl.value)
            n_base.append(nb.value)
        return res, n_links, n_base

    def search_pq(self, flags_storage: DenseStorage, pq: "PqOracle", queries_preprocessed, top, ef, with_stats=False):
        res, stats = [], []
        for qv in f32(np.atleast_2d(queries_preprocessed)):
            lut = pq.lut(qv)
            s = Scorer()
            s.kind, s.st, s.pq, s.pq_codes = 2, C.pointer(flags_storage.st), C.pointer(pq.pq), pq.codes.ctypes.data
            s.pq_lut, s.isa = lut.ctypes.data, pq.isa
            r, ns = self._run(s, top, ef)
            res.append(r)
            stats.append(ns)
        return (res, stats) if with_stats else res


def _hnsw_search_tq(self, flags_storage: DenseStorage, tq: "TqOracle", queries_preprocessed, top, ef):
    res = []
    for qv in f32(np.atleast_2d(queries_preprocessed)):
        e = _lib.qo_tq_precompute_query(tq.h, _p(qv))
        s = Scorer()
        s.kind, s.st, s.tq, s.tq_rows, s.tq_query, s.tq_invert = 5, C.pointer(flags_storage.st), tq.h, tq.rows.ctypes.data, e, 1 if tq.invert else 0
        res.append(self._run(s, top, ef)[0])
        _lib.qo_tq_query_free(e)
    return res


Hnsw.search_tq = _hnsw_search_tq


class _TqQueryKeep:
    """owns one qo_tq_query (freed with the Python object)"""
    def __init__(self, e):
        self.e = e

    def __del__(self):
        if self.e:
            _lib.qo_tq_query_free(self.e)
            self.e = None


class MultiOracle:
    """Multi-vector points with MaxSim over an inner scorer (qo_scorer kind 4): `MultiMetricQueryScorer` over dense inner rows, or
    `QuantizedMultivectorStorage` over SQ / BQ / PQ inner rows.  inner = ("dense", DenseStorage) | ("sq", DenseStorage, SqOracle) |
    ("bq", DenseStorage, BqOracle) | ("pq", DenseStorage, PqOracle) | ("tq", DenseStorage, TqOracle) where the DenseStorage holds the PREPROCESSED inner rows (for the
    quantized kinds only its row count and, for PQ builds, its rows matter).  offsets: [n_points + 1] ascending inner-row offsets."""

    def __init__(self, inner, offsets, point_deleted=None):
        self.inner, self.kind = inner, inner[0]
        self.offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.n_points = len(self.offsets) - 1
        # the POINT-level storage: n, deleted flags (rows unused)
        self.points = DenseStorage(F32, DOT, np.zeros((self.n_points, 1), dtype=np.float32), point_deleted=point_deleted)

    def _token(self, qv):
        """one inner scorer for one PREPROCESSED inner query vector; returns (Scorer, keep-alive)"""
        s = Scorer()
        st = self.inner[1]
        if self.kind == "dense":
            enc = np.ascontiguousarray(cast(st.dtype, f32(qv)[None, :]))[0]
            s.kind, s.st, s.query = 0, C.pointer(st.st), enc.ctypes.data
            return s, enc
        if self.kind == "sq":
            sq = self.inner[2]
            codes, off = sq.encode_query(f32(qv))
            s.kind, s.st, s.sq, s.sq_rows = 1, C.pointer(st.st), C.pointer(sq.sq), sq.rows.ctypes.data
            s.sq_query, s.sq_query_offset, s.isa = codes.ctypes.data, off, sq.isa
            return s, codes
        if self.kind == "bq":
            bq = self.inner[2]
            qb = bq.encode(f32(qv)[None, :])[0]
            s.kind, s.st, s.bq_rows, s.bq_query = 3, C.pointer(st.st), bq.rows.ctypes.data, qb.ctypes.data
            s.bq_dim, s.bq_distance, s.bq_invert = bq.dim, bq.distance, bq.invert
            return s, qb
        if self.kind == "tq":
            tq = self.inner[2]
            e = _lib.qo_tq_precompute_query(tq.h, _p(f32(qv)))
            s.kind, s.st, s.tq, s.tq_rows, s.tq_query, s.tq_invert = 5, C.pointer(st.st), tq.h, tq.rows.ctypes.data, e, 1 if tq.invert else 0
            return s, _TqQueryKeep(e)
        pq = self.inner[2]
        lut = pq.lut(f32(qv))
        s.kind, s.st, s.pq, s.pq_codes = 2, C.pointer(st.st), C.pointer(pq.pq), pq.codes.ctypes.data
        s.pq_lut, s.isa = lut.ctypes.data, pq.isa
        return s, lut

    def scorer(self, multi_query_preprocessed):
        """qo_scorer of kind 4 for one multi-query ([tokens, dim] preprocessed f32); returns (Scorer, keep-alive)"""
        toks = [self._token(q) for q in f32(np.atleast_2d(multi_query_preprocessed))]
        arr = (Scorer * max(1, len(toks)))(*[t[0] for t in toks])
        s = Scorer()
        s.kind, s.st = 4, C.pointer(self.points.st)
        s.mv_tokens, s.mv_n_tokens, s.mv_offsets = C.addressof(arr), len(toks), self.offsets.ctypes.data
        return s, (arr, toks)

    def template(self):
        """the storage as a scorer template (builds: only score_internal is used, through mv_tokens[0])"""
        return self.scorer(np.zeros((1, self.inner[1].rows.shape[1]), dtype=np.float32))

    def score_points(self, multi_queries_preprocessed, ids):
        out = np.empty((len(multi_queries_preprocessed), len(ids)), dtype=np.float32)
        for j, mq in enumerate(multi_queries_preprocessed):
            s, keep = self.scorer(mq)
            for i, p in enumerate(ids):
                out[j, i] = _lib.qo_scorer_score_point(C.byref(s), int(p))
        return out

    def score_internal(self, a, b):
        s, keep = self.template()
        return np.float32(_lib.qo_scorer_score_internal(C.byref(s), int(a), int(b)))

    def build(self, m=8, m0=None, ef_construct=32, entry_points_num=4, seed=42) -> "Hnsw":
        """GraphLayersBuilder over the multi-vector points through score_internal_max_similarity (sequential, as the reference's tests)."""
        g = Hnsw.__new__(Hnsw)
        g.storage, g.m, g.m0, g.n = None, m, (2 * m if m0 is None else m0), self.n_points
        s, keep = self.template()
        g._keep = (s, keep, self)
        g.h = _lib.qo_hnsw_build_with(C.byref(s), g.m, g.m0, ef_construct, entry_points_num, 1, seed)
        return g

    def search(self, graph: "Hnsw", multi_queries_preprocessed, top, ef):
        res, stats = [], []
        for mq in multi_queries_preprocessed:
            s, keep = self.scorer(mq)
            r, ns = graph._run(s, top, ef)
            res.append(r)
            stats.append(ns)
        return res, stats


def links_heuristic(sorted_candidates, level_m, score_table):
    c = np.ascontiguousarray(sorted_candidates, dtype=ScoredPointOffset)
    t = f32(score_table)
    out = np.zeros(max(level_m, 1), dtype=np.uint32)
    n = _lib.qo_links_heuristic(_p(c), len(c), level_m, _p(t), t.shape[0], _p(out))
    return out[:n].tolist()


def links_connect(links, new_point, target, level_m, score_table):
    t = f32(score_table)
    buf = np.zeros(level_m + 1, dtype=np.uint32)
    buf[:len(links)] = links
    n = _lib.qo_links_connect(_p(buf), len(links), new_point, target, level_m, _p(t), t.shape[0])
    return buf[:n].tolist()


def links_connect_heuristic(links, new_point, target, level_m, score_table):
    t = f32(score_table)
    buf = np.zeros(level_m + 1, dtype=np.uint32)
    buf[:len(links)] = links
    n = _lib.qo_links_connect_heuristic(_p(buf), len(links), new_point, target, level_m, _p(t), t.shape[0])
    return buf[:n].tolist()


# ---- compressed graph-links files (oracle/qdrant_oracle_links.c) ----------------------------------------------------
_u8c, _u32c, _u64c = C.c_uint8, C.c_uint32, C.c_uint64
_sig("qo_bitpack_write", _u64c, [_P, _P, _u32c, _P, _u64c])
_sig("qo_bitpack_read", None, [_P, _u64c, _P, _u32c, _P])
_sig("qo_pack_links", _u64c, [_P, _u32c, _u8c, _u32c, _P, _u64c])
_sig("qo_iterate_packed_links", _u32c, [_P, _u64c, _u8c, _u32c, _P, _u32c])
_sig("qo_packed_links_size", _u64c, [_P, _u64c, _u8c, _u32c, _u32c])
_sig("qo_ordered_compress", _u64c, [_P, _u64c, _P, _u64c, _P])
_sig("qo_ordered_compress_with", _u64c, [_P, _u64c, _u8c, _u8c, _u8c, _P, _u64c])
_sig("qo_ordered_get", _u64c, [_P, _u64c, _u8c, _u8c, _u8c, _u64c])
_sig("qo_links_serialize_compressed", _u64c, [_u32c, _u32c, _u32c, _u32c, _P, _P, _P, _P, C.c_int, _u64c, _u8c, _P, _u64c, _u8c, _P,
                                              _P, _u64c])


def bitpack_write(values, bits) -> bytes:
    """BitWriter (lib/common/common/src/bitpacking.rs:11-57): write(value, bits)... finish()."""
    v, b = np.ascontiguousarray(values, dtype=np.uint64), np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.zeros(8 * len(v) + 8, dtype=np.uint8)
    n = _lib.qo_bitpack_write(_p(v), _p(b), len(v), _p(out), len(out))
    return out[:n].tobytes()


def bitpack_read(data: bytes, bits):
    """BitReader (bitpacking.rs:60-131): set_bits(b); read() per entry of `bits`."""
    d, b = np.frombuffer(data, dtype=np.uint8).copy(), np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.zeros(len(b), dtype=np.uint64)
    _lib.qo_bitpack_read(_p(d), len(d), _p(b), len(b), _p(out))
    return out


def pack_links(raw_links, bits_per_unsorted, sorted_count):
    """pack_links (bitpacking_links.rs:38-82) -> (bytes, raw_links as the reference leaves them: first sorted_count sorted)."""
    raw = np.ascontiguousarray(raw_links, dtype=np.uint32).copy()
    out = np.zeros(8 * len(raw) + 16, dtype=np.uint8)
    n = _lib.qo_pack_links(_p(raw), len(raw), bits_per_unsorted, sorted_count, _p(out), len(out))
    return out[:n].tobytes(), raw


def iterate_packed_links(data: bytes, bits_per_unsorted, sorted_count):
    """iterate_packed_links(..).collect() (bitpacking_links.rs:90-207)."""
    d = np.frombuffer(data, dtype=np.uint8).copy()
    out = np.zeros(max(1, len(d)), dtype=np.uint32)
    n = _lib.qo_iterate_packed_links(_p(d) if len(d) else None, len(d), bits_per_unsorted, sorted_count, _p(out), len(out))
    return out[:n].copy()


def packed_links_size(data: bytes, bits_per_unsorted, sorted_count, total_count):
    d = np.frombuffer(data, dtype=np.uint8).copy()
    return int(_lib.qo_packed_links_size(_p(d) if len(d) else None, len(d), bits_per_unsorted, sorted_count, total_count))


def ordered_compress(values):
    """bitpacking_ordered::compress (bitpacking_ordered.rs:69-73) -> (bytes, (base_bits, delta_bits, chunk_len_log2))."""
    v = np.ascontiguousarray(values, dtype=np.uint64)
    prm = np.zeros(3, dtype=np.uint8)
    n = _lib.qo_ordered_compress(_p(v) if len(v) else None, len(v), None, 0, _p(prm))
    out = np.zeros(n, dtype=np.uint8)
    _lib.qo_ordered_compress(_p(v) if len(v) else None, len(v), _p(out), n, _p(prm))
    return out.tobytes(), tuple(int(x) for x in prm)


def ordered_get(data: bytes, length, params, index):
    d = np.frombuffer(data, dtype=np.uint8).copy()
    return int(_lib.qo_ordered_get(_p(d), length, params[0], params[1], params[2], index))


def compressed_links_file(p: "PlainLinks", base_vectors=None, link_vectors=None, base_align=1, link_align=1) -> bytes:
    """serialize_graph_links(edges, Compressed | CompressedWithVectors, HnswM{m, m0}) (graph_links/serializer.rs:23-243) of the
    graph held in the plain arrays `p`.  With vectors: base_vectors [n, base_size] u8, link_vectors [n, link_size] u8."""
    re, lo = np.ascontiguousarray(p.reindex, dtype=np.uint32), np.ascontiguousarray(p.level_offsets, dtype=np.uint64)
    off, nb = np.ascontiguousarray(p.offsets, dtype=np.uint64), np.ascontiguousarray(p.neighbors, dtype=np.uint32)
    wv = base_vectors is not None
    bv = np.ascontiguousarray(base_vectors, dtype=np.uint8) if wv else None
    lv = np.ascontiguousarray(link_vectors, dtype=np.uint8) if wv else None
    args = [p.m, p.m0, len(re), len(lo) - 1, _p(re), _p(lo), _p(off), _p(nb) if len(nb) else None, 1 if wv else 0,
            bv.shape[1] if wv else 0, base_align, _p(bv), lv.shape[1] if wv else 0, link_align, _p(lv)]
    n = _lib.qo_links_serialize_compressed(*args, None, 0)
    out = np.zeros(n, dtype=np.uint8)
    _lib.qo_links_serialize_compressed(*args, _p(out), n)
    return out.tobytes()


# ---- multi-dense vectors: MaxSim (query_scorer/mod.rs:70-97) -------------------------------------------------------------
_sig("qo_max_similarity", _f, [_P, C.c_uint32, C.c_uint32, C.c_uint64])


def max_similarity(sims):
    """score_max_similarity over sims[a, b] = similarity(query inner vector a, point inner vector b)."""
    t = np.ascontiguousarray(sims, dtype=np.float32)
    return np.float32(_lib.qo_max_similarity(_p(t), t.shape[0], t.shape[1], t.shape[1]))


def multi_scores(storage: "DenseStorage", inner_queries, query_first, point_offsets, ids):
    """MultiMetricQueryScorer::score_stored over the oracle: inner_queries [nqi, dim] ORIGINAL vectors, multi-query j = inner queries
    [query_first[j], query_first[j + 1]); point p = inner rows [point_offsets[p], point_offsets[p + 1]) of `storage`."""
    n_rows = storage.rows.shape[0]
    sims = storage.score_points(inner_queries, np.arange(n_rows, dtype=np.uint32))       # [nqi, n_rows], the leaves' bits
    out = np.empty((len(query_first) - 1, len(ids)), dtype=np.float32)
    for j in range(len(query_first) - 1):
        for c, p in enumerate(ids):
            out[j, c] = max_similarity(sims[query_first[j]:query_first[j + 1], int(point_offsets[p]):int(point_offsets[p + 1])])
    return out


# ---- TurboQuant (oracle/qdrant_oracle_tq.c) ---------------------------------------------------------------------------
TQ_BITS4, TQ_BITS2, TQ_BITS1_5, TQ_BITS1 = range(4)
_sig("qo_tq_padded_dim_for", C.c_uint32, [C.c_uint32, C.c_int])
_sig("qo_tq_permutation_map", None, [C.c_uint64, C.c_uint32, _P])
_sig("qo_tq_chunk_sizes", C.c_uint32, [C.c_uint32, _P])
_sig("qo_tq_wht", None, [_P, C.c_uint32])
_sig("qo_tq_new", _P, [C.c_uint32, C.c_int, C.c_int, C.c_int])
_sig("qo_tq_new_plus", _P, [C.c_uint32, C.c_int, C.c_int, C.c_int, _P, _P])
_sig("qo_tq_query_ec_correction", _f, [_P])
_sig("qo_tq_free", None, [_P])
_sig("qo_tq_padded_dim", C.c_uint32, [_P])
_sig("qo_tq_quantized_size", C.c_uint32, [_P])
_sig("qo_tq_rotate", None, [_P, _P])
_sig("qo_tq_quantize", None, [_P, _P, _P])
_sig("qo_tq_precompute_query", _P, [_P, _P])
_sig("qo_tq_query_free", None, [_P])
_sig("qo_tq_query_export", None, [_P, _P, _P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int64)])
_sig("qo_tq_score_precomputed", _f, [_P, _P, _P])
_sig("qo_tq_score_symmetric", _f, [_P, _P, _P])
_sig("qo_tq_dequantize", None, [_P, _P, _P])
_sig("qo_tq_rotate_inverse", None, [_P, _P])


class TqOracle:
    """TurboQuantizer (TQMode::Normal) + the EncodedVectorsTQ glue (`invert`)."""

    def __init__(self, distance, dim, bits, rotation_unpadded=False, invert=None, shift=None, scale=None):
        self.distance, self.dim, self.bits = distance, dim, bits
        self.invert = (distance in (EUCLID, MANHATTAN)) if invert is None else bool(invert)   # quantized_vectors.rs:232
        if shift is not None:                           # TQMode::Plus with the given ErrorCorrection
            self.shift, self.scale = f32(shift), f32(scale)
            self.h = _lib.qo_tq_new_plus(dim, bits, distance, 1 if rotation_unpadded else 0, _p(self.shift), _p(self.scale))
        else:
            self.shift = self.scale = None
            self.h = _lib.qo_tq_new(dim, bits, distance, 1 if rotation_unpadded else 0)
        self.padded_dim = _lib.qo_tq_padded_dim(self.h)
        self.row_bytes = _lib.qo_tq_quantized_size(self.h)
        self.rows = None

    def __del__(self):
        try:
            _lib.qo_tq_free(self.h)
        except Exception:
            pass

    def rotate(self, x):
        buf = np.zeros(self.padded_dim, dtype=np.float64)
        buf[:len(x)] = x
        _lib.qo_tq_rotate(self.h, _p(buf))
        return buf

    def encode_rows(self, vectors):
        v = f32(np.atleast_2d(vectors))
        out = np.zeros((v.shape[0], self.row_bytes), dtype=np.uint8)
        for i in range(v.shape[0]):
            _lib.qo_tq_quantize(self.h, _p(v[i]), _p(out[i]))
        self.rows = out
        return out

    def query(self, q):
        """(q_signed [padded_dim] i32, postprocess_scale, l2_norm, sum_q) of precompute_query"""
        e = _lib.qo_tq_precompute_query(self.h, _p(f32(q)))
        qs = np.zeros(self.padded_dim, dtype=np.int32)
        ps, l2, sq = C.c_float(), C.c_float(), C.c_int64()
        _lib.qo_tq_query_export(self.h, e, _p(qs), C.byref(ps), C.byref(l2), C.byref(sq))
        _lib.qo_tq_query_free(e)
        return qs, np.float32(ps.value), np.float32(l2.value), sq.value

    def score_points(self, queries, ids):
        """EncodedVectorsTQ::score_point for every (query, id): ORIGINAL (un-rotated) query vectors, as the storage stores them."""
        q = f32(np.atleast_2d(queries))
        out = np.empty((q.shape[0], len(ids)), dtype=np.float32)
        for i in range(q.shape[0]):
            e = _lib.qo_tq_precompute_query(self.h, _p(q[i]))
            for j, p in enumerate(ids):
                s = _lib.qo_tq_score_precomputed(self.h, e, _p(self.rows[int(p)]))
                out[i, j] = -s if self.invert else s
            _lib.qo_tq_query_free(e)
        return out

    def dequantize(self, row, rotate_back=True):
        """TurboQuantizer::dequantize::<f64> (+ apply_inverse_rotation): [padded_dim] f64"""
        out = np.zeros(self.padded_dim, dtype=np.float64)
        _lib.qo_tq_dequantize(self.h, _p(np.ascontiguousarray(row, dtype=np.uint8)), _p(out))
        if rotate_back:
            _lib.qo_tq_rotate_inverse(self.h, _p(out))
        return out

    def rotate_inverse(self, x):
        buf = np.array(x, dtype=np.float64)
        _lib.qo_tq_rotate_inverse(self.h, _p(buf))
        return buf

    def score_internal(self, a, b):
        out = np.empty(len(a), dtype=np.float32)
        for k, (i, j) in enumerate(zip(a, b)):
            s = _lib.qo_tq_score_symmetric(self.h, _p(self.rows[int(i)]), _p(self.rows[int(j)]))
            out[k] = -s if self.invert else s
        return out


_sig("qo_vector_stats", None, [_P, C.c_uint64, C.c_uint32, _P, _P, _P, _P])


def vector_stats(rows):
    """VectorStats::build: (min, max, mean, stddev) per dimension, streaming Welford over the rows in order."""
    v = f32(np.atleast_2d(rows))
    n, dim = v.shape
    out = [np.zeros(dim, dtype=np.float32) for _ in range(4)]
    _lib.qo_vector_stats(_p(v) if n else None, n, dim, *[_p(o) for o in out])
    return tuple(out)


_sig("qo_tq_plus_quantiles", None, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_float)])
_sig("qo_tq_preprocess", None, [_P, _P, _P])
_sig("qo_p2_quantile", C.c_double, [C.c_double, _P, C.c_uint64, _P])


def p2_quantile(q, values, with_grid=False):
    v = np.ascontiguousarray(values, dtype=np.float64)
    grid = np.zeros(7, dtype=np.float64)
    r = _lib.qo_p2_quantile(float(q), _p(v) if len(v) else None, len(v), _p(grid))
    return (r, grid) if with_grid else r

_sig("qo_tq_plus_fit", None, [_P, _P, C.c_uint32, _P, _P])


def tq_plus_fit_p2(distance, dim, bits, sample, rotation_unpadded=False):
    """shift / scale of TQ+ as the reference's first pass estimates them: one pair of 7-marker P-square estimators per rotated coordinate, fed
    with the sampled vectors in iteration order (which vectors: the reference's Permutor - an input here)."""
    plain = TqOracle(distance, dim, bits, rotation_unpadded=rotation_unpadded)
    v = f32(np.atleast_2d(sample))
    shift, scale = np.zeros(plain.padded_dim, dtype=np.float32), np.zeros(plain.padded_dim, dtype=np.float32)
    _lib.qo_tq_plus_fit(plain.h, _p(v), v.shape[0], _p(shift), _p(scale))
    return shift, scale


def tq_plus_quantiles(bits):
    a, b, c = C.c_double(), C.c_double(), C.c_float()
    _lib.qo_tq_plus_quantiles(bits, C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, np.float32(c.value)


def tq_plus_fit(distance, dim, bits, vectors, rotation_unpadded=False):
    """shift / scale of TQ+ as EncodedVectorsTQ::encode derives them (encoded_vectors_tq.rs:156-240) from per-coordinate quantiles of the rotated,
    length-rescaled vectors at Phi(+-c_outer) - here exact quantiles of the given vectors instead of the reference's P-square estimates over a
    random sample (the parameters are an INPUT of the scorer either way: they are persisted in the metadata)."""
    from math import erf, sqrt
    plain = TqOracle(distance, dim, bits, rotation_unpadded=rotation_unpadded)
    pd = plain.padded_dim
    rot = np.stack([plain.rotate(v.astype(np.float64)) for v in f32(vectors)])
    if distance != COSINE:
        ln = np.sqrt((rot * rot).sum(axis=1, keepdims=True))
        rot = np.where(ln > 0, rot * (np.sqrt(pd) / np.where(ln > 0, ln, 1.0)), rot)
    else:
        rot = rot * np.sqrt(pd)
    c_outer = {TQ_BITS4: 2.733, TQ_BITS2: 1.510, TQ_BITS1_5: 0.7978846, TQ_BITS1: 0.7978846}[bits]
    p = 0.5 * (1.0 + erf(c_outer / sqrt(2.0)))
    q_lo, q_hi = np.quantile(rot.astype(np.float32), 1.0 - p, axis=0), np.quantile(rot.astype(np.float32), p, axis=0)
    shift = (-(q_lo + q_hi) / 2.0).astype(np.float32)
    denom = (q_hi - q_lo).astype(np.float32)
    scale = np.where(denom > 1e-3, np.float32(2.0 * c_outer) / np.where(denom > 1e-3, denom, 1.0), 1.0).astype(np.float32)
    return shift, scale


# ---- scorers as values: any storage kind, custom queries over them (qo_scorer kind 6) ----------------------------------
class ScorerFactory:
    """One qo_scorer per (preprocessed) query vector over a storage of the oracle:
      ("dense", DenseStorage) | ("sq", flags DenseStorage, SqOracle) | ("pq", flags, PqOracle) | ("bq", flags, BqOracle) | ("tq", flags, TqOracle)
      | ("multi", MultiOracle)  (the "query vector" is then a [tokens, dim] multi-vector).
    Returned scorers come with what they point at (keep it alive as long as the scorer is used)."""

    def __init__(self, *spec):
        self.spec, self.kind = spec, spec[0]
        self.flags = spec[1].points if self.kind == "multi" else spec[1]
        self._tq_queries = []

    def __del__(self):
        for e in self._tq_queries:
            _lib.qo_tq_query_free(e)

    def scorer(self, qpre):
        s = Scorer()
        if self.kind == "multi":
            return self.spec[1].scorer(qpre)
        st = self.spec[1]
        qv = f32(qpre)
        if self.kind == "dense":
            enc = np.ascontiguousarray(cast(st.dtype, qv[None, :]))[0]
            s.kind, s.st, s.query = 0, C.pointer(st.st), enc.ctypes.data
            return s, enc
        quant = self.spec[2]
        if self.kind == "sq":
            codes, off = quant.encode_query(qv)
            s.kind, s.st, s.sq, s.sq_rows = 1, C.pointer(st.st), C.pointer(quant.sq), quant.rows.ctypes.data
            s.sq_query, s.sq_query_offset, s.isa = codes.ctypes.data, off, quant.isa
            return s, codes
        if self.kind == "bq":
            qb = quant.encode(qv[None, :])[0]
            s.kind, s.st, s.bq_rows, s.bq_query = 3, C.pointer(st.st), quant.rows.ctypes.data, qb.ctypes.data
            s.bq_dim, s.bq_distance, s.bq_invert = quant.dim, quant.distance, quant.invert
            return s, qb
        if self.kind == "pq":
            lut = quant.lut(qv)
            s.kind, s.st, s.pq, s.pq_codes = 2, C.pointer(st.st), C.pointer(quant.pq), quant.codes.ctypes.data
            s.pq_lut, s.isa = lut.ctypes.data, quant.isa
            return s, lut
        e = _lib.qo_tq_precompute_query(quant.h, _p(qv))
        self._tq_queries.append(e)
        s.kind, s.st, s.tq, s.tq_rows, s.tq_query, s.tq_invert = 5, C.pointer(st.st), quant.h, quant.rows.ctypes.data, e, 1 if quant.invert else 0
        return s, qv

    def custom(self, examples_pre, kind, n_a, n_b, coefs=None):
        """qo_scorer kind 6: a custom query over this storage; `examples_pre` in flat_iter() order, preprocessed."""
        ex = [self.scorer(e) for e in examples_pre]
        arr = (Scorer * max(1, len(ex)))(*[e[0] for e in ex])
        cf = None if coefs is None else f32(coefs)
        s = Scorer()
        s.kind, s.st = 6, C.pointer(self.flags.st)
        s.cq_examples, s.cq_kind, s.cq_n_a, s.cq_n_b = C.addressof(arr), kind, n_a, n_b
        s.cq_coefs = None if cf is None else cf.ctypes.data
        return s, (arr, ex, cf)


def scorer_score_points(scorer, ids):
    return np.array([_lib.qo_scorer_score_point(C.byref(scorer), int(i)) for i in ids], dtype=np.float32)


def _hnsw_search_scorer(self, scorer, top, ef):
    """GraphLayers::search with any qo_scorer as the points scorer -> (results, points scored)"""
    return self._run(scorer, top, ef)


Hnsw.search_scorer = _hnsw_search_scorer

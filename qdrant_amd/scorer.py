"""Host-side mirror of the reference's scorer interface on top of the C-ABI.

Same names, argument meaning and error behaviour as
  lib/segment/src/vector_storage/raw_scorer.rs:39-58      (trait RawScorer, new_raw_scorer)
  lib/segment/src/index/hnsw_index/point_scorer.rs:307-472 (BatchFilteredSearcher)
  lib/common/common/src/types.rs:12-31                     (ScoredPointOffset)
so that tests/ reads like the reference's own tests.  Everything here is plumbing: all scoring
happens in libqdrant_amd.so on the GPU; there is no CPU path.
"""
import ctypes as C
import enum
import os
from typing import Iterable, List, Optional, Sequence

import numpy as np

from . import _ffi as F


class Distance(enum.IntEnum):  # lib/segment/src/types.rs:313-322
    Cosine = F.COSINE
    Euclid = F.EUCLID
    Dot = F.DOT
    Manhattan = F.MANHATTAN


class VectorStorageDatatype(enum.IntEnum):
    Float32 = F.DTYPE_F32
    Float16 = F.DTYPE_F16
    Uint8 = F.DTYPE_U8


ScoredPointOffset = np.dtype([("idx", np.uint32), ("score", np.float32)])  # types.rs:12-17, 8 bytes

_NP_ELEM = {F.DTYPE_F32: np.float32, F.DTYPE_F16: np.float16, F.DTYPE_U8: np.uint8}


def device_count() -> int:
    n = C.c_int32(0)
    F.check(F.lib().qmx_device_count(C.byref(n)))
    return n.value


def _bits_to_words(bits) -> Optional[np.ndarray]:
    """bool array -> BitSlice<u64, Lsb0> words (lib/common/common/src/bitvec.rs:6-7)."""
    if bits is None:
        return None
    bits = np.asarray(bits, dtype=bool)
    pad = (-len(bits)) % 64
    b = np.concatenate([bits, np.zeros(pad, dtype=bool)]) if pad else bits
    return np.packbits(b.reshape(-1, 8), axis=1, bitorder="little").reshape(-1).view(np.uint64).copy()


class VectorStorage:
    """Device-resident dense vector storage (the read side of `DenseVectorStorageRead`,
    lib/segment/src/vector_storage/vector_storage_base.rs:265-316).

    `vectors` are the STORED rows, i.e. already passed through `Distance::preprocess_vector`
    at insert time (lib/segment/src/data_types/named_vectors.rs:350-368) and cast to `datatype`.
    A numpy array is uploaded; a torch CUDA tensor is adopted in place.
    """

    def __init__(self, vectors, distance: Distance, datatype: VectorStorageDatatype = VectorStorageDatatype.Float32,
                 device_id: int = 0, flags: int = 0, dim: Optional[int] = None):
        self._h = C.c_void_p()
        self.distance = Distance(distance)
        self.datatype = VectorStorageDatatype(datatype)
        on_device = hasattr(vectors, "data_ptr") and getattr(vectors, "is_cuda", False)
        if not on_device:
            vectors = np.ascontiguousarray(vectors, dtype=_NP_ELEM[int(datatype)])
        self._keep = vectors
        n = int(vectors.shape[0])
        self.dim = int(dim if dim is not None else vectors.shape[1])
        self.count = n
        desc = F.SegmentDesc()
        desc.dtype = int(datatype)
        desc.distance = int(distance)
        desc.dim = self.dim
        desc.flags = flags | (F.SEG_DATA_ON_DEVICE if on_device else 0)
        desc.n = n
        desc.row_stride_bytes = 0
        desc.data = F.ptr(vectors)
        desc.device_id = device_id
        F.check(F.lib().qmx_segment_create(C.byref(desc), C.byref(self._h)))
        if on_device is False:
            self._keep = None  # uploaded: the host copy may go away (ownership)

    @classmethod
    def from_files(cls, vectors_path: str, dim: int, distance: Distance, datatype: VectorStorageDatatype = VectorStorageDatatype.Float32,
                   deleted_path: Optional[str] = None, device_id: int = 0, flags: int = 0):
        """Open the reference's immutable dense vector file ("data" header + rows) and, optionally, its "drop" flags file
        (dense/immutable_dense_vectors.rs:25-27, 90, 364-378) straight onto the device (`qmx_segment_create_from_files`)."""
        self = cls.__new__(cls)
        self._h, self._keep = C.c_void_p(), None
        self.distance, self.datatype, self.dim = Distance(distance), VectorStorageDatatype(datatype), int(dim)
        desc = F.SegmentDesc()
        desc.dtype, desc.distance, desc.dim, desc.flags, desc.n, desc.device_id = int(datatype), int(distance), int(dim), flags, 0, device_id
        F.check(F.lib().qmx_segment_create_from_files(C.byref(desc), os.fsencode(vectors_path),
                                                     None if deleted_path is None else os.fsencode(deleted_path), C.byref(self._h)))
        rb = C.c_uint64()
        F.check(F.lib().qmx_segment_row_bytes(self._h, C.byref(rb)))
        self.count = (os.path.getsize(vectors_path) - 4) // rb.value
        return self

    def total_vector_count(self) -> int:
        return self.count

    def info(self) -> dict:
        """`qmx_segment_get_info`: which derived copy the prefilter streams ("i8" / "half" / "pair" / None) and, under `SEG_AUTO_COPY`, what the
        trial at create measured."""
        i = F.SegmentInfo()
        F.check(F.lib().qmx_segment_get_info(self._h, C.byref(i)))
        name = {0: None, F.SEG_I8_COPY: "i8", F.SEG_HALF_COPY: "half", F.SEG_SPLIT_COPY: "pair"}[i.derived_copy]
        return {"derived_copy": name, "chosen_by_trial": bool(i.chosen_by_trial), "derived_copy_bytes": int(i.derived_copy_bytes),
                "i8_scale_balance": float(i.i8_scale_balance), "trial_i8_ms": float(i.trial_i8_ms), "trial_half_ms": float(i.trial_half_ms),
                "trial_i8_verified_rows": float(i.trial_i8_verified_rows), "trial_i8_fallback_queries": int(i.trial_i8_fallback_queries)}

    def set_deleted(self, point_deleted=None, vec_deleted=None):
        """`NotDeletedChecker{point_deleted, vec_deleted}` (raw_scorer.rs:580-603); bool arrays."""
        pw, vw = _bits_to_words(point_deleted), _bits_to_words(vec_deleted)
        F.check(F.lib().qmx_segment_set_deleted(
            self._h, F.ptr(pw), 0 if point_deleted is None else len(point_deleted),
            F.ptr(vw), 0 if vec_deleted is None else len(vec_deleted)))

    def get_dense(self, ids: Sequence[int]) -> np.ndarray:
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.empty((len(ids), self.dim), dtype=_NP_ELEM[int(self.datatype)])
        F.check(F.lib().qmx_segment_read_rows(self._h, F.ptr(ids), len(ids), F.ptr(out)))
        return out

    def close(self):
        if self._h:
            F.lib().qmx_segment_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ScalarQuantizer:
    """`MetadataInt8` + `VectorParameters` of `EncodedVectorsU8` (lib/quantization/src/encoded_vectors_u8.rs:84-91):
    the GIVEN (alpha, offset) — parity is defined on given parameters because the reference's quantile
    estimate samples randomly (quantile.rs:35-82)."""

    def __init__(self, dim: int, distance: Distance, alpha: float, offset: float):
        self.dim = int(dim)
        self.distance = Distance(distance)
        self.alpha = np.float32(alpha)
        self.offset = np.float32(offset)
        self.actual_dim = (self.dim + 15) // 16 * 16                       # get_actual_dim :622-624
        # invert: Euclid / Manhattan (vector_storage/quantized/quantized_vectors.rs:232)
        self.invert = self.distance in (Distance.Euclid, Distance.Manhattan)
        a = self.alpha
        if self.distance in (Distance.Dot, Distance.Cosine):
            m = a * a                                                      # :205-221
        elif self.distance == Distance.Manhattan:
            m = a
        else:
            m = np.float32(-2.0) * a * a
        self.multiplier = np.float32(-m if self.invert else m)

    @classmethod
    def from_min_max(cls, data, dim: int, distance: Distance):
        """quantile = None: alpha_offset_from_min_max over the whole data (:523-533), deterministic."""
        d = np.asarray(data, dtype=np.float32)
        mn, mx = np.float32(d.min()), np.float32(d.max())
        return cls(dim, distance, (mx - mn) / np.float32(127.0), mn)

    @classmethod
    def fit(cls, data, dim: int, distance: Distance, device_id: int = 0):
        """Same fit on the device (`qmx_sq_fit_min_max`); `data`: numpy [n, dim] f32 or a torch CUDA tensor."""
        on_device = hasattr(data, "data_ptr") and getattr(data, "is_cuda", False)
        d = data if on_device else np.ascontiguousarray(data, dtype=np.float32)
        p = F.SqParams()
        F.check(F.lib().qmx_sq_fit_min_max(device_id, int(distance), F.ptr(d), int(d.shape[0]), int(dim), C.byref(p)))
        return cls(dim, distance, p.alpha, p.offset)

    @classmethod
    def fit_quantile(cls, data, dim: int, distance: Distance, quantile: float, sample=None, count: Optional[int] = None, device_id: int = 0):
        """`EncodedVectorsU8::encode` with `quantile = Some(q)` (encoded_vectors_u8.rs:193-208): the interval of
        `find_quantile_interval` (quantile.rs:35-84) over `sample` (default: the first SAMPLE_SIZE = 5 000 rows of `data`; the
        reference samples at random), falling back to the min / max fit where the reference does."""
        d = np.ascontiguousarray(data, dtype=np.float32)
        smp = d[:5000] if sample is None else np.ascontiguousarray(sample, dtype=np.float32)
        p, found = F.SqParams(), C.c_int32(0)
        F.check(F.lib().qmx_sq_fit_quantile(device_id, int(distance), F.ptr(smp), int(smp.shape[0]), int(dim),
                                            int(d.shape[0] if count is None else count), float(quantile), C.byref(p), C.byref(found)))
        if not found.value:
            return cls.fit(d, dim, distance, device_id)
        return cls(dim, distance, p.alpha, p.offset)

    def params(self) -> "F.SqParams":
        p = F.SqParams()
        p.actual_dim = self.actual_dim
        p.alpha = float(self.alpha)
        p.offset = float(self.offset)
        p.multiplier = float(self.multiplier)
        p.invert = 1 if self.invert else 0
        return p

    def quantized_vector_size(self) -> int:
        return self.actual_dim + 4

    def encode(self, vectors, device_id: int = 0) -> np.ndarray:
        """`EncodedVectorsU8::encode` row loop (:236-296) on device: [n, dim] f32 -> [n, 4 + actual_dim] u8 rows."""
        v = np.ascontiguousarray(vectors, dtype=np.float32)
        out = np.empty((v.shape[0], self.quantized_vector_size()), dtype=np.uint8)
        p = self.params()
        F.check(F.lib().qmx_sq_encode(device_id, int(self.distance), C.byref(p), F.ptr(v), v.shape[0], self.dim, F.ptr(out)))
        return out


class EncodedVectorsU8(VectorStorage):
    """Device-resident `EncodedVectorsU8` storage = what `QuantizedVectors::raw_scorer` scores against
    (vector_storage/quantized/quantized_vectors.rs:65-81).  `rows`: [n, 4 + actual_dim] u8 in the reference
    layout `[f32 vector_offset][codes]`."""

    def __init__(self, rows, quantizer: ScalarQuantizer, device_id: int = 0):
        self._h = C.c_void_p()
        self.quantizer = quantizer
        self.distance = quantizer.distance
        self.datatype = None
        on_device = hasattr(rows, "data_ptr") and getattr(rows, "is_cuda", False)   # a torch CUDA tensor: split on the device, no host copy
        if not on_device:
            rows = np.ascontiguousarray(rows, dtype=np.uint8)
        assert rows.shape[1] == quantizer.quantized_vector_size()
        self.dim = quantizer.dim
        self.count = int(rows.shape[0])
        self._keep = None
        self._sq = quantizer.params()
        desc = F.SegmentDesc()
        desc.dtype = F.DTYPE_SQ_U8
        desc.distance = int(quantizer.distance)
        desc.dim = quantizer.dim
        desc.flags = 0
        desc.n = self.count
        desc.row_stride_bytes = 0
        desc.data = F.ptr(rows)
        desc.device_id = device_id
        desc.sq = C.pointer(self._sq)
        F.check(F.lib().qmx_segment_create(C.byref(desc), C.byref(self._h)))

    def get_quantized_vector(self, ids: Sequence[int]) -> np.ndarray:
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.empty((len(ids), self.quantizer.quantized_vector_size()), dtype=np.uint8)
        F.check(F.lib().qmx_segment_read_rows(self._h, F.ptr(ids), len(ids), F.ptr(out)))
        return out


class ProductQuantizer:
    """`Metadata{centroids, vector_division, vector_parameters}` of `EncodedVectorsPQ`
    (lib/quantization/src/encoded_vectors_pq.rs:46-51).  Centroids are an INPUT: the reference's k-means
    re-seeds empty clusters randomly (kmeans.rs:113-120), so parity is defined on given centroids."""

    def __init__(self, dim: int, distance: Distance, chunk_size: int, centroids, lut_mfma: bool = False):
        self.dim = int(dim)
        self.distance = Distance(distance)
        self.chunk_size = int(chunk_size)
        self.centroids = np.ascontiguousarray(centroids, dtype=np.float32)   # [n_centroids, dim], flattened by chunks
        assert self.centroids.ndim == 2 and self.centroids.shape[1] == self.dim
        self.n_centroids = int(self.centroids.shape[0])
        self.m = (self.dim + self.chunk_size - 1) // self.chunk_size          # get_vector_division :164-169
        self.invert = self.distance in (Distance.Euclid, Distance.Manhattan)
        self.lut_mfma = bool(lut_mfma)

    def params(self) -> "F.PqParams":
        p = F.PqParams()
        p.chunk_size = self.chunk_size
        p.n_centroids = self.n_centroids
        p.centroids = self.centroids.ctypes.data
        p.invert = 1 if self.invert else 0
        p.lut_mfma = 1 if self.lut_mfma else 0
        return p

    def quantized_vector_size(self) -> int:
        return self.m

    def encode(self, vectors, device_id: int = 0) -> np.ndarray:
        """`EncodedVectorsPQ::encode_vector` (:301-329) on device: [n, dim] f32 -> [n, m] u8 codes."""
        v = np.ascontiguousarray(vectors, dtype=np.float32)
        out = np.empty((v.shape[0], self.m), dtype=np.uint8)
        p = self.params()
        F.check(F.lib().qmx_pq_encode(device_id, C.byref(p), F.ptr(v), v.shape[0], self.dim, F.ptr(out)))
        return out


def pq_train(sample, dim: int, chunk_size: int, n_centroids: int = 256, max_iterations: int = 100, accuracy: float = 1e-5,
             threads: int = 1, device_id: int = 0):
    """`find_centroids` / `kmeans` on a given sample, on the device (qmx_pq_train) -> (centroids [n_centroids, dim], iterations [m])."""
    s = np.ascontiguousarray(sample, dtype=np.float32)
    cen = np.zeros((n_centroids, dim), dtype=np.float32)
    m = (dim + chunk_size - 1) // chunk_size
    iters = np.zeros(m, dtype=np.uint32)
    F.check(F.lib().qmx_pq_train(device_id, F.ptr(s), s.shape[0], dim, chunk_size, n_centroids, max_iterations, float(accuracy), threads,
                                 F.ptr(cen), F.ptr(iters)))
    return cen, iters


class EncodedVectorsPQ(VectorStorage):
    """Device-resident `EncodedVectorsPQ` storage: rows = [n, m] u8 centroid indices."""

    def __init__(self, codes, quantizer: ProductQuantizer, device_id: int = 0):
        self._h = C.c_void_p()
        self.quantizer = quantizer
        self.distance = quantizer.distance
        self.datatype = None
        on_device = hasattr(codes, "data_ptr") and getattr(codes, "is_cuda", False)   # a torch CUDA tensor is adopted in place
        if not on_device:
            codes = np.ascontiguousarray(codes, dtype=np.uint8)
        assert codes.shape[1] == quantizer.m
        self.dim = quantizer.dim
        self.count = int(codes.shape[0])
        self._keep = codes if on_device else None
        self._pq = quantizer.params()
        desc = F.SegmentDesc()
        desc.dtype = F.DTYPE_PQ
        desc.distance = int(quantizer.distance)
        desc.dim = quantizer.dim
        desc.flags = F.SEG_DATA_ON_DEVICE if on_device else 0
        desc.n = self.count
        desc.row_stride_bytes = 0
        desc.data = F.ptr(codes)
        desc.device_id = device_id
        desc.pq = C.pointer(self._pq)
        F.check(F.lib().qmx_segment_create(C.byref(desc), C.byref(self._h)))

    def get_quantized_vector(self, ids: Sequence[int]) -> np.ndarray:
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.empty((len(ids), self.quantizer.m), dtype=np.uint8)
        F.check(F.lib().qmx_segment_read_rows(self._h, F.ptr(ids), len(ids), F.ptr(out)))
        return out


def vector_stats(vectors, dim: int, device_id: int = 0):
    """`VectorStats::build` (vector_stats.rs): (min, max, mean, stddev) of every dimension over the vectors in order, on the device."""
    on_device = hasattr(vectors, "data_ptr") and getattr(vectors, "is_cuda", False)
    if not on_device:
        vectors = np.ascontiguousarray(vectors, dtype=np.float32).reshape(-1, dim)
    n = int(vectors.shape[0])
    out = [np.empty(dim, dtype=np.float32) for _ in range(4)]
    F.check(F.lib().qmx_vector_stats(device_id, F.ptr(vectors) if n else None, n, dim, *[F.ptr(o) for o in out]))
    return tuple(out)


class BinaryQuantizer:
    """`Metadata{vector_parameters, encoding, query_encoding: SameAsStorage, vector_stats}` of `EncodedVectorsBin<u128>`
    (lib/quantization/src/encoded_vectors_binary.rs:43-78).  `invert` defaults to the segment's choice (quantized_vectors.rs:232:
    Euclid | Manhattan).  `encoding`: 0 one bit, 1 two bits, 2 one and a half bits; the latter two take the per-dimension
    `mean` / `stddev` of the storage (`VectorStats`, an input like the SQ interval), None = no stats."""

    def __init__(self, dim: int, distance: Distance, invert: Optional[bool] = None, encoding: int = 0, mean=None, stddev=None,
                 query_encoding: int = 0):
        self.query_encoding = int(query_encoding)     # QueryEncoding: 0 SameAsStorage, 1 Scalar4bits, 2 Scalar8bits
        self.dim = int(dim)
        self.distance = Distance(distance)
        natural = self.distance in (Distance.Euclid, Distance.Manhattan)
        self.invert = natural if invert is None else bool(invert)
        self._toggle = self.invert != natural
        self.encoding = int(encoding)
        self.mean = None if mean is None else np.ascontiguousarray(mean, dtype=np.float32)
        self.stddev = None if stddev is None else np.ascontiguousarray(stddev, dtype=np.float32)

    @classmethod
    def fit(cls, vectors, dim: int, distance: Distance, encoding: int, invert: Optional[bool] = None, query_encoding: int = 0,
            device_id: int = 0) -> "BinaryQuantizer":
        """The quantizer of a new storage: `VectorStats::build` over ALL its vectors in order (qmx_vector_stats on the device: streaming Welford in
        f64, the oracle's bits) for Encoding::TwoBits / OneAndHalfBits; Encoding::OneBit needs no statistics.  `vectors`: numpy or a torch CUDA tensor."""
        if int(encoding) == 0:
            return cls(dim, distance, invert, 0, None, None, query_encoding)
        mean, stddev = vector_stats(vectors, dim, device_id)[2:]
        return cls(dim, distance, invert, encoding, mean, stddev, query_encoding)

    def params(self) -> "F.BqParams":
        p = F.BqParams()
        p.encoding = self.encoding
        p.query_encoding = self.query_encoding
        p.mean = None if self.mean is None else self.mean.ctypes.data
        p.stddev = None if self.stddev is None else self.stddev.ctypes.data
        return p

    def quantized_vector_size(self) -> int:
        """get_quantized_vector_size_from_params::<u128>(dim, encoding) (:829-840)."""
        return int(F.lib().qmx_bq_row_bytes(self.dim, self.encoding))

    def encode(self, vectors, device_id: int = 0) -> np.ndarray:
        """`encode_vector` (:535-672) on device: [n, dim] f32 -> [n, quantized_vector_size] bytes."""
        v = np.ascontiguousarray(vectors, dtype=np.float32)
        out = np.empty((v.shape[0], self.quantized_vector_size()), dtype=np.uint8)
        p = self.params()
        F.check(F.lib().qmx_bq_encode_ex(device_id, C.byref(p), F.ptr(v), v.shape[0], self.dim, F.ptr(out)))
        return out


class EncodedVectorsBin(VectorStorage):
    """Device-resident `EncodedVectorsBin<u128>` storage (1 bit per dimension): rows = [n, ceil(dim / 128) * 16] bytes."""

    def __init__(self, rows, quantizer: BinaryQuantizer, device_id: int = 0):
        self._h = C.c_void_p()
        self.quantizer = quantizer
        self.distance = quantizer.distance
        self.datatype = None
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        assert rows.shape[1] == quantizer.quantized_vector_size()
        self.dim = quantizer.dim
        self.count = int(rows.shape[0])
        self._keep = None
        desc = F.SegmentDesc()
        desc.dtype = F.DTYPE_BQ
        desc.distance = int(quantizer.distance)
        desc.dim = quantizer.dim
        desc.flags = F.SEG_BQ_TOGGLE_INVERT if quantizer._toggle else 0
        self._bq = quantizer.params()
        desc.bq = C.pointer(self._bq)
        desc.n = self.count
        desc.row_stride_bytes = 0
        desc.data = F.ptr(rows)
        desc.device_id = device_id
        F.check(F.lib().qmx_segment_create(C.byref(desc), C.byref(self._h)))

    def get_quantized_vector(self, ids: Sequence[int]) -> np.ndarray:
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.empty((len(ids), self.quantizer.quantized_vector_size()), dtype=np.uint8)
        F.check(F.lib().qmx_segment_read_rows(self._h, F.ptr(ids), len(ids), F.ptr(out)))
        return out


class TurboQuantizer:
    """`Metadata{vector_parameters, bits, mode: Normal, rotation}` of `EncodedVectorsTQ` (lib/quantization/src/encoded_vectors_tq.rs:33-46) =
    `TurboQuantizer::new(dim, bits, TQMode::Normal, distance, rotation, None)` (turboquant/quantization.rs:127-158).  `bits`: 0 = Bits4,
    1 = Bits2, 2 = Bits1_5, 3 = Bits1 (TQBits).  Rows are produced by the reference's `quantize` (an input here, like PQ codes)."""

    def __init__(self, dim: int, distance: Distance, bits: int, rotation_unpadded: bool = False, invert: Optional[bool] = None, shift=None, scale=None):
        self.dim, self.distance, self.bits = int(dim), Distance(distance), int(bits)
        # TQMode::Plus: the storage's persisted ErrorCorrection (shift / scale per rotated coordinate), or None = TQMode::Normal
        self.shift = None if shift is None else np.ascontiguousarray(shift, dtype=np.float32)
        self.scale = None if scale is None else np.ascontiguousarray(scale, dtype=np.float32)
        self.plus_mode = self.shift is not None
        self.rotation_unpadded = bool(rotation_unpadded)
        self.invert = (self.distance in (Distance.Euclid, Distance.Manhattan)) if invert is None else bool(invert)
        value_bits = {0: 4, 1: 2, 2: 1, 3: 1}[self.bits]
        mult = {0: 2, 1: 4, 2: 8, 3: 8}[self.bits]
        d = self.dim * 3 // 2 if self.bits == 2 else self.dim
        self.padded_dim = (d + mult - 1) // mult * mult
        self.code_bytes = self.padded_dim * value_bits // 8

    def params(self) -> "F.TqParams":
        p = F.TqParams()
        p.bits, p.rotation_unpadded, p.invert = self.bits, 1 if self.rotation_unpadded else 0, 1 if self.invert else 0
        p.plus_mode = 1 if getattr(self, "plus_mode", False) else 0
        if self.shift is not None:
            p.ec_shift, p.ec_scale = self.shift.ctypes.data, self.scale.ctypes.data
        return p

    @classmethod
    def fit_plus(cls, sample, dim: int, distance: Distance, bits: int, rotation_unpadded: bool = False, invert: Optional[bool] = None,
                 device_id: int = 0) -> "TurboQuantizer":
        """TQMode::Plus: the error correction fitted on the device as `EncodedVectorsTQ::encode`'s first pass fits it (P-square estimates of the
        quantiles Phi(-+c_outer) of every rotated coordinate, qmx_tq_fit_plus).  `sample`: the sampled vectors as stored (cosine rows normalised), in
        ascending index order - the reference draws `sample_size` = 8 192 / 4 096 / 2 048 (4 / 2 / 1 bits) indices with its Permutor."""
        plain = cls(dim, distance, bits, rotation_unpadded, invert)
        v = np.ascontiguousarray(np.atleast_2d(sample), dtype=np.float32).reshape(-1, dim)
        shift, scale = np.empty(plain.padded_dim, dtype=np.float32), np.empty(plain.padded_dim, dtype=np.float32)
        p = plain.params()
        F.check(F.lib().qmx_tq_fit_plus(device_id, int(plain.distance), plain.dim, C.byref(p), F.ptr(v) if len(v) else None, len(v), F.ptr(shift),
                                        F.ptr(scale)))
        return cls(dim, distance, bits, rotation_unpadded, invert, shift=shift, scale=scale)

    def quantized_vector_size(self) -> int:
        """TurboQuantizer::quantized_size_for (turboquant/encoding.rs:172-190)."""
        return self.code_bytes + (8 if self.distance == Distance.Euclid else 4) + (4 if getattr(self, "plus_mode", False) else 0)

    def encode(self, vectors, device_id: int = 0) -> np.ndarray:
        """`TurboQuantizer::quantize` on device: [n, dim] f32 (as stored: cosine rows normalised) -> [n, quantized_vector_size] bytes."""
        v = np.ascontiguousarray(vectors, dtype=np.float32)
        out = np.empty((v.shape[0], self.quantized_vector_size()), dtype=np.uint8)
        p = self.params()
        F.check(F.lib().qmx_tq_encode(device_id, int(self.distance), self.dim, C.byref(p), F.ptr(v), v.shape[0], F.ptr(out)))
        return out


class EncodedVectorsTQ(VectorStorage):
    """Device-resident `EncodedVectorsTQ` storage: rows = [n, quantized_vector_size] bytes as `TurboQuantizer::quantize` writes them."""

    def __init__(self, rows, quantizer: TurboQuantizer, device_id: int = 0):
        self._h = C.c_void_p()
        self.quantizer = quantizer
        self.distance = quantizer.distance
        self.datatype = None
        on_device = hasattr(rows, "data_ptr") and getattr(rows, "is_cuda", False)   # a torch CUDA tensor: split on the device, no host copy
        if not on_device:
            rows = np.ascontiguousarray(rows, dtype=np.uint8)
        assert rows.shape[1] == quantizer.quantized_vector_size()
        self.dim = quantizer.dim
        self.count = int(rows.shape[0])
        self._keep = None
        desc = F.SegmentDesc()
        desc.dtype = F.DTYPE_TQ
        desc.distance = int(quantizer.distance)
        desc.dim = quantizer.dim
        desc.flags = 0
        self._tq = quantizer.params()
        desc.tq = C.pointer(self._tq)
        desc.n = self.count
        desc.row_stride_bytes = 0
        desc.data = F.ptr(rows)
        desc.device_id = device_id
        F.check(F.lib().qmx_segment_create(C.byref(desc), C.byref(self._h)))

    def get_quantized_vector(self, ids: Sequence[int]) -> np.ndarray:
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.empty((len(ids), self.quantizer.quantized_vector_size()), dtype=np.uint8)
        F.check(F.lib().qmx_segment_read_rows(self._h, F.ptr(ids), len(ids), F.ptr(out)))
        return out


class MultiDenseVectorStorage:
    """`MultiDenseVectorStorage` with `MultiVectorComparator::MaxSim` (vector_storage/multi_dense/, query_scorer/mod.rs:70-97): the
    inner vectors of all points flattened into one dense block on the device + per-point offsets."""

    def __init__(self, inner_vectors, point_offsets, distance: Distance, datatype: VectorStorageDatatype = VectorStorageDatatype.Float32,
                 device_id: int = 0, inner_storage=None):
        self.inner = inner_storage if inner_storage is not None else VectorStorage(inner_vectors, distance, datatype, device_id=device_id)
        self.offsets = np.ascontiguousarray(point_offsets, dtype=np.uint64)
        self.count = len(self.offsets) - 1
        self.point_deleted = None

    def set_deleted(self, point_deleted):
        self.point_deleted = None if point_deleted is None else np.ascontiguousarray(point_deleted, dtype=bool)

    def _queries(self, multi_queries):
        qs = [np.atleast_2d(np.asarray(q, dtype=np.float32)) for q in multi_queries]
        first = np.zeros(len(qs) + 1, dtype=np.uint32)
        first[1:] = np.cumsum([len(q) for q in qs])
        return new_raw_scorer(np.concatenate(qs, axis=0), self.inner), first

    def score_points(self, multi_queries, ids) -> np.ndarray:
        """`MultiMetricQueryScorer::score_stored_batch` for every multi-query: [n_queries, len(ids)] f32."""
        scorer, first = self._queries(multi_queries)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.empty((len(first) - 1, len(ids)), dtype=np.float32)
        F.check(F.lib().qmx_multi_score_points(scorer._h, F.ptr(first), len(first) - 1, F.ptr(self.offsets), self.count, F.ptr(ids), len(ids),
                                               F.ptr(out)))
        return out

    def peek_top_all(self, multi_queries, top: int, ids=None) -> List[np.ndarray]:
        """`BatchFilteredSearcher::peek_top_all` over the MaxSim scorers (brute force)."""
        scorer, first = self._queries(multi_queries)
        nq = len(first) - 1
        out = np.zeros((nq, top), dtype=ScoredPointOffset)
        counts = np.zeros(nq, dtype=np.uint32)
        idarr = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
        words = _bits_to_words(self.point_deleted)
        F.check(F.lib().qmx_multi_search_topk(scorer._h, F.ptr(first), nq, F.ptr(self.offsets), self.count, F.ptr(words),
                                              0 if self.point_deleted is None else len(self.point_deleted), top, F.ptr(idarr),
                                              0 if idarr is None else len(idarr), F.ptr(out), F.ptr(counts)))
        return [out[i, :counts[i]].copy() for i in range(nq)]

    def _custom(self, queries):
        """custom queries whose examples are multi-vectors -> (inner scorer, example_first, descriptors)"""
        flat, descs, first, coefs = [], (F.CustomQuery * len(queries))(), 0, []
        for i, q in enumerate(queries):
            descs[i].kind, descs[i].first, descs[i].n_a, descs[i].n_b, descs[i].coef_first = q.kind, first, q.n_a, q.n_b, len(coefs)
            flat += [np.atleast_2d(np.asarray(e, dtype=np.float32)) for e in q.examples]
            first += len(q.examples)
            if q.coefs is not None:
                coefs += q.coefs.tolist()
        scorer, efirst = self._queries(flat)
        if coefs:
            cf = np.asarray(coefs, dtype=np.float32)
            F.check(F.lib().qmx_custom_set_coefficients(scorer._h, F.ptr(cf), len(cf)))
        return scorer, efirst, descs

    def custom_score_points(self, queries, ids) -> np.ndarray:
        """`MultiCustomQueryScorer` / `QuantizedMultiCustomQueryScorer`: `queries` = MultiCustomQuery-like objects (CustomQuery whose examples are
        [tokens, dim] arrays) -> [n_queries, len(ids)] f32."""
        scorer, efirst, descs = self._custom(queries)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.empty((len(queries), len(ids)), dtype=np.float32)
        F.check(F.lib().qmx_multi_custom_score_points(scorer._h, F.ptr(efirst), len(efirst) - 1, descs, len(queries), F.ptr(self.offsets), self.count,
                                                      F.ptr(ids), len(ids), F.ptr(out)))
        return out

    def custom_peek_top(self, queries, top: int, ids=None) -> List[np.ndarray]:
        scorer, efirst, descs = self._custom(queries)
        nq = len(queries)
        out = np.zeros((nq, top), dtype=ScoredPointOffset)
        counts = np.zeros(nq, dtype=np.uint32)
        idarr = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
        words = _bits_to_words(self.point_deleted)
        F.check(F.lib().qmx_multi_custom_search_topk(scorer._h, F.ptr(efirst), len(efirst) - 1, descs, nq, F.ptr(self.offsets), self.count, F.ptr(words),
                                                     0 if self.point_deleted is None else len(self.point_deleted), top, F.ptr(idarr),
                                                     0 if idarr is None else len(idarr), F.ptr(out), F.ptr(counts)))
        return [out[i, :counts[i]].copy() for i in range(nq)]

    def custom_search_hnsw(self, graph, queries, top: int, ef: int, with_counters: bool = False):
        """`GraphLayers::search` over the multi-vector POINTS with a `MultiCustomQueryScorer` per custom query (examples = multi-vectors), on device."""
        scorer, efirst, descs = self._custom(queries)
        nq = len(queries)
        out = np.zeros((nq, max(top, 1)), dtype=ScoredPointOffset)
        counts = np.zeros(nq, dtype=np.uint32)
        words = _bits_to_words(self.point_deleted)
        ctr = F.Counters()
        F.check(F.lib().qmx_multi_custom_hnsw_search(graph._h, scorer._h, F.ptr(efirst), len(efirst) - 1, descs, nq, F.ptr(self.offsets), self.count,
                                                     F.ptr(words), 0 if self.point_deleted is None else len(self.point_deleted), top, ef, F.ptr(out),
                                                     F.ptr(counts), C.byref(ctr)))
        res = [out[i, :counts[i]].copy() for i in range(nq)]
        return (res, ctr) if with_counters else res

    def search_hnsw(self, graph, multi_queries, top: int, ef: int, with_counters: bool = False):
        """`GraphLayers::search` over the multi-vector POINTS with the MaxSim scorer of every multi-query (`MultiMetricQueryScorer` /
        `QuantizedMultiQueryScorer` behind `FilteredScorer`), on device."""
        scorer, first = self._queries(multi_queries)
        nq = len(first) - 1
        out = np.zeros((nq, top), dtype=ScoredPointOffset)
        counts = np.zeros(nq, dtype=np.uint32)
        words = _bits_to_words(self.point_deleted)
        ctr = F.Counters()
        F.check(F.lib().qmx_multi_hnsw_search(graph._h, scorer._h, F.ptr(first), nq, F.ptr(self.offsets), self.count, F.ptr(words),
                                              0 if self.point_deleted is None else len(self.point_deleted), top, ef, F.ptr(out), F.ptr(counts),
                                              C.byref(ctr)))
        res = [out[i, :counts[i]].copy() for i in range(nq)]
        return (res, ctr) if with_counters else res


class QuantizedMultivectorStorage(MultiDenseVectorStorage):
    """`QuantizedMultivectorStorage<QuantizedStorage, Offsets>` (vector_storage/quantized/quantized_multivector_storage/mod.rs:76-130): the
    quantized INNER rows (an `EncodedVectorsU8` / `EncodedVectorsPQ` / `EncodedVectorsBin` storage on the device) + `MultivectorOffset`s as
    ascending `[n_points + 1]` row offsets; MaxSim over the quantized scores (`score_point_max_similarity`, :339-363)."""

    def __init__(self, quantized_inner_storage, point_offsets):
        super().__init__(None, point_offsets, None, inner_storage=quantized_inner_storage)


def load_quantizer(meta_json, dtype: int):
    """`EncodedVectors{U8,PQ,Bin}::load`'s metadata half: the text of a segment's "quantized.meta.json"
    (vector_storage/quantized/quantized_vectors/config.rs:13) -> ScalarQuantizer | ProductQuantizer | BinaryQuantizer,
    parsed by the library (`qmx_quant_meta_parse`, host only).  Every value is taken from the file as written
    (alpha, offset, multiplier, invert, centroids, stats), not re-derived."""
    data = meta_json.encode() if isinstance(meta_json, str) else bytes(meta_json)
    m = F.QuantMeta()
    F.check(F.lib().qmx_quant_meta_parse(int(dtype), data, len(data), C.byref(m)))
    try:
        distance = Distance(m.distance)
        if dtype == F.DTYPE_SQ_U8:
            q = ScalarQuantizer(m.dim, distance, m.sq.alpha, m.sq.offset)
            q.actual_dim, q.multiplier, q.invert = int(m.sq.actual_dim), np.float32(m.sq.multiplier), bool(m.sq.invert)
            return q
        if dtype == F.DTYPE_PQ:
            cen = np.ctypeslib.as_array(C.cast(m.pq.centroids, C.POINTER(C.c_float)), (m.pq.n_centroids, m.dim)).copy()
            q = ProductQuantizer(m.dim, distance, m.pq.chunk_size, cen)
            q.invert = bool(m.pq.invert)
            return q
        if dtype == F.DTYPE_TQ:
            shift = scale = None
            if m.tq.ec_shift:
                q0 = TurboQuantizer(m.dim, distance, int(m.tq.bits))
                shift = np.ctypeslib.as_array(C.cast(m.tq.ec_shift, C.POINTER(C.c_float)), (q0.padded_dim,)).copy()
                scale = np.ctypeslib.as_array(C.cast(m.tq.ec_scale, C.POINTER(C.c_float)), (q0.padded_dim,)).copy()
            q = TurboQuantizer(m.dim, distance, int(m.tq.bits), bool(m.tq.rotation_unpadded), bool(m.tq.invert), shift, scale)
            q.plus_mode = bool(m.tq.plus_mode)
            return q
        mean = stddev = None
        if m.bq.mean:
            mean = np.ctypeslib.as_array(C.cast(m.bq.mean, C.POINTER(C.c_float)), (m.dim,)).copy()
            stddev = np.ctypeslib.as_array(C.cast(m.bq.stddev, C.POINTER(C.c_float)), (m.dim,)).copy()
        return BinaryQuantizer(m.dim, distance, bool(m.invert), int(m.bq.encoding), mean, stddev, int(m.bq_query_encoding))
    finally:
        F.lib().qmx_quant_meta_free(C.byref(m))


class RawScorer:
    """`Box<dyn RawScorer>` for a batch of `QueryVector::Nearest` queries (raw_scorer.rs:39-58).
    One instance holds `nq` scorers; single-query use is nq == 1."""

    def __init__(self, handle, storage: VectorStorage, nq: int):
        self._h = handle
        self.storage = storage
        self.nq = nq

    def set_filter(self, allowed=None):
        """Payload filter of `ScorerFilters` as an allow mask (bool array over point ids); None clears it."""
        if allowed is None:
            F.check(F.lib().qmx_query_set_filter(self._h, None, 0))
            return
        words = _bits_to_words(np.asarray(allowed, dtype=bool))
        F.check(F.lib().qmx_query_set_filter(self._h, F.ptr(words), len(allowed)))

    def score_points(self, points: Sequence[int]) -> np.ndarray:
        """scores[qi, i] = similarity(query qi, stored point points[i])  (raw_scorer.rs:40)."""
        ids = np.ascontiguousarray(points, dtype=np.uint32)
        scores = np.empty((self.nq, len(ids)), dtype=np.float32)
        F.check(F.lib().qmx_score_points(self._h, F.ptr(ids), len(ids), F.ptr(scores), None))
        return scores

    def score_point(self, point: int, query_index: int = 0) -> float:
        out = C.c_float()
        F.check(F.lib().qmx_score_point(self._h, query_index, int(point), C.byref(out)))
        return out.value

    def score_internal(self, point_a, point_b) -> np.ndarray:
        a = np.ascontiguousarray(np.atleast_1d(point_a), dtype=np.uint32)
        b = np.ascontiguousarray(np.atleast_1d(point_b), dtype=np.uint32)
        out = np.empty(len(a), dtype=np.float32)
        F.check(F.lib().qmx_score_internal(self.storage._h, F.ptr(a), F.ptr(b), len(a), F.ptr(out)))
        return out

    def score_points_ragged(self, ids_per_query: Sequence[Sequence[int]]) -> List[np.ndarray]:
        """Query qi scores its own id list: the HNSW hop of many concurrent searches in one launch
        (graph_layers.rs:125-139 produce <= m0 ids per hop per search)."""
        assert len(ids_per_query) == self.nq
        lens = [len(x) for x in ids_per_query]
        offsets = np.zeros(self.nq + 1, dtype=np.uint32)
        offsets[1:] = np.cumsum(lens)
        ids = (np.concatenate([np.asarray(x, dtype=np.uint32) for x in ids_per_query])
               if offsets[-1] else np.zeros(0, dtype=np.uint32))
        scores = np.empty(int(offsets[-1]), dtype=np.float32)
        F.check(F.lib().qmx_score_points_ragged(self._h, F.ptr(ids), F.ptr(offsets), F.ptr(scores), None))
        return [scores[offsets[i]:offsets[i + 1]] for i in range(self.nq)]

    def score_bytes(self, rows) -> np.ndarray:
        """`QueryScorerBytes::score_bytes` (query_scorer/mod.rs:48-68): rows given as raw bytes in the
        storage's reference row layout."""
        rows = np.ascontiguousarray(rows)
        n = rows.shape[0]
        stride = rows.strides[0]
        scores = np.empty((self.nq, n), dtype=np.float32)
        F.check(F.lib().qmx_score_bytes(self._h, F.ptr(rows), n, stride, F.ptr(scores)))
        return scores

    def rescore(self, ids, top: int, counts=None) -> List[np.ndarray]:
        """The rescoring tail of `postprocess_search_result` (vector_index_search_common.rs:73-90):
        query qi re-scores ids[qi] with THIS (original-vector) scorer, sorted descending, truncated to top."""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        assert ids.ndim == 2 and ids.shape[0] == self.nq
        cnt = None if counts is None else np.ascontiguousarray(counts, dtype=np.uint32)
        out = np.zeros((self.nq, top), dtype=ScoredPointOffset)
        oc = np.zeros(self.nq, dtype=np.uint32)
        F.check(F.lib().qmx_rescore(self._h, F.ptr(ids), F.ptr(cnt), ids.shape[1], top, F.ptr(out), F.ptr(oc)))
        return [out[i, :oc[i]].copy() for i in range(self.nq)]

    def encoded_query(self, query_index: int = 0) -> np.ndarray:
        if isinstance(self.storage, EncodedVectorsPQ):   # EncodedQueryPQ {lut: Vec<f32>} = [m][n_centroids]
            qz = self.storage.quantizer
            out = np.empty((qz.m, qz.n_centroids), dtype=np.float32)
            F.check(F.lib().qmx_query_read_encoded(self._h, query_index, F.ptr(out), out.nbytes, None))
            return out
        if isinstance(self.storage, EncodedVectorsBin):  # EncodedBinVector {encoded_vector: Vec<u128>} as bytes
            # a scalar-encoded query (Scalar4bits / Scalar8bits) holds 4 / 8 bit planes per row word; internal queries are rows
            qz = self.storage.quantizer
            planes = {0: 1, 1: 4, 2: 8}[qz.query_encoding] if not getattr(self, "_internal", False) else 1
            out = np.empty(qz.quantized_vector_size() * planes, dtype=np.uint8)
            F.check(F.lib().qmx_query_read_encoded(self._h, query_index, F.ptr(out), out.nbytes, None))
            return out
        if isinstance(self.storage, EncodedVectorsU8):   # EncodedQueryU8 {offset: f32, encoded_query: Vec<u8>}
            nbytes = 4 + self.storage.quantizer.actual_dim
            out = np.empty(nbytes, dtype=np.uint8)
            F.check(F.lib().qmx_query_read_encoded(self._h, query_index, F.ptr(out), nbytes, None))
            return out
        nbytes = self.storage.dim * np.dtype(_NP_ELEM[int(self.storage.datatype)]).itemsize
        out = np.empty(nbytes, dtype=np.uint8)
        written = C.c_uint64()
        F.check(F.lib().qmx_query_read_encoded(self._h, query_index, F.ptr(out), nbytes, C.byref(written)))
        return out[:written.value].view(_NP_ELEM[int(self.storage.datatype)])

    def last_counters(self) -> "F.Counters":
        """qmx_query_last_counters: the `HardwareCounterCell` increments of the batch's last brute-force search (synchronises)."""
        c = F.Counters()
        F.check(F.lib().qmx_query_last_counters(self._h, C.byref(c)))
        return c

    def close(self):
        if self._h:
            F.lib().qmx_query_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def new_raw_scorer(query, storage: VectorStorage) -> RawScorer:
    """`new_raw_scorer(QueryVector::Nearest(query), storage, hc)` (raw_scorer.rs:60-114).
    `query`: [dim] or [nq, dim] f32 ORIGINAL vectors; preprocessing + cast happen on device."""
    if hasattr(query, "data_ptr") and getattr(query, "is_cuda", False):    # a contiguous [nq, dim] f32 torch CUDA tensor: no host copy
        q = query
        assert q.dim() == 2 and q.is_contiguous() and str(q.dtype) == "torch.float32"
    else:
        q = np.ascontiguousarray(np.atleast_2d(query), dtype=np.float32)
    if q.shape[1] != storage.dim:
        raise ValueError(f"query dim {q.shape[1]} != storage dim {storage.dim}")
    h = C.c_void_p()
    F.check(F.lib().qmx_query_create(storage._h, F.ptr(q), q.shape[0], C.byref(h)))
    return RawScorer(h, storage, q.shape[0])


def new_raw_scorer_internal(point_ids, storage: VectorStorage) -> RawScorer:
    """`FilteredScorer::new_internal` (point_scorer.rs:183-218): stored points as queries."""
    ids = np.ascontiguousarray(np.atleast_1d(point_ids), dtype=np.uint32)
    h = C.c_void_p()
    F.check(F.lib().qmx_query_create_internal(storage._h, F.ptr(ids), len(ids), C.byref(h)))
    scorer = RawScorer(h, storage, len(ids))
    scorer._internal = True
    return scorer


class CustomQuery:
    """One `QueryVector::{RecommendBestScore, RecommendSumScores, Discover, Context}` (vector_storage/query/*.rs);
    vectors are ORIGINAL (un-preprocessed) f32, as for `new_raw_scorer`."""

    def __init__(self, kind: int, examples, n_a: int, n_b: int, coefs=None):
        self.kind, self.examples, self.n_a, self.n_b = kind, [np.asarray(v, dtype=np.float32) for v in examples], n_a, n_b
        self.coefs = None if coefs is None else np.asarray(coefs, dtype=np.float32)

    @classmethod
    def recommend_best_score(cls, positives, negatives):
        return cls(F.CUSTOM_RECO_BEST_SCORE, list(positives) + list(negatives), len(positives), len(negatives))

    @classmethod
    def recommend_sum_scores(cls, positives, negatives):
        return cls(F.CUSTOM_RECO_SUM_SCORES, list(positives) + list(negatives), len(positives), len(negatives))

    @classmethod
    def discover(cls, target, pairs):
        return cls(F.CUSTOM_DISCOVER, [target] + [v for p in pairs for v in p], 1, len(pairs))

    @classmethod
    def context(cls, pairs):
        return cls(F.CUSTOM_CONTEXT, [v for p in pairs for v in p], 0, len(pairs))

    @classmethod
    def feedback_naive(cls, target, feedback, a: float, b: float, c: float, margin: float = 0.0):
        """`NaiveFeedbackQuery::into_query` (feedback_query.rs:43-47, 117-145, 159-174): `feedback` = [(vector, score)];
        every ordered pair with score difference above `margin` becomes a context pair with
        partial_computation = confidence.powf(b) * c (f32; libm's powf, here numpy's)."""
        pairs, coefs = [], [np.float32(a)]
        for i, (pv, ps) in enumerate(feedback):                  # itertools' permutations(2) order
            for j, (nv, ns) in enumerate(feedback):
                if i == j:
                    continue
                confidence = np.float32(ps) - np.float32(ns)
                if confidence <= np.float32(margin):
                    continue
                pairs.append((pv, nv))
                coefs.append(np.float32(np.power(confidence, np.float32(b), dtype=np.float32) * np.float32(c)))
        return cls(F.CUSTOM_FEEDBACK, [target] + [v for p in pairs for v in p], 1, len(pairs), coefs)


class CustomRawScorer:
    """`new_raw_scorer(QueryVector::<custom>, storage)` for a batch of custom queries (raw_scorer.rs:60-114 ->
    CustomQueryScorer): all example vectors form one device query batch, every custom query is a slice of it."""

    def __init__(self, queries: Sequence[CustomQuery], storage: VectorStorage):
        self.storage = storage
        flat, descs, first, coefs = [], (F.CustomQuery * len(queries))(), 0, []
        for i, q in enumerate(queries):
            descs[i].kind, descs[i].first, descs[i].n_a, descs[i].n_b, descs[i].coef_first = q.kind, first, q.n_a, q.n_b, len(coefs)
            flat += q.examples
            first += len(q.examples)
            if q.coefs is not None:
                coefs += q.coefs.tolist()
        self._descs, self.nq = descs, len(queries)
        self.examples = new_raw_scorer(np.stack(flat) if flat else np.zeros((0, storage.dim), dtype=np.float32), storage)
        if coefs:
            cf = np.asarray(coefs, dtype=np.float32)
            F.check(F.lib().qmx_custom_set_coefficients(self.examples._h, F.ptr(cf), len(cf)))

    def score_points(self, points: Sequence[int]) -> np.ndarray:
        ids = np.ascontiguousarray(points, dtype=np.uint32)
        out = np.empty((self.nq, len(ids)), dtype=np.float32)
        F.check(F.lib().qmx_custom_score_points(self.examples._h, self._descs, self.nq, F.ptr(ids), len(ids), F.ptr(out)))
        return out

    def peek_top(self, top: int, points=None) -> List[np.ndarray]:
        out = np.zeros((self.nq, top), dtype=ScoredPointOffset)
        counts = np.zeros(self.nq, dtype=np.uint32)
        ids = None if points is None else np.ascontiguousarray(points, dtype=np.uint32)
        F.check(F.lib().qmx_custom_search_topk(self.examples._h, self._descs, self.nq, top, F.ptr(ids), 0 if ids is None else len(ids),
                                               F.ptr(out), F.ptr(counts)))
        return [out[i, :counts[i]].copy() for i in range(self.nq)]

    def search_hnsw(self, graph, top: int, ef: int, with_scored: bool = False):
        """`GraphLayers::search(top, ef, Hnsw, points_scorer = the custom scorer)` for every custom query of the batch, on device
        (qmx_custom_hnsw_search): dense, SQ, PQ, BQ and TurboQuant storages."""
        out = np.zeros((self.nq, max(top, 1)), dtype=ScoredPointOffset)
        counts = np.zeros(self.nq, dtype=np.uint32)
        ctr = F.Counters()
        F.check(F.lib().qmx_custom_hnsw_search(graph._h, self.examples._h, self._descs, self.nq, top, ef, F.ptr(out), F.ptr(counts), None, C.byref(ctr)))
        res = [out[i, :counts[i]].copy() for i in range(self.nq)]
        return (res, int(ctr.vectors_scored)) if with_scored else res


def search_quantized(searched: RawScorer, original: Optional[RawScorer], top: int, oversampling: float = 0.0, rescore: bool = True,
                     graph=None, hnsw_ef: int = 0, ids=None, is_stopped=None, acorn: bool = False, counters=None, raw_output: bool = False):
    """`PlainVectorIndexReadView::search` (graph is None) or the graph arm of `HNSWIndexReadView::search`, with
    `get_oversampled_top` and `postprocess_search_result` (vector_index_search_common.rs:27-91) in one device-side call."""
    p = F.SearchParams()
    p.top, p.oversampling, p.rescore, p.hnsw_ef, p.acorn = int(top), float(oversampling), 1 if rescore else 0, int(hnsw_ef), 1 if acorn else 0
    nq = searched.nq
    out = np.zeros((nq, top), dtype=ScoredPointOffset)
    counts = np.zeros(nq, dtype=np.uint32)
    stop = None
    if is_stopped is not None:
        stop = is_stopped if isinstance(is_stopped, np.ndarray) else np.array([1 if is_stopped else 0], dtype=np.uint8)
    idarr = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
    F.check(F.lib().qmx_search_quantized(None if graph is None else graph._h, searched._h, None if original is None else original._h,
                                         C.byref(p), F.ptr(idarr), 0 if idarr is None else len(idarr), F.ptr(out), F.ptr(counts),
                                         F.ptr(stop), None if counters is None else C.byref(counters)))
    if raw_output:          # ([nq, top] ScoredPointOffset, [nq] counts) as the library wrote them: what a caller timing the call wants (no per-query slicing)
        return out, counts
    return [out[i, :counts[i]].copy() for i in range(nq)]


class BatchFilteredSearcher:
    """`BatchFilteredSearcher` (point_scorer.rs:307-472): one scorer + one bounded queue per query."""

    def __init__(self, queries, vectors: VectorStorage, top: int, quantized_vectors=None):
        if top == 0:
            raise ValueError("length must be greater than zero")  # FixedLengthPriorityQueue::new panics
        self.top = int(top)
        self.storage = quantized_vectors if quantized_vectors is not None else vectors
        self.scorer = new_raw_scorer(queries, self.storage)
        self.counters = F.Counters()

    @classmethod
    def new_for_test(cls, vectors: Sequence, vector_storage: VectorStorage, top: int):
        return cls(vectors, vector_storage, top)

    def _run(self, ids, is_stopped) -> List[np.ndarray]:
        nq = self.scorer.nq
        out = np.zeros((nq, self.top), dtype=ScoredPointOffset)
        counts = np.zeros(nq, dtype=np.uint32)
        stop = None
        if is_stopped is not None:
            stop = is_stopped if isinstance(is_stopped, np.ndarray) else np.array([1 if is_stopped else 0], dtype=np.uint8)
        if ids is not None:
            ids = np.ascontiguousarray(list(ids) if not isinstance(ids, np.ndarray) else ids, dtype=np.uint32)
        F.check(F.lib().qmx_search_topk(self.scorer._h, self.top, F.ptr(ids), 0 if ids is None else len(ids),
                                        F.ptr(out), F.ptr(counts), F.ptr(stop), C.byref(self.counters)))
        return [out[i, :counts[i]].copy() for i in range(nq)]

    def peek_top_all(self, is_stopped=None) -> List[np.ndarray]:
        """Score every non-deleted point (point_scorer.rs:408-421)."""
        return self._run(None, is_stopped)

    def peek_top_iter(self, points: Iterable[int], is_stopped=None) -> List[np.ndarray]:
        """Candidate stream given explicitly (point_scorer.rs:423-472)."""
        return self._run(points, is_stopped)

"""ctypes binding of include/qdrant_amd.h (libqdrant_amd.so, built in-tree by `make`).

The product path has no fallback: if the HIP library is missing this module raises, and every
call that cannot reach a gfx950 device raises `QmxError` with the library's status code.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libqdrant_amd.so")
TESTDATA_LIB_PATH = os.path.join(_HERE, "libqmx_testdata.so")     # synthetic row generators of bench.py / tools / tests: NOT part of the product library

# status codes (qmx_status)
OK, ERR_OUT_OF_MEMORY, ERR_OUT_OF_BOUNDS, ERR_NOT_SUPPORTED, ERR_NOT_READY, ERR_TIMEOUT, ERR_OTHER, \
    ERR_CANCELLED, ERR_BAD_ARG, ERR_NO_DEVICE = range(10)
STATUS_NAMES = ["OK", "OUT_OF_MEMORY", "OUT_OF_BOUNDS", "NOT_SUPPORTED", "NOT_READY", "TIMEOUT", "OTHER",
                "CANCELLED", "BAD_ARG", "NO_DEVICE"]

DTYPE_F32, DTYPE_F16, DTYPE_U8, DTYPE_SQ_U8, DTYPE_PQ, DTYPE_BQ, DTYPE_TQ = range(7)
COSINE, EUCLID, DOT, MANHATTAN = range(4)

SEG_DATA_ON_DEVICE = 0x1
SEG_U8_SCALAR_ORDER = 0x2
SEG_TIME_KERNELS = 0x4
SEG_BQ_TOGGLE_INVERT = 0x8
SEG_SPLIT_COPY = 0x10
SEG_HALF_COPY = 0x20
SEG_I8_COPY = 0x40
SEG_AUTO_COPY = 0x80
SEG_PQ_PREFILTER_COPY = 0x100


class ScoredPoint(C.Structure):
    _fields_ = [("idx", C.c_uint32), ("score", C.c_float)]


class Counters(C.Structure):
    _fields_ = [("vectors_scored", C.c_uint64), ("bytes_read", C.c_uint64), ("kernel_launches", C.c_uint64),
                ("kernel_ms", C.c_float), ("reserved", C.c_float),
                ("prefilter_candidates", C.c_uint64), ("verified_rows", C.c_uint64), ("fallback_queries", C.c_uint32), ("prefilter_queries", C.c_uint32)]


class SqParams(C.Structure):
    _fields_ = [("actual_dim", C.c_uint32), ("alpha", C.c_float), ("offset", C.c_float), ("multiplier", C.c_float),
                ("invert", C.c_uint8), ("pad_", C.c_uint8 * 3)]


class PqParams(C.Structure):
    _fields_ = [("chunk_size", C.c_uint32), ("n_centroids", C.c_uint32), ("centroids", C.c_void_p),
                ("invert", C.c_uint8), ("lut_mfma", C.c_uint8), ("pad_", C.c_uint8 * 2)]


class BqParams(C.Structure):
    _fields_ = [("encoding", C.c_uint32), ("query_encoding", C.c_uint32), ("mean", C.c_void_p), ("stddev", C.c_void_p)]


BQ_ONE_BIT, BQ_TWO_BITS, BQ_ONE_AND_HALF_BITS = range(3)
BQ_QUERY_SAME_AS_STORAGE, BQ_QUERY_SCALAR_4BITS, BQ_QUERY_SCALAR_8BITS = range(3)


class TqParams(C.Structure):
    _fields_ = [("bits", C.c_uint32), ("rotation_unpadded", C.c_uint32), ("invert", C.c_uint8), ("plus_mode", C.c_uint8), ("pad_", C.c_uint8 * 2),
                ("reserved", C.c_uint32), ("ec_shift", C.c_void_p), ("ec_scale", C.c_void_p)]


TQ_BITS4, TQ_BITS2, TQ_BITS1_5, TQ_BITS1 = range(4)


class SegmentInfo(C.Structure):
    """qmx_segment_info: the derived copy a segment holds and what the QMX_SEG_AUTO_COPY trial measured."""
    _fields_ = [("derived_copy", C.c_uint32), ("chosen_by_trial", C.c_uint32), ("derived_copy_bytes", C.c_uint64), ("i8_scale_balance", C.c_float),
                ("trial_i8_ms", C.c_float), ("trial_half_ms", C.c_float), ("trial_i8_verified_rows", C.c_float), ("trial_i8_fallback_queries", C.c_uint32),
                ("reserved", C.c_uint32)]


class SegmentDesc(C.Structure):
    _fields_ = [("dtype", C.c_uint32), ("distance", C.c_uint32), ("dim", C.c_uint32), ("flags", C.c_uint32),
                ("n", C.c_uint64), ("row_stride_bytes", C.c_uint64), ("data", C.c_void_p),
                ("device_id", C.c_int32), ("reserved", C.c_int32), ("sq", C.POINTER(SqParams)),
                ("pq", C.POINTER(PqParams)), ("bq", C.POINTER(BqParams)), ("tq", C.POINTER(TqParams))]


class HnswDesc(C.Structure):
    _fields_ = [("m", C.c_uint32), ("m0", C.c_uint32), ("n_points", C.c_uint32), ("n_levels", C.c_uint32),
                ("reindex", C.c_void_p), ("level_offsets", C.c_void_p), ("offsets", C.c_void_p), ("n_offsets", C.c_uint64),
                ("neighbors", C.c_void_p), ("n_neighbors", C.c_uint64),
                ("entry_point_ids", C.c_void_p), ("entry_point_levels", C.c_void_p), ("n_entry_points", C.c_uint32),
                ("n_extra_entry_points", C.c_uint32), ("extra_entry_point_ids", C.c_void_p),
                ("extra_entry_point_levels", C.c_void_p), ("device_id", C.c_int32), ("reserved", C.c_int32)]


class QuantMeta(C.Structure):
    """qmx_quant_meta: a parsed quantized.meta.json (library-owned arrays, qmx_quant_meta_free)."""
    _fields_ = [("dtype", C.c_uint32), ("dim", C.c_uint32), ("distance", C.c_uint32), ("invert", C.c_uint8),
                ("has_deprecated_count", C.c_uint8), ("bq_query_encoding", C.c_uint8), ("pad_", C.c_uint8),
                ("deprecated_count", C.c_uint64), ("sq", SqParams), ("pq", PqParams), ("bq", BqParams), ("tq", TqParams), ("owner", C.c_void_p)]


class GraphLinks(C.Structure):
    """qmx_graph_links: a links file decoded on the host (library-owned arrays, qmx_graph_links_free)."""
    _fields_ = [("format", C.c_uint32), ("m", C.c_uint32), ("m0", C.c_uint32), ("n_points", C.c_uint32), ("n_levels", C.c_uint32),
                ("reserved", C.c_uint32), ("n_offsets", C.c_uint64), ("n_neighbors", C.c_uint64),
                ("reindex", C.POINTER(C.c_uint32)), ("level_offsets", C.POINTER(C.c_uint64)), ("offsets", C.POINTER(C.c_uint64)),
                ("neighbors", C.POINTER(C.c_uint32)), ("owner", C.c_void_p)]


class HnswBuildParams(C.Structure):
    _fields_ = [("m", C.c_uint32), ("m0", C.c_uint32), ("ef_construct", C.c_uint32), ("entry_points_num", C.c_uint32),
                ("seed", C.c_uint64), ("max_batch", C.c_uint32), ("reserved", C.c_uint32)]


class HnswInfo(C.Structure):
    _fields_ = [("m", C.c_uint32), ("m0", C.c_uint32), ("n_points", C.c_uint32), ("n_levels", C.c_uint32),
                ("n_offsets", C.c_uint64), ("n_neighbors", C.c_uint64), ("n_entry_points", C.c_uint32),
                ("n_extra_entry_points", C.c_uint32)]


class SearchParams(C.Structure):
    _fields_ = [("top", C.c_uint32), ("oversampling", C.c_float), ("rescore", C.c_uint8), ("acorn", C.c_uint8), ("pad_", C.c_uint8 * 2), ("hnsw_ef", C.c_uint32)]


class CustomQuery(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("first", C.c_uint32), ("n_a", C.c_uint32), ("n_b", C.c_uint32), ("coef_first", C.c_uint32)]


CUSTOM_RECO_BEST_SCORE, CUSTOM_RECO_SUM_SCORES, CUSTOM_DISCOVER, CUSTOM_CONTEXT, CUSTOM_FEEDBACK = range(5)


class QmxError(RuntimeError):
    def __init__(self, status, message):
        self.status = status
        name = STATUS_NAMES[status] if 0 <= status < len(STATUS_NAMES) else str(status)
        super().__init__(f"qmx status {status} ({name}): {message}")


# every exported symbol of include/qdrant_amd.h with its signature (checked by tests/test_abi.py)
_P = C.c_void_p
TESTDATA_SIGNATURES = {
    "qmx_synth_fill_f32": (C.c_int32, [C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]),
    "qmx_synth_fill_latent_f32": (C.c_int32, [C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]),
}

SIGNATURES = {
    "qmx_abi_version": (C.c_uint32, []),
    "qmx_set_option": (C.c_int32, [C.c_char_p, C.c_int64]),
    "qmx_get_option": (C.c_int32, [C.c_char_p, C.POINTER(C.c_int64)]),
    "qmx_device_count": (C.c_int32, [C.POINTER(C.c_int32)]),
    "qmx_last_error": (C.c_int32, [C.c_char_p, C.c_size_t]),
    "qmx_segment_create": (C.c_int32, [C.POINTER(SegmentDesc), C.POINTER(_P)]),
    "qmx_segment_create_chunked": (C.c_int32, [C.POINTER(SegmentDesc), _P, C.c_uint64, C.c_uint32, C.POINTER(_P)]),
    "qmx_segment_destroy": (C.c_int32, [_P]),
    "qmx_segment_set_deleted": (C.c_int32, [_P, _P, C.c_uint64, _P, C.c_uint64]),
    "qmx_segment_read_rows": (C.c_int32, [_P, _P, C.c_uint32, _P]),
    "qmx_segment_row_bytes": (C.c_int32, [_P, C.POINTER(C.c_uint64)]),
    "qmx_segment_get_info": (C.c_int32, [_P, C.POINTER(SegmentInfo)]),
    "qmx_preprocess_f32": (C.c_int32, [C.c_int32, C.c_uint32, _P, C.c_uint64, C.c_uint32, _P]),
    "qmx_cast_f32": (C.c_int32, [C.c_int32, C.c_uint32, _P, C.c_uint64, _P]),
    "qmx_query_create": (C.c_int32, [_P, _P, C.c_uint32, C.POINTER(_P)]),
    "qmx_query_create_internal": (C.c_int32, [_P, _P, C.c_uint32, C.POINTER(_P)]),
    "qmx_query_update": (C.c_int32, [_P, _P]),
    "qmx_query_set_filter": (C.c_int32, [_P, _P, C.c_uint64]),
    "qmx_query_destroy": (C.c_int32, [_P]),
    "qmx_query_set_stream": (C.c_int32, [_P, _P]),
    "qmx_query_synchronize": (C.c_int32, [_P]),
    "qmx_query_set_timing": (C.c_int32, [_P, C.c_int32]),
    "qmx_query_timing": (C.c_int32, [_P, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "qmx_query_last_kernel": (C.c_int32, [_P, C.c_char_p, C.c_size_t]),
    "qmx_query_read_encoded": (C.c_int32, [_P, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "qmx_score_points": (C.c_int32, [_P, _P, C.c_uint32, _P, C.POINTER(Counters)]),
    "qmx_score_points_ragged": (C.c_int32, [_P, _P, _P, _P, C.POINTER(Counters)]),
    "qmx_score_point": (C.c_int32, [_P, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]),
    "qmx_score_internal": (C.c_int32, [_P, _P, _P, C.c_uint32, _P]),
    "qmx_score_bytes": (C.c_int32, [_P, _P, C.c_uint32, C.c_uint64, _P]),
    "qmx_search_topk": (C.c_int32, [_P, C.c_uint32, _P, C.c_uint64, _P, _P, _P, C.POINTER(Counters)]),
    "qmx_search_topk_async": (C.c_int32, [_P, C.c_uint32, _P, C.c_uint64, _P, _P]),
    "qmx_query_last_counters": (C.c_int32, [_P, C.POINTER(Counters)]),
    "qmx_rescore": (C.c_int32, [_P, _P, _P, C.c_uint32, C.c_uint32, _P, _P]),
    "qmx_search_quantized": (C.c_int32, [_P, _P, _P, C.POINTER(SearchParams), _P, C.c_uint64, _P, _P, _P, C.POINTER(Counters)]),
    "qmx_custom_score_points": (C.c_int32, [_P, _P, C.c_uint32, _P, C.c_uint32, _P]),
    "qmx_custom_search_topk": (C.c_int32, [_P, _P, C.c_uint32, C.c_uint32, _P, C.c_uint64, _P, _P]),
    "qmx_custom_hnsw_search": (C.c_int32, [_P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P, _P, C.POINTER(Counters)]),
    "qmx_multi_custom_score_points": (C.c_int32, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, _P]),
    "qmx_multi_custom_search_topk": (C.c_int32, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint64, C.c_uint32, _P, C.c_uint64, _P, _P]),
    "qmx_multi_custom_hnsw_search": (C.c_int32, [_P, _P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint64, C.c_uint32, C.c_uint32, _P, _P,
                                                 C.POINTER(Counters)]),
    "qmx_merge_topk": (C.c_int32, [C.c_int32, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P]),
    "qmx_merge_topk_async": (C.c_int32, [C.c_int32, _P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P]),
    "qmx_topk_record_bytes": (C.c_uint64, [C.c_uint32, C.c_uint32]),
    "qmx_merge_topk_packed_async": (C.c_int32, [C.c_int32, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P]),
    "qmx_sharded_search_topk": (C.c_int32, [_P, C.c_uint32, C.c_uint32, _P, _P, _P, _P, C.POINTER(Counters)]),
    "qmx_sharded_search_topk_async": (C.c_int32, [_P, C.c_uint32, C.c_uint32, _P, _P, _P]),
    "qmx_sharded_hnsw_search": (C.c_int32, [_P, _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P, _P, _P, C.POINTER(Counters)]),
    "qmx_sharded_query_update": (C.c_int32, [_P, C.c_uint32, _P]),
    "qmx_hnsw_create": (C.c_int32, [C.POINTER(HnswDesc), C.POINTER(_P)]),
    "qmx_hnsw_create_from_plain_file": (C.c_int32, [_P, C.c_uint64, C.POINTER(HnswDesc), C.POINTER(_P)]),
    "qmx_hnsw_create_from_file": (C.c_int32, [_P, C.c_uint64, C.POINTER(HnswDesc), C.POINTER(_P)]),
    "qmx_quant_meta_parse": (C.c_int32, [C.c_uint32, C.c_char_p, C.c_uint64, C.POINTER(QuantMeta)]),
    "qmx_quant_meta_free": (None, [C.POINTER(QuantMeta)]),
    "qmx_graph_links_decode": (C.c_int32, [_P, C.c_uint64, C.POINTER(GraphLinks)]),
    "qmx_graph_links_free": (None, [C.POINTER(GraphLinks)]),
    "qmx_hnsw_destroy": (C.c_int32, [_P]),
    "qmx_hnsw_build": (C.c_int32, [_P, C.POINTER(HnswBuildParams), C.POINTER(_P)]),
    "qmx_hnsw_build_quantized": (C.c_int32, [_P, _P, C.POINTER(HnswBuildParams), C.POINTER(_P)]),
    "qmx_sharded_hnsw_build": (C.c_int32, [C.POINTER(_P), C.POINTER(_P), C.c_uint32, C.POINTER(HnswBuildParams), C.POINTER(_P), C.POINTER(C.c_int32)]),
    "qmx_hnsw_get_info": (C.c_int32, [_P, C.POINTER(HnswInfo)]),
    "qmx_hnsw_export_plain": (C.c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "qmx_hnsw_search": (C.c_int32, [_P, _P, C.c_uint32, C.c_uint32, _P, _P, _P, C.POINTER(Counters)]),
    "qmx_hnsw_search_acorn": (C.c_int32, [_P, _P, C.c_uint32, C.c_uint32, _P, _P, _P, C.POINTER(Counters)]),
    "qmx_hnsw_search_async": (C.c_int32, [_P, _P, C.c_uint32, C.c_uint32, _P, _P, _P]),
    "qmx_hnsw_search_traced": (C.c_int32, [_P, _P, C.c_uint32, C.c_uint32, _P, _P, _P, C.c_uint32, _P]),
    "qmx_sq_encode": (C.c_int32, [C.c_int32, C.c_uint32, C.POINTER(SqParams), _P, C.c_uint64, C.c_uint32, _P]),
    "qmx_pq_train": (C.c_int32, [C.c_int32, _P, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, _P, _P]),
    "qmx_sq_fit_min_max": (C.c_int32, [C.c_int32, C.c_uint32, _P, C.c_uint64, C.c_uint32, C.POINTER(SqParams)]),
    "qmx_segment_create_from_files": (C.c_int32, [C.POINTER(SegmentDesc), C.c_char_p, C.c_char_p, C.POINTER(_P)]),
    "qmx_sq_fit_quantile": (C.c_int32, [C.c_int32, C.c_uint32, _P, C.c_uint64, C.c_uint32, C.c_uint64, C.c_float, C.POINTER(SqParams), C.POINTER(C.c_int32)]),
    "qmx_multi_score_points": (C.c_int32, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, _P]),
    "qmx_multi_search_topk": (C.c_int32, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint64, C.c_uint32, _P, C.c_uint64, _P, _P]),
    "qmx_tq_encode": (C.c_int32, [C.c_int32, C.c_uint32, C.c_uint32, _P, _P, C.c_uint64, _P]),
    "qmx_vector_stats": (C.c_int32, [C.c_int32, _P, C.c_uint64, C.c_uint32, _P, _P, _P, _P]),
    "qmx_tq_fit_plus": (C.c_int32, [C.c_int32, C.c_uint32, C.c_uint32, _P, _P, C.c_uint64, _P, _P]),
    "qmx_hnsw_search_with_vectors": (C.c_int32, [_P, _P, _P, C.c_uint32, C.c_uint32, _P, _P, _P, _P]),
    "qmx_multi_hnsw_build": (C.c_int32, [_P, _P, C.c_uint32, _P, C.c_uint64, _P, _P]),
    "qmx_multi_hnsw_build_quantized": (C.c_int32, [_P, _P, _P, C.c_uint32, _P, C.c_uint64, _P, _P]),
    "qmx_multi_hnsw_search": (C.c_int32, [_P, _P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint64, C.c_uint32, C.c_uint32, _P, _P, _P]),
    "qmx_custom_set_coefficients": (C.c_int32, [_P, _P, C.c_uint32]),
    "qmx_bq_encode_ex": (C.c_int32, [C.c_int32, C.POINTER(BqParams), _P, C.c_uint64, C.c_uint32, _P]),
    "qmx_bq_row_bytes": (C.c_uint64, [C.c_uint32, C.c_uint32]),
    "qmx_bq_encode": (C.c_int32, [C.c_int32, _P, C.c_uint64, C.c_uint32, _P]),
    "qmx_pq_encode": (C.c_int32, [C.c_int32, C.POINTER(PqParams), _P, C.c_uint64, C.c_uint32, _P]),
}

_lib = None


def lib():
    """Loads libqdrant_amd.so; fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950). "
                "qdrant_amd has no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        # the harnesses' data generators live in their own library; when it is there its entry points are reachable through the same handle
        # (lib.qmx_synth_fill_f32 / lib.qmx_synth_fill_latent_f32), so that bench.py and the tests read as before
        if os.path.exists(TESTDATA_LIB_PATH):
            td = C.CDLL(TESTDATA_LIB_PATH)
            for name, (res, args) in TESTDATA_SIGNATURES.items():
                fn = getattr(td, name)
                fn.restype = res
                fn.argtypes = args
                setattr(handle, name, fn)
        _lib = handle
    return _lib


def last_error():
    buf = C.create_string_buffer(1024)
    lib().qmx_last_error(buf, len(buf))
    return buf.value.decode(errors="replace")


def check(status):
    if status != OK:
        raise QmxError(status, last_error())


def set_option(name, value):
    """qmx_set_option: pick between result-identical kernel paths (value < 0 restores the load-time value)."""
    check(lib().qmx_set_option(name.encode(), int(value)))


def get_option(name):
    v = C.c_int64()
    check(lib().qmx_get_option(name.encode(), C.byref(v)))
    return v.value


def last_kernel(query_handle):
    """qmx_query_last_kernel: the demangled symbol of the scoring kernel the batch's last search launched."""
    buf = C.create_string_buffer(1024)
    check(lib().qmx_query_last_kernel(query_handle, buf, len(buf)))
    return buf.value.decode(errors="replace")


def ptr(x):
    """Raw pointer of a numpy array (host), a torch tensor (host or device), an int, or None."""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):  # torch.Tensor
        return C.c_void_p(x.data_ptr())
    if hasattr(x, "ctypes"):  # numpy.ndarray
        return C.c_void_p(x.ctypes.data)
    if isinstance(x, C.Array) or hasattr(x, "_b_base_"):
        return C.cast(x, C.c_void_p)
    raise TypeError(f"cannot take a pointer of {type(x)}")

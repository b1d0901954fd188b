"""qdrant_amd — MI355X-native (gfx950, HIP) implementation of Qdrant's batched vector-scoring path.

Layout: `csrc/` (HIP kernels + the C-ABI of include/qdrant_amd.h), `_ffi.py` (ctypes binding),
`scorer.py` (host-side mirror of the reference's RawScorer / BatchFilteredSearcher interface).
"""
from ._ffi import (COSINE, DOT, DTYPE_BQ, DTYPE_F16, DTYPE_F32, DTYPE_PQ, DTYPE_SQ_U8, DTYPE_U8, EUCLID, MANHATTAN,  # noqa: F401
                   QmxError, get_option, lib, set_option)
from .scorer import (BatchFilteredSearcher, Distance, EncodedVectorsPQ, EncodedVectorsU8, ProductQuantizer, RawScorer, ScalarQuantizer,  # noqa: F401
                     ScoredPointOffset, VectorStorage, VectorStorageDatatype, device_count, new_raw_scorer,
                     new_raw_scorer_internal, pq_train, search_quantized, CustomQuery, CustomRawScorer, BinaryQuantizer, EncodedVectorsBin, load_quantizer, MultiDenseVectorStorage, QuantizedMultivectorStorage, TurboQuantizer, EncodedVectorsTQ, vector_stats)
from .hnsw import GraphLayers, decode_links_file  # noqa: F401

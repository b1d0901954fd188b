/*
 * qdrant_amd.h — C-ABI of the MI355X-native vector-scoring library (libqdrant_amd.so).
 *
 * This is the drop-in boundary for ONE hot path of qdrant/qdrant v1.19.0: the batched
 * vector-scoring path behind the `Metric` / `QueryScorer` / `RawScorer` traits of lib/segment
 * and the `EncodedVectors` trait of lib/quantization.  Every entry point below names the
 * reference interface (file:line under the qdrant tree) it replaces.  A Rust maintainer binds
 * these with a plain `unsafe extern "C"` block, exactly like the reference
 * already binds its own C SIMD leaves (lib/quantization/src/encoded_vectors_u8.rs:833-846).
 *
 * Conventions
 *  - plain pointers + sizes; no C++/torch types; all functions return `int32_t` status
 *    (`QMX_OK` == 0).  Nothing throws or aborts across the ABI.  `qmx_last_error` returns a
 *    thread-local message for the last failing call on this thread.
 *  - data pointers (`vectors`, `queries`, `ids`, `out`, ...) may point to HOST memory or to
 *    DEVICE (HBM) memory of the segment's GPU; the library inspects the pointer.  Control
 *    structs (descriptors, params, counters) are always host memory.
 *  - a `qmx_segment` is immutable after creation and safe for concurrent use from any number
 *    of host threads (reference: storages are shared `&` across search threads,
 *    lib/segment/src/index/plain_vector_index/mod.rs:48-52).  A `qmx_query` (one batch of
 *    queries = a batch of `RawScorer`s) belongs to one thread at a time, like
 *    `Box<dyn RawScorer>` (lib/segment/src/vector_storage/raw_scorer.rs:60-64).
 *  - there is NO CPU fallback inside the library: without a usable gfx950 device every call
 *    returns QMX_ERR_NO_DEVICE and the caller keeps its CPU scorer, mirroring the reference's
 *    "GPU error => fall back to CPU" contract (lib/segment/src/index/hnsw_index/hnsw/gpu_build.rs:134-139).
 */
#ifndef QDRANT_AMD_H
#define QDRANT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QMX_API __attribute__((visibility("default")))

/* Status codes.  1..6 mirror `gpu::GpuError` (lib/gpu/src/lib.rs:50-72); CANCELLED mirrors
 * `OperationError::Cancelled` raised by `check_process_stopped`
 * (lib/segment/src/common/operation_error.rs:399-411). */
typedef enum qmx_status {
    QMX_OK = 0,
    QMX_ERR_OUT_OF_MEMORY = 1,
    QMX_ERR_OUT_OF_BOUNDS = 2,
    QMX_ERR_NOT_SUPPORTED = 3,
    QMX_ERR_NOT_READY = 4,
    QMX_ERR_TIMEOUT = 5,
    QMX_ERR_OTHER = 6,
    QMX_ERR_CANCELLED = 7,
    QMX_ERR_BAD_ARG = 8,
    QMX_ERR_NO_DEVICE = 9
} qmx_status;

/* Element type of the stored block.  F32/F16/U8 = `VectorStorageDatatype`
 * (lib/segment/src/types.rs, Float32/Float16/Uint8); SQ_U8 = `EncodedVectorsU8`
 * (lib/quantization/src/encoded_vectors_u8.rs:33-37); PQ = `EncodedVectorsPQ`
 * (lib/quantization/src/encoded_vectors_pq.rs:33-37). */
typedef enum qmx_dtype {
    QMX_DTYPE_F32 = 0,
    QMX_DTYPE_F16 = 1,
    QMX_DTYPE_U8 = 2,
    QMX_DTYPE_SQ_U8 = 3,
    QMX_DTYPE_PQ = 4,
    QMX_DTYPE_BQ = 5,  /* EncodedVectorsBin<u128>; Encoding and QueryEncoding in qmx_bq_params (default OneBit, SameAsStorage)
                          (lib/quantization/src/encoded_vectors_binary.rs): rows of ceil(dim / 128) * 16 bytes,
                          bit i = vector[i] > 0; invert derives from the distance (quantized_vectors.rs:232) */
    QMX_DTYPE_TQ = 6   /* EncodedVectorsTQ (lib/quantization/src/encoded_vectors_tq.rs over turboquant/): rows as TurboQuantizer::quantize
                          writes them, [codes: padded_dim * bits / 8 bytes][scaling_factor f32][l2_length f32 for Euclid]
                          (turboquant/encoding.rs:117-134, 172-258); qmx_tq_params required */
} qmx_dtype;

/* Same order as `enum Distance` (lib/segment/src/types.rs:313-322). */
typedef enum qmx_distance {
    QMX_DISTANCE_COSINE = 0,
    QMX_DISTANCE_EUCLID = 1,
    QMX_DISTANCE_DOT = 2,
    QMX_DISTANCE_MANHATTAN = 3
} qmx_distance;

/* `ScoredPointOffset` — #[repr(C)] {idx: u32, score: f32}, 8 bytes
 * (lib/common/common/src/types.rs:12-17).  Usable verbatim across the FFI. */
typedef struct qmx_scored_point {
    uint32_t idx;
    float score;
} qmx_scored_point;

/* What the Rust shim feeds into `HardwareCounterCell`
 * (lib/segment/src/vector_storage/query_scorer/metric_query_scorer.rs:75-76,84-85). */
typedef struct qmx_counters {
    uint64_t vectors_scored; /* (row, query) pairs scored                                  */
    uint64_t bytes_read;     /* algorithmic bytes of stored rows streamed / gathered       */
    uint64_t kernel_launches;
    float kernel_ms;         /* HIP-event time of the scoring kernels of the call (0 if not timed) */
    float reserved;
    /* Brute force through the prefilter (QMX_SEG_HALF_COPY / QMX_SEG_SPLIT_COPY blocks, f32 batches of more than 64 queries): what it cost on THIS
     * data.  All zero on the exact track.  bytes_read counts what was really streamed: the derived copy, the sample, the re-scored rows and the
     * exact passes of the queries that fell back - not rows x row_bytes. */
    uint64_t prefilter_candidates; /* (row, query) pairs the approximate scan let through (live rows)              */
    uint64_t verified_rows;        /* of those, re-scored exactly from the f32 rows                                */
    uint32_t fallback_queries;     /* queries whose candidate / verification lists overflowed (masses of near-equal
                                      scores): these, and only these, took the exact scan of the block             */
    uint32_t prefilter_queries;    /* queries of the batch served by the prefilter                                 */
} qmx_counters;

/* Flags for qmx_segment_desc.flags */
#define QMX_SEG_DATA_ON_DEVICE 0x1u /* `data` is device memory; adopt it without copying (caller keeps it alive) */
/* u8 metrics: how the exact i32 lane sums become f32.  0 (default) = AVX2 order of the live
 * x86 reference path (8 i32 lanes -> cvtepi32_ps -> f32 hsum, spaces/metric_uint/avx2/dot.rs:51-52);
 * 1 = scalar order (one i32 total cast once, spaces/metric_uint/simple_dot.rs:58-69).  The two
 * differ only when a partial sum exceeds 2^24. */
#define QMX_SEG_U8_SCALAR_ORDER 0x2u
/* time every scoring kernel with HIP events on its stream and report it in qmx_counters.kernel_ms */
#define QMX_SEG_TIME_KERNELS 0x4u
/* BQ only.  VectorParameters.invert of a segment's quantized storage is `distance == Euclid | Manhattan`
 * (lib/segment/src/vector_storage/quantized/quantized_vectors.rs:232) and that is what a BQ segment uses by default;
 * the reference's own quantization tests also build the opposite pairing (lib/quantization/tests/integration/
 * test_binary.rs:77 `test_binary_dot_inverted`, :129 l1 not inverted): this flag toggles `invert`. */
#define QMX_SEG_BQ_TOGGLE_INVERT 0x8u
/* f32 dot / cosine blocks: also keep the block as f16 PAIRS (x 2^e = h + l exactly to 2^-22, 4 bytes per element like the original) in the
 * layout the matrix cores read.  Batches of more than 64 queries then scan that copy at the HBM rate and re-score the few survivors from
 * the f32 original: the results stay the reference's bits (qdrant_amd/csrc/scan_split.hip).  Costs a second copy of the block in HBM; if it
 * does not fit the segment is created without it.  (The reference keeps derived copies of a storage the same way: its quantized storages.) */
#define QMX_SEG_SPLIT_COPY 0x10u
/* The same with the HIGH parts only (x 2^e ~ h, 2 bytes per element, half the block again in HBM): the prefilter then reads half the bytes
 * and multiplies once per element instead of three times; its band is 10x wider (each operand is within 2^-11 of its value), so a few more
 * rows are re-scored exactly.  Results are still the reference's bits; data whose best scores crowd inside 1e-3 |q| |row| of each other
 * overflow the verification list and take the exact scan instead. */
#define QMX_SEG_HALF_COPY 0x20u
/* The same with INT8 codes (1 byte per element: a quarter of the block again in HBM; dim a multiple of 128, at most 2048 like the two above - beyond 768 floats the per-query exact fallback runs in passes of 32 queries instead of 64): one scale per column, one
 * per query, the integer product on the int8 matrix cores.  The band is a worst-case bound of the two roundings (about 0.7 standard deviations of the
 * score on unit Gaussian rows), so the pass renews an EXACT lower bound of the k-th best score after each of its launches and re-scores the rows
 * whose approximate score lies within one band of it: a few hundred per query.  Results are still the reference's bits; the same per-query exact
 * fallback.  Takes precedence over the two flags above; a block with an element that is not finite gets no int8 copy. */
#define QMX_SEG_I8_COPY 0x40u
/* Let the library choose the derived copy for THIS block (takes precedence over the three flags above).  The int8 copy is the fastest where its
 * worst-case band is narrow (Gaussian-like and low-intrinsic-dimension rows, rows with a few dominant coordinates: the column scales are balanced
 * for those) and the slowest where it is wide (heavy-tailed elements: most rows inside the band are re-scored, or the lists overflow and the query
 * pays the prefilter AND the exact scan).  So the choice is measured at create: the int8 copy is built, 128 stored rows (a strided sample: queries
 * distributed like the rows) are searched through it, and unless they verify few rows each the half copy is built beside it, the same batch is timed
 * through both and the faster copy stays (the other is freed).  Costs ~50 - 100 ms per 10 M x 768 block on top of the copies' passes; what was
 * chosen and what the trial measured: qmx_segment_get_info (0 ms = that trial did not run or failed).  Results are the reference's bits either way.
 * The choice is TIMING-dependent: other work on the device during the trial can tip it, so the half copy must beat the int8 copy by 10 % to replace it,
 * and callers must not assert on which copy a segment holds - only footprint and latency differ. */
#define QMX_SEG_AUTO_COPY 0x80u
/* PQ blocks of 2^18 rows and more carry a rotated copy of their codes (ceil32(m) bytes per row) that the 8-bit prefilter of batches of 4 and more
 * queries scans (pq_prefilter.hip).  It is built by default when it at most doubles the codes' footprint (m >= 16); this flag asks for it on narrower
 * codes too (m = 8: 32 bytes per row beside the 8 of the codes). */
#define QMX_SEG_PQ_PREFILTER_COPY 0x100u

/* SQ-int8 parameters = `MetadataInt8` (lib/quantization/src/encoded_vectors_u8.rs:84-91).
 * Parity is defined on GIVEN (alpha, offset): the reference's quantile estimate samples
 * randomly (lib/quantization/src/quantile.rs:35-82). */
typedef struct qmx_sq_params {
    uint32_t actual_dim; /* dim rounded up to 16 (encoded_vectors_u8.rs:622-624) */
    float alpha;
    float offset;
    float multiplier;
    uint8_t invert;      /* VectorParameters.invert: true for Euclid / Manhattan
                            (lib/segment/src/vector_storage/quantized/quantized_vectors.rs:232) */
    uint8_t pad_[3];
} qmx_sq_params;

/* PQ codebook = `Metadata{centroids, vector_division}` (encoded_vectors_pq.rs:46-51):
 * 256 centroids, each a full `dim`-long f32 vector "flattened by chunks". */
typedef struct qmx_pq_params {
    uint32_t chunk_size;     /* f32 elements per chunk; m = ceil(dim / chunk_size) (get_vector_division :164-169) */
    uint32_t n_centroids;    /* CENTROIDS_COUNT = 256 (or fewer when count <= 256, :354-362) */
    const float *centroids;  /* [n_centroids][dim], host or device */
    uint8_t invert;
    uint8_t lut_mfma;        /* 1: build the query LUT with f32 MFMA (fma chain, <=1e-5 rel vs reference);
                                0: exact reference order (mul then add, sequential)  */
    uint8_t pad_[2];
} qmx_pq_params;

/* One stored vector block = what `DenseVectorStorageRead` exposes
 * (lib/segment/src/vector_storage/vector_storage_base.rs:265-316): N rows of `dim` elements,
 * row-major, fixed stride (immutable file layout: dense/immutable_dense_vectors.rs:100-115 —
 * pass `file_base + 4` to skip the "data" header).  For SQ_U8 rows are the reference layout
 * `[f32 vector_offset][u8 code x actual_dim]` (encoded_vectors_u8.rs:22-24,626-629), for PQ
 * `m` code bytes (encoded_vectors_pq.rs:617-619). */
/* Binary quantization beyond one bit: `Encoding::{TwoBits, OneAndHalfBits}` (lib/quantization/src/encoded_vectors_binary.rs:62-78,
 * 570-672): bit i = value above the per-dimension zero band, a second bit plane marks values above it (1.5 bits: one such bit
 * per PAIR of dimensions, OR-ed).  The band comes from `VectorStats` (vector_stats.rs: mean and stddev per dimension, computed by the
 * reference while it reads the vectors): given here, like the SQ interval and the PQ centroids.  NULL mean / stddev = no stats.
 * Scoring is the one-bit xor-popcount over the longer rows, with the ORIGINAL dim in calculate_metric. */
typedef enum qmx_bq_encoding { QMX_BQ_ONE_BIT = 0, QMX_BQ_TWO_BITS = 1, QMX_BQ_ONE_AND_HALF_BITS = 2 } qmx_bq_encoding;
/* `QueryEncoding` (encoded_vectors_binary.rs:48-54): how qmx_query_create encodes a query against a BQ segment.  SameAsStorage: the
 * row encoding (xor-popcount of two bit rows).  Scalar4bits / Scalar8bits (asymmetric quantization, `encode_scalar_query_vector`
 * :692-756): each query value keeps 4 / 8 bits over [-max_abs, max_abs], stored as bit planes; the score is the plane-weighted
 * xor-popcount divided by 2^bits - 1 (`xor_popcnt_scalar` :337-409, `calculate_metric` :783-788).  Stored <-> stored scores
 * (qmx_query_create_internal, qmx_score_internal) are one-bit whatever the query encoding (:892-917). */
typedef enum qmx_bq_query_encoding { QMX_BQ_QUERY_SAME_AS_STORAGE = 0, QMX_BQ_QUERY_SCALAR_4BITS = 1, QMX_BQ_QUERY_SCALAR_8BITS = 2 } qmx_bq_query_encoding;
typedef struct qmx_bq_params {
    uint32_t encoding;      /* qmx_bq_encoding */
    uint32_t query_encoding; /* qmx_bq_query_encoding */
    const float *mean;      /* [dim] host or device, or NULL */
    const float *stddev;    /* [dim] host or device, or NULL */
} qmx_bq_params;

/* TurboQuant (`TurboQuantizer`, lib/quantization/src/turboquant/quantization.rs:14-158; `Metadata`, encoded_vectors_tq.rs:33-46).
 * TQMode::Normal, and TQMode::Plus with the storage's persisted error correction (`shift` / `scale` per rotated coordinate: fitted by qmx_tq_fit_plus
 * as the reference does, with P-square quantile estimators over sampled vectors, encoded_vectors_tq.rs:156-240).  Distances Dot, Cosine, Euclid - and Manhattan
 * (DistanceType::L1: the reference has no integer kernel for it, score_precomputed :596-607 dequantises the row, rotates it back and sums |q - v|,
 * score_symmetric :429-440 likewise on the difference of two rows; served here for qmx_score_points[_ragged], qmx_search_topk, qmx_score_internal,
 * qmx_search_quantized, qmx_tq_encode and qmx_hnsw_search - the walk for rotations over a multiple of 16 coordinates up to 1024, of 32 up to 2048 or of 64
 * up to 4096; other lengths, custom / multi-vector walks and the BUILD through such a storage: QMX_ERR_NOT_SUPPORTED).  Queries are rotated (HadamardRotation, f64, the
 * reference's fixed permutation seeds) and integer-encoded on the device (`precompute_query` :496-567 with the x86_64 constants of
 * turboquant/simd/query{4,2,1}bit); scores are `score_precomputed` (:569-620) negated when `invert`; qmx_score_internal is
 * `score_symmetric` (:395-445).  `encode_internal_vector` is None (:453-459): qmx_query_create_internal is QMX_ERR_NOT_SUPPORTED, as for PQ. */
typedef enum qmx_tq_bits { QMX_TQ_BITS4 = 0, QMX_TQ_BITS2 = 1, QMX_TQ_BITS1_5 = 2, QMX_TQ_BITS1 = 3 } qmx_tq_bits;   /* TQBits, turboquant/mod.rs:15-20 */
typedef struct qmx_tq_params {
    uint32_t bits;               /* qmx_tq_bits */
    uint32_t rotation_unpadded;  /* TQRotation: 0 = Padded (rotate all padded_dim coordinates), 1 = Unpadded (only the first dim) */
    uint8_t invert;              /* VectorParameters.invert */
    uint8_t plus_mode;           /* TQMode::Plus: needs ec_shift / ec_scale */
    uint8_t pad_[2];
    uint32_t reserved;
    const float *ec_shift;       /* TQ+ `ErrorCorrectionMetadata.shift` [padded_dim] (encoded_vectors_tq.rs:93-96), HOST array, or NULL */
    const float *ec_scale;       /* ... `.scale` [padded_dim] */
} qmx_tq_params;

/* The first pass of `EncodedVectorsTQ::encode` under TQMode::Plus (encoded_vectors_tq.rs:156-234): every sampled vector is rotated and rescaled to
 * norm sqrt(padded_dim) (`TurboQuantizer::preprocess_into`, quantization.rs:169-207) and each rotated coordinate feeds a pair of 7-marker P-square
 * estimators (`find_quantile_interval_per_coordinate_with_preprocess`, quantile.rs:130-281; p_square.rs, its x86_64 AVX2 + FMA arithmetic) tracking the
 * quantiles Phi(-+c_outer); shift = -(q_lo + q_hi) / 2, scale = 2 c_outer / (q_hi - q_lo) (1 below MIN_QUANTILE_WIDTH).  `sample` = the sampled vectors
 * [n_sample][dim] f32 (host or device) in ascending index order - WHICH vectors is the reference's `Permutor` over `bits.sample_size()` = 2 048 / 4 096 /
 * 8 192 indices: an input, like the PQ k-means sample.  `params`: bits and rotation; its error-correction fields are ignored.  shift_out / scale_out:
 * [padded_dim] f32 (host or device), the arrays `qmx_tq_params.ec_shift / ec_scale` and the metadata take.  Equal to the oracle's restatement bit for bit. */
QMX_API int32_t qmx_tq_fit_plus(int32_t device_id, uint32_t distance, uint32_t dim, const qmx_tq_params *params, const float *sample, uint64_t n_sample,
                                float *shift_out, float *scale_out);

/* `TurboQuantizer::quantize` (turboquant/quantization.rs:211-296; TQMode::Normal, and TQMode::Plus when `params` carries ec_shift / ec_scale: the shift / scale are applied to the rotated
 * coordinates and `xm` joins the row's extras) for a batch, on the device: vectors [n][dim] f32 as the storage
 * holds them (cosine: normalised; host or device) -> out_rows [n][quantized size] bytes (host or device), the rows a QMX_DTYPE_TQ segment takes.
 * Rotation, length rescale and the two f64 sums (l2 length, centroid norm) follow the reference's operation order: rows are byte-identical to
 * the oracle's restatement. */
QMX_API int32_t qmx_tq_encode(int32_t device_id, uint32_t distance, uint32_t dim, const qmx_tq_params *params, const float *vectors, uint64_t n,
                              void *out_rows);

typedef struct qmx_segment_desc {
    uint32_t dtype;            /* qmx_dtype */
    uint32_t distance;         /* qmx_distance */
    uint32_t dim;              /* ORIGINAL vector dimension */
    uint32_t flags;            /* QMX_SEG_* */
    uint64_t n;                /* rows */
    uint64_t row_stride_bytes; /* 0 = tightly packed */
    const void *data;          /* host or device rows */
    int32_t device_id;         /* HIP device ordinal */
    int32_t reserved;
    const qmx_sq_params *sq;   /* required for QMX_DTYPE_SQ_U8 */
    const qmx_pq_params *pq;   /* required for QMX_DTYPE_PQ */
    const qmx_bq_params *bq;   /* optional for QMX_DTYPE_BQ: NULL = Encoding::OneBit */
    const qmx_tq_params *tq;   /* required for QMX_DTYPE_TQ */
} qmx_segment_desc;

/* The quantizers' metadata file ("quantized.meta.json", vector_storage/quantized/quantized_vectors/config.rs:13) as
 * serde_json wrote it: `MetadataInt8` (encoded_vectors_u8.rs:43-47, 84-91), PQ `Metadata` (encoded_vectors_pq.rs:46-51),
 * BQ `Metadata` (encoded_vectors_binary.rs:112-125), each with `VectorParameters` (encoded_vectors.rs:28-39).
 * qmx_quant_meta_parse is the read side of `EncodedVectors*::load`: it fills the parameter struct qmx_segment_desc takes
 * for `dtype` (the other two stay zero).  Floats go through f64 and narrow to f32 exactly like serde_json, so values the
 * reference wrote come back bit-identical.  Pointers inside (`pq.centroids`, `bq.mean`, `bq.stddev`) are library-owned
 * host arrays, valid until qmx_quant_meta_free.  Host only: needs no device.
 * PQ: `vector_division` must be the uniform division the reference itself produces (get_vector_division :164-169),
 * anything else is QMX_ERR_NOT_SUPPORTED. */
typedef struct qmx_quant_meta {
    uint32_t dtype;                 /* QMX_DTYPE_SQ_U8 | QMX_DTYPE_PQ | QMX_DTYPE_BQ | QMX_DTYPE_TQ, as asked  */
    uint32_t dim;                   /* VectorParameters.dim                                                  */
    uint32_t distance;              /* qmx_distance of VectorParameters.distance_type (L1 = Manhattan, L2 = Euclid) */
    uint8_t invert;                 /* VectorParameters.invert                                               */
    uint8_t has_deprecated_count;   /* VectorParameters.count was present                                    */
    uint8_t bq_query_encoding;      /* QueryEncoding: 0 SameAsStorage, 1 Scalar4bits, 2 Scalar8bits          */
    uint8_t pad_;
    uint64_t deprecated_count;
    qmx_sq_params sq;
    qmx_pq_params pq;
    qmx_bq_params bq;
    qmx_tq_params tq;               /* EncodedVectorsTQ `Metadata` (encoded_vectors_tq.rs:33-46): bits, mode (plus_mode), rotation; invert copied */
    void *owner;
} qmx_quant_meta;
QMX_API int32_t qmx_quant_meta_parse(uint32_t dtype, const char *json, uint64_t n_bytes, qmx_quant_meta *out);
QMX_API void qmx_quant_meta_free(qmx_quant_meta *meta);

typedef struct qmx_segment qmx_segment;
typedef struct qmx_query qmx_query;
typedef struct qmx_hnsw qmx_hnsw;

/* ---- library / device ------------------------------------------------------------------ */

/* Number of usable gfx950 devices (replaces `GpuDevicesMaganer` enumeration,
 * lib/segment/src/index/hnsw_index/gpu/gpu_devices_manager.rs:11-15). */
QMX_API int32_t qmx_device_count(int32_t *out_count);
/* Copies the calling thread's last error message (NUL-terminated) into buf. */
QMX_API int32_t qmx_last_error(char *buf, size_t buf_len);
/* ABI version; bumped on any signature change. */
QMX_API uint32_t qmx_abi_version(void);
/* Kernel-path options (process-wide).  Each selects between kernels that return IDENTICAL results - the parity tests run
 * both sides of every option - so they are tuning / triage switches, never correctness switches: "no_mfma_scan", "no_mfma16",
 * "no_prescan", "prescan_shift", "hnsw_log_cap", "no_split_scan", "split_min_queries", "no_split256", "no_pq_pair", "no_pq_prefilter",
 * "pq_prefilter_min_queries", "hnsw_pq_per_cu", "tq_rotate_block", "no_topk_small", "verify_max_per_query", "hnsw_pq_direct_walk",
 * "hnsw_pq_table_build", "hnsw_no_pq_prefilter", "hnsw_no_lds_visited", "pq_lut_no_lds", "hnsw_per_cu", "tq_wide_min_queries", "tq_wide_high_digit", "sq_wide_min_queries", "i8_resident", "debug"
 * - and one that selects the ORDER AMONG EQUAL SCORES of the plain HNSW walk: "hnsw_reference_heap_order" (see qmx_hnsw_search_traced) -
 * (qdrant_amd/csrc/common.hpp says what each selects; round 6 removed the experiments that had measured slower twice: their numbers stay under
 * profiles/).  Initial values come from the environment variables QMX_<NAME> read
 * ONCE when the library is loaded; value < 0 restores that initial value.  Unknown name => QMX_ERR_BAD_ARG.  (No reference
 * counterpart: the reference selects its SIMD leaf by cpu feature detection, spaces/simple.rs:15-33.) */
QMX_API int32_t qmx_set_option(const char *name, int64_t value);
QMX_API int32_t qmx_get_option(const char *name, int64_t *out_value);

/* ---- segment (device-resident copy of one vector storage) --------------------------------- */

/* Uploads (or adopts) the block.  Replaces opening a `VectorStorageEnum` for scoring
 * (raw_scorer.rs:60-114 matches on it) / `GpuVectorStorage::new`
 * (hnsw_index/gpu/gpu_vector_storage/mod.rs).  OOM => QMX_ERR_OUT_OF_MEMORY, caller keeps CPU. */
QMX_API int32_t qmx_segment_create(const qmx_segment_desc *desc, qmx_segment **out);
/* Direct ingestion of the reference's storage files:
 *   dense f32 / f16 / u8: the immutable dense vector file = 4-byte header "data", then row-major rows
 *     (lib/segment/src/vector_storage/dense/immutable_dense_vectors.rs:25-27, 90, 105-110: num_vectors = (len - 4) / dim / sizeof(T));
 *   SQ / PQ / BQ: the quantized storage file = flat rows of quantized_vector_size bytes, no header
 *     (vector_storage/quantized/quantized_storage.rs:25-70); desc->sq / desc->pq carry the metadata JSON's values;
 *   deleted_path (optional): the "drop" flags file = 4-byte header "drop", padding to 8 bytes, then the BitSlice<u64, Lsb0>
 *     words (immutable_dense_vectors.rs:27, 364-378) -> the vector-deleted flags of the segment.
 * desc->data is ignored; desc->n = 0 takes the row count from the file size, otherwise the file must hold at least n rows.
 * The file is streamed through a pinned staging buffer: no host copy of the block is kept. */
QMX_API int32_t qmx_segment_create_from_files(const qmx_segment_desc *desc, const char *vectors_path, const char *deleted_path,
                                              qmx_segment **out);

/* Same for a CHUNKED appendable storage (`ChunkedVectors<T>`, lib/segment/src/vector_storage/chunked_vectors.rs: fixed-size
 * chunks of CHUNK_SIZE = 32 MiB, vector_storage/common.rs:27; row `key` lives in chunk key / rows_per_chunk at row
 * key % rows_per_chunk): `chunks[c]` points at chunk c (host or device), every chunk but the last holds `rows_per_chunk`
 * rows; `desc->data` is ignored, `desc->n` is the total row count.  The rows are gathered into one HBM block.
 * Quantized chunked storages too (`QuantizedChunkedMmapStorage`, vector_storage/quantized/quantized_chunked_mmap_storage/read_only.rs:20,
 * read_write.rs:18: the appendable form of the quantized storages): SQ / PQ / BQ / TQ dtypes with the quantizer's parameters in `desc`, rows in the
 * quantizer's own layout. */
QMX_API int32_t qmx_segment_create_chunked(const qmx_segment_desc *desc, const void *const *chunks, uint64_t rows_per_chunk,
                                           uint32_t n_chunks, qmx_segment **out);
QMX_API int32_t qmx_segment_destroy(qmx_segment *seg);
/* Deleted flags = the two `BitSlice<u64, Lsb0>` of `NotDeletedChecker`
 * (raw_scorer.rs:580-603; layout lib/common/common/src/bitvec.rs:6-7).  A point past
 * `n_point_bits` counts as deleted, a vector past `n_vec_bits` as not deleted (:596-603).
 * NULL/0 for both = nothing deleted, every row < n live.  Not thread-safe against running
 * searches on the same segment (the reference takes these by shared borrow per search). */
QMX_API int32_t qmx_segment_set_deleted(qmx_segment *seg, const uint64_t *point_deleted,
                                        uint64_t n_point_bits, const uint64_t *vec_deleted,
                                        uint64_t n_vec_bits);
/* Reads rows back (device -> host), for tests: `get_dense` / `get_quantized_vector`. */
QMX_API int32_t qmx_segment_read_rows(const qmx_segment *seg, const uint32_t *ids, uint32_t n,
                                      void *out_rows /* [n][row_bytes] reference layout */);
QMX_API int32_t qmx_segment_row_bytes(const qmx_segment *seg, uint64_t *out);
/* What a segment holds beside its rows: the derived copy of an f32 dot / cosine block (QMX_SEG_*_COPY flags) and, under QMX_SEG_AUTO_COPY, what the
 * trial at create measured.  (The reference's counterpart is the segment telemetry, `VectorIndexSearchesTelemetry` / `SegmentInfo`: which index and
 * quantization serve a segment is reported, not guessed.) */
typedef struct qmx_segment_info {
    uint32_t derived_copy;          /* 0 = none, else ONE of QMX_SEG_I8_COPY / QMX_SEG_HALF_COPY / QMX_SEG_SPLIT_COPY: the copy the prefilter streams */
    uint32_t chosen_by_trial;       /* 1 = QMX_SEG_AUTO_COPY decided it                                                                            */
    uint64_t derived_copy_bytes;    /* HBM the copy takes                                                                                          */
    float i8_scale_balance;         /* int8 copy: the range ratio G the column scales were balanced to (0 = every column at max |x| / 127)         */
    float trial_i8_ms;              /* AUTO: the 128-query trial batch through the int8 copy, milliseconds (best of three)                         */
    float trial_half_ms;            /* ... through the half copy (0 = not tried: the int8 trial verified few rows per query)                       */
    float trial_i8_verified_rows;   /* ... exactly re-scored rows per query of the int8 trial                                                       */
    uint32_t trial_i8_fallback_queries; /* ... of its 128 queries, how many overflowed their lists and took the exact scan                          */
    uint32_t reserved;
} qmx_segment_info;
QMX_API int32_t qmx_segment_get_info(const qmx_segment *seg, qmx_segment_info *out);

/* ---- Metric::preprocess ---------------------------------------------------------------- */

/* `Metric<T>::preprocess` on a batch (lib/segment/src/spaces/metric.rs:15-16): cosine
 * normalisation with the reference's skip rule (spaces/tools.rs:14-16) and AVX accumulation
 * order (spaces/simple_avx.rs:127-165); identity for the other distances.  `out` may alias `in`. */
QMX_API int32_t qmx_preprocess_f32(int32_t device_id, uint32_t distance, const float *in,
                                   uint64_t n, uint32_t dim, float *out);
/* Element casts of `PrimitiveVectorElement::slice_from_float_cow`
 * (lib/segment/src/data_types/primitive.rs:76-79 f16 RNE; :127-129 u8 saturating truncation). */
QMX_API int32_t qmx_cast_f32(int32_t device_id, uint32_t dst_dtype, const float *in, uint64_t count,
                             void *out);

/* ---- query batch = batch of RawScorers ---------------------------------------------------- */

/* `new_raw_scorer(QueryVector::Nearest(q), storage)` for each of `nq` queries
 * (raw_scorer.rs:60-114 -> MetricQueryScorer::new, metric_query_scorer.rs:35-58): preprocess
 * once, cast to the element type; for SQ/PQ segments `EncodedVectors::encode_query`
 * (encoded_vectors_u8.rs:583-619; encoded_vectors_pq.rs:519-541 — the LUT), all on device.
 * `queries`: [nq][dim] f32 ORIGINAL (un-preprocessed) vectors. */
QMX_API int32_t qmx_query_create(const qmx_segment *seg, const float *queries, uint32_t nq,
                                 qmx_query **out);
/* `FilteredScorer::new_internal` (hnsw_index/point_scorer.rs:183-218): the stored vector
 * `point_ids[i]` becomes query i (SQ: `encode_internal_vector`, encoded_vectors_u8.rs:715-728;
 * PQ returns None there, so the caller passes the original vector to qmx_query_create). */
QMX_API int32_t qmx_query_create_internal(const qmx_segment *seg, const uint32_t *point_ids,
                                          uint32_t nq, qmx_query **out);
/* Re-encodes a NEW batch of the same size into an existing handle (the next request reusing the
 * scorer slot): only enqueues the preprocess / cast kernels on the query's stream, no allocation.
 * `queries` host or device. */
QMX_API int32_t qmx_query_update(qmx_query *q, const float *queries);
/* The payload filter of `ScorerFilters` (hnsw_index/point_scorer.rs:78-85: `check_vector(id)` = not deleted AND
 * `filter_context.check(id)`) for this query batch, evaluated by the caller's payload index into an allow bitmap
 * (`BitSlice<u64, Lsb0>`, bit id = point id passes; ids past n_bits are rejected).  Applies to the brute-force
 * stream (like `peek_top_iter` over filtered points), to qmx_hnsw_search (filtered walk: rejected points are neither
 * scored nor traversed, as in `FilteredScorer::score_points`) — not to the plain score_points entry points, which
 * score what they are given.  NULL clears the filter.  The bitmap is copied. */
QMX_API int32_t qmx_query_set_filter(qmx_query *q, const uint64_t *allowed, uint64_t n_bits);
QMX_API int32_t qmx_query_destroy(qmx_query *q);
/* Run this query batch's kernels on a caller-owned hipStream_t (NULL = the query's own stream). */
QMX_API int32_t qmx_query_set_stream(qmx_query *q, void *hip_stream);
QMX_API int32_t qmx_query_synchronize(qmx_query *q);
/* enabled != 0: bracket every scoring (scan / gather) kernel of this query batch with a HIP event
 * pair on its stream.  Recording never synchronises; the synchronous entry points report the sum
 * in qmx_counters.kernel_ms, and qmx_query_timing synchronises the stream, returns the total and
 * launch count since the previous call and resets them (how bench.py derives roofline.achieved). */
QMX_API int32_t qmx_query_set_timing(qmx_query *q, int32_t enabled);
QMX_API int32_t qmx_query_timing(qmx_query *q, float *total_ms, uint32_t *n_launches);
/* The (demangled) symbol of the scoring kernel the last qmx_search_topk* / qmx_hnsw_search* call of this batch launched - what a
 * profiler trace of the same call shows - so that measurements name the kernel they timed instead of guessing the dispatch.
 * Empty string before the first search.  (The reference's counterpart is its per-leaf dispatch, spaces/simple.rs:15-33.) */
QMX_API int32_t qmx_query_last_kernel(const qmx_query *q, char *buf, size_t buf_len);
/* Encoded form read-back for tests: f32/f16/u8 preprocessed+cast query; SQ: [f32 offset][codes];
 * PQ: the LUT [m][n_centroids] f32. */
QMX_API int32_t qmx_query_read_encoded(const qmx_query *q, uint32_t query_index, void *out,
                                       uint64_t out_bytes, uint64_t *written);

/* ---- RawScorer surface (gather scoring) ---------------------------------------------------- */

/* `RawScorer::score_points(points, scores)` (raw_scorer.rs:40, 561-564) for every query of the
 * batch against the SAME id list: scores[qi * n + i] = similarity(query qi, row ids[i]).
 * ids must be < n rows (QMX_ERR_OUT_OF_BOUNDS otherwise; the reference panics). */
QMX_API int32_t qmx_score_points(qmx_query *q, const uint32_t *ids, uint32_t n, float *scores,
                                 qmx_counters *counters);
/* Ragged form for HNSW hops batched across concurrent searches: query qi scores
 * ids[offsets[qi] .. offsets[qi+1]) into scores at the same positions
 * (graph_layers.rs:305-313,125-139 produce <= m0 ids per hop per query). */
QMX_API int32_t qmx_score_points_ragged(qmx_query *q, const uint32_t *ids, const uint32_t *offsets,
                                        float *scores, qmx_counters *counters);
/* `RawScorer::score_point` (raw_scorer.rs:43). */
QMX_API int32_t qmx_score_point(qmx_query *q, uint32_t query_index, uint32_t id, float *out);
/* `RawScorer::score_internal(a, b)` stored<->stored (raw_scorer.rs:50;
 * metric_query_scorer.rs:94-99; SQ encoded_vectors_u8.rs:675-705; PQ encoded_vectors_pq.rs:574-618),
 * batched: out[i] = score(a[i], b[i]). */
QMX_API int32_t qmx_score_internal(const qmx_segment *seg, const uint32_t *a, const uint32_t *b,
                                   uint32_t n, float *out);
/* `QueryScorerBytes::score_bytes` (query_scorer/mod.rs:48-68): rows given as raw bytes in the
 * segment's reference row layout (inline link vectors, graph_layers.rs:336-389). */
QMX_API int32_t qmx_score_bytes(qmx_query *q, const void *rows, uint32_t n, uint64_t stride_bytes,
                                float *scores /* [nq][n] */);

/* ---- brute-force search ---------------------------------------------------------------------- */

/* `BatchFilteredSearcher::peek_top_iter` (hnsw_index/point_scorer.rs:423-472) as driven by
 * `PlainVectorIndexReadView::search` (plain_vector_index/read_view/search.rs:54-135):
 *   ids == NULL : candidate stream = every row whose point-deleted bit is zero
 *                 (`iter_not_deleted`, :400-406), vector-deleted rows skipped (`check_vector`);
 *   ids != NULL : candidate stream = that list (payload-filtered ids, search.rs:104-108),
 *                 still subject to the deleted flags.
 * For each query keeps the `top` best by score (`FixedLengthPriorityQueue`,
 * lib/common/common/src/fixed_length_priority_queue.rs:47-59) and returns them sorted by
 * descending score (`into_sorted_vec`, :63-65); among equal scores the lower id comes first
 * (the reference's order among equals is heap-dependent).
 *   top        : 1..65536.  Up to 64 entries live in one register list per wavefront; a larger `top` runs
 *                ceil(top / 64) passes over the candidates, pass p keeping the best 64 keys strictly below the
 *                last key of pass p - 1 (keys are unique: score, then offset), so the result is the same list.
 *   out        : [nq][top] ScoredPointOffset;  out_counts : [nq] number of valid entries.
 *   is_stopped : polled between kernel launches; non-zero => QMX_ERR_CANCELLED
 *                (`check_process_stopped`, point_scorer.rs:433,437).  May be NULL. */
QMX_API int32_t qmx_search_topk(qmx_query *q, uint32_t top, const uint32_t *ids, uint64_t n_ids,
                                qmx_scored_point *out, uint32_t *out_counts,
                                const volatile uint8_t *is_stopped, qmx_counters *counters);
/* Same, but only enqueues on the query's stream; `out`/`out_counts` must be device memory.
 * Complete with qmx_query_synchronize. */
QMX_API int32_t qmx_search_topk_async(qmx_query *q, uint32_t top, const uint32_t *ids,
                                      uint64_t n_ids, qmx_scored_point *out_dev,
                                      uint32_t *out_counts_dev);
/* The counters of the last qmx_search_topk[_async] enqueued on this batch (synchronises its stream): what the asynchronous form cannot
 * hand back at enqueue time - the `HardwareCounterCell` increments of the search (metric_query_scorer.rs:75-85), incl. the prefilter's
 * candidates / re-scored rows / fallback queries. */
QMX_API int32_t qmx_query_last_counters(qmx_query *q, qmx_counters *out);

/* `postprocess_search_result` rescoring (lib/segment/src/index/vector_index_search_common.rs:73-90):
 * query qi re-scores ids[qi * n_per_query .. +counts[qi]) with the ORIGINAL-vector scorer `q`,
 * sorts descending, truncates to `top`. */
QMX_API int32_t qmx_rescore(qmx_query *q, const uint32_t *ids, const uint32_t *counts,
                            uint32_t n_per_query, uint32_t top, qmx_scored_point *out,
                            uint32_t *out_counts);

/* `SearchParams.quantization` subset + `hnsw_ef` (lib/segment/src/types.rs: QuantizationSearchParams{ignore, rescore,
 * oversampling}); `ignore` / `exact` are expressed by passing the original-vector batch as the searched one. */
typedef struct qmx_search_params {
    uint32_t top;
    float oversampling;   /* > 1.0: the quantized stage returns (oversampling * top) as usize candidates (:27-46) */
    uint8_t rescore;      /* re-score the candidates with the original vectors (default_rescoring or the request's) */
    uint8_t acorn;        /* only with a graph: SearchAlgorithm::Acorn instead of Hnsw (hnsw/read_view/search.rs:42-90) */
    uint8_t pad_[2];
    uint32_t hnsw_ef;     /* only with a graph; raised to the oversampled top (graph_layers.rs:549) */
} qmx_search_params;

/* One call for `PlainVectorIndexReadView::search` (plain_vector_index/read_view/search.rs:54-135, g == NULL) or the
 * graph arm of `HNSWIndexReadView::search` (hnsw/read_view/search.rs:30-179, g != NULL), including
 * `get_oversampled_top` and `postprocess_search_result` (vector_index_search_common.rs:27-91): stage 1 searches the
 * batch `searched` (the quantized scorers when `is_quantized_search`, else the original ones) for the oversampled top,
 * stage 2 re-scores those candidates with `original` (same queries over the original vectors; may be NULL when
 * rescore == 0), sorts descending and truncates to `top`.  Candidates stay on the device between the stages.
 * ids / n_ids: optional candidate list of the brute-force stage (payload-filtered points), ignored with a graph. */
QMX_API int32_t qmx_search_quantized(const qmx_hnsw *g, qmx_query *searched, qmx_query *original, const qmx_search_params *params,
                                     const uint32_t *ids, uint64_t n_ids, qmx_scored_point *out, uint32_t *out_counts,
                                     const volatile uint8_t *is_stopped, qmx_counters *counters);

/* ---- custom queries (recommend / discover / context) ---------------------------------------------- */

/* `QueryVector::{RecommendBestScore, RecommendSumScores, Discover, Context, FeedbackNaive}` scored by `CustomQueryScorer`
 * (lib/segment/src/vector_storage/query_scorer/custom_query_scorer.rs:44-121): score(point) =
 * query.score_by(|example| Metric::similarity(example, point)).  The EXAMPLE vectors of all custom queries of a request
 * form one ordinary query batch (`qmx_query_create`: each example is preprocessed and cast like a Nearest query,
 * custom_query_scorer.rs:58-66); a qmx_custom_query names its slice of that batch in the reference's `flat_iter()`
 * order: reco: n_a positives then n_b negatives (vector_storage/query/reco_query.rs:25-27, 68-131); discover: the
 * target (n_a = 1) then n_b (positive, negative) pairs (discover_query.rs:34-73); context: n_b pairs, n_a = 0
 * (context_query.rs:53-62, 95-118); feedback (`FeedbackQuery`, feedback_query.rs:147-227): the target (n_a = 1) then the n_b
 * (positive, negative) context pairs that `NaiveFeedbackCoefficients::extract_context_pairs` (:117-145) kept, with the
 * coefficients [a, partial_computation_0 .. partial_computation_{n_b-1}] at `coef_first` of the batch's coefficient array
 * (qmx_custom_set_coefficients): score = a * sim(target) + sum_i partial_computation_i * (sim(positive_i) - sim(negative_i)),
 * f32, in that order.  partial_computation = confidence.powf(b) * c is computed by the caller (libm's powf). */
typedef enum qmx_custom_kind {
    QMX_CUSTOM_RECO_BEST_SCORE = 0,
    QMX_CUSTOM_RECO_SUM_SCORES = 1,
    QMX_CUSTOM_DISCOVER = 2,
    QMX_CUSTOM_CONTEXT = 3,
    QMX_CUSTOM_FEEDBACK = 4
} qmx_custom_kind;
typedef struct qmx_custom_query {
    uint32_t kind;        /* qmx_custom_kind */
    uint32_t first;       /* index of the query's first example inside the example batch */
    uint32_t n_a;
    uint32_t n_b;
    uint32_t coef_first;  /* QMX_CUSTOM_FEEDBACK: index of `a` inside the coefficient array; ignored otherwise */
} qmx_custom_query;

/* Coefficients of the feedback queries of an example batch (copied; host or device memory). */
QMX_API int32_t qmx_custom_set_coefficients(qmx_query *examples, const float *coefs, uint32_t n);

/* Multi-dense vectors with the MaxSim comparator (`MultiVectorConfig{comparator: MaxSim}`, `score_max_similarity`,
 * lib/segment/src/vector_storage/query_scorer/mod.rs:70-97; used by `MultiMetricQueryScorer`, query_scorer/multi_metric_query_scorer.rs):
 * the storage keeps every point's inner vectors flattened (`MultiDenseVectorStorage`: vectors + per-point offsets), so the segment here
 * is the block of INNER rows and point p = inner rows [point_offsets[p], point_offsets[p + 1]).  `inner` is a query batch over that
 * segment holding the inner vectors of all multi-queries back to back; multi-query j = inner queries [query_first[j], query_first[j + 1]).
 *   score(j, p) = sum over the query's inner vectors a (in order, from 0.0) of max over the point's inner vectors b of similarity(a, b)
 * with the segment's metric (no post-processing, as the reference).  The similarities are the dense scan's (bit-identical to the
 * reference's leaves) and the two loops are the reference's: scores are bit-exact.  query_first [n_queries + 1] and point_offsets
 * [n_points + 1] are host arrays.  The inner-row similarity matrix ([inner queries][inner rows] f32) is materialised in HBM (<= 48 GB).
 *   qmx_multi_score_points: scores[j * n + i] for point ids[i].
 *   qmx_multi_search_topk : brute force over `ids` or every point; `point_deleted` = optional BitSlice<u64, Lsb0> over POINTS
 *                           (bit set = deleted; points past n_deleted_bits count as deleted, as NotDeletedChecker); out [n_queries][top]. */
QMX_API int32_t qmx_multi_score_points(qmx_query *inner, const uint32_t *query_first, uint32_t n_queries, const uint64_t *point_offsets,
                                       uint32_t n_points, const uint32_t *ids, uint32_t n, float *scores);
QMX_API int32_t qmx_multi_search_topk(qmx_query *inner, const uint32_t *query_first, uint32_t n_queries, const uint64_t *point_offsets,
                                      uint32_t n_points, const uint64_t *point_deleted, uint64_t n_deleted_bits, uint32_t top,
                                      const uint32_t *ids, uint64_t n_ids, qmx_scored_point *out, uint32_t *out_counts);

/* `GraphLayers::search_with_vectors` (graph_layers.rs:564-596; taken by hnsw/read_view/search.rs:88-134 when the graph has inline storage -
 * `GraphLinksFormat::CompressedWithVectors` - and the search is quantized): the walk (`search_entry_with_vectors` :391-449,
 * `search_on_level_with_vectors` :336-389) is steered by the LINKS scorer - the quantized vectors stored next to each link, i.e. the rows of
 * the quantized segment `links` was made over - while every candidate the level-0 loop pops (the one that ends the loop included) is scored
 * by the BASE scorer - the full vector stored in front of the node's links, i.e. the rows of the original segment `base` was made over -
 * into a second `SearchContext(ef)`; its best `top` are the result: rescoring fused into the walk.  On the device the link and base vectors
 * are read from the two segments in HBM (the same bytes the file carries inline); the walk kernel lists the popped candidates and the pair
 * kernel scores them (exact bits of the base scorer).  max(top, ef) <= 4096; a search that pops more than 32 max(ef, top) + 256 candidates
 * => QMX_ERR_NOT_SUPPORTED.  counters->vectors_scored = link vectors + base vectors scored.  Ties: the reference's loop ends at the first popped
 * candidate whose score is BELOW the lower bound (:358); a candidate that `nearest` evicted before it was expanded and whose score EQUALS the bound is
 * expanded there - and here: every such candidate is kept (the evictions of the latest score, which are the only ones that can still tie with the bound,
 * in four registers and a per-slot stack of 4 096 entries; a search that needs more reports QMX_ERR_NOT_SUPPORTED, never a silent drop) and expanded in the
 * device's order among equal scores (lower id first) where the reference's is its BinaryHeap's: results can differ only inside runs of equal link
 * scores, like every other tie of the walk (tests/test_gpu_hnsw_with_vectors_ties.py pins the rule on BQ links). */
QMX_API int32_t qmx_hnsw_search_with_vectors(const qmx_hnsw *g, qmx_query *links, qmx_query *base, uint32_t top, uint32_t ef,
                                             qmx_scored_point *out, uint32_t *out_counts, const volatile uint8_t *is_stopped,
                                             qmx_counters *counters);

/* QUANTIZED multi-vectors (`QuantizedMultivectorStorage`, lib/segment/src/vector_storage/quantized/quantized_multivector_storage/mod.rs:76-393,
 * `MultivectorOffset{start, count}` :39-42): the two calls above accept an `inner` batch over an SQ, PQ or BQ segment of the quantized INNER
 * rows; the similarities are then the quantized scorer's (`quantized_storage.score(inner_query, vector)`), max / sum as
 * `score_point_max_similarity` (:339-363).
 *
 * The HNSW walk over multi-vector points (`GraphLayers::search` with `MultiMetricQueryScorer`, query_scorer/multi_metric_query_scorer.rs:18-127, or the
 * quantized `QuantizedMultiQueryScorer`, quantized/quantized_multi_query_scorer.rs): `g` is the graph over the n_points POINTS; every hop candidate
 * is scored by MaxSim of the multi-query against the point's inner rows, inside the walk kernel (dense, SQ and BQ inner rows; PQ:
 * QMX_ERR_NOT_SUPPORTED).  A multi-query's inner vectors must fit the LDS (16 + tokens x query-entry bytes <= 150 KiB).  Deleted flags are per
 * point, as in qmx_multi_search_topk; a filter set with qmx_query_set_filter on `inner` is read as a bitmap over POINTS.  out [n_queries][top],
 * counters->vectors_scored = points scored. */
QMX_API int32_t qmx_multi_hnsw_search(const qmx_hnsw *g, qmx_query *inner, const uint32_t *query_first, uint32_t n_queries,
                                      const uint64_t *point_offsets, uint32_t n_points, const uint64_t *point_deleted, uint64_t n_deleted_bits,
                                      uint32_t top, uint32_t ef, qmx_scored_point *out, uint32_t *out_counts, qmx_counters *counters);

/* `RawScorer::score_points` of custom scorers: scores[qi * n + i] = custom query qi against stored point ids[i]. */
QMX_API int32_t qmx_custom_score_points(qmx_query *examples, const qmx_custom_query *queries, uint32_t n_queries,
                                        const uint32_t *ids, uint32_t n, float *scores);
/* The brute-force search with custom scorers (`BatchFilteredSearcher` over them, point_scorer.rs:423-472): candidates =
 * `ids` or every live point; deleted flags and the batch's payload filter apply; out [n_queries][top]. */
QMX_API int32_t qmx_custom_search_topk(qmx_query *examples, const qmx_custom_query *queries, uint32_t n_queries, uint32_t top,
                                       const uint32_t *ids, uint64_t n_ids, qmx_scored_point *out, uint32_t *out_counts);

/* `GraphLayers::search` with a custom query as the points scorer: raw_scorer.rs:228-333 builds a CustomQueryScorer (dense storages), a
 * QuantizedCustomQueryScorer (SQ / PQ / BQ: quantized/quantized_custom_query_scorer.rs:13-113) or a TurboCustomQueryScorer
 * (query_scorer/turbo_custom_query_scorer.rs:17-113) for whatever storage `ex` is bound to - every example encoded as that storage's query -
 * and graph_layers.rs:108-149 walks with it.  Search qi = custom query queries[qi]; the walk scores a hop candidate against every example of
 * the query and combines (`score_by`).  Same results contract as qmx_hnsw_search; qmx_custom_set_coefficients first for feedback queries. */
QMX_API int32_t qmx_custom_hnsw_search(const qmx_hnsw *g, qmx_query *ex, const qmx_custom_query *queries, uint32_t n_queries, uint32_t top,
                                       uint32_t ef, qmx_scored_point *out, uint32_t *out_counts, const volatile uint8_t *is_stopped,
                                       qmx_counters *counters);

/* Custom queries whose examples are MULTI-VECTORS (`MultiCustomQueryScorer`, query_scorer/multi_custom_query_scorer.rs:19-130; over quantized
 * inner rows `QuantizedMultiCustomQueryScorer`, quantized/quantized_multi_custom_query_scorer.rs:19-96): similarity(example, point) =
 * score_max_similarity (query_scorer/mod.rs:70-97), then the query's score_by.  `inner` holds the inner vectors of all examples; example e = inner
 * query vectors [example_first[e], example_first[e + 1]); queries[i].first / n_a / n_b count EXAMPLES; points as in qmx_multi_score_points. */
QMX_API int32_t qmx_multi_custom_score_points(qmx_query *inner, const uint32_t *example_first, uint32_t n_examples,
                                              const qmx_custom_query *queries, uint32_t n_queries, const uint64_t *point_offsets,
                                              uint32_t n_points, const uint32_t *ids, uint32_t n, float *scores /* [n_queries][n] */);
QMX_API int32_t qmx_multi_custom_search_topk(qmx_query *inner, const uint32_t *example_first, uint32_t n_examples,
                                             const qmx_custom_query *queries, uint32_t n_queries, const uint64_t *point_offsets,
                                             uint32_t n_points, const uint64_t *point_deleted, uint64_t n_deleted_bits, uint32_t top,
                                             const uint32_t *ids, uint64_t n_ids, qmx_scored_point *out, uint32_t *out_counts);
/* ... and as the scorer of the HNSW walk over the multi-vector POINTS (`g` as for qmx_multi_hnsw_search): every hop candidate is scored by MaxSim against
 * every example of the query, then `score_by`.  Dense and SQ inner rows; a query's examples must fit the LDS (header + 16 bytes and the tokens per example
 * <= 150 KiB).  Deleted flags are per point; a filter set on `inner` is read as a bitmap over points. */
QMX_API int32_t qmx_multi_custom_hnsw_search(const qmx_hnsw *g, qmx_query *inner, const uint32_t *example_first, uint32_t n_examples,
                                             const qmx_custom_query *queries, uint32_t n_queries, const uint64_t *point_offsets, uint32_t n_points,
                                             const uint64_t *point_deleted, uint64_t n_deleted_bits, uint32_t top, uint32_t ef, qmx_scored_point *out,
                                             uint32_t *out_counts, qmx_counters *counters);

/* k-way merge of per-segment / per-GPU result lists = `BatchResultAggregator`
 * (lib/shard/src/search_result_aggregator.rs:50-121) with all point versions equal:
 * lists[(l * nq + qi) * k ..], idx already globalised by the caller.  Items are pushed in list
 * order; an id already seen in an earlier item is dropped whatever its score (`seen.insert`,
 * :28-37).  n_lists * k <= 16384.  Runs on `device_id`. */
QMX_API int32_t qmx_merge_topk(int32_t device_id, const qmx_scored_point *lists,
                               const uint32_t *list_counts, uint32_t n_lists, uint32_t nq,
                               uint32_t k, qmx_scored_point *out, uint32_t *out_counts);

/* Same merge, enqueued on `hip_stream` with every buffer in device memory (the multi-GPU path:
 * lists = the RCCL all-gather of per-GPU top-k).  `list_idx_base_dev[l]` (optional) is added to
 * every idx of list l: segment-local offsets -> (segment, offset) globalised ids. */
QMX_API int32_t qmx_merge_topk_async(int32_t device_id, void *hip_stream, const qmx_scored_point *lists_dev,
                                     const uint32_t *list_counts_dev, const uint32_t *list_idx_base_dev,
                                     uint32_t n_lists, uint32_t nq, uint32_t k, qmx_scored_point *out_dev,
                                     uint32_t *out_counts_dev);

/* The same merge over PACKED records: one record per list = one rank's whole answer to a batch in ONE buffer, so that the exchange step of the
 * segment-sharded search is a single all-gather (`ncclAllGather` of qmx_topk_record_bytes(nq, k) bytes per rank):
 *     record = [nq][k] qmx_scored_point, then [nq] uint32 counts, then 0 or 4 bytes of padding (records are multiples of 8 bytes).
 * A rank fills its record with qmx_search_topk_async(q, k, ids, n_ids, out = record, out_counts = record + nq * k * 8) - no packing kernel.
 * `records_dev` = n_lists records back to back (the all-gather's output); everything else as qmx_merge_topk_async.  The reference merges the
 * per-segment lists of a batch in one aggregator pass too (`BatchResultAggregator::update_batch_results`, search_result_aggregator.rs:91-106). */
QMX_API uint64_t qmx_topk_record_bytes(uint32_t nq, uint32_t k);
QMX_API int32_t qmx_merge_topk_packed_async(int32_t device_id, void *hip_stream, const void *records_dev, const uint32_t *list_idx_base_dev,
                                            uint32_t n_lists, uint32_t nq, uint32_t k, qmx_scored_point *out_dev, uint32_t *out_counts_dev);

/* ---- one batch against N segments (N devices), one call ---------------------------------------- */

/* `SegmentsSearcher::search` (lib/collection/src/collection_manager/segments_searcher.rs:250-285) runs one search task per segment and
 * hands the per-segment lists to `BatchResultAggregator` (lib/shard/src/search_result_aggregator.rs:50-121).  This is that fan-out and
 * merge for the ONE host process that owns all segments: queries[i] is the batch bound to segment i (qmx_query_create(segment_i, ...), the
 * same nq everywhere, the same queries: qmx_sharded_query_update); the segments may live on different devices (one 10 M-row segment per
 * MI355X, BASELINE configs[4]) or on one.  Every segment's brute-force top-k is enqueued on its own batch's stream - the devices work
 * concurrently -, its Q x top x 8 B list is copied to the device of queries[0] (hipMemcpyPeerAsync over xGMI when it lives elsewhere: the
 * one exchange step of the path), and the k-way merge (qmx_merge_topk's kernel, ids + id_bases[i]) runs there.
 *   id_bases : [n_segments] added to segment-local offsets (NULL = zeros).  With the segments created over consecutive row ranges of ONE block
 *              (desc.data = block + r_i * stride, desc.n = rows of the range) and id_bases[i] = r_i the result is the single-segment search of
 *              that block, list for list: the row-split (strong-scaling) form of SURVEY 8e.
 *   out / out_counts : [nq][top] / [nq], host or device (device: on the device of queries[0]).
 * n_segments * top <= 16384.  Errors as qmx_search_topk. */
QMX_API int32_t qmx_sharded_search_topk(qmx_query *const *queries, uint32_t n_segments, uint32_t top, const uint32_t *id_bases,
                                        qmx_scored_point *out, uint32_t *out_counts, const volatile uint8_t *is_stopped,
                                        qmx_counters *counters);
/* Same, enqueue only: out_dev / out_counts_dev on the device of queries[0]; complete with qmx_query_synchronize(queries[0]) (its stream is
 * ordered behind every segment's stream by the merge). */
QMX_API int32_t qmx_sharded_search_topk_async(qmx_query *const *queries, uint32_t n_segments, uint32_t top, const uint32_t *id_bases,
                                              qmx_scored_point *out_dev, uint32_t *out_counts_dev);
/* The local stage as an HNSW walk of each segment's own graph (graphs[i] built over segment i; hnsw/read_view/search.rs) instead of the scan. */
QMX_API int32_t qmx_sharded_hnsw_search(const qmx_hnsw *const *graphs, qmx_query *const *queries, uint32_t n_segments, uint32_t top,
                                        uint32_t ef, const uint32_t *id_bases, qmx_scored_point *out, uint32_t *out_counts,
                                        const volatile uint8_t *is_stopped, qmx_counters *counters);
/* qmx_query_update of every segment's batch with the same queries (host or device memory; device memory must be readable from every
 * segment's device). */
QMX_API int32_t qmx_sharded_query_update(qmx_query *const *queries, uint32_t n_segments, const float *batch);

/* ---- HNSW search on device -------------------------------------------------------------------- */

/* One built graph = `GraphLayers` (lib/segment/src/index/hnsw_index/graph_layers.rs:58-72): the
 * plain `GraphLinks` arrays exactly as `GraphLinksSerializer` lays them out
 * (graph_links/serializer.rs:52-176, read by graph_links/view.rs) plus `EntryPoints`
 * (entry_points.rs:10-16).  Links of point p on level 0 are
 * neighbors[offsets[p] .. offsets[p + 1]); on level l > 0 the slot is
 * level_offsets[l] + reindex[p].  All arrays host or device; they are copied to HBM. */
typedef struct qmx_hnsw_desc {
    uint32_t m;                       /* HnswM.m  (links per point on levels > 0)  */
    uint32_t m0;                      /* HnswM.m0 (links per point on level 0)     */
    uint32_t n_points;                /* == rows of the segment searched            */
    uint32_t n_levels;                /* levels_count = max level + 1               */
    const uint32_t *reindex;          /* [n_points]                                 */
    const uint64_t *level_offsets;    /* [n_levels + 1]; last = n_offsets - 1       */
    const uint64_t *offsets;          /* [n_offsets]                                */
    uint64_t n_offsets;
    const uint32_t *neighbors;        /* [n_neighbors]                              */
    uint64_t n_neighbors;
    const uint32_t *entry_point_ids;  /* EntryPoints::entry_points, in order        */
    const uint32_t *entry_point_levels;
    uint32_t n_entry_points;
    uint32_t n_extra_entry_points;    /* EntryPoints::extra_entry_points (iter_unsorted order), may be 0 */
    const uint32_t *extra_entry_point_ids;
    const uint32_t *extra_entry_point_levels;
    int32_t device_id;
    int32_t reserved;
} qmx_hnsw_desc;

/* Uploads the graph (replaces `GraphLayers::load` for the search side).  Immutable afterwards and
 * safe for concurrent searches from any number of threads (each with its own qmx_query). */
QMX_API int32_t qmx_hnsw_create(const qmx_hnsw_desc *desc, qmx_hnsw **out);
/* Same, from the bytes of a PLAIN `links.bin` / `graph links` file as written by
 * `serialize_graph_links(.., GraphLinksFormatParam::Plain, ..)` (graph_links/serializer.rs:52-209):
 * HeaderPlain {point_count, levels_count, total_neighbors_count, total_offset_count,
 * offsets_padding_bytes, [u8; 24]} (graph_links/header.rs:9-20, 64 bytes), level_offsets
 * [levels_count] u64, reindex [point_count] u32, neighbors u32, padding, offsets u64.
 * `desc` supplies m, m0, the entry points and the device; its array fields are ignored.
 * Compressed formats (header version 0xFFFF_FFFF_FFFF_FF01/02) => QMX_ERR_NOT_SUPPORTED here:
 * use qmx_hnsw_create_from_file, which reads all three formats. */
QMX_API int32_t qmx_hnsw_create_from_plain_file(const void *bytes, uint64_t n_bytes, const qmx_hnsw_desc *desc,
                                                qmx_hnsw **out);
QMX_API int32_t qmx_hnsw_destroy(qmx_hnsw *g);
/* The graph-links file in ANY of the reference's three formats (`GraphLinksFormat`,
 * graph_links/format.rs): Plain (above), Compressed (HeaderCompressed, header.rs:22-37, version
 * 0xFFFF_FFFF_FFFF_FF01: links bit-packed by `pack_links`, lib/common/common/src/bitpacking_links.rs:38-82,
 * offsets by `bitpacking_ordered::compress`, bitpacking_ordered.rs:69-105) and CompressedWithVectors
 * (HeaderCompressedWithVectors, header.rs:39-54, version ..FF02: per node [base vector on level 0]
 * [varint count][packed links][padding][link vectors][padding], serializer.rs:127-171).
 * Replaces `GraphLinksView::load` + `links()` (graph_links/view.rs:110-208, 244-275): the file is
 * decoded ONCE on the host into the plain arrays below (the order of a node's links is the
 * iterator's: the first level_m links ascending, as `pack_links` sorted them) and uploaded; the
 * inline vectors of the ..FF02 format are skipped (the rows are resident in HBM already).
 * Host-only, needs no device: every section is bounds-checked against n_bytes. */
typedef struct qmx_graph_links {
    uint32_t format;                  /* 0 Plain, 1 Compressed, 2 CompressedWithVectors           */
    uint32_t m, m0;                   /* HnswM of the header (compressed formats); 0 for Plain     */
    uint32_t n_points, n_levels;
    uint32_t reserved;
    uint64_t n_offsets, n_neighbors;
    const uint32_t *reindex;          /* [n_points]                                               */
    const uint64_t *level_offsets;    /* [n_levels + 1]; last = n_offsets - 1                     */
    const uint64_t *offsets;          /* [n_offsets], in links (not bytes)                        */
    const uint32_t *neighbors;        /* [n_neighbors]                                            */
    void *owner;                      /* library-owned storage behind the four arrays             */
} qmx_graph_links;
QMX_API int32_t qmx_graph_links_decode(const void *bytes, uint64_t n_bytes, qmx_graph_links *out);
QMX_API void qmx_graph_links_free(qmx_graph_links *links);
/* qmx_graph_links_decode + qmx_hnsw_create.  `desc` supplies the entry points and the device (and m, m0
 * for a Plain file); for the compressed formats m / m0 come from the header and a non-zero desc->m / m0
 * that disagrees is QMX_ERR_BAD_ARG. */
QMX_API int32_t qmx_hnsw_create_from_file(const void *bytes, uint64_t n_bytes, const qmx_hnsw_desc *desc, qmx_hnsw **out);


/* `GraphLayers::search(top, ef, SearchAlgorithm::Hnsw, FilteredScorer, None, is_stopped)`
 * (graph_layers.rs:530-562) for every query of the batch `q`, entirely on device: entry point
 * selection under the segment's deleted flags, greedy descent through the upper levels
 * (`search_entry`, :247-317), the ef-bounded beam on level 0 (`search_on_level`, :108-149) and
 * `into_iter_sorted().take(top)`.  `q` may belong to a dense, SQ or PQ segment (the quantized
 * scorer of `is_quantized_search`); rescoring with the original vectors is a separate
 * qmx_rescore call, as in hnsw/read_view/search.rs.  ef is raised to `top` (:549); max(top, ef) <= 4096
 * (up to 512 the beam lives in registers, beyond it in LDS).
 *   out : [nq][top], out_counts : [nq].  Results equal the reference's whenever the scores met
 *   on the walk are distinct.  AMONG EQUAL SCORES the default walk orders candidates by ascending id where the reference's order is that of
 *   its two binary heaps (search_context.rs:8-40): on integer-score storages (SQ, u8, BQ, 1-bit TurboQuant) the id lists can differ from the
 *   reference's inside runs of equal scores, and a walk that parts from the reference's at a tie may visit other points afterwards.  For the
 *   reference's own lists set the option "hnsw_reference_heap_order" (the two heaps kept on the device in std's sift order: the reference's
 *   ids, score bits and pop sequence; see qmx_hnsw_search_traced for what it costs and what is pinned).
 *   counters->vectors_scored = points scored over all searches (HardwareCounter cpu).  PQ walks through per-search LUTs also report the hop
 *   prefilter: counters->prefilter_candidates = level-0 candidates that met the 8-bit bound, counters->verified_rows = those scored exactly. */
QMX_API int32_t qmx_hnsw_search(const qmx_hnsw *g, qmx_query *q, uint32_t top, uint32_t ef,
                                qmx_scored_point *out, uint32_t *out_counts,
                                const volatile uint8_t *is_stopped, qmx_counters *counters);
/* The same with `SearchAlgorithm::Acorn` (graph_layers.rs:154-243, 550-559; chosen by hnsw/read_view/search.rs:42-90 when the
 * request enables it and the filter is selective enough): on level 0 a link that fails `check_vector` (deleted flags + the
 * payload filter set with qmx_query_set_filter) is not scored but explored - its own links are offered as 2-hop neighbours -
 * with the reference's two visited lists and per-node limits.  m0 <= 128. */
QMX_API int32_t qmx_hnsw_search_acorn(const qmx_hnsw *g, qmx_query *q, uint32_t top, uint32_t ef,
                                      qmx_scored_point *out, uint32_t *out_counts,
                                      const volatile uint8_t *is_stopped, qmx_counters *counters);
/* qmx_hnsw_search (host outputs) that also lists, per search, the candidates the level-0 loop of `search_on_level` pops from `candidates` and
 * expands (graph_layers.rs:120-147: every `candidate` that passes the `candidate.score < lower_bound` test), in order, with their scores:
 *   pops : [nq][pop_cap], pop_counts : [nq] (a count above pop_cap means the list is incomplete).
 * Test / verification aid: two walks of one graph agree exactly as long as their pop sequences agree, and where two sequences first differ the
 * two popped candidates show WHY (equal scores = the reference's order among ties; anything else = a defect).  See also the option
 * "hnsw_reference_heap_order" (qmx_set_option): the walk then keeps `nearest` and `candidates` as the reference's two binary heaps in std's
 * sift order (search_context.rs:8-40, fixed_length_priority_queue.rs:47-59) and returns the reference's lists among equal scores too - one lane
 * works the heaps, so it is slow: a verification mode for the plain walk (dense, SQ, PQ, BQ, TurboQuant scorers). */
QMX_API int32_t qmx_hnsw_search_traced(const qmx_hnsw *g, qmx_query *q, uint32_t top, uint32_t ef,
                                       qmx_scored_point *out, uint32_t *out_counts,
                                       qmx_scored_point *pops, uint32_t pop_cap, uint32_t *pop_counts);
/* qmx_hnsw_search, only enqueued on the query's stream; outputs in device memory.
 * `out_scored_dev` ([nq] points scored per search) may be NULL. */
QMX_API int32_t qmx_hnsw_search_async(const qmx_hnsw *g, qmx_query *q, uint32_t top, uint32_t ef,
                                      qmx_scored_point *out_dev, uint32_t *out_counts_dev,
                                      uint32_t *out_scored_dev);

/* ---- HNSW build on device ------------------------------------------------------------------------ */

/* `HnswConfig` subset used by `GraphLayersBuilder` (graph_layers_builder.rs:230-262). */
typedef struct qmx_hnsw_build_params {
    uint32_t m;                 /* links per point on levels > 0                                  */
    uint32_t m0;                /* links per point on level 0 (the reference uses 2 m); <= 128    */
    uint32_t ef_construct;      /* beam width of the insertion searches; <= 512                   */
    uint32_t entry_points_num;  /* extra entry points kept (highest levels), reference default 10 */
    uint64_t seed;              /* level draw: level(i) = round(-ln U(seed, i) / ln max(m, 2))    */
    uint32_t max_batch;         /* points inserted concurrently (0 = default 16384); a batch never exceeds
                                   1/32 of the points already linked                              */
    uint32_t reserved;
} qmx_hnsw_build_params;

/* Sizes of the plain arrays of a graph (for qmx_hnsw_export_plain). */
typedef struct qmx_hnsw_info {
    uint32_t m, m0, n_points, n_levels;
    uint64_t n_offsets, n_neighbors;
    uint32_t n_entry_points, n_extra_entry_points;
} qmx_hnsw_info;

/* Builds the HNSW graph of a segment on its GPU — dense f32 / f16 / u8 (u8: dot, euclid, manhattan), or SQ-int8 / BQ, where the
 * build scores through the quantized scorer as the reference does for a quantized segment (hnsw/build.rs:334-341:
 * `FilteredScorer::new_internal` over `QuantizedVectors`, stored rows as queries through `encode_internal_vector`) —: the work of
 * `GraphLayersBuilder::link_new_point` (graph_layers_builder.rs:417-474) for every non-deleted point, batch-parallel — the counterpart of
 * the reference's rayon / Vulkan builders (hnsw/build.rs:355, hnsw/gpu_build.rs).  Insertion order inside a batch
 * is concurrent, so the graph is not link-for-link the sequential CPU graph; it obeys the same invariants (<= m0 / m
 * links, no self links, no duplicates, links only to points of at least that level) and is checked by recall.  With
 * params->max_batch = 1 (one insertion per launch) it IS the sequential graph, link for link, wherever scores do not tie.
 * The result is searchable at once (qmx_hnsw_search) and exportable as plain GraphLinks arrays. */
QMX_API int32_t qmx_hnsw_build(const qmx_segment *seg, const qmx_hnsw_build_params *params, qmx_hnsw **out);
/* The build of a QUANTIZED segment whose storage cannot turn a stored row into a query — product quantization
 * (EncodedVectorsPQ::encode_internal_vector -> None, encoded_vectors_pq.rs:620-623) and TurboQuant (EncodedVectorsTQ, likewise).
 * FilteredScorer::new_internal (hnsw_index/point_scorer.rs:183-218) then scores the searches of an insertion with
 * quantized_vectors.raw_scorer(ORIGINAL vector of the point) = its LUT (PQ) / its precompute_query (TQ), while the heuristic and the back
 * links use the storage's score_internal (PQ: centroid <-> centroid distances, :574-618; TQ: score_symmetric, turboquant/quantization.rs:395-494).
 * `original` = the f32 segment the quantized segment was encoded from (same rows, same device; QMX_ERR_NOT_SUPPORTED without it); for
 * every other dtype it may be NULL and the call equals qmx_hnsw_build. */
QMX_API int32_t qmx_hnsw_build_quantized(const qmx_segment *quantized, const qmx_segment *original, const qmx_hnsw_build_params *params,
                                         qmx_hnsw **out);

/* Index build over INDEPENDENT segments, fanned out over their devices (north_star: "index build over independent segments shards across the 8 GPUs").
 * The reference builds one segment's graph per optimizer task, each task locking one GPU of the device pool for the duration of its build
 * (`GpuDevicesMaganer::lock_device`, index/hnsw_index/gpu/gpu_devices_manager.rs:120-143; the build itself: hnsw/build.rs:53 `build_hnsw_on_gpu`
 * under the rayon pool of :199,355) - builds of different segments share nothing.  Here: one host thread per segment inside the call, each running
 * qmx_hnsw_build_quantized(segments[i], originals ? originals[i] : NULL, params, &out_graphs[i]) on its segment's device (`originals` as there:
 * the f32 segments PQ / TurboQuant segments were encoded from, NULL entries - or a NULL array - where not needed).  Segments may share a
 * device (their builds then interleave on it) or sit one per device (the 8-GPU node: eight builds side by side, no exchange of any kind).
 * out_status[i] (may be NULL) receives each build's own status; the call returns the first failure in segment order, or QMX_OK.  Graphs of failed
 * builds are NULL; the others are complete and owned by the caller whatever the call returned.  Every graph equals what the single call builds
 * (tests/test_gpu_threads.py: N threads x N graphs on one device == the sequential builds, link for link). */
QMX_API int32_t qmx_sharded_hnsw_build(const qmx_segment *const *segments, const qmx_segment *const *originals, uint32_t n_segments,
                                       const qmx_hnsw_build_params *params, qmx_hnsw **out_graphs, int32_t *out_status);

/* The HNSW build over multi-vector points (hnsw/build.rs:334-341 through `FilteredScorer::new_internal`, point_scorer.rs:183-218: every score of
 * the build is `MultiMetricQueryScorer::score_internal`, multi_metric_query_scorer.rs, or for quantized inner rows `score_internal_max_similarity`,
 * quantized_multivector_storage/mod.rs:366-393): as qmx_hnsw_build, with point p = inner rows [point_offsets[p], point_offsets[p + 1]) of `inner`
 * (f32, f16, SQ or BQ rows; PQ and TurboQuant rows: qmx_multi_hnsw_build_quantized below) and the deleted flags per POINT (host arrays, as in qmx_multi_hnsw_search; the
 * inner segment's own flags are not read).  The graph has n_points points; search it with qmx_multi_hnsw_search. */
QMX_API int32_t qmx_multi_hnsw_build(const qmx_segment *inner, const uint64_t *point_offsets, uint32_t n_points, const uint64_t *point_deleted,
                                     uint64_t n_deleted_bits, const qmx_hnsw_build_params *params, qmx_hnsw **out);
/* The same over inner rows that cannot be turned back into queries - PQ and TurboQuant (`QuantizedMultivectorStorage::encode_internal_vector`,
 * quantized_multivector_storage/mod.rs:458-470, is None as soon as one inner row's is: encoded_vectors_pq.rs:624, encoded_vectors_tq.rs:453; TurboQuant
 * over Manhattan: QMX_ERR_NOT_SUPPORTED).  As qmx_hnsw_build_quantized does for single vectors, the searches of
 * an insertion then score through the query scorer of the point's ORIGINAL multi-vector (point_scorer.rs:183-218: MaxSim over the LUTs / precomputed queries
 * of its inner vectors) and only stored <-> stored pairs through score_internal_max_similarity: `original_inner` = the f32 segment the inner rows were encoded from
 * (same dim, same device, at least as many rows).  Inner rows that are their own queries (f32, f16, SQ, BQ): `original_inner` may be NULL and the call is
 * qmx_multi_hnsw_build. */
QMX_API int32_t qmx_multi_hnsw_build_quantized(const qmx_segment *inner, const qmx_segment *original_inner, const uint64_t *point_offsets, uint32_t n_points,
                                               const uint64_t *point_deleted, uint64_t n_deleted_bits, const qmx_hnsw_build_params *params, qmx_hnsw **out);
QMX_API int32_t qmx_hnsw_get_info(const qmx_hnsw *g, qmx_hnsw_info *out);
/* Copies the plain arrays of a graph built by qmx_hnsw_build into caller (host) buffers sized per qmx_hnsw_get_info:
 * reindex [n_points], level_offsets [n_levels + 1], offsets [n_offsets], neighbors [n_neighbors], entry points. */
QMX_API int32_t qmx_hnsw_export_plain(const qmx_hnsw *g, uint32_t *reindex, uint64_t *level_offsets, uint64_t *offsets,
                                      uint32_t *neighbors, uint32_t *entry_point_ids, uint32_t *entry_point_levels,
                                      uint32_t *extra_entry_point_ids, uint32_t *extra_entry_point_levels);

/* ---- quantizers -------------------------------------------------------------------------------- */

/* `EncodedVectorsU8::encode` row loop (encoded_vectors_u8.rs:236-296) for given params:
 * in [n][dim] f32 -> out [n][4 + actual_dim] reference rows. */
QMX_API int32_t qmx_sq_encode(int32_t device_id, uint32_t distance, const qmx_sq_params *params,
                              const float *in, uint64_t n, uint32_t dim, void *out_rows);
/* `VectorStats::build` (lib/quantization/src/vector_stats.rs:27-117): the per-dimension statistics Encoding::TwoBits / OneAndHalfBits of the binary
 * quantizer encode against (encoded_vectors_binary.rs:456: over ALL vectors of the storage, in order) - streaming Welford in f64 (mean, sample stddev)
 * and f32 min / max; vectors [n][dim] f32 (host or device); outputs [dim] f32 (host or device; min_out / max_out may be NULL).  One thread per dimension
 * runs the reference's sequential update: the oracle's bits.  The arrays are what qmx_bq_params.mean / .stddev take. */
QMX_API int32_t qmx_vector_stats(int32_t device_id, const float *vectors, uint64_t n, uint32_t dim, float *min_out, float *max_out, float *mean_out,
                                 float *stddev_out);
/* `EncodedVectorsBin::encode_vector` for Encoding::OneBit (encoded_vectors_binary.rs:535-568) with the u128 store type:
 * in [n][dim] f32 (already metric-preprocessed, as the storage's rows are) -> out [n][ceil(dim / 128) * 16] bytes. */
QMX_API int32_t qmx_bq_encode(int32_t device_id, const float *in, uint64_t n, uint32_t dim, void *out_rows);
/* The same for any encoding: rows of qmx_bq_row_bytes(dim, encoding) bytes (get_quantized_vector_size_from_params::<u128>, :829-840). */
QMX_API int32_t qmx_bq_encode_ex(int32_t device_id, const qmx_bq_params *params, const float *in, uint64_t n, uint32_t dim, void *out_rows);
QMX_API uint64_t qmx_bq_row_bytes(uint32_t dim, uint32_t encoding);
/* PQ codebook training = `kmeans` (lib/quantization/src/kmeans.rs:9-169) for every chunk of `find_centroids`
 * (encoded_vectors_pq.rs:342-407) on a GIVEN sample [n][dim] (the reference draws <= KMEANS_SAMPLE_SIZE = 10 000
 * vectors with a randomly keyed Permutor: unpinned, so the sample is an input): first-k init, update_indexes,
 * update_centroids with `threads` row ranges accumulated in f64 in row order and summed in range order (the
 * reference's per-rayon-thread CentroidsCounter; pass its `max_kmeans_threads`), stop per chunk when
 * sum(|old - new|) < accuracy (KMEANS_ACCURACY = 1e-5) or after max_iterations (KMEANS_MAX_ITERATIONS = 100).
 * An empty cluster keeps its centroid (the reference re-seeds it with a random vector: unpinned).
 * out_centroids [n_centroids][dim] (host or device) = `Metadata.centroids`; out_iterations [m] (host, may be NULL). */
QMX_API int32_t qmx_pq_train(int32_t device_id, const float *sample, uint64_t n, uint32_t dim, uint32_t chunk_size,
                             uint32_t n_centroids, uint32_t max_iterations, float accuracy, uint32_t threads,
                             float *out_centroids, uint32_t *out_iterations);
/* The `quantile = None` parameter fit of `EncodedVectorsU8::encode` (encoded_vectors_u8.rs:193, 516-527 ->
 * `find_min_max_from_iter`, quantile.rs:19-33; multiplier :210-226): global min / max of the data on device (NaN never
 * wins a comparison, as in the reference's fold), alpha = (max - min) / 127, offset = min.  Deterministic, unlike the
 * quantile estimate.  in [n][dim] f32 host or device; fills every field of `out`. */
QMX_API int32_t qmx_sq_fit_min_max(int32_t device_id, uint32_t distance, const float *in, uint64_t n, uint32_t dim,
                                   qmx_sq_params *out);
/* The `quantile = Some(q)` fit (encoded_vectors_u8.rs:194-205 -> `find_quantile_interval`, quantile.rs:35-84) on a GIVEN
 * sample: the reference draws <= SAMPLE_SIZE = 5 000 random vectors (`take_random_vectors`), which vectors is not
 * reproducible, so the sample is an input (as for qmx_pq_train); `count` = vectors in the storage.  Given the sample the
 * interval is an exact order statistic: sorted[cut + 1] .. sorted[len - cut - 1] of the flattened sample with
 * cut = max(1, min((len - 1) / 2, (n_sample as f32 * (1 - q) / 2) as usize)).  *found = 0 where the reference returns None
 * (count < 127, q >= 1, fewer than 4 values, fewer than 2 left): the caller then keeps the min / max fit, as the reference does.
 * sample [n_sample][dim] f32, host or device. */
QMX_API int32_t qmx_sq_fit_quantile(int32_t device_id, uint32_t distance, const float *sample, uint64_t n_sample, uint32_t dim,
                                    uint64_t count, float quantile, qmx_sq_params *out, int32_t *found);
/* `EncodedVectorsPQ::encode_vector` (encoded_vectors_pq.rs:301-329): L2 argmin per chunk,
 * first minimum wins.  in [n][dim] f32 -> out [n][m] u8. */
QMX_API int32_t qmx_pq_encode(int32_t device_id, const qmx_pq_params *params, const float *in,
                              uint64_t n, uint32_t dim, uint8_t *out_codes);

#ifdef __cplusplus
}
#endif
#endif /* QDRANT_AMD_H */

/*
 * qdrant_oracle.h — CPU restatement of qdrant v1.19.0's vector-scoring hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may link or call this.  The product library (libqdrant_amd.so) never does.
 *
 * Every function cites the reference file:line (paths relative to the qdrant tree) it follows.
 * x86 intrinsics are identical in C and Rust (`std::arch::x86_64` == <immintrin.h>), so the
 * AVX / SSE paths are restated instruction-for-instruction: same accumulators, same horizontal
 * sum order.  Built with -ffp-contract=off so that scalar `a*b + c` stays un-fused like rustc.
 *
 * Pinning: checked in tests/test_oracle_golden.py against every known-answer vector the
 * reference's own unit tests hold for this path and, for the SQ integer
 * leaves, against the reference's own C kernels compiled into oracle/_ref/.
 * Unpinned by the reference (stated per row in DESIGN.md 4): order among EQUAL scores in the bounded
 * heap (Rust std BinaryHeap, restated here from its published algorithm); accumulation ORDER of the f32 / u8 leaves (its literals
 * are small integers); f16 leaves; PQ, k-means, the HNSW builder / search and the TurboQuant sections incl. the TQ+ P-square fit
 * (the reference's tests for those are tolerance / property tests: mirrored, no literal to hold the bits).
 */
#ifndef QDRANT_ORACLE_H
#define QDRANT_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { QO_COSINE = 0, QO_EUCLID = 1, QO_DOT = 2, QO_MANHATTAN = 3 }; /* types.rs:313-322 */
enum { QO_F32 = 0, QO_F16 = 1, QO_U8 = 2 };
enum { QO_ISA_AUTO = 0, QO_ISA_AVX = 1, QO_ISA_SSE = 2, QO_ISA_SCALAR = 3 };

typedef struct { uint32_t idx; float score; } qo_scored_point; /* common/src/types.rs:12-17 */

/* ---- f32 metrics: spaces/simple.rs, simple_avx.rs, simple_sse.rs ---- */
float qo_dot_f32(const float *a, const float *b, size_t n, int isa);
float qo_euclid_f32(const float *a, const float *b, size_t n, int isa);
float qo_manhattan_f32(const float *a, const float *b, size_t n, int isa);
/* Metric::similarity with the reference's dispatch thresholds (simple.rs:15,22,129-155) */
float qo_similarity_f32(int distance, const float *q, const float *v, size_t n);
/* cosine_preprocess(_avx/_sse) ; out may alias in.  isa as above. */
void qo_cosine_preprocess_f32(const float *in, float *out, size_t n, int isa);
/* Metric::preprocess for `distance` (identity unless cosine) with reference dispatch */
void qo_preprocess_f32(int distance, const float *in, float *out, size_t n);
/* MetricPostProcessing::postprocess (simple.rs:74-78,118-122,162-166,208-212) */
float qo_postprocess(int distance, float score);

/* ---- f16: spaces/metric_f16 ---- */
float qo_dot_f16(const uint16_t *a, const uint16_t *b, size_t n, int isa);
float qo_euclid_f16(const uint16_t *a, const uint16_t *b, size_t n, int isa);
float qo_manhattan_f16(const uint16_t *a, const uint16_t *b, size_t n, int isa);
float qo_similarity_f16(int distance, const uint16_t *q, const uint16_t *v, size_t n);
void qo_f32_to_f16(const float *in, uint16_t *out, size_t n); /* f16::from_f32, RNE (primitive.rs:77-79) */
void qo_f16_to_f32(const uint16_t *in, float *out, size_t n);

/* ---- u8: spaces/metric_uint ---- */
float qo_dot_u8(const uint8_t *a, const uint8_t *b, size_t n, int isa);
float qo_cosine_u8(const uint8_t *a, const uint8_t *b, size_t n, int isa);
float qo_euclid_u8(const uint8_t *a, const uint8_t *b, size_t n, int isa);
float qo_manhattan_u8(const uint8_t *a, const uint8_t *b, size_t n, int isa);
float qo_similarity_u8(int distance, const uint8_t *q, const uint8_t *v, size_t n, int isa);
void qo_f32_to_u8(const float *in, uint8_t *out, size_t n); /* `x as u8` (primitive.rs:127-129) */

/* ---- bounded top-k: FixedLengthPriorityQueue over Rust BinaryHeap<Reverse<T>> ---- */
typedef struct qo_topk qo_topk;
qo_topk *qo_topk_new(size_t length);
void qo_topk_free(qo_topk *);
void qo_topk_push(qo_topk *, uint32_t idx, float score);            /* fixed_length_priority_queue.rs:47-59 */
size_t qo_topk_into_sorted(qo_topk *, qo_scored_point *out);        /* :63-65 ; consumes content */
int qo_topk_push_ex(qo_topk *, uint32_t idx, float score, qo_scored_point *removed); /* push's Option<T>: 0 None, 1 Some(evicted), 2 Some(value) */
int qo_topk_top(const qo_topk *, qo_scored_point *out);              /* top() :80-82, 0 if empty */
size_t qo_topk_len(const qo_topk *);
const qo_scored_point *qo_topk_data(const qo_topk *);                /* iter_unsorted() order */
int qo_ordered_float_cmp(float a, float b);                          /* OrderedFloat::cmp */

/* ---- brute force: BatchFilteredSearcher::peek_top_iter (point_scorer.rs:423-472) ---- */
typedef struct {
    int dtype;            /* QO_F32 / QO_F16 / QO_U8 */
    int distance;
    int u8_isa;           /* isa used for u8 similarity (QO_ISA_AUTO = AVX2 as on this host) */
    const void *rows;     /* [n][dim] elements */
    size_t n, dim;
    const uint64_t *point_deleted; size_t n_point_bits; /* NULL => all rows live */
    const uint64_t *vec_deleted;   size_t n_vec_bits;
} qo_storage;
/* queries: already preprocessed + cast to the element type, [nq][dim] elements.
 * ids == NULL: stream = iter_zeros(point_deleted); else that list.  out [nq][top]; counts [nq].
 * Returns 0, or 7 if *is_stopped became non-zero. */
int qo_peek_top_iter(const qo_storage *st, const void *queries, size_t nq, size_t top,
                     const uint32_t *ids, size_t n_ids, qo_scored_point *out, uint32_t *counts,
                     const volatile uint8_t *is_stopped);
/* same scan with `threads` pthreads over disjoint row ranges and a final merge (the reference's
 * segment-parallel model, segments_searcher.rs:250-285); for the CPU baseline only */
int qo_peek_top_parallel(const qo_storage *st, const void *queries, size_t nq, size_t top,
                         qo_scored_point *out, uint32_t *counts, int threads);
/* RawScorer::score_points for one query */
void qo_score_points(const qo_storage *st, const void *query, const uint32_t *ids, size_t n, float *out);

/* ---- SQ int8: lib/quantization/src/encoded_vectors_u8.rs ---- */
typedef struct {
    uint32_t dim, actual_dim;
    int distance;         /* QO_* ; Cosine is treated like Dot */
    int invert;
    float alpha, offset, multiplier;
} qo_sq;
/* alpha/offset from global min/max (find_alpha_offset_size_dim, :523-533) + multiplier (:205-221) */
void qo_sq_init(qo_sq *sq, int distance, int invert, uint32_t dim, const float *data, size_t n);
void qo_sq_init_params(qo_sq *sq, int distance, int invert, uint32_t dim, float alpha, float offset);
uint8_t qo_sq_encode_value(const qo_sq *sq, float v);                       /* :94-98 */
float qo_sq_get_shift(const qo_sq *sq);                                     /* :116-134 */
void qo_sq_encode_row(const qo_sq *sq, const float *v, uint8_t *out_row);   /* :236-296 ; row = 4 + actual_dim */
void qo_sq_encode_query(const qo_sq *sq, const float *q, uint8_t *codes, float *q_offset); /* :583-619 */
/* score_bytes (:777-810) -> score_point_avx/sse/simple ; isa selects the integer leaf.
 * use_ref != 0 calls the reference's own C kernels through fn pointers set by qo_sq_set_ref_kernels */
float qo_sq_score(const qo_sq *sq, const uint8_t *q_codes, float q_offset, const uint8_t *row, int isa);
float qo_sq_score_internal(const qo_sq *sq, const uint8_t *row_i, const uint8_t *row_j, int isa); /* :675-705 */
float qo_sq_dot_avx(const uint8_t *q, const uint8_t *v, uint32_t dim);      /* cpp/avx2.c:25-63 restated */
float qo_sq_l1_avx(const uint8_t *q, const uint8_t *v, uint32_t dim);       /* cpp/avx2.c:65-122 */
float qo_sq_dot_sse(const uint8_t *q, const uint8_t *v, uint32_t dim);      /* cpp/sse.c:28-52 */
float qo_sq_l1_sse(const uint8_t *q, const uint8_t *v, uint32_t dim);       /* cpp/sse.c:472-513 */
typedef float (*qo_sq_leaf_fn)(const uint8_t *, const uint8_t *, uint32_t);
void qo_sq_set_ref_kernels(qo_sq_leaf_fn dot_avx, qo_sq_leaf_fn l1_avx); /* from oracle/_ref */

/* ---- PQ: lib/quantization/src/encoded_vectors_pq.rs ---- */
typedef struct {
    uint32_t dim, chunk_size, m, n_centroids;
    int distance, invert;
    const float *centroids; /* [n_centroids][dim] */
} qo_pq;
void qo_pq_init(qo_pq *pq, int distance, int invert, uint32_t dim, uint32_t chunk_size,
                uint32_t n_centroids, const float *centroids);
void qo_pq_encode_vector(const qo_pq *pq, const float *v, uint8_t *codes);      /* :301-329 */
void qo_pq_encode_query(const qo_pq *pq, const float *q, float *lut);           /* :519-541 ; lut [m][n_centroids] */
float qo_pq_score(const qo_pq *pq, const float *lut, const uint8_t *codes, int isa); /* :409-443 sse, :478-493 simple */
float qo_pq_score_internal(const qo_pq *pq, const uint8_t *ci, const uint8_t *cj);   /* :574-618 */
/* deterministic Lloyd k-means for tests (kmeans.rs:9-169 shape; the reference re-seeds empty
 * clusters randomly, so centroids are an INPUT to parity, never compared) */
void qo_pq_train(uint32_t dim, uint32_t chunk_size, uint32_t n_centroids, const float *data,
                 size_t n, int iters, float *centroids_out);
float qo_custom_combine(int kind, uint32_t n_a, uint32_t n_b, const float *sims);   /* Query::score_by of the custom queries */
/* FeedbackQuery::score_by (feedback_query.rs:198-226): sims = [target, pos_0, neg_0, ...], coefs = [a, partial_computation_0, ...] */
float qo_custom_feedback(uint32_t n_pairs, const float *sims, const float *coefs);
void qo_pq_train_ex(uint32_t dim, uint32_t chunk_size, uint32_t n_centroids, const float *data, size_t n, uint32_t max_iters,
                    float accuracy, uint32_t threads, float *centroids_out, uint32_t *iters_done);   /* kmeans.rs:9-169 on a given sample */

/* find_quantile_interval (lib/quantization/src/quantile.rs:35-84) on a GIVEN sample [n_sample][dim] (the reference's choice of
 * vectors is random); count = vectors in the storage.  Returns 1 and the interval, or 0 where the reference returns None. */
int qo_sq_quantile_interval(const float *sample, size_t n_sample, uint32_t dim, size_t count, float quantile, float *min_out, float *max_out);

/* ---- BQ: EncodedVectorsBin<u128>, Encoding::OneBit, QueryEncoding::SameAsStorage (lib/quantization/src/encoded_vectors_binary.rs) ---- */
size_t qo_bq_row_bytes(uint32_t dim);                                            /* :829-840 with u128::get_storage_size :412-419 */
void qo_bq_encode_row(uint32_t dim, const float *v, uint8_t *out);               /* encode_one_bit_vector :558-568 */
uint32_t qo_bq_xor_popcnt(const uint8_t *q, const uint8_t *v, uint32_t n_u128);  /* cpp/sse.c:54-75 */
/* calculate_metric :766-810 with query_bits_count == 1; distance as QO_* codes; invert = VectorParameters.invert */
float qo_bq_score(int distance, int invert, uint32_t dim, const uint8_t *q, const uint8_t *v);
/* Encoding::{OneBit = 0, TwoBits = 1, OneAndHalfBits = 2} (:62-78): row size (:829-840), encode_vector (:535-672; mean / stddev = the
 * VectorStats of the storage, NULL = none), calculate_metric over the longer rows with the ORIGINAL dim (:766-810) */
size_t qo_bq_row_bytes_ex(uint32_t dim, int encoding);
void qo_bq_encode_row_ex(uint32_t dim, int encoding, const float *mean, const float *stddev, const float *v, uint8_t *out);
float qo_bq_score_ex(int distance, int invert, uint32_t dim, int encoding, const uint8_t *q, const uint8_t *v);
/* QueryEncoding::Scalar4bits / Scalar8bits (encoded_vectors_binary.rs:692-756 encode, :403-409 xor_popcnt_scalar, :783-810 metric) */
size_t qo_bq_encode_scalar_query(uint32_t dim, int encoding, uint32_t bits, const float *query, uint8_t *out);
uint64_t qo_bq_xor_popcnt_scalar(const uint8_t *vector, const uint8_t *query, uint32_t n_u128, uint32_t bits);
float qo_bq_score_scalar(int distance, int invert, uint32_t dim, int encoding, uint32_t bits, const uint8_t *scalar_query, const uint8_t *v);

/* ---- TurboQuant: lib/quantization/src/turboquant/ behind EncodedVectorsTQ (oracle/qdrant_oracle_tq.c; TQMode::Normal) ----
 * bits: TQBits in the reference's order {Bits4 = 0, Bits2 = 1, Bits1_5 = 2, Bits1 = 3}; distance: QO_DOT | QO_COSINE | QO_EUCLID | QO_MANHATTAN (DistanceType::L1: the dequantise-and-rotate-back fallback). */
typedef struct qo_tq qo_tq;
typedef struct qo_tq_query qo_tq_query;
uint32_t qo_tq_padded_dim_for(uint32_t dim, int bits);                                 /* encoding.rs:194-201 */
void qo_tq_permutation_map(uint64_t seed, uint32_t count, uint32_t *map);             /* permutation.rs:108-115 on the identity */
uint32_t qo_tq_chunk_sizes(uint32_t dim, uint32_t *out);                               /* rotation.rs:222-233 */
void qo_tq_wht(double *x, uint32_t n);                                                 /* rotation.rs:158-176 */
qo_tq *qo_tq_new(uint32_t dim, int bits, int distance, int rotation_unpadded);         /* TurboQuantizer::new (quantization.rs:127-158) */
qo_tq *qo_tq_new_plus(uint32_t dim, int bits, int distance, int rotation_unpadded, const float *shift, const float *scale);   /* TQMode::Plus with given ErrorCorrection */
float qo_tq_query_ec_correction(const qo_tq_query *e);
void qo_tq_free(qo_tq *t);
uint32_t qo_tq_padded_dim(const qo_tq *t);
uint32_t qo_tq_quantized_size(const qo_tq *t);                                         /* encoding.rs:172-190 */
void qo_tq_rotate(const qo_tq *t, double *x);                                          /* HadamardRotation::apply on x[..rotation dim] */
void qo_tq_quantize(const qo_tq *t, const float *vec, uint8_t *out);                   /* TurboQuantizer::quantize */
qo_tq_query *qo_tq_precompute_query(const qo_tq *t, const float *query);               /* precompute_query */
void qo_tq_query_free(qo_tq_query *e);
void qo_tq_query_export(const qo_tq *t, const qo_tq_query *e, int32_t *q_out, float *postprocess_scale, float *l2_norm, int64_t *sum_q);
/* TQ+ fit (TQMode::Plus first pass, encoded_vectors_tq.rs:156-234): per-coordinate P-square quantile estimators over the sampled vectors */
double qo_p2_quantile(double q, const double *values, uint64_t n, double *grid);   /* P2Quantile<7>: new, push*, estimate */
void qo_tq_plus_quantiles(int bits, double *min_q, double *max_q, float *c_outer);
void qo_tq_preprocess(const qo_tq *t, const float *vec, double *buf);                  /* preprocess_into */
void qo_tq_plus_fit(const qo_tq *t, const float *sample, uint32_t n_sample, float *shift, float *scale);
float qo_tq_score_precomputed(const qo_tq *t, const qo_tq_query *e, const uint8_t *vec);   /* score_precomputed (before `invert`) */
float qo_tq_score_symmetric(const qo_tq *t, const uint8_t *v1, const uint8_t *v2);         /* score_symmetric (before `invert`) */
void qo_tq_dequantize(const qo_tq *t, const uint8_t *vec, double *out);                   /* dequantize::<f64> (quantization.rs:321-376), rotated space */
void qo_tq_rotate_inverse(const qo_tq *t, double *x);                                     /* HadamardRotation::apply_inverse on x[..rotation dim] */

/* ---- cross-segment merge: BatchResultAggregator (lib/shard/src/search_result_aggregator.rs:50-121) ----
 * lists[(l * nq + qi) * k ..] with counts[l * nq + qi] valid entries; idx_base[l] (optional) is added to
 * every idx of list l (segment-local offset -> global id).  Points are pushed list by list, each list in
 * its own (descending) order; an id already seen for that query is skipped (:33-36).  All versions equal. */
void qo_merge_topk(const qo_scored_point *lists, const uint32_t *counts, const uint32_t *idx_base,
                   uint32_t n_lists, uint32_t nq, uint32_t k, qo_scored_point *out, uint32_t *out_counts);

/* VectorStats::build (lib/quantization/src/vector_stats.rs:27-117): streaming Welford (f64) mean / sample stddev and f32 min / max of every dimension over
 * ALL vectors in order - the statistics Encoding::TwoBits / OneAndHalfBits of the binary quantizer encode against (encoded_vectors_binary.rs:456). */
void qo_vector_stats(const float *rows, uint64_t n, uint32_t dim, float *min, float *max, float *mean, float *stddev);

/* ---- scorer = FilteredScorer{RawScorer, NotDeletedChecker} (hnsw_index/point_scorer.rs:53-63) ---- */
typedef struct qo_scorer {
    int kind;                 /* 0 dense (Metric over st->rows), 1 SQ (EncodedVectorsU8), 2 PQ (EncodedVectorsPQ), 3 BQ (EncodedVectorsBin<u128>),
                                 4 multi-vector MaxSim over an inner scorer of one of the kinds above (fields mv_*), 5 TurboQuant (fields tq_*),
                                 6 custom query (Recommend / Discover / Context / Feedback) over example scorers of any other kind (fields cq_*) */
    const qo_storage *st;     /* dense rows for kind 0; the deleted flags and n for every kind */
    const void *query;        /* kind 0: preprocessed + cast query, [dim] elements */
    const qo_sq *sq; const uint8_t *sq_rows; const uint8_t *sq_query; float sq_query_offset;
    const qo_pq *pq; const uint8_t *pq_codes; const float *pq_lut;
    int isa;                  /* leaf used for SQ / PQ scoring */
    const uint8_t *bq_rows; const uint8_t *bq_query; uint32_t bq_dim; int bq_distance; int bq_invert;   /* kind 3 */
    /* kind 4: `st` = the POINT-level storage (n = points, deleted flags of points; rows unused).  Point p = inner rows
     * [mv_offsets[p], mv_offsets[p + 1]) of the inner storage; the query = mv_n_tokens inner scorers (one per inner query vector, each a
     * complete scorer of the inner kind over the inner rows).  score_point = score_max_similarity (query_scorer/mod.rs:70-97, dense) =
     * QuantizedMultivectorStorage::score_point_max_similarity (quantized_multivector_storage/mod.rs:339-363); score_internal =
     * score_internal_max_similarity (:366-393) / MultiMetricQueryScorer::score_internal through mv_tokens[0] as the inner template. */
    const struct qo_scorer *mv_tokens; uint32_t mv_n_tokens; const uint64_t *mv_offsets;
    /* kind 5: TurboQuant (EncodedVectorsTQ): rows of qo_tq_quantized_size bytes; the query = TurboQuantizer::precompute_query of the
     * preprocessed query; score_internal = score_symmetric; `invert` applied on top (encoded_vectors_tq.rs) */
    const struct qo_tq *tq; const uint8_t *tq_rows; const struct qo_tq_query *tq_query; int tq_invert;
    /* kind 6: CustomQueryScorer (query_scorer/custom_query_scorer.rs:44-121), QuantizedCustomQueryScorer (quantized/quantized_custom_query_scorer.rs:13-113),
     * TurboCustomQueryScorer (query_scorer/turbo_custom_query_scorer.rs:17-113), MultiCustomQueryScorer (query_scorer/multi_custom_query_scorer.rs:19-130; examples
     * of kind 4): score_point = query.score_by(|example| example_scorer.score_point(id)); the examples in the query's flat_iter() order, each a complete
     * scorer (the example transformed / encoded as that storage's query); cq_kind as qo_custom_combine (4 = feedback with cq_coefs).  `st` = the flags. */
    const struct qo_scorer *cq_examples; uint32_t cq_kind, cq_n_a, cq_n_b; const float *cq_coefs;
} qo_scorer;
float qo_scorer_score_point(const qo_scorer *s, uint32_t id);              /* RawScorer::score_point */
float qo_scorer_score_internal(const qo_scorer *s, uint32_t a, uint32_t b); /* RawScorer::score_internal */
int qo_scorer_check_vector(const qo_scorer *s, uint32_t id);               /* raw_scorer.rs:596-603 */

/* ---- HNSW: hnsw_index/{graph_layers.rs, graph_layers_builder.rs, links_container.rs, entry_points.rs} ---- */
typedef struct qo_hnsw qo_hnsw;
/* GraphLayersBuilder::new + set_levels for every point (level = round(-ln(U) / ln(max(m,2))),
 * graph_layers_builder.rs:320,388-396, U from our own splitmix64 stream: the reference draws from
 * rand's thread rng) + link_new_point(0..n) single-threaded with the internal scorer of each point
 * (FilteredScorer::new_internal: query = stored row, dense storages only). */
qo_hnsw *qo_hnsw_build(const qo_storage *st, uint32_t m, uint32_t m0, uint32_t ef_construct,
                       uint32_t entry_points_num, int use_heuristic, uint64_t seed);
/* same algorithm, `threads` workers inserting concurrently after the first 256 points
 * (hnsw/build.rs:285-356): like the reference's rayon build the result depends on thread timing. */
qo_hnsw *qo_hnsw_build_parallel(const qo_storage *st, uint32_t m, uint32_t m0, uint32_t ef_construct,
                                uint32_t entry_points_num, int use_heuristic, uint64_t seed, int threads);
/* the build of a QUANTIZED segment (hnsw/build.rs:334-341, point_scorer.rs:183-218): tmpl = the quantized storage as a scorer
 * template over tmpl->st (original dense storage); PQ scores each insertion's searches with the LUT of the original vector */
qo_hnsw *qo_hnsw_build_with(const qo_scorer *tmpl, uint32_t m, uint32_t m0, uint32_t ef_construct, uint32_t entry_points_num,
                            int use_heuristic, uint64_t seed);
void qo_hnsw_free(qo_hnsw *g);
uint32_t qo_hnsw_point_level(const qo_hnsw *g, uint32_t id);
uint32_t qo_hnsw_max_level(const qo_hnsw *g);
/* links of (point, level) into out (capacity m0); returns the count */
uint32_t qo_hnsw_links(const qo_hnsw *g, uint32_t id, uint32_t level, uint32_t *out);
/* primary entry points (EntryPoints.entry_points): ids / levels, returns count (call with NULLs to size) */
uint32_t qo_hnsw_entry_points(const qo_hnsw *g, uint32_t *ids, uint32_t *levels, uint32_t cap);
uint32_t qo_hnsw_extra_entry_points(const qo_hnsw *g, uint32_t *ids, uint32_t *levels, uint32_t cap);
qo_hnsw *qo_hnsw_import_plain(uint32_t n, uint32_t m, uint32_t m0, uint32_t n_levels, const uint32_t *reindex,
                              const uint64_t *level_offsets, const uint64_t *offsets, const uint32_t *neighbors,
                              const uint32_t *ep_ids, const uint32_t *ep_levels, uint32_t n_ep,
                              const uint32_t *xp_ids, const uint32_t *xp_levels, uint32_t n_xp);   /* GraphLayers::load, plain links */
/* the plain GraphLinks view (graph_links/view.rs:42-60,211-218 ; serializer.rs:52-87):
 *   reindex[n]            point -> rank in descending-level order
 *   level_offsets[L + 1]  index into `offsets` where each level starts (level 0 at 0, n entries), last = total
 *   offsets[total + 1]    start of each (level, slot) neighbour run in `neighbors`
 *   neighbors[]           u32 ids
 * Call once with NULL arrays to get sizes. */
void qo_hnsw_export_plain(const qo_hnsw *g, uint32_t *n_levels, uint64_t *n_offsets, uint64_t *n_neighbors,
                          uint32_t *reindex, uint64_t *level_offsets, uint64_t *offsets, uint32_t *neighbors);
/* GraphLayers::search (graph_layers.rs:530-562) with SearchAlgorithm::Hnsw: entry point, greedy
 * search_entry down to level 0, search_on_level(ef = max(ef, top)), sorted take(top).
 * Returns the number of results; *n_scored (optional) counts score_point evaluations. */
uint32_t qo_hnsw_search(const qo_hnsw *g, const qo_scorer *scorer, uint32_t top, uint32_t ef,
                        qo_scored_point *out, uint64_t *n_scored);
/* the same with SearchAlgorithm: 0 = Hnsw, 1 = Acorn (search_on_level_acorn, graph_layers.rs:154-243) */
uint32_t qo_hnsw_search_algo(const qo_hnsw *g, const qo_scorer *scorer, uint32_t top, uint32_t ef, int algorithm, qo_scored_point *out,
                             uint64_t *n_scored);
/* the same (SearchAlgorithm::Hnsw) + the candidates the level-0 loop pops and expands, in order; *n_pops may exceed pop_cap */
uint32_t qo_hnsw_search_traced(const qo_hnsw *g, const qo_scorer *scorer, uint32_t top, uint32_t ef, qo_scored_point *out, uint64_t *n_scored,
                               qo_scored_point *pops, uint32_t pop_cap, uint32_t *n_pops);
/* GraphLayers::search_with_vectors (graph_layers.rs:564-596) for graphs with inline storage: the walk is steered by `links_scorer` (the
 * quantized link vectors), every popped candidate is scored by `base_scorer` (its full base vector) and the best `top` of those are returned. */
uint32_t qo_hnsw_search_with_vectors(const qo_hnsw *g, const qo_scorer *links_scorer, const qo_scorer *base_scorer, uint32_t top, uint32_t ef,
                                     qo_scored_point *out, uint64_t *n_scored, uint64_t *n_base_scored);
/* LinksContainer::fill_from_sorted_with_heuristic (links_container.rs:47-71) and ::connect (:74-103) on
 * an explicit pairwise score table score[a * n + b]; for the reference's literal test (:312-391). */
uint32_t qo_links_heuristic(const qo_scored_point *sorted_candidates, uint32_t n_cand, uint32_t level_m,
                            const float *score_table, uint32_t n, uint32_t *out_links);
uint32_t qo_links_connect(uint32_t *links, uint32_t len, uint32_t new_point, uint32_t target, uint32_t level_m,
                          const float *score_table, uint32_t n);
/* LinksContainer::connect_with_heuristic (:139-222; restated as its own reference implementation connect_with_heuristic_simple :107-132,
 * which the reference's test_connect_new_point_with_heuristic holds equal) over a score table */
uint32_t qo_links_connect_heuristic(uint32_t *links, uint32_t len, uint32_t new_point, uint32_t target, uint32_t lm,
                                    const float *score_table, uint32_t n);

/* score_max_similarity (query_scorer/mod.rs:70-97) over a similarity table sims[a * stride + b] */
float qo_max_similarity(const float *sims, uint32_t n_a, uint32_t n_b, uint64_t stride);

/* ---- compressed graph-links files (qdrant_oracle_links.c; bitpacking.rs, bitpacking_links.rs, bitpacking_ordered.rs,
 *      graph_links/serializer.rs + header.rs) ---- */
uint64_t qo_bitpack_write(const uint64_t *values, const uint8_t *bits, uint32_t n, uint8_t *out, uint64_t cap);
void qo_bitpack_read(const uint8_t *data, uint64_t len, const uint8_t *bits, uint32_t n, uint64_t *values);
uint64_t qo_pack_links(uint32_t *raw_links, uint32_t n, uint8_t bits_per_unsorted, uint32_t sorted_count, uint8_t *out, uint64_t cap);
uint32_t qo_iterate_packed_links(const uint8_t *links, uint64_t len, uint8_t bits_per_unsorted, uint32_t sorted_count, uint32_t *out,
                                 uint32_t cap);
uint64_t qo_packed_links_size(const uint8_t *data, uint64_t len, uint8_t bits_per_unsorted, uint32_t sorted_count, uint32_t total_count);
uint64_t qo_ordered_compress(const uint64_t *values, uint64_t n, uint8_t *out, uint64_t cap, uint8_t *params3);
uint64_t qo_ordered_compress_with(const uint64_t *values, uint64_t n, uint8_t base_bits, uint8_t delta_bits, uint8_t chunk_len_log2,
                                  uint8_t *out, uint64_t cap);
uint64_t qo_ordered_get(const uint8_t *data, uint64_t length, uint8_t base_bits, uint8_t delta_bits, uint8_t chunk_len_log2, uint64_t index);
uint64_t qo_links_serialize_compressed(uint32_t m, uint32_t m0, uint32_t n_points, uint32_t n_levels, const uint32_t *reindex,
                                       const uint64_t *level_offsets, const uint64_t *offsets, const uint32_t *neighbors,
                                       int with_vectors, uint64_t base_size, uint8_t base_align, const uint8_t *base_vectors,
                                       uint64_t link_size, uint8_t link_align, const uint8_t *link_vectors, uint8_t *out, uint64_t cap);

/* ---- synthetic data shared bit-for-bit with the device generator ---- */
float qo_synth_value(uint64_t seed, uint64_t row, uint32_t col, uint32_t dim);
void qo_synth_fill_f32(uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim, float *out);
void qo_synth_fill_latent_f32(uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim, uint32_t K, float noise, float *out);

#ifdef __cplusplus
}
#endif
#endif

// api_custom.hip — the C-ABI of include/qdrant_amd.h, custom queries and multi-vector points.
// (One of the api_*.hip translation units; what they share: api_internal.hpp.)
#include "api_internal.hpp"

extern "C" {

// ---------------------------------------------------------------------------------------------
// custom queries (custom_query.hip)
// ---------------------------------------------------------------------------------------------
// `n_examples`: how many examples the batch holds (for multi-vector examples: the number of example multi-vectors, not of inner vectors)
static int32_t custom_validate(const qmx_query *ex, const qmx_custom_query *queries, uint32_t n_queries, uint32_t n_examples, uint32_t *max_examples) {
    QMX_REQUIRE(!is_device_ptr(queries), QMX_ERR_BAD_ARG, "the custom query descriptors are a host array (they are validated here)");
    if (max_examples) *max_examples = 0;
    for (uint32_t i = 0; i < n_queries; ++i) {
        const qmx_custom_query &c = queries[i];
        QMX_REQUIRE(c.kind <= QMX_CUSTOM_FEEDBACK, QMX_ERR_BAD_ARG, "bad custom query kind %u", c.kind);
        QMX_REQUIRE(c.kind != QMX_CUSTOM_FEEDBACK || c.n_a == 1, QMX_ERR_BAD_ARG, "a feedback query has exactly one target");
        QMX_REQUIRE(c.kind != QMX_CUSTOM_FEEDBACK || (uint64_t)c.coef_first + 1 + c.n_b <= ex->n_cq_coefs, QMX_ERR_OUT_OF_BOUNDS,
                    "feedback query %u reaches past the %u coefficients set with qmx_custom_set_coefficients", i, ex->n_cq_coefs);
        const uint64_t ne = c.kind <= QMX_CUSTOM_RECO_SUM_SCORES ? (uint64_t)c.n_a + c.n_b : (uint64_t)c.n_a + 2ull * c.n_b;
        QMX_REQUIRE(c.kind != QMX_CUSTOM_DISCOVER || c.n_a == 1, QMX_ERR_BAD_ARG, "a discover query has exactly one target");
        QMX_REQUIRE(c.kind != QMX_CUSTOM_CONTEXT || c.n_a == 0, QMX_ERR_BAD_ARG, "a context query has pairs only");
        QMX_REQUIRE((uint64_t)c.first + ne <= n_examples, QMX_ERR_OUT_OF_BOUNDS, "custom query %u reaches past the %u examples of the batch", i, n_examples);
        if (max_examples) *max_examples = std::max<uint32_t>(*max_examples, (uint32_t)ne);
    }
    return QMX_OK;
}

static int32_t custom_prepare(qmx_query *ex, const qmx_custom_query *queries, uint32_t n_queries, const uint32_t *d_ids, uint64_t n) {
    QMX_TRY(custom_validate(ex, queries, n_queries, ex->nq, nullptr));
    QMX_REQUIRE((uint64_t)ex->nq * n * 4 <= (48ull << 30), QMX_ERR_NOT_SUPPORTED, "example similarity matrix of %llu x %u floats is too large",
                (unsigned long long)n, ex->nq);
    QMX_TRY(ex->cq_sims.reserve((size_t)ex->nq * n * 4));
    QMX_TRY(ex->cq_scores.reserve((size_t)n_queries * n * 4));
    QMX_TRY(ex->cq_desc.reserve((size_t)n_queries * sizeof(qmx_custom_query)));
    QMX_HIP(hipMemcpyAsync(ex->cq_desc.p, queries, (size_t)n_queries * sizeof(qmx_custom_query), hipMemcpyDefault, ex->stream));
    QMX_TRY(score_ids_device(ex, d_ids, n, (float *)ex->cq_sims.p, nullptr));     // similarity(example, point), every example x candidate
    return launch_custom_combine(ex->stream, (const qmx_custom_query *)ex->cq_desc.p, n_queries, (const float *)ex->cq_sims.p, n, (const float *)ex->cq_coefs.p,
                                 (float *)ex->cq_scores.p);
}

int32_t qmx_custom_set_coefficients(qmx_query *ex, const float *coefs, uint32_t n) {
    QMX_REQUIRE(ex && (n == 0 || coefs), QMX_ERR_BAD_ARG, "NULL argument");
    QMX_HIP(hipSetDevice(ex->device));
    ex->n_cq_coefs = 0;
    if (n == 0) return QMX_OK;
    QMX_TRY(ex->cq_coefs.reserve((size_t)n * sizeof(float)));
    QMX_HIP(hipMemcpyAsync(ex->cq_coefs.p, coefs, (size_t)n * sizeof(float), hipMemcpyDefault, ex->stream));
    QMX_HIP(hipStreamSynchronize(ex->stream));   // the caller's buffer may go away
    ex->n_cq_coefs = n;
    return QMX_OK;
}

int32_t qmx_custom_score_points(qmx_query *ex, const qmx_custom_query *queries, uint32_t n_queries, const uint32_t *ids, uint32_t n, float *scores) {
    QMX_REQUIRE(ex && (n_queries == 0 || queries) && (n == 0 || (ids && scores)), QMX_ERR_BAD_ARG, "NULL argument");
    QMX_HIP(hipSetDevice(ex->device));
    if (n == 0 || n_queries == 0) return QMX_OK;
    const void *d_ids = nullptr;
    QMX_TRY(stage_in(ex, ex->ids, ids, (size_t)n * 4, &d_ids));
    QMX_TRY(custom_prepare(ex, queries, n_queries, (const uint32_t *)d_ids, n));
    QMX_TRY(copy_out(ex->stream, scores, ex->cq_scores.p, (size_t)n_queries * n * 4));
    return check_err_flag(ex);
}

int32_t qmx_custom_search_topk(qmx_query *ex, const qmx_custom_query *queries, uint32_t n_queries, uint32_t top, const uint32_t *ids, uint64_t n_ids,
                               qmx_scored_point *out, uint32_t *out_counts) {
    QMX_REQUIRE(ex && (n_queries == 0 || queries) && out && out_counts, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(top >= 1 && top <= MAX_TOP, QMX_ERR_NOT_SUPPORTED, "top %u not in 1..%u", top, MAX_TOP);
    QMX_HIP(hipSetDevice(ex->device));
    if (n_queries == 0) return QMX_OK;
    const void *d_ids = nullptr;
    uint64_t n = ex->seg->scan_rows();
    if (ids) {
        n = n_ids;
        if (n_ids) QMX_TRY(stage_in(ex, ex->ids, ids, (size_t)n_ids * 4, &d_ids));
    }
    const bool out_dev = is_device_ptr(out), cnt_dev = is_device_ptr(out_counts);
    qmx_scored_point *d_out = out;
    uint32_t *d_oc = out_counts;
    if (!out_dev) { QMX_TRY(ex->out.reserve((size_t)n_queries * top * sizeof(qmx_scored_point))); d_out = (qmx_scored_point *)ex->out.p; }
    if (!cnt_dev) { QMX_TRY(ex->counts.reserve((size_t)n_queries * 4)); d_oc = (uint32_t *)ex->counts.p; }
    if (n == 0) {
        QMX_HIP(hipMemsetAsync(d_oc, 0, (size_t)n_queries * 4, ex->stream));
    } else {
        QMX_TRY(custom_prepare(ex, queries, n_queries, (const uint32_t *)d_ids, n));
        DeletedView del = ex->seg->deleted_view();
        if (ex->has_filter) { del.allowed = (const uint64_t *)ex->filter.p; del.n_allowed_bits = ex->n_filter_bits; }
        QMX_TRY(launch_custom_topk(ex->stream, (const float *)ex->cq_scores.p, n, (const uint32_t *)d_ids, del, n_queries, top, d_out, d_oc));
    }
    if (!out_dev) QMX_TRY(copy_out(ex->stream, out, d_out, (size_t)n_queries * top * sizeof(qmx_scored_point)));
    if (!cnt_dev) QMX_TRY(copy_out(ex->stream, out_counts, d_oc, (size_t)n_queries * 4));
    return check_err_flag(ex);
}

// GraphLayers::search with a custom query as the points scorer (graph_layers.rs:108-149 walks with whatever scorer raw_scorer.rs:228-333 built)
int32_t qmx_custom_hnsw_search(const qmx_hnsw *g, qmx_query *ex, const qmx_custom_query *queries, uint32_t n_queries, uint32_t top, uint32_t ef,
                               qmx_scored_point *out, uint32_t *out_counts, const volatile uint8_t *is_stopped, qmx_counters *counters) {
    QMX_REQUIRE(g && ex && (n_queries == 0 || queries) && out && out_counts, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_TRY(hnsw_check(g, ex, top, ef));
    QMX_HIP(hipSetDevice(ex->device));
    if (counters) memset(counters, 0, sizeof(*counters));
    if (n_queries == 0) return QMX_OK;
    if (is_stopped && *is_stopped) {
        set_error("search cancelled");
        return QMX_ERR_CANCELLED;
    }
    uint32_t max_examples = 0;
    QMX_TRY(custom_validate(ex, queries, n_queries, ex->nq, &max_examples));
    const bool out_dev = is_device_ptr(out), cnt_dev = is_device_ptr(out_counts);
    if (g->n_points == 0) {   // get_entry_point() -> None
        if (cnt_dev) QMX_HIP(hipMemset(out_counts, 0, (size_t)n_queries * 4));
        else memset(out_counts, 0, (size_t)n_queries * 4);
        return QMX_OK;
    }
    QMX_TRY(ex->cq_desc.reserve((size_t)n_queries * sizeof(qmx_custom_query)));
    QMX_HIP(hipMemcpyAsync(ex->cq_desc.p, queries, (size_t)n_queries * sizeof(qmx_custom_query), hipMemcpyHostToDevice, ex->stream));
    CustomWalk cw{(const qmx_custom_query *)ex->cq_desc.p, (const float *)ex->cq_coefs.p, n_queries, max_examples};
    qmx_scored_point *d_out = out;
    uint32_t *d_counts = out_counts;
    if (!out_dev) { QMX_TRY(ex->out.reserve((size_t)n_queries * top * sizeof(qmx_scored_point))); d_out = (qmx_scored_point *)ex->out.p; }
    if (!cnt_dev) { QMX_TRY(ex->counts.reserve((size_t)n_queries * 4)); d_counts = (uint32_t *)ex->counts.p; }
    QMX_TRY(ex->hnsw_scored.reserve((size_t)n_queries * 4));
    const bool timed = ex->timing || (ex->seg->flags & QMX_SEG_TIME_KERNELS) != 0;
    QMX_TRY(hnsw_enqueue(g, ex, top, ef, d_out, d_counts, (uint32_t *)ex->hnsw_scored.p, timed, false, nullptr, nullptr, &cw));
    if (!out_dev) QMX_TRY(copy_out(ex->stream, out, d_out, (size_t)n_queries * top * sizeof(qmx_scored_point)));
    if (!cnt_dev) QMX_TRY(copy_out(ex->stream, out_counts, d_counts, (size_t)n_queries * 4));
    QMX_TRY(check_err_flag(ex));    // synchronises (the caller's descriptors may go away)
    if (counters) {
        std::vector<uint32_t> sc(n_queries);
        QMX_HIP(hipMemcpy(sc.data(), ex->hnsw_scored.p, (size_t)n_queries * 4, hipMemcpyDeviceToHost));
        uint64_t total = 0;
        for (uint32_t v : sc) total += v;
        counters->vectors_scored = total;            // POINTS scored (each costs one similarity per example of its query)
        counters->kernel_launches = 1;
        if (timed) { const float before = ex->timing_ms; QMX_TRY(timing_fold(ex)); counters->kernel_ms = ex->timing_ms - before; }
    }
    return QMX_OK;
}

// ---------------------------------------------------------------------------------------------
// multi-dense vectors: MaxSim (score_max_similarity, query_scorer/mod.rs:70-97)
// ---------------------------------------------------------------------------------------------
static int32_t multi_prepare(qmx_query *inner, const uint32_t *query_first, uint32_t n_queries, const uint64_t *point_offsets, uint32_t n_points,
                             const uint32_t *d_ids, uint64_t n) {
    const qmx_segment *s = inner->seg;
    const bool qf_dev = is_device_ptr(query_first), off_dev = is_device_ptr(point_offsets);
    QMX_REQUIRE(!qf_dev && !off_dev, QMX_ERR_BAD_ARG, "query_first and point_offsets are host arrays (they are validated here)");
    QMX_REQUIRE(query_first[0] <= query_first[n_queries] && query_first[n_queries] <= inner->nq, QMX_ERR_OUT_OF_BOUNDS,
                "multi-queries reach past the %u inner query vectors of the batch", inner->nq);
    for (uint32_t j = 0; j < n_queries; ++j)
        QMX_REQUIRE(query_first[j] <= query_first[j + 1], QMX_ERR_BAD_ARG, "query_first is not ascending at %u", j);
    const uint64_t n_rows = s->n;
    for (uint32_t p = 0; p < n_points; ++p)
        QMX_REQUIRE(point_offsets[p] <= point_offsets[p + 1], QMX_ERR_BAD_ARG, "point_offsets is not ascending at %u", p);
    QMX_REQUIRE(point_offsets[n_points] <= n_rows, QMX_ERR_OUT_OF_BOUNDS, "point_offsets reach past the %llu inner rows of the segment",
                (unsigned long long)n_rows);
    QMX_REQUIRE((uint64_t)inner->nq * n_rows * 4 <= (48ull << 30), QMX_ERR_NOT_SUPPORTED, "similarity matrix of %llu x %u floats is too large",
                (unsigned long long)n_rows, inner->nq);
    QMX_TRY(inner->mv_qfirst.reserve((size_t)(n_queries + 1) * 4));
    QMX_TRY(inner->mv_offsets.reserve((size_t)(n_points + 1) * 8));
    QMX_HIP(hipMemcpyAsync(inner->mv_qfirst.p, query_first, (size_t)(n_queries + 1) * 4, hipMemcpyHostToDevice, inner->stream));
    QMX_HIP(hipMemcpyAsync(inner->mv_offsets.p, point_offsets, (size_t)(n_points + 1) * 8, hipMemcpyHostToDevice, inner->stream));
    QMX_TRY(inner->cq_sims.reserve((size_t)inner->nq * n_rows * 4));
    QMX_TRY(inner->cq_scores.reserve((size_t)n_queries * n * 4));
    // similarity(inner query, inner row) for every pair, with the dense scan's lane policies (score mode: deleted flags are not consulted)
    QMX_TRY(score_ids_device(inner, nullptr, n_rows, (float *)inner->cq_sims.p, nullptr));
    return launch_maxsim(inner->stream, (const float *)inner->cq_sims.p, n_rows, (const uint32_t *)inner->mv_qfirst.p, n_queries,
                         (const uint64_t *)inner->mv_offsets.p, n_points, d_ids, n, (float *)inner->cq_scores.p, inner->d_err);
}

int32_t qmx_multi_score_points(qmx_query *inner, const uint32_t *query_first, uint32_t n_queries, const uint64_t *point_offsets, uint32_t n_points,
                               const uint32_t *ids, uint32_t n, float *scores) {
    QMX_REQUIRE(inner && query_first && point_offsets && (n == 0 || (ids && scores)), QMX_ERR_BAD_ARG, "NULL argument");
    QMX_HIP(hipSetDevice(inner->device));
    if (n == 0 || n_queries == 0) return QMX_OK;
    const void *d_ids = nullptr;
    QMX_TRY(stage_in(inner, inner->ids, ids, (size_t)n * 4, &d_ids));
    QMX_TRY(multi_prepare(inner, query_first, n_queries, point_offsets, n_points, (const uint32_t *)d_ids, n));
    QMX_TRY(copy_out(inner->stream, scores, inner->cq_scores.p, (size_t)n_queries * n * 4));
    return check_err_flag(inner);
}

int32_t qmx_multi_search_topk(qmx_query *inner, const uint32_t *query_first, uint32_t n_queries, const uint64_t *point_offsets, uint32_t n_points,
                              const uint64_t *point_deleted, uint64_t n_deleted_bits, uint32_t top, const uint32_t *ids, uint64_t n_ids,
                              qmx_scored_point *out, uint32_t *out_counts) {
    QMX_REQUIRE(inner && query_first && point_offsets && out && out_counts, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(top >= 1 && top <= MAX_TOP, QMX_ERR_NOT_SUPPORTED, "top %u not in 1..%u", top, MAX_TOP);
    QMX_HIP(hipSetDevice(inner->device));
    if (n_queries == 0) return QMX_OK;
    const void *d_ids = nullptr;
    uint64_t n = n_points;
    if (ids) {
        n = n_ids;
        if (n_ids) QMX_TRY(stage_in(inner, inner->ids, ids, (size_t)n_ids * 4, &d_ids));
    }
    const bool out_dev = is_device_ptr(out), cnt_dev = is_device_ptr(out_counts);
    qmx_scored_point *d_out = out;
    uint32_t *d_oc = out_counts;
    if (!out_dev) { QMX_TRY(inner->out.reserve((size_t)n_queries * top * sizeof(qmx_scored_point))); d_out = (qmx_scored_point *)inner->out.p; }
    if (!cnt_dev) { QMX_TRY(inner->counts.reserve((size_t)n_queries * 4)); d_oc = (uint32_t *)inner->counts.p; }
    if (n == 0) {
        QMX_HIP(hipMemsetAsync(d_oc, 0, (size_t)n_queries * 4, inner->stream));
    } else {
        QMX_TRY(multi_prepare(inner, query_first, n_queries, point_offsets, n_points, (const uint32_t *)d_ids, n));
        // deletion is per POINT here (the id tracker's bitslice over multi-vector points), not per inner row
        DeletedView del;
        memset(&del, 0, sizeof(del));
        del.n_rows = n_points;
        if (point_deleted && n_deleted_bits) {
            const void *d_bits = nullptr;
            QMX_TRY(stage_in(inner, inner->mv_deleted, point_deleted, (size_t)((n_deleted_bits + 63) / 64) * 8, &d_bits));
            del.point_deleted = (const uint64_t *)d_bits;
            del.n_point_bits = n_deleted_bits;
        }
        QMX_TRY(launch_custom_topk(inner->stream, (const float *)inner->cq_scores.p, n, (const uint32_t *)d_ids, del, n_queries, top, d_out, d_oc));
    }
    if (!out_dev) QMX_TRY(copy_out(inner->stream, out, d_out, (size_t)n_queries * top * sizeof(qmx_scored_point)));
    if (!cnt_dev) QMX_TRY(copy_out(inner->stream, out_counts, d_oc, (size_t)n_queries * 4));
    return check_err_flag(inner);
}

// Custom queries whose examples are multi-vectors (MultiCustomQueryScorer, query_scorer/multi_custom_query_scorer.rs:19-130; over quantized inner rows
// QuantizedMultiCustomQueryScorer, quantized/quantized_multi_custom_query_scorer.rs:19-96): similarity(example, point) = score_max_similarity, then the
// query's score_by.  The MaxSim row of every example (multi_prepare, as for plain multi-queries), then the combination over those rows.
static int32_t multi_custom_prepare(qmx_query *inner, const uint32_t *example_first, uint32_t n_examples, const qmx_custom_query *queries, uint32_t n_queries,
                                    const uint64_t *point_offsets, uint32_t n_points, const uint32_t *d_ids, uint64_t n) {
    QMX_TRY(custom_validate(inner, queries, n_queries, n_examples, nullptr));
    QMX_TRY(multi_prepare(inner, example_first, n_examples, point_offsets, n_points, d_ids, n));       // cq_scores[e * n + c] = MaxSim(example e, candidate c)
    QMX_TRY(inner->cq_multi.reserve((size_t)n_queries * n * 4));
    QMX_TRY(inner->cq_desc.reserve((size_t)n_queries * sizeof(qmx_custom_query)));
    QMX_HIP(hipMemcpyAsync(inner->cq_desc.p, queries, (size_t)n_queries * sizeof(qmx_custom_query), hipMemcpyHostToDevice, inner->stream));
    return launch_custom_combine(inner->stream, (const qmx_custom_query *)inner->cq_desc.p, n_queries, (const float *)inner->cq_scores.p, n,
                                 (const float *)inner->cq_coefs.p, (float *)inner->cq_multi.p);
}

int32_t qmx_multi_custom_score_points(qmx_query *inner, const uint32_t *example_first, uint32_t n_examples, const qmx_custom_query *queries, uint32_t n_queries,
                                      const uint64_t *point_offsets, uint32_t n_points, const uint32_t *ids, uint32_t n, float *scores) {
    QMX_REQUIRE(inner && example_first && point_offsets && (n_queries == 0 || queries) && (n == 0 || (ids && scores)), QMX_ERR_BAD_ARG, "NULL argument");
    QMX_HIP(hipSetDevice(inner->device));
    if (n == 0 || n_queries == 0) return QMX_OK;
    const void *d_ids = nullptr;
    QMX_TRY(stage_in(inner, inner->ids, ids, (size_t)n * 4, &d_ids));
    QMX_TRY(multi_custom_prepare(inner, example_first, n_examples, queries, n_queries, point_offsets, n_points, (const uint32_t *)d_ids, n));
    QMX_TRY(copy_out(inner->stream, scores, inner->cq_multi.p, (size_t)n_queries * n * 4));
    return check_err_flag(inner);
}

int32_t qmx_multi_custom_search_topk(qmx_query *inner, const uint32_t *example_first, uint32_t n_examples, const qmx_custom_query *queries, uint32_t n_queries,
                                     const uint64_t *point_offsets, uint32_t n_points, const uint64_t *point_deleted, uint64_t n_deleted_bits, uint32_t top,
                                     const uint32_t *ids, uint64_t n_ids, qmx_scored_point *out, uint32_t *out_counts) {
    QMX_REQUIRE(inner && example_first && point_offsets && (n_queries == 0 || queries) && out && out_counts, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(top >= 1 && top <= MAX_TOP, QMX_ERR_NOT_SUPPORTED, "top %u not in 1..%u", top, MAX_TOP);
    QMX_HIP(hipSetDevice(inner->device));
    if (n_queries == 0) return QMX_OK;
    const void *d_ids = nullptr;
    uint64_t n = n_points;
    if (ids) {
        n = n_ids;
        if (n_ids) QMX_TRY(stage_in(inner, inner->ids, ids, (size_t)n_ids * 4, &d_ids));
    }
    const bool out_dev = is_device_ptr(out), cnt_dev = is_device_ptr(out_counts);
    qmx_scored_point *d_out = out;
    uint32_t *d_oc = out_counts;
    if (!out_dev) { QMX_TRY(inner->out.reserve((size_t)n_queries * top * sizeof(qmx_scored_point))); d_out = (qmx_scored_point *)inner->out.p; }
    if (!cnt_dev) { QMX_TRY(inner->counts.reserve((size_t)n_queries * 4)); d_oc = (uint32_t *)inner->counts.p; }
    if (n == 0) {
        QMX_HIP(hipMemsetAsync(d_oc, 0, (size_t)n_queries * 4, inner->stream));
    } else {
        QMX_TRY(multi_custom_prepare(inner, example_first, n_examples, queries, n_queries, point_offsets, n_points, (const uint32_t *)d_ids, n));
        DeletedView del;      // deletion is per POINT (the id tracker's bitslice over multi-vector points)
        memset(&del, 0, sizeof(del));
        del.n_rows = n_points;
        if (point_deleted && n_deleted_bits) {
            const void *d_bits = nullptr;
            QMX_TRY(stage_in(inner, inner->mv_deleted, point_deleted, (size_t)((n_deleted_bits + 63) / 64) * 8, &d_bits));
            del.point_deleted = (const uint64_t *)d_bits;
            del.n_point_bits = n_deleted_bits;
        }
        QMX_TRY(launch_custom_topk(inner->stream, (const float *)inner->cq_multi.p, n, (const uint32_t *)d_ids, del, n_queries, top, d_out, d_oc));
    }
    if (!out_dev) QMX_TRY(copy_out(inner->stream, out, d_out, (size_t)n_queries * top * sizeof(qmx_scored_point)));
    if (!cnt_dev) QMX_TRY(copy_out(inner->stream, out_counts, d_oc, (size_t)n_queries * 4));
    return check_err_flag(inner);
}

// GraphLayers::search over multi-vector POINTS with a custom query whose examples are multi-vectors (MultiCustomQueryScorer behind FilteredScorer)
int32_t qmx_multi_custom_hnsw_search(const qmx_hnsw *g, qmx_query *inner, const uint32_t *example_first, uint32_t n_examples, const qmx_custom_query *queries,
                                     uint32_t n_queries, const uint64_t *point_offsets, uint32_t n_points, const uint64_t *point_deleted, uint64_t n_deleted_bits,
                                     uint32_t top, uint32_t ef, qmx_scored_point *out, uint32_t *out_counts, qmx_counters *counters) {
    QMX_REQUIRE(g && inner && example_first && point_offsets && (n_queries == 0 || queries) && out && out_counts, QMX_ERR_BAD_ARG, "NULL argument");
    const qmx_segment *s = inner->seg;
    QMX_REQUIRE(g->device == s->device, QMX_ERR_BAD_ARG, "graph lives on device %d, the segment on %d", g->device, s->device);
    QMX_REQUIRE(g->n_points <= n_points, QMX_ERR_OUT_OF_BOUNDS, "graph has %u points, the multi-vector storage %u", g->n_points, n_points);
    QMX_REQUIRE(top >= 1, QMX_ERR_BAD_ARG, "top must be > 0");
    QMX_REQUIRE(std::max(top, ef) <= HNSW_MAX_EF, QMX_ERR_NOT_SUPPORTED, "max(top, ef) = %u > %u not supported yet", std::max(top, ef), HNSW_MAX_EF);
    QMX_REQUIRE(!is_device_ptr(example_first) && !is_device_ptr(point_offsets), QMX_ERR_BAD_ARG, "example_first and point_offsets are host arrays");
    QMX_HIP(hipSetDevice(inner->device));
    if (counters) memset(counters, 0, sizeof(*counters));
    if (n_queries == 0) return QMX_OK;
    QMX_REQUIRE(example_first[n_examples] <= inner->nq, QMX_ERR_OUT_OF_BOUNDS, "examples reach past the %u inner query vectors of the batch", inner->nq);
    for (uint32_t e = 0; e < n_examples; ++e) QMX_REQUIRE(example_first[e] <= example_first[e + 1], QMX_ERR_BAD_ARG, "example_first is not ascending at %u", e);
    for (uint32_t p = 0; p < n_points; ++p) QMX_REQUIRE(point_offsets[p] <= point_offsets[p + 1], QMX_ERR_BAD_ARG, "point_offsets is not ascending at %u", p);
    QMX_REQUIRE(point_offsets[n_points] <= s->n, QMX_ERR_OUT_OF_BOUNDS, "point_offsets reach past the %llu inner rows of the segment", (unsigned long long)s->n);
    uint32_t max_examples = 0;
    QMX_TRY(custom_validate(inner, queries, n_queries, n_examples, &max_examples));
    uint64_t lds_need = 0;        // the largest staged block: header + offset table + per example (16-byte MaxSim header + its tokens)
    for (uint32_t i = 0; i < n_queries; ++i) {
        const qmx_custom_query &c = queries[i];
        const uint32_t ne = c.kind <= QMX_CUSTOM_RECO_SUM_SCORES ? c.n_a + c.n_b : c.n_a + 2 * c.n_b;
        uint64_t need = 32 + ((4ull * ne + 15) & ~15ull);
        for (uint32_t e = 0; e < ne; ++e) need += 16 + (uint64_t)(example_first[c.first + e + 1] - example_first[c.first + e]) * inner->q_stride;
        lds_need = std::max(lds_need, need);
    }
    QMX_REQUIRE(lds_need <= HNSW_LDS_QUERY_MAX, QMX_ERR_NOT_SUPPORTED, "a custom query of %llu bytes of example tokens does not fit the LDS",
                (unsigned long long)lds_need);
    const bool out_dev = is_device_ptr(out), cnt_dev = is_device_ptr(out_counts);
    if (g->n_points == 0) {
        if (cnt_dev) QMX_HIP(hipMemset(out_counts, 0, (size_t)n_queries * 4));
        else memset(out_counts, 0, (size_t)n_queries * 4);
        return QMX_OK;
    }
    QMX_TRY(inner->mv_qfirst.reserve((size_t)(n_examples + 1) * 4));
    QMX_TRY(inner->mv_offsets.reserve((size_t)(n_points + 1) * 8));
    QMX_TRY(inner->cq_desc.reserve((size_t)n_queries * sizeof(qmx_custom_query)));
    QMX_HIP(hipMemcpyAsync(inner->mv_qfirst.p, example_first, (size_t)(n_examples + 1) * 4, hipMemcpyHostToDevice, inner->stream));
    QMX_HIP(hipMemcpyAsync(inner->mv_offsets.p, point_offsets, (size_t)(n_points + 1) * 8, hipMemcpyHostToDevice, inner->stream));
    QMX_HIP(hipMemcpyAsync(inner->cq_desc.p, queries, (size_t)n_queries * sizeof(qmx_custom_query), hipMemcpyHostToDevice, inner->stream));
    MultiWalk mw;
    memset(&mw, 0, sizeof(mw));
    mw.d_qfirst = (const uint32_t *)inner->mv_qfirst.p;
    mw.d_offsets = (const uint64_t *)inner->mv_offsets.p;
    mw.n_queries = n_queries;
    mw.max_tokens = 1;
    mw.del.n_rows = n_points;
    if (point_deleted && n_deleted_bits) {
        const void *d_bits = nullptr;
        QMX_TRY(stage_in(inner, inner->mv_deleted, point_deleted, (size_t)((n_deleted_bits + 63) / 64) * 8, &d_bits));
        mw.del.point_deleted = (const uint64_t *)d_bits;
        mw.del.n_point_bits = n_deleted_bits;
    }
    if (inner->has_filter) { mw.del.allowed = (const uint64_t *)inner->filter.p; mw.del.n_allowed_bits = inner->n_filter_bits; }
    CustomWalk cw{(const qmx_custom_query *)inner->cq_desc.p, (const float *)inner->cq_coefs.p, n_queries, max_examples, (uint32_t)lds_need};
    qmx_scored_point *d_out = out;
    uint32_t *d_counts = out_counts;
    if (!out_dev) { QMX_TRY(inner->out.reserve((size_t)n_queries * top * sizeof(qmx_scored_point))); d_out = (qmx_scored_point *)inner->out.p; }
    if (!cnt_dev) { QMX_TRY(inner->counts.reserve((size_t)n_queries * 4)); d_counts = (uint32_t *)inner->counts.p; }
    QMX_TRY(inner->hnsw_scored.reserve((size_t)n_queries * 4));
    const bool timed = inner->timing || (s->flags & QMX_SEG_TIME_KERNELS) != 0;
    QMX_TRY(hnsw_enqueue(g, inner, top, ef, d_out, d_counts, (uint32_t *)inner->hnsw_scored.p, timed, false, &mw, nullptr, &cw));
    if (!out_dev) QMX_TRY(copy_out(inner->stream, out, d_out, (size_t)n_queries * top * sizeof(qmx_scored_point)));
    if (!cnt_dev) QMX_TRY(copy_out(inner->stream, out_counts, d_counts, (size_t)n_queries * 4));
    QMX_TRY(check_err_flag(inner));
    if (counters) {
        std::vector<uint32_t> sc(n_queries);
        QMX_HIP(hipMemcpy(sc.data(), inner->hnsw_scored.p, (size_t)n_queries * 4, hipMemcpyDeviceToHost));
        uint64_t total = 0;
        for (uint32_t v : sc) total += v;
        counters->vectors_scored = total;
        counters->kernel_launches = 1;
        if (timed) { const float before = inner->timing_ms; QMX_TRY(timing_fold(inner)); counters->kernel_ms = inner->timing_ms - before; }
    }
    return QMX_OK;
}

}  // extern "C"

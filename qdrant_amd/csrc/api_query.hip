// api_query.hip — the C-ABI of include/qdrant_amd.h, query batches and scoring (score_points, pairs, rescore, score_internal, score_bytes).
// (One of the api_*.hip translation units; what they share: api_internal.hpp.)
#include "api_internal.hpp"

extern "C" {

// ---------------------------------------------------------------------------------------------
// query batch
// ---------------------------------------------------------------------------------------------
static int32_t query_alloc(const qmx_segment *seg, uint32_t nq, qmx_query **out, bool internal = false) {
    qmx_query *q = new (std::nothrow) qmx_query();
    QMX_REQUIRE(q, QMX_ERR_OUT_OF_MEMORY, "host allocation failed");
    q->seg = seg;
    q->device = seg->device;
    q->nq = nq;
    q->nq_padded = ((nq + MAX_QT_TOPK - 1) / MAX_QT_TOPK) * MAX_QT_TOPK;
    if (q->nq_padded == 0) q->nq_padded = MAX_QT_TOPK;
    if (seg->dtype == QMX_DTYPE_PQ) q->nq_padded = std::max<uint32_t>(nq, 1);   // LUTs are never read past nq
    // tile entry = elements zero-padded to whole 128-byte segments + the aux block
    // a scalar-encoded BQ query holds `bits` planes per row word (a stored row as the query has one: score_internal is 1-bit)
    q->bq_bits = (seg->dtype == QMX_DTYPE_BQ && !internal) ? seg->bq_query_bits : 1;
    q->aux_off = (uint32_t)((seg->scan_dim * elem_bytes(seg->dtype) * q->bq_bits + 127) & ~127u);
    if (seg->dtype == QMX_DTYPE_TQ) tq_entry_layout(seg, &q->bq_bits, &q->tq_qbytes_off, &q->aux_off);
    if (seg->dtype == QMX_DTYPE_BQ && q->bq_bits > 1 && seg->fast_layout()) {   // behind the planes: the values as bytes (scan_sq_mfma.hip BqOps), 8 per row byte
        q->tq_qbytes_off = q->aux_off;
        q->aux_off += ((seg->scan_dim + 63) & ~63u) * 8;
    }
    q->q_stride = q->aux_off + QUERY_AUX_BYTES;
    if (seg->dtype == QMX_DTYPE_SQ_U8 || seg->dtype == QMX_DTYPE_F16 || seg->dtype == QMX_DTYPE_TQ || (seg->dtype == QMX_DTYPE_BQ && q->tq_qbytes_off))
        q->q_stride = lds_tile_stride(q->q_stride);
    if (seg->dtype == QMX_DTYPE_PQ) {   // the encoded query is the LUT [m][n_centroids] f32 (EncodedQueryPQ)
        q->q_stride = (uint32_t)(((size_t)seg->pq_m * seg->pq.n_centroids * sizeof(float) + 15) & ~(size_t)15);
        q->aux_off = 0;
    }
    auto fail = [&](hipError_t e, const char *what) {
        int32_t rc = hip_status(e, what, __FILE__, __LINE__);
        qmx_query_destroy(q);
        return rc;
    };
    hipError_t e = hipStreamCreateWithFlags(&q->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) return fail(e, "hipStreamCreate");
    q->stream = q->own_stream;
    const size_t qbytes = (size_t)q->nq_padded * q->q_stride;
    e = hipMalloc(&q->d_queries, qbytes);
    if (e != hipSuccess) return fail(e, "hipMalloc(queries)");
    e = hipMemsetAsync(q->d_queries, 0, qbytes, q->stream);
    if (e != hipSuccess) return fail(e, "hipMemset(queries)");
    e = hipMalloc((void **)&q->d_err, sizeof(int));
    if (e != hipSuccess) return fail(e, "hipMalloc(err)");
    e = hipMemsetAsync(q->d_err, 0, sizeof(int), q->stream);
    if (e != hipSuccess) return fail(e, "hipMemset(err)");
    *out = q;
    return QMX_OK;
}

// MetricQueryScorer::new (metric_query_scorer.rs:35-58) for every query of the batch: preprocess
// once, cast to the element type, pack into the LDS-tile layout.  Enqueued on the query's stream.
static int32_t query_encode(qmx_query *q, const float *queries) {
    const qmx_segment *seg = q->seg;
    const uint32_t nq = q->nq;
    if (nq == 0) return QMX_OK;
    const size_t fbytes = (size_t)nq * seg->dim * sizeof(float);
    const void *d_src = nullptr;
    QMX_TRY(stage_in(q, q->misc, queries, fbytes, &d_src));
    QMX_TRY(q->enc.reserve(fbytes));
    float *d_f32 = (float *)q->enc.p;
    // u8 storages never normalise (metric_uint/simple_cosine.rs:53-55)
    const bool normalise = seg->distance == QMX_DISTANCE_COSINE && seg->dtype != QMX_DTYPE_U8;
    if (normalise) {
        QMX_TRY(launch_cosine_preprocess_f32(q->stream, (const float *)d_src, d_f32, nq, seg->dim));
    } else {
        QMX_HIP(hipMemcpyAsync(d_f32, d_src, fbytes, hipMemcpyDeviceToDevice, q->stream));
    }
    if (seg->dtype <= QMX_DTYPE_U8)
        return launch_pack_queries(q->stream, (int)seg->dtype, (int)seg->distance, d_f32, 0, seg->dim * 4, nq, seg->dim,
                                   q->d_queries, q->q_stride, q->aux_off);
    if (seg->dtype == QMX_DTYPE_SQ_U8)   // EncodedVectorsU8::encode_query (encoded_vectors_u8.rs:583-619)
        return launch_sq_encode(q->stream, (int)seg->distance, seg->sq, seg->dim, d_f32, nq, (uint8_t *)q->d_queries, q->q_stride,
                                nullptr, nullptr, 1, q->aux_off);
    if (seg->dtype == QMX_DTYPE_BQ && q->bq_bits > 1)   // encode_query_vector, Scalar4bits / Scalar8bits (:683-756)
        return launch_bq_encode_scalar_query(q->stream, d_f32, nq, seg->dim, seg->bq_encoding, q->bq_bits, (uint8_t *)q->d_queries, q->q_stride, q->tq_qbytes_off,
                                             q->aux_off, (seg->scan_dim + 63) & ~63u);
    if (seg->dtype == QMX_DTYPE_BQ)      // encode_query_vector, SameAsStorage (encoded_vectors_binary.rs:673-690) = encode_one_bit_vector
        return launch_bq_encode(q->stream, d_f32, nq, seg->dim, seg->bq_encoding, seg->d_bq_mean, seg->d_bq_stddev, (uint8_t *)q->d_queries, q->q_stride);
    if (tq_l1(seg)) return QMX_OK;       // DistanceType::L1 scores against the query as given (quantization.rs:532-535): q->enc holds it
    if (seg->dtype == QMX_DTYPE_TQ) {    // TurboQuantizer::precompute_query (turboquant/quantization.rs:496-567)
        QMX_TRY(q->tq_rot.reserve((size_t)nq * seg->tq_padded_dim * sizeof(double)));
        QMX_TRY(launch_tq_rotate(q->stream, d_f32, nq, tq_rotation(seg), (double *)q->tq_rot.p));
        return launch_tq_query_encode(q->stream, (double *)q->tq_rot.p, nq, seg->tq_padded_dim, seg->tq_value_bits,
                                      seg->distance == QMX_DISTANCE_EUCLID ? 1 : 0, q->d_queries, q->q_stride, q->aux_off, seg->d_tq_shift, seg->d_tq_scale,
                                      q->tq_qbytes_off);
    }
    if (seg->dtype == QMX_DTYPE_PQ)      // EncodedVectorsPQ::encode_query (encoded_vectors_pq.rs:519-541)
        return launch_pq_lut(q->stream, seg->distance, seg->dim, seg->pq, seg->d_centroids, d_f32, nq, (float *)q->d_queries);
    set_error("query encode for dtype %u not built yet", seg->dtype);
    return QMX_ERR_NOT_SUPPORTED;
}

int32_t qmx_query_create(const qmx_segment *seg, const float *queries, uint32_t nq, qmx_query **out) {
    QMX_REQUIRE(seg && out && (nq == 0 || queries), QMX_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    QMX_HIP(hipSetDevice(seg->device));
    qmx_query *q = nullptr;
    QMX_TRY(query_alloc(seg, nq, &q));
    int32_t rc = query_encode(q, queries);
    if (rc == QMX_OK) {
        hipError_t e = hipStreamSynchronize(q->stream);
        if (e != hipSuccess) rc = hip_status(e, "sync", __FILE__, __LINE__);
    }
    if (rc != QMX_OK) {
        qmx_query_destroy(q);
        return rc;
    }
    *out = q;
    return QMX_OK;
}

int32_t qmx_query_update(qmx_query *q, const float *queries) {
    QMX_REQUIRE(q && (q->nq == 0 || queries), QMX_ERR_BAD_ARG, "NULL argument");
    QMX_HIP(hipSetDevice(q->seg->device));
    return query_encode(q, queries);
}

int32_t qmx_query_create_internal(const qmx_segment *seg, const uint32_t *point_ids, uint32_t nq, qmx_query **out) {
    QMX_REQUIRE(seg && out && (nq == 0 || point_ids), QMX_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    QMX_HIP(hipSetDevice(seg->device));
    QMX_REQUIRE(seg->dtype <= QMX_DTYPE_SQ_U8 || seg->dtype == QMX_DTYPE_BQ, QMX_ERR_NOT_SUPPORTED,
                "dtype %u has no internal encoding (EncodedVectorsPQ::encode_internal_vector returns None): pass the original vector to qmx_query_create",
                seg->dtype);
    qmx_query *q = nullptr;
    QMX_TRY(query_alloc(seg, nq, &q, true));
    int32_t rc = QMX_OK;
    do {
        if (nq == 0) break;
        const void *d_ids = nullptr;
        if ((rc = stage_in(q, q->ids, point_ids, (size_t)nq * 4, &d_ids)) != QMX_OK) break;
        if (seg->dtype == QMX_DTYPE_SQ_U8) {   // encode_internal_vector (encoded_vectors_u8.rs:715-728)
            float shift = (seg->distance == QMX_DISTANCE_DOT || seg->distance == QMX_DISTANCE_COSINE)
                              ? (float)seg->sq.actual_dim * seg->sq.offset * seg->sq.offset : 0.0f;
            if (seg->sq.invert) shift = -shift;
            if ((rc = launch_sq_internal_query(q->stream, seg->d_rows, seg->d_row_offsets, seg->sq.actual_dim, (const uint32_t *)d_ids,
                                               nq, seg->n, shift, q->d_queries, q->q_stride, q->aux_off, q->d_err)) != QMX_OK) break;
            if ((rc = check_err_flag(q)) != QMX_OK) break;
            break;
        }
        // the stored row IS the query (already preprocessed at insert): FilteredScorer::new_internal
        if ((rc = q->misc.reserve((size_t)nq * seg->row_bytes)) != QMX_OK) break;
        if ((rc = launch_gather_rows(q->stream, seg->d_rows, seg->row_stride, seg->row_bytes, (const uint32_t *)d_ids, nq,
                                     seg->n, q->misc.p, q->d_err)) != QMX_OK) break;
        // BQ: EncodedVectorsBin::encode_internal_vector (encoded_vectors_binary.rs:923-934) = the stored bits, packed as bytes
        const bool bq = seg->dtype == QMX_DTYPE_BQ;
        if ((rc = launch_pack_queries(q->stream, bq ? (int)QMX_DTYPE_U8 : (int)seg->dtype, bq ? (int)QMX_DISTANCE_DOT : (int)seg->distance,
                                      q->misc.p, 1, (uint32_t)seg->row_bytes, nq, bq ? (uint32_t)seg->row_bytes : seg->dim, q->d_queries,
                                      q->q_stride, q->aux_off)) != QMX_OK) break;
        if ((rc = check_err_flag(q)) != QMX_OK) break;
    } while (0);
    if (rc != QMX_OK) {
        qmx_query_destroy(q);
        return rc;
    }
    *out = q;
    return QMX_OK;
}

int32_t qmx_query_destroy(qmx_query *q) {
    if (!q) return QMX_OK;
    (void)hipSetDevice(q->device);
    if (q->stream) (void)hipStreamSynchronize(q->stream);
    if (q->d_queries) (void)hipFree(q->d_queries);
    if (q->d_err) (void)hipFree(q->d_err);
    q->partial.release();
    q->out.release();
    q->counts.release();
    q->ids.release();
    q->scores.release();
    q->misc.release();
    q->enc.release();
    q->bounds.release();
    q->gthr.release();
    q->cq_coefs.release();
    q->filter.release();
    q->cq_sims.release();
    q->mv_qfirst.release();
    q->mv_offsets.release();
    q->mv_deleted.release();
    q->cq_scores.release();
    q->cq_desc.release();
    q->cq_multi.release();
    q->sp_bq.release(); q->sp_f32.release(); q->sp_cand.release(); q->sp_cnt.release(); q->sp_ver.release(); q->sp_vscores.release(); q->sp_sample.release(); q->sp_wl.release(); q->xcnt.release(); q->tq_rot.release(); q->sp_plan.release(); q->sp_fq.release(); q->sp_probe.release(); q->sp_pscores.release(); q->pq_table.release(); q->sh_lists.release(); q->sh_out.release();
    if (q->sh_done) (void)hipEventDestroy(q->sh_done);
    if (q->sh_merged) (void)hipEventDestroy(q->sh_merged);
    q->cand.release();
    q->cand_cnt.release();
    q->cand_ids.release();
    q->hnsw_vis.release();
    q->hnsw_log.release();
    q->hnsw_scored.release();
    q->hnsw_pq8.release();
    q->hnsw_lutx.release();
    q->hnsw_next.release();
    q->hnsw_refc.release();
    for (auto &p : q->evs) {
        if (p.a) (void)hipEventDestroy(p.a);
        if (p.b) (void)hipEventDestroy(p.b);
    }
    if (q->own_stream) (void)hipStreamDestroy(q->own_stream);
    delete q;
    return QMX_OK;
}

int32_t qmx_query_set_filter(qmx_query *q, const uint64_t *allowed, uint64_t n_bits) {
    QMX_REQUIRE(q, QMX_ERR_BAD_ARG, "NULL query");
    QMX_HIP(hipSetDevice(q->device));
    if (!allowed) {
        q->has_filter = false;
        q->n_filter_bits = 0;
        return QMX_OK;
    }
    const size_t words = (size_t)((n_bits + 63) / 64);
    QMX_TRY(q->filter.reserve(std::max<size_t>(words, 1) * 8));
    if (words) QMX_HIP(hipMemcpyAsync(q->filter.p, allowed, words * 8, hipMemcpyDefault, q->stream));
    QMX_HIP(hipStreamSynchronize(q->stream));     // the caller's buffer may go away
    q->n_filter_bits = n_bits;
    q->has_filter = true;
    return QMX_OK;
}

int32_t qmx_query_set_stream(qmx_query *q, void *hip_stream) {
    QMX_REQUIRE(q, QMX_ERR_BAD_ARG, "NULL query");
    QMX_HIP(hipStreamSynchronize(q->stream));
    q->stream = hip_stream ? (hipStream_t)hip_stream : q->own_stream;
    return QMX_OK;
}

int32_t qmx_query_set_timing(qmx_query *q, int32_t enabled) {
    QMX_REQUIRE(q, QMX_ERR_BAD_ARG, "NULL query");
    q->timing = enabled != 0;
    return QMX_OK;
}

int32_t qmx_query_timing(qmx_query *q, float *total_ms, uint32_t *n_launches) {
    QMX_REQUIRE(q, QMX_ERR_BAD_ARG, "NULL query");
    QMX_HIP(hipSetDevice(q->seg->device));
    QMX_HIP(hipStreamSynchronize(q->stream));
    QMX_TRY(timing_fold(q));
    if (total_ms) *total_ms = q->timing_ms;
    if (n_launches) *n_launches = q->timing_launches;
    q->timing_ms = 0.f;
    q->timing_launches = 0;
    return QMX_OK;
}

int32_t qmx_query_last_kernel(const qmx_query *q, char *buf, size_t buf_len) {
    QMX_REQUIRE(q && buf && buf_len > 0, QMX_ERR_BAD_ARG, "NULL argument");
    buf[0] = 0;
    if (!q->last_kernel) return QMX_OK;
    const char *mangled = hipKernelNameRefByPtr(q->last_kernel, q->stream);
    if (!mangled) return QMX_OK;
    int status = 0;
    char *dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
    snprintf(buf, buf_len, "%s", (status == 0 && dem) ? dem : mangled);
    free(dem);
    return QMX_OK;
}

int32_t qmx_query_synchronize(qmx_query *q) {
    QMX_REQUIRE(q, QMX_ERR_BAD_ARG, "NULL query");
    QMX_HIP(hipSetDevice(q->seg->device));
    QMX_HIP(hipStreamSynchronize(q->stream));
    return QMX_OK;
}

int32_t qmx_query_read_encoded(const qmx_query *q, uint32_t query_index, void *out, uint64_t out_bytes, uint64_t *written) {
    QMX_REQUIRE(q && out, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(query_index < q->nq, QMX_ERR_OUT_OF_BOUNDS, "query index %u >= %u", query_index, q->nq);
    QMX_HIP(hipSetDevice(q->seg->device));
    const bool sq = q->seg->dtype == QMX_DTYPE_SQ_U8;
    const uint64_t ebytes = q->seg->dtype == QMX_DTYPE_PQ ? (uint64_t)q->seg->pq_m * q->seg->pq.n_centroids * sizeof(float)
                                                          : (uint64_t)q->seg->scan_dim * elem_bytes(q->seg->dtype) * q->bq_bits;
    const uint64_t bytes = ebytes + (sq ? 4 : 0);
    QMX_REQUIRE(out_bytes >= bytes, QMX_ERR_BAD_ARG, "buffer too small: need %llu", (unsigned long long)bytes);
    QMX_HIP(hipStreamSynchronize(q->stream));
    const char *entry = (const char *)q->d_queries + (size_t)query_index * q->q_stride;
    if (sq) QMX_HIP(hipMemcpy(out, entry + q->aux_off, 4, hipMemcpyDefault));   // EncodedQueryU8{offset, encoded_query}
    QMX_HIP(hipMemcpy((char *)out + (sq ? 4 : 0), entry, ebytes, hipMemcpyDefault));
    if (written) *written = bytes;
    return QMX_OK;
}

// ---------------------------------------------------------------------------------------------
// scoring
// ---------------------------------------------------------------------------------------------
void fill_args(const qmx_query *q, uint32_t tile0, uint32_t nq_tile, ScanArgs &a) {
    const qmx_segment *s = q->seg;
    memset(&a, 0, sizeof(a));
    a.rows = s->d_rows;
    a.n_rows = s->n;
    a.row_stride = s->row_stride;
    a.dim = s->scan_dim;
    a.nq = nq_tile;
    a.queries = (const char *)q->d_queries + (size_t)tile0 * q->q_stride;
    a.q_stride = q->q_stride;
    a.aux_off = q->aux_off;
    {   // SIMD body / scalar tail split of the reference leaf (AVX loops step 32 elements)
        const uint32_t eb = elem_bytes(s->dtype);
        const uint32_t full = s->dtype <= QMX_DTYPE_U8 ? s->scan_dim - s->scan_dim % 32 : s->scan_dim;
        const uint32_t body_bytes = full * eb;
        a.nseg = body_bytes / 128;
        a.rem_pieces = (body_bytes % 128) / 16;
        a.tail_start = full;
    }
    a.del = s->deleted_view();
    if (q->has_filter) {
        a.del.allowed = (const uint64_t *)q->filter.p;
        a.del.n_allowed_bits = q->n_filter_bits;
    }
    a.err_flag = q->d_err;
    a.tq_l1 = s->d_tq_l1;
    a.flags = s->flags;
    a.sq_multiplier = s->sq.multiplier;
    a.row_offsets = s->d_row_offsets;
    a.pq_m = s->pq_m;
    a.pq_ncent = s->pq.n_centroids;
    a.pq_pair = s->d_pq_pair;
    a.pq_invert = s->pq.invert;
    if (s->dtype == QMX_DTYPE_PQ) {      // the codebook itself (the LUT-free walk, pq.hip HopPQDirect)
        a.pq_centroids = s->d_centroids;
        a.pq_dim = s->dim;
        a.pq_chunk = s->pq.chunk_size;
        a.pq_kind = (s->distance == QMX_DISTANCE_DOT || s->distance == QMX_DISTANCE_COSINE) ? 0u : s->distance == QMX_DISTANCE_MANHATTAN ? 1u : 2u;
    }
    a.bq_dim = s->dim;
    // calculate_metric's match: (Dot | Cosine, invert = false) and (L1 | L2, invert = true) -> zeros - xor; the toggled pairs -> xor - zeros
    a.bq_flip = (s->flags & QMX_SEG_BQ_TOGGLE_INVERT) ? 1 : 0;
    a.bq_qbits = q->bq_bits;
    a.tq_sf = s->d_tq_sf;
    a.tq_l2 = s->d_tq_l2;
    a.tq_bits = s->tq_value_bits;
    a.tq_invert = s->tq_invert ? 1 : 0;
    a.tq_planes = (s->tq_value_bits == 1 && s->d_tq_shift) ? 16 : 8;
    a.tq_qbytes_off = q->tq_qbytes_off;
    // |low + 128 high| <= 8127 * 128 * dims (4 / 2 bits); |2 v.q - sum q| <= 3 * 32767 * dims (1 bit, 16-bit TQ+ queries)
    a.tq_i32 = s->dtype == QMX_DTYPE_TQ && (s->tq_value_bits == 1 ? s->tq_padded_dim <= 16384 : s->tq_padded_dim <= 2000) ? 1 : 0;
}

int32_t launch_scan(const qmx_query *q, int qt, ScanMode mode, const ScanArgs &a, uint32_t *grid) {
    const qmx_segment *s = q->seg;
    if (s->dtype <= QMX_DTYPE_U8) {
        QMX_REQUIRE(s->fast_layout(), QMX_ERR_NOT_SUPPORTED,
                    "dtype %u dim %u: an adopted device block needs a 16-byte aligned base and row stride (got stride %llu); "
                    "let qmx_segment_create upload it instead", s->dtype, s->dim, (unsigned long long)s->row_stride);
        if (qt >= 8 && mfma_scan_ok(s))
            return s->dtype == QMX_DTYPE_F32 ? launch_scan_f32_mfma(q->stream, qt, mode, a, s->num_cus, grid)
                                              : launch_scan_f16_mfma(q->stream, qt, mode, a, s->num_cus, grid);
        return launch_scan_dense(q->stream, (int)s->dtype, (int)s->distance, qt, mode, a, s->num_cus, grid);
    }
    if (s->dtype == QMX_DTYPE_SQ_U8) {
        // (4 queries already pay for the padded 16-query matrix-core pass: 1.28 ms against 1.64 ms on the VALU kernel, 10 M x 768)
        if (qt >= 4 && mfma_scan_ok(s)) return launch_scan_sq_mfma(q->stream, std::max(qt, 8), mode, a, s->num_cus, grid);
        return launch_scan_sq(q->stream, (int)s->distance, std::min(qt, (int)MAX_QT), mode, a, s->num_cus, grid);
    }
    if (s->dtype == QMX_DTYPE_PQ) return launch_scan_pq(q->stream, mode, a, s->num_cus, grid);
    if (s->dtype == QMX_DTYPE_BQ) {
        QMX_REQUIRE(s->fast_layout(), QMX_ERR_NOT_SUPPORTED, "adopted BQ block is not 16-byte aligned");
        if (qt >= 4 && bq_mfma_ok(q)) return launch_scan_bq_mfma(q->stream, std::max(qt, 8), mode, a, s->num_cus, grid);
        return launch_scan_bq(q->stream, std::min(qt, (int)MAX_QT), mode, a, s->num_cus, grid);
    }
    if (s->dtype == QMX_DTYPE_TQ) {
        // 4 queries already pay for the padded 16-query matrix-core pass (integer arithmetic either way: the same bits)
        if (qt >= 4 && mfma_scan_ok(s)) return launch_scan_tq_mfma(q->stream, std::max(qt, 8), mode, a, s->num_cus, grid);
        return launch_scan_tq(q->stream, std::min(qt, 4), mode, a, s->num_cus, grid);
    }
    set_error("dtype %u not built yet", s->dtype);
    return QMX_ERR_NOT_SUPPORTED;
}

// TurboQuant over Manhattan: scores of queries [q0, q0 + nq) against the candidates d_ids[0..n) (rows 0..n without ids) into d_scores[(qi - q0) * stride + i],
// or - sel - of the (query, candidate) items of a PairSel into d_scores[i].  Batches of 65 536 rows: dequantise, rotate back, sum |q - v|.
int32_t tq_l1_scores_device(qmx_query *q, uint32_t q0, uint32_t nq, const uint32_t *d_ids, uint64_t n, float *d_scores, uint64_t stride, const PairSel *sel) {
    const qmx_segment *s = q->seg;
    const uint64_t B = 65536;
    QMX_TRY(q->tq_rot.reserve((size_t)std::min<uint64_t>(n, B) * s->tq_padded_dim * sizeof(double)));
    double *buf = (double *)q->tq_rot.p;
    const float *d_q = (const float *)q->enc.p + (size_t)q0 * s->dim;
    for (uint64_t r0 = 0; r0 < n; r0 += B) {
        const uint32_t cnt = (uint32_t)std::min<uint64_t>(B, n - r0);
        QMX_TRY(launch_tq_l1_dequant(q->stream, s->d_rows, s->row_stride, s->d_tq_sf, d_ids ? d_ids + r0 : nullptr, r0, cnt, s->n, s->tq_padded_dim, s->tq_value_bits,
                                     s->d_tq_shift, s->d_tq_scale, buf, q->d_err, sel));
        QMX_TRY(launch_tq_rotate_f64(q->stream, buf, cnt, tq_rotation_inverse(s)));
        QMX_TRY(launch_tq_l1_scores(q->stream, buf, cnt, s->tq_padded_dim, s->dim, sel ? (const float *)q->enc.p : d_q, s->dim, 0, nq, d_scores, stride, r0,
                                    s->tq_invert ? 1 : 0, sel, r0));
    }
    return QMX_OK;
}

// The score matrix of queries [tile0, tile0 + nq_tile) of the batch against the candidates ids[0..n) (rows 0..n without ids): scores[(qi - tile0) * stride + i].
// One launch per tile_qt queries; the f32 matrix-core kernel takes them all in one launch (scan_mfma.hip: score mode loops over its query tiles).
int32_t score_matrix_enqueue(const qmx_query *q, uint32_t tile0, uint32_t nq_tile, const uint32_t *d_ids, uint64_t n, float *d_scores, uint64_t stride,
                                    uint32_t *launches) {
    const qmx_segment *s = q->seg;
    if (tq_l1(s)) {
        if (launches) *launches += 3 * (uint32_t)((n + 65535) / 65536);
        return tq_l1_scores_device(const_cast<qmx_query *>(q), tile0, nq_tile, d_ids, n, d_scores, stride, nullptr);
    }
    const uint32_t SQT = tile_qt(s, q);
    const bool loops = s->dtype == QMX_DTYPE_F32 && SQT >= 8 && mfma_scan_ok(s);
    const uint32_t step = loops ? nq_tile : SQT;
    for (uint32_t st0 = 0; st0 < nq_tile; st0 += step) {
        const uint32_t nq_sub = std::min<uint32_t>(step, nq_tile - st0);
        ScanArgs pre;
        fill_args(q, tile0 + st0, nq_sub, pre);
        pre.ids = d_ids;
        pre.n_cand = n;
        pre.top = 1;
        pre.scores = d_scores + (size_t)st0 * stride;
        pre.scores_stride = stride;
        uint32_t pgrid = 0;
        QMX_TRY(launch_scan(q, (int)std::min<uint32_t>(pow2_ceil(nq_sub), std::max<uint32_t>(SQT, 8)), SCAN_SCORES, pre, &pgrid));
        if (launches) ++*launches;
    }
    return QMX_OK;
}

// scores[qi * n + i] for every query of the batch
int32_t score_ids_device(qmx_query *q, const uint32_t *d_ids, uint64_t n, float *d_scores, qmx_counters *counters) {
    const qmx_segment *s = q->seg;
    const uint32_t TQ = tile_qt(s, q);
    uint32_t launches = 0;
    QMX_TRY(score_matrix_enqueue(q, 0, q->nq, d_ids, n, d_scores, n, &launches));
    if (counters) {
        counters->kernel_launches += launches;
        counters->vectors_scored += (uint64_t)q->nq * n;
        counters->bytes_read += (uint64_t)((q->nq + TQ - 1) / TQ) * n * s->row_bytes;
    }
    return QMX_OK;
}

int32_t qmx_score_points(qmx_query *q, const uint32_t *ids, uint32_t n, float *scores, qmx_counters *counters) {
    QMX_REQUIRE(q && (n == 0 || (ids && scores)), QMX_ERR_BAD_ARG, "NULL argument");
    QMX_HIP(hipSetDevice(q->seg->device));
    if (counters) memset(counters, 0, sizeof(*counters));
    if (n == 0 || q->nq == 0) return QMX_OK;
    const void *d_ids = nullptr;
    QMX_TRY(stage_in(q, q->ids, ids, (size_t)n * 4, &d_ids));
    const size_t sbytes = (size_t)q->nq * n * sizeof(float);
    float *d_scores = scores;
    const bool out_dev = is_device_ptr(scores);
    if (!out_dev) {
        QMX_TRY(q->scores.reserve(sbytes));
        d_scores = (float *)q->scores.p;
    }
    QMX_TRY(score_ids_device(q, (const uint32_t *)d_ids, n, d_scores, counters));
    if (!out_dev) QMX_HIP(hipMemcpyAsync(scores, d_scores, sbytes, hipMemcpyDeviceToHost, q->stream));
    return check_err_flag(q);
}

int32_t qmx_score_point(qmx_query *q, uint32_t query_index, uint32_t id, float *out) {
    QMX_REQUIRE(q && out, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(query_index < q->nq, QMX_ERR_OUT_OF_BOUNDS, "query index out of range");
    std::vector<float> tmp(q->nq);
    QMX_TRY(qmx_score_points(q, &id, 1, tmp.data(), nullptr));
    *out = tmp[query_index];
    return QMX_OK;
}

// ---------------------------------------------------------------------------------------------
// pair scoring (ragged score_points, rescoring, score_internal)
// ---------------------------------------------------------------------------------------------
int32_t score_pairs_device(qmx_query *q, const PairSel &sel, const uint32_t *d_ids, uint64_t n_items, float *d_scores,
                                  bool timed) {
    const qmx_segment *s = q->seg;
    ScanArgs a;
    fill_args(q, 0, q->nq, a);
    a.ids = d_ids;
    a.n_cand = n_items;
    a.scores = d_scores;
    size_t slot = 0;
    if (timed) QMX_TRY(timing_begin(q, &slot));
    int32_t rc;
    if (tq_l1(s)) {
        rc = tq_l1_scores_device(q, 0, q->nq, d_ids, n_items, d_scores, 0, &sel);
    } else if (s->dtype <= QMX_DTYPE_U8) {
        QMX_REQUIRE(s->fast_layout(), QMX_ERR_NOT_SUPPORTED, "adopted device block is not 16-byte aligned");
        rc = launch_pairs_dense(q->stream, (int)s->dtype, (int)s->distance, a, sel, n_items, s->num_cus);
    } else if (s->dtype == QMX_DTYPE_SQ_U8) {
        rc = launch_pairs_sq(q->stream, (int)s->distance, a, sel, n_items, s->num_cus);
    } else if (s->dtype == QMX_DTYPE_PQ) {
        rc = launch_pairs_pq(q->stream, a, sel, n_items, s->num_cus);
    } else if (s->dtype == QMX_DTYPE_TQ) {
        rc = launch_pairs_tq(q->stream, a, sel, n_items, s->num_cus);
    } else if (s->dtype == QMX_DTYPE_BQ) {
        rc = launch_pairs_bq(q->stream, a, sel, n_items, s->num_cus);
    } else {
        set_error("dtype %u not built yet", s->dtype);
        rc = QMX_ERR_NOT_SUPPORTED;
    }
    if (timed && rc == QMX_OK) QMX_TRY(timing_end(q, slot));
    return rc;
}

int32_t qmx_score_points_ragged(qmx_query *q, const uint32_t *ids, const uint32_t *offsets, float *scores, qmx_counters *counters) {
    QMX_REQUIRE(q && offsets, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_HIP(hipSetDevice(q->device));
    if (counters) memset(counters, 0, sizeof(*counters));
    if (q->nq == 0) return QMX_OK;
    std::vector<uint32_t> off(q->nq + 1);
    QMX_HIP(hipMemcpy(off.data(), offsets, off.size() * 4, hipMemcpyDefault));
    const uint64_t total = off[q->nq];
    if (total == 0) return QMX_OK;
    QMX_REQUIRE(ids && scores, QMX_ERR_BAD_ARG, "NULL argument");
    std::vector<uint32_t> qsel(total);
    for (uint32_t qi = 0; qi < q->nq; ++qi) {
        QMX_REQUIRE(off[qi] <= off[qi + 1] && off[qi + 1] <= total, QMX_ERR_BAD_ARG, "offsets must be non-decreasing");
        for (uint32_t j = off[qi]; j < off[qi + 1]; ++j) qsel[j] = qi;
    }
    const void *d_ids = nullptr;
    QMX_TRY(stage_in(q, q->ids, ids, (size_t)total * 4, &d_ids));
    QMX_TRY(q->misc.reserve((size_t)total * 4));
    QMX_HIP(hipMemcpyAsync(q->misc.p, qsel.data(), (size_t)total * 4, hipMemcpyHostToDevice, q->stream));
    const bool out_dev = is_device_ptr(scores);
    float *d_scores = scores;
    if (!out_dev) {
        QMX_TRY(q->scores.reserve((size_t)total * 4));
        d_scores = (float *)q->scores.p;
    }
    PairSel sel{(const uint32_t *)q->misc.p, 0, nullptr};
    const bool timed = q->timing;
    QMX_TRY(score_pairs_device(q, sel, (const uint32_t *)d_ids, total, d_scores, timed));
    if (!out_dev) QMX_HIP(hipMemcpyAsync(scores, d_scores, (size_t)total * 4, hipMemcpyDeviceToHost, q->stream));
    QMX_TRY(check_err_flag(q));   // synchronises: qsel / staged ids may go away
    if (counters) {
        counters->vectors_scored = total;
        counters->bytes_read = total * q->seg->row_bytes;
        counters->kernel_launches = 1;
        if (timed) { const float before = q->timing_ms; QMX_TRY(timing_fold(q)); counters->kernel_ms = q->timing_ms - before; }
    }
    return QMX_OK;
}

int32_t qmx_rescore(qmx_query *q, const uint32_t *ids, const uint32_t *counts, uint32_t n_per_query, uint32_t top,
                    qmx_scored_point *out, uint32_t *out_counts) {
    QMX_REQUIRE(q && ids && out && out_counts, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(top >= 1 && top <= MAX_TOP, QMX_ERR_NOT_SUPPORTED, "top %u not in 1..%u", top, MAX_TOP);
    QMX_HIP(hipSetDevice(q->device));
    if (q->nq == 0) return QMX_OK;
    if (n_per_query == 0) {
        if (is_device_ptr(out_counts)) QMX_HIP(hipMemsetAsync(out_counts, 0, (size_t)q->nq * 4, q->stream));
        else for (uint32_t i = 0; i < q->nq; ++i) out_counts[i] = 0;
        return QMX_OK;
    }
    const uint64_t total = (uint64_t)q->nq * n_per_query;
    const void *d_ids = nullptr, *d_counts = nullptr;
    QMX_TRY(stage_in(q, q->ids, ids, (size_t)total * 4, &d_ids));
    QMX_TRY(stage_in(q, q->misc, counts, counts ? (size_t)q->nq * 4 : 0, &d_counts));
    QMX_TRY(q->scores.reserve((size_t)total * 4));
    PairSel sel{nullptr, n_per_query, (const uint32_t *)d_counts};
    QMX_TRY(score_pairs_device(q, sel, (const uint32_t *)d_ids, total, (float *)q->scores.p, false));
    const bool out_dev = is_device_ptr(out), cnt_dev = is_device_ptr(out_counts);
    qmx_scored_point *d_out = out;
    uint32_t *d_oc = out_counts;
    if (!out_dev) { QMX_TRY(q->out.reserve((size_t)q->nq * top * sizeof(qmx_scored_point))); d_out = (qmx_scored_point *)q->out.p; }
    if (!cnt_dev) { QMX_TRY(q->counts.reserve((size_t)q->nq * 4)); d_oc = (uint32_t *)q->counts.p; }
    // sort descending, truncate to top (vector_index_search_common.rs:85-88)
    QMX_TRY(launch_sort_scored(q->stream, (const float *)q->scores.p, (const uint32_t *)d_ids, (const uint32_t *)d_counts, n_per_query,
                               q->nq, top, d_out, d_oc));
    if (!out_dev) QMX_TRY(copy_out(q->stream, out, d_out, (size_t)q->nq * top * sizeof(qmx_scored_point)));
    if (!cnt_dev) QMX_TRY(copy_out(q->stream, out_counts, d_oc, (size_t)q->nq * 4));
    return check_err_flag(q);
}

int32_t qmx_score_internal(const qmx_segment *seg, const uint32_t *a_ids, const uint32_t *b_ids, uint32_t n, float *out) {
    QMX_REQUIRE(seg && (n == 0 || (a_ids && b_ids && out)), QMX_ERR_BAD_ARG, "NULL argument");
    if (n == 0) return QMX_OK;
    if (seg->dtype == QMX_DTYPE_PQ || seg->dtype == QMX_DTYPE_TQ) {   // centroid <-> centroid (encoded_vectors_pq.rs:574-618 | TurboQuantizer::score_symmetric); no query involved
        QMX_HIP(hipSetDevice(seg->device));
        DevBuf ba, bb, bo, be;
        int32_t rc = QMX_OK;
        do {
            if ((rc = ba.reserve((size_t)n * 4)) != QMX_OK || (rc = bb.reserve((size_t)n * 4)) != QMX_OK ||
                (rc = bo.reserve((size_t)n * 4)) != QMX_OK || (rc = be.reserve(4)) != QMX_OK) break;
            hipError_t e = hipMemcpy(ba.p, a_ids, (size_t)n * 4, hipMemcpyDefault);
            if (e == hipSuccess) e = hipMemcpy(bb.p, b_ids, (size_t)n * 4, hipMemcpyDefault);
            if (e == hipSuccess) e = hipMemset(be.p, 0, 4);
            if (e != hipSuccess) { rc = hip_status(e, "stage ids", __FILE__, __LINE__); break; }
            if (tq_l1(seg)) {    // score_symmetric's L1 arm (quantization.rs:429-440): both rows dequantised, ONE inverse rotation of the difference, sum |x| over padded_dim
                DevBuf da, db;
                const uint32_t pd = seg->tq_padded_dim;
                const uint64_t B = 32768;
                if ((rc = da.reserve((size_t)std::min<uint64_t>(n, B) * pd * 8)) == QMX_OK) rc = db.reserve((size_t)std::min<uint64_t>(n, B) * pd * 8);
                for (uint64_t r0 = 0; r0 < n && rc == QMX_OK; r0 += B) {
                    const uint32_t cnt = (uint32_t)std::min<uint64_t>(B, n - r0);
                    rc = launch_tq_l1_dequant(nullptr, seg->d_rows, seg->row_stride, seg->d_tq_sf, (const uint32_t *)ba.p + r0, 0, cnt, seg->n, pd, seg->tq_value_bits,
                                              seg->d_tq_shift, seg->d_tq_scale, (double *)da.p, (int *)be.p, nullptr);
                    if (rc == QMX_OK)
                        rc = launch_tq_l1_dequant(nullptr, seg->d_rows, seg->row_stride, seg->d_tq_sf, (const uint32_t *)bb.p + r0, 0, cnt, seg->n, pd, seg->tq_value_bits,
                                                  seg->d_tq_shift, seg->d_tq_scale, (double *)db.p, (int *)be.p, nullptr);
                    if (rc == QMX_OK) rc = launch_tq_l1_diff(nullptr, (double *)da.p, (const double *)db.p, (uint64_t)cnt * pd);
                    if (rc == QMX_OK) rc = launch_tq_rotate_f64(nullptr, (double *)da.p, cnt, tq_rotation_inverse(seg));
                    if (rc == QMX_OK)
                        rc = launch_tq_l1_scores(nullptr, (const double *)da.p, cnt, pd, pd, nullptr, pd, 0, 1, (float *)bo.p, 0, r0, seg->tq_invert ? 1 : 0, nullptr, 0);
                }
                if (rc == QMX_OK && hipDeviceSynchronize() != hipSuccess) rc = QMX_ERR_OTHER;
                da.release(); db.release();
            }
            else if (seg->dtype == QMX_DTYPE_TQ)
            {
                TqEc ec{seg->d_tq_weights, seg->d_tq_xm, seg->tq_weight_scale, seg->tq_mm_const};
                rc = launch_tq_internal(nullptr, seg->d_rows, (uint32_t)seg->row_stride, seg->d_tq_sf, seg->d_tq_l2, seg->tq_code_bytes, seg->tq_value_bits,
                                        seg->tq_invert ? 1 : 0, seg->n, (const uint32_t *)ba.p, (const uint32_t *)bb.p, n, (float *)bo.p, (int *)be.p,
                                        seg->d_tq_weights ? &ec : nullptr);
            }
            else
                rc = launch_pq_internal(nullptr, seg->distance, seg->dim, seg->pq, seg->d_centroids, seg->d_pq_pair, seg->d_rows, seg->row_stride, seg->n,
                                        (const uint32_t *)ba.p, (const uint32_t *)bb.p, n, (float *)bo.p, (int *)be.p);
            if (rc != QMX_OK) break;
            int flag = 0;
            e = hipMemcpy(&flag, be.p, 4, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(out, bo.p, (size_t)n * 4, hipMemcpyDefault);
            if (e != hipSuccess) { rc = hip_status(e, "copy scores", __FILE__, __LINE__); break; }
            if (flag) { set_error("point offset out of range for this segment"); rc = QMX_ERR_OUT_OF_BOUNDS; }
        } while (0);
        ba.release(); bb.release(); bo.release(); be.release();
        return rc;
    }
    // query i = stored point a[i] (FilteredScorer::new_internal), then the diagonal pairs (i, b[i])
    qmx_query *q = nullptr;
    QMX_TRY(qmx_query_create_internal(seg, a_ids, n, &q));
    int32_t rc = QMX_OK;
    do {
        const void *d_ids = nullptr;
        if ((rc = stage_in(q, q->ids, b_ids, (size_t)n * 4, &d_ids)) != QMX_OK) break;
        const bool out_dev = is_device_ptr(out);
        float *d_scores = out;
        if (!out_dev) {
            if ((rc = q->scores.reserve((size_t)n * 4)) != QMX_OK) break;
            d_scores = (float *)q->scores.p;
        }
        PairSel sel{nullptr, 0, nullptr};
        if ((rc = score_pairs_device(q, sel, (const uint32_t *)d_ids, n, d_scores, false)) != QMX_OK) break;
        if (!out_dev) {
            hipError_t e = hipMemcpyAsync(out, d_scores, (size_t)n * 4, hipMemcpyDeviceToHost, q->stream);
            if (e != hipSuccess) { rc = hip_status(e, "copy scores", __FILE__, __LINE__); break; }
        }
        rc = check_err_flag(q);
    } while (0);
    qmx_query_destroy(q);
    return rc;
}

int32_t qmx_score_bytes(qmx_query *q, const void *rows, uint32_t n, uint64_t stride_bytes, float *scores) {
    QMX_REQUIRE(q && (n == 0 || (rows && scores)), QMX_ERR_BAD_ARG, "NULL argument");
    QMX_HIP(hipSetDevice(q->device));
    if (n == 0 || q->nq == 0) return QMX_OK;
    const qmx_segment *s = q->seg;
    // a transient block in the segment's own device layout (aligned rows / SQ split), scored by the scan kernel
    qmx_segment_desc d;
    memset(&d, 0, sizeof(d));
    d.dtype = s->dtype;
    d.distance = s->distance;
    d.dim = s->dim;
    d.flags = s->flags & ~QMX_SEG_DATA_ON_DEVICE;
    d.n = n;
    d.row_stride_bytes = stride_bytes;
    d.data = rows;
    d.device_id = s->device;
    d.sq = &s->sq;
    qmx_pq_params pq = s->pq;
    pq.centroids = s->d_centroids;
    d.pq = &pq;
    qmx_bq_params bq = {s->bq_encoding, 0, nullptr, nullptr};   // the row size follows the encoding; scoring needs no stats
    d.bq = &bq;
    qmx_segment *tmp = nullptr;
    QMX_TRY(qmx_segment_create(&d, &tmp));
    int32_t rc = QMX_OK;
    do {
        const size_t sbytes = (size_t)q->nq * n * sizeof(float);
        const bool out_dev = is_device_ptr(scores);
        float *d_scores = scores;
        if (!out_dev) {
            if ((rc = q->scores.reserve(sbytes)) != QMX_OK) break;
            d_scores = (float *)q->scores.p;
        }
        for (uint32_t tile0 = 0; tile0 < q->nq && rc == QMX_OK; tile0 += MAX_QT) {
            const uint32_t nq_tile = std::min<uint32_t>(MAX_QT, q->nq - tile0);
            ScanArgs a;
            fill_args(q, tile0, nq_tile, a);
            a.rows = tmp->d_rows;
            a.n_rows = n;
            a.row_stride = tmp->row_stride;
            a.row_offsets = tmp->d_row_offsets;
            a.del = tmp->deleted_view();
            a.n_cand = n;
            a.top = 1;
            a.scores = d_scores + (size_t)tile0 * n;
            a.scores_stride = n;
            uint32_t grid = 0;
            rc = launch_scan(q, (int)pow2_ceil(nq_tile), SCAN_SCORES, a, &grid);
        }
        if (rc != QMX_OK) break;
        if (!out_dev) {
            hipError_t e = hipMemcpyAsync(scores, d_scores, sbytes, hipMemcpyDeviceToHost, q->stream);
            if (e != hipSuccess) { rc = hip_status(e, "copy scores", __FILE__, __LINE__); break; }
        }
        rc = check_err_flag(q);
    } while (0);
    (void)hipStreamSynchronize(q->stream);
    qmx_segment_destroy(tmp);
    return rc;
}

}  // extern "C"

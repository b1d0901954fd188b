// api_segment.hip — the C-ABI of include/qdrant_amd.h, segments (upload, derived copies, files), preprocess / casts, quantizer encoders and fits.
// (One of the api_*.hip translation units; what they share: api_internal.hpp.)
#include "api_internal.hpp"

extern "C" {

// ---------------------------------------------------------------------------------------------
// segment
// ---------------------------------------------------------------------------------------------
static void segment_free(qmx_segment *seg) {
    if (seg->owns_rows && seg->d_rows) (void)hipFree(seg->d_rows);
    if (seg->d_point_deleted) (void)hipFree(seg->d_point_deleted);
    if (seg->d_vec_deleted) (void)hipFree(seg->d_vec_deleted);
    if (seg->d_centroids) (void)hipFree(seg->d_centroids);
    if (seg->d_pq_pair) (void)hipFree(seg->d_pq_pair);
    if (seg->d_pq_rot) (void)hipFree(seg->d_pq_rot);
    if (seg->d_rows_split) (void)hipFree(seg->d_rows_split);
    if (seg->d_i8_scale) (void)hipFree(seg->d_i8_scale);
    if (seg->d_i8_stats) (void)hipFree(seg->d_i8_stats);
    if (seg->d_row_offsets) (void)hipFree(seg->d_row_offsets);
    if (seg->d_bq_mean) (void)hipFree(seg->d_bq_mean);
    if (seg->d_bq_stddev) (void)hipFree(seg->d_bq_stddev);
    if (seg->d_sq_bi) (void)hipFree(seg->d_sq_bi);
    if (seg->d_tq_sf) (void)hipFree(seg->d_tq_sf);
    if (seg->d_tq_l2) (void)hipFree(seg->d_tq_l2);
    if (seg->d_tq_xm) (void)hipFree(seg->d_tq_xm);
    if (seg->d_tq_shift) (void)hipFree(seg->d_tq_shift);
    if (seg->d_tq_scale) (void)hipFree(seg->d_tq_scale);
    if (seg->d_tq_weights) (void)hipFree(seg->d_tq_weights);
    if (seg->d_tq_tables) (void)hipFree(seg->d_tq_tables);
    if (seg->d_tq_l1) (void)hipFree(seg->d_tq_l1);
    if (seg->d_tq_norms) (void)hipFree(seg->d_tq_norms);
    delete seg;
}

static int32_t segment_upload(qmx_segment *s, const qmx_segment_desc *desc) {
    const uint64_t src_stride = desc->row_stride_bytes ? desc->row_stride_bytes : s->row_bytes;
    QMX_REQUIRE(src_stride >= s->row_bytes, QMX_ERR_BAD_ARG, "row_stride_bytes %llu < row size %llu",
                (unsigned long long)src_stride, (unsigned long long)s->row_bytes);
    const bool on_device = (desc->flags & QMX_SEG_DATA_ON_DEVICE) != 0;
    if (s->dtype == QMX_DTYPE_SQ_U8) {
        // split [f32 offset][codes] rows into a 16-byte aligned code block + an offset column
        const uint32_t ad = s->sq.actual_dim;
        s->row_stride = ad;
        QMX_HIP(hipMalloc(&s->d_rows, (size_t)std::max<uint64_t>(1, s->n) * ad));
        s->owns_rows = true;
        QMX_HIP(hipMalloc((void **)&s->d_row_offsets, (size_t)std::max<uint64_t>(1, s->n) * sizeof(float)));
        if (s->n == 0) return QMX_OK;
        const void *d_src = desc->data;
        DevBuf tmp;
        if (!on_device && !is_device_ptr(desc->data)) {
            QMX_TRY(tmp.reserve((size_t)s->n * src_stride));
            hipError_t e = hipMemcpy(tmp.p, desc->data, (size_t)(s->n - 1) * src_stride + s->row_bytes, hipMemcpyHostToDevice);
            if (e != hipSuccess) { tmp.release(); return hip_status(e, "hipMemcpy(SQ rows)", __FILE__, __LINE__); }
            d_src = tmp.p;
        }
        int32_t rc = launch_sq_split(nullptr, d_src, src_stride, s->n, ad, s->d_rows, s->d_row_offsets);
        if (rc == QMX_OK && hipDeviceSynchronize() != hipSuccess) rc = QMX_ERR_OTHER;
        tmp.release();
        return rc;
    }
    if (s->dtype == QMX_DTYPE_TQ) {
        // split [codes][scaling_factor][l2_length] rows into a 16-byte aligned, zero padded code block + the extras columns
        const bool has_l2 = s->distance == QMX_DISTANCE_EUCLID;
        s->row_stride = s->scan_dim;
        QMX_HIP(hipMalloc(&s->d_rows, (size_t)std::max<uint64_t>(1, s->n) * s->row_stride));
        s->owns_rows = true;
        QMX_HIP(hipMalloc((void **)&s->d_tq_sf, (size_t)std::max<uint64_t>(1, s->n) * sizeof(float)));
        if (has_l2) QMX_HIP(hipMalloc((void **)&s->d_tq_l2, (size_t)std::max<uint64_t>(1, s->n) * sizeof(float)));
        if (s->d_tq_shift) QMX_HIP(hipMalloc((void **)&s->d_tq_xm, (size_t)std::max<uint64_t>(1, s->n) * sizeof(float)));
        if (s->n == 0) return QMX_OK;
        const void *d_src = desc->data;
        DevBuf tmp;
        if (!on_device && !is_device_ptr(desc->data)) {
            QMX_TRY(tmp.reserve((size_t)s->n * src_stride));
            hipError_t e = hipMemcpy(tmp.p, desc->data, (size_t)(s->n - 1) * src_stride + s->row_bytes, hipMemcpyHostToDevice);
            if (e != hipSuccess) { tmp.release(); return hip_status(e, "hipMemcpy(TQ rows)", __FILE__, __LINE__); }
            d_src = tmp.p;
        }
        int32_t rc = launch_tq_split(nullptr, d_src, src_stride, s->n, s->tq_code_bytes, (uint32_t)s->row_stride, has_l2 ? 1 : 0, s->d_rows, s->d_tq_sf,
                                     s->d_tq_l2, s->d_tq_xm);
        if (rc == QMX_OK && hipDeviceSynchronize() != hipSuccess) rc = QMX_ERR_OTHER;
        tmp.release();
        return rc;
    }
    if (on_device) {
        s->d_rows = const_cast<void *>(desc->data);
        s->row_stride = src_stride;
        s->owns_rows = false;
        return QMX_OK;
    }
    // rows are re-packed at a 16-byte multiple so the 16-B lane loads stay aligned
    s->row_stride = (s->row_bytes + 15) & ~15ull;
    const size_t bytes = (size_t)std::max<uint64_t>(1, s->n) * s->row_stride;
    QMX_HIP(hipMalloc(&s->d_rows, bytes));
    s->owns_rows = true;
    if (s->n) {
        if (s->row_stride != s->row_bytes) QMX_HIP(hipMemset(s->d_rows, 0, bytes));
        QMX_HIP(hipMemcpy2D(s->d_rows, s->row_stride, desc->data, src_stride, s->row_bytes, s->n, hipMemcpyDefault));
    }
    return QMX_OK;
}

// PQ blocks large enough for the 6-bit prefilter (pq_prefilter.hip): the rotated copy of the codes, m_pad bytes per row next to the m of the block
// (10 M x 96: 0.96 GB, one pass).  Out of memory is not an error: the exact kernel serves.
// The rotated copy of a PQ block's codes that the 8-bit prefilter scans (pq_prefilter.hip): ceil32(m) bytes per row beside the m-byte codes.  Built where
// it pays and costs little: blocks of 2^18 rows and more, at most twice the codes' own size (m >= 16: an m = 8 block would grow five-fold for it; opt in
// with QMX_SEG_PQ_PREFILTER_COPY).  Out of memory or a failed pass is not an error: the exact kernel serves every batch size.
static int32_t segment_pq_rot(qmx_segment *s) {
    if (s->dtype != QMX_DTYPE_PQ || s->n < (1u << 18) || !pq_prefilter_shape_ok(s->pq_m, s->pq.n_centroids) || option(OPT_NO_PQ_PREFILTER)) return QMX_OK;
    const uint32_t m_pad = (s->pq_m + 31u) & ~31u;
    if (m_pad > 2 * s->pq_m && !(s->flags & QMX_SEG_PQ_PREFILTER_COPY)) return QMX_OK;
    if (hipMalloc(&s->d_pq_rot, pq_rot_bytes(s->n, s->pq_m)) != hipSuccess) {
        (void)hipGetLastError();
        s->d_pq_rot = nullptr;
        return QMX_OK;
    }
    if (launch_pq_rotate(nullptr, s->d_rows, s->row_stride, s->n, s->pq_m, s->d_pq_rot) != QMX_OK || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        ::qmx::clear_stale_error();
        (void)hipFree(s->d_pq_rot);
        s->d_pq_rot = nullptr;
    }
    return QMX_OK;
}

// one pass over an f32 dot / cosine block that the split prefilter may serve: the power-of-two scale of its rows and the norm bound of
// the verification band (4.5 ms per 30 GB; nothing for other storages)
// ---- derived copies of an f32 dot / cosine block (scan_split.hip): what the prefilters stream instead of the f32 rows ----
static void segment_drop_copy(qmx_segment *s) {
    if (s->d_rows_split) (void)hipFree(s->d_rows_split);
    if (s->d_i8_scale) (void)hipFree(s->d_i8_scale);
    if (s->d_i8_stats) (void)hipFree(s->d_i8_stats);
    s->d_rows_split = nullptr;
    s->d_i8_scale = nullptr;
    s->d_i8_stats = nullptr;
    s->split_i8 = false;
    s->split_half = false;
    s->copy_bytes = 0;
    (void)hipGetLastError();
}
static bool segment_i8_eligible(const qmx_segment *s) { return s->split_stats && split_i8_dim_ok(s->dim) && split_fallback_qt(s->dim) != 0; }    // (dims the prefilter path serves: search_enqueue)
static bool segment_f16_eligible(const qmx_segment *s) { return s->split_stats && s->dim % 128 == 0 && split_fallback_qt(s->dim) != 0; }
// the int8 copy: column maxima / sums of squares (one pass), the scales (host: split_i8_choose_scales), the worst row's code norms under them (a second
// pass), the codes (a third).  false: out of memory, or an element that is not finite - no copy is left behind
static bool segment_build_i8(qmx_segment *s) {
    uint32_t *d_colmax = nullptr;
    float *d_colsq = nullptr;
    uint32_t h[4] = {0, 0, 1, 0};
    std::vector<float> colmax(s->dim), colsq(s->dim), scale(s->dim);
    bool ok = hipMalloc((void **)&d_colmax, (size_t)s->dim * 4) == hipSuccess && hipMalloc((void **)&d_colsq, (size_t)s->dim * 4) == hipSuccess &&
              hipMalloc((void **)&s->d_i8_scale, (size_t)s->dim * 4) == hipSuccess && hipMalloc((void **)&s->d_i8_stats, 16) == hipSuccess;
    if (ok) ok = launch_split_i8_colstats(nullptr, s->d_rows, s->row_stride, s->n, s->dim, d_colmax, d_colsq) == QMX_OK &&
                 hipMemcpy(colmax.data(), d_colmax, (size_t)s->dim * 4, hipMemcpyDeviceToHost) == hipSuccess &&
                 hipMemcpy(colsq.data(), d_colsq, (size_t)s->dim * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if (ok) {
        s->i8_balance = split_i8_choose_scales(colmax.data(), colsq.data(), s->n, s->dim, scale.data());
        ok = hipMemcpy(s->d_i8_scale, scale.data(), (size_t)s->dim * 4, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok) ok = launch_split_i8_rowstats(nullptr, s->d_rows, s->row_stride, s->n, s->dim, s->d_i8_scale, s->d_i8_stats) == QMX_OK &&
                 hipMemcpy(h, s->d_i8_stats, 16, hipMemcpyDeviceToHost) == hipSuccess && h[2] == 0;
    if (ok) ok = hipMalloc(&s->d_rows_split, split_i8_copy_bytes(s->n, s->dim)) == hipSuccess;
    if (ok) ok = launch_split_i8_copy(nullptr, s->d_rows, s->row_stride, s->n, s->dim, s->d_i8_scale, s->d_rows_split) == QMX_OK &&
                 hipDeviceSynchronize() == hipSuccess;
    if (d_colmax) (void)hipFree(d_colmax);
    if (d_colsq) (void)hipFree(d_colsq);
    (void)hipGetLastError();
    if (!ok) {
        segment_drop_copy(s);
        return false;
    }
    s->split_i8 = true;
    s->copy_bytes = split_i8_copy_bytes(s->n, s->dim);
    return true;
}
// the f16 copies (one pass: read 4 B, write 4 or 2 B per element).  Out of memory is not an error: the converting kernel serves
static int32_t segment_build_f16(qmx_segment *s, bool half) {
    s->split_half = half;
    if (hipMalloc(&s->d_rows_split, split_copy_bytes(s->n, s->dim, half)) != hipSuccess) {
        (void)hipGetLastError();
        s->d_rows_split = nullptr;
        s->split_half = false;
        return QMX_OK;
    }
    QMX_TRY(launch_split_copy(nullptr, s->d_rows, s->row_stride, s->n, s->dim, split_row_scale(s->row_maxabs), s->d_rows_split, half));
    QMX_HIP(hipDeviceSynchronize());
    s->copy_bytes = split_copy_bytes(s->n, s->dim, half);
    return QMX_OK;
}

// QMX_SEG_AUTO_COPY: which copy serves THIS block is measured, not guessed.  The int8 copy halves the half copy's bytes per query but its band is a
// worst-case bound that scales with sum_i |q_i| max_r |x_ri|: on rows with heavy-tailed elements more rows fall inside it than the verification is
// worth (or than its lists take: the query then pays the prefilter AND the exact scan).  So: build the int8 copy, search 128 stored rows (a strided
// sample of the block: queries distributed like the rows) for their 10 nearest through it, read the counters; a block whose queries verify few rows
// keeps it without further ado, any other gets the half copy built beside it, the same batch is timed through both, and the faster one stays.
constexpr uint32_t AUTO_TRIAL_QUERIES = 128, AUTO_TRIAL_TOP = 10, AUTO_EASY_VERIFIED = 1024;
static int32_t auto_trial(qmx_segment *s, const float *d_trial_queries, float *ms_out, qmx_counters *c_out) {
    qmx_query *q = nullptr;
    QMX_TRY(qmx_query_create(s, d_trial_queries, AUTO_TRIAL_QUERIES, &q));
    std::vector<qmx_scored_point> out((size_t)AUTO_TRIAL_QUERIES * AUTO_TRIAL_TOP);
    std::vector<uint32_t> counts(AUTO_TRIAL_QUERIES);
    int32_t rc = QMX_OK;
    float best = 3.0e38f;
    for (int rep = 0; rep < 3 && rc == QMX_OK; ++rep) {            // (the first run pays the scratch allocations: the best of three is the step)
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = QMX_ERR_OTHER;
        if (rc == QMX_OK && hipEventRecord(e0, q->stream) != hipSuccess) rc = QMX_ERR_OTHER;
        if (rc == QMX_OK) rc = qmx_search_topk(q, AUTO_TRIAL_TOP, nullptr, 0, out.data(), counts.data(), nullptr, c_out);
        if (rc == QMX_OK && (hipEventRecord(e1, q->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) rc = QMX_ERR_OTHER;
        float ms = 0.0f;
        if (rc == QMX_OK && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms < best) best = ms;
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
    if (rc == QMX_OK) rc = qmx_query_last_counters(q, c_out);
    qmx_query_destroy(q);
    *ms_out = (rc == QMX_OK && best < 3.0e38f) ? best : 0.0f;          // (a trial that failed measured nothing: 0 in qmx_segment_get_info, not a sentinel)
    return rc;
}
static int32_t segment_auto_copy(qmx_segment *s) {
    s->auto_choice = true;
    if (!segment_i8_eligible(s) || !segment_build_i8(s)) {
        if (segment_f16_eligible(s)) QMX_TRY(segment_build_f16(s, true));
        return QMX_OK;
    }
    // the trial batch: rows n / 256, 3 n / 256, ... (stored rows are preprocessed: the query path normalises them again - a no-op up to round-off)
    float *d_tq = nullptr;
    if (hipMalloc((void **)&d_tq, (size_t)AUTO_TRIAL_QUERIES * s->dim * 4) != hipSuccess) {
        (void)hipGetLastError();
        return QMX_OK;                                                  // (no room for a trial: the int8 copy stays, its fallback is exact whatever happens)
    }
    const uint64_t step = s->n / AUTO_TRIAL_QUERIES;
    bool ok = true;
    for (uint32_t i = 0; i < AUTO_TRIAL_QUERIES && ok; ++i)
        ok = hipMemcpyAsync(d_tq + (size_t)i * s->dim, (const unsigned char *)s->d_rows + (step * i + step / 2) * s->row_stride, (size_t)s->dim * 4,
                            hipMemcpyDeviceToDevice, nullptr) == hipSuccess;
    ok = ok && hipDeviceSynchronize() == hipSuccess;
    qmx_counters c_i8{}, c_half{};
    int32_t rc = ok ? auto_trial(s, d_tq, &s->auto_i8_ms, &c_i8) : QMX_ERR_OTHER;
    if (rc == QMX_OK) {
        s->auto_i8_verified = (float)c_i8.verified_rows / (float)AUTO_TRIAL_QUERIES;
        s->auto_i8_fallback = c_i8.fallback_queries;
        const bool easy = c_i8.fallback_queries == 0 && c_i8.verified_rows <= (uint64_t)AUTO_EASY_VERIFIED * AUTO_TRIAL_QUERIES;
        if (!easy && segment_f16_eligible(s)) {
            // the half copy beside it: park the int8 copy, build, time, keep the faster
            void *i8_rows = s->d_rows_split;
            float *i8_scale = s->d_i8_scale;
            uint32_t *i8_stats = s->d_i8_stats;
            const uint64_t i8_bytes = s->copy_bytes;
            s->d_rows_split = nullptr; s->d_i8_scale = nullptr; s->d_i8_stats = nullptr; s->split_i8 = false;
            rc = segment_build_f16(s, true);
            if (rc == QMX_OK && s->d_rows_split) rc = auto_trial(s, d_tq, &s->auto_half_ms, &c_half);
            // The choice is by wall-clock time, so work sharing the device during the trial can tip it: the half copy (twice the bytes of the int8 copy in
            // HBM) has to win by a margin - 10 % - before it replaces the int8 copy.  Results are exact either way; only footprint and latency differ.
            const bool half_wins = rc == QMX_OK && s->d_rows_split && s->auto_half_ms > 0.0f && s->auto_i8_ms > 0.0f && s->auto_half_ms < 0.9f * s->auto_i8_ms;
            if (half_wins) {
                (void)hipFree(i8_rows); (void)hipFree(i8_scale); (void)hipFree(i8_stats);
            } else {
                if (s->d_rows_split) (void)hipFree(s->d_rows_split);
                s->d_rows_split = i8_rows; s->d_i8_scale = i8_scale; s->d_i8_stats = i8_stats; s->split_i8 = true; s->split_half = false;
                s->copy_bytes = i8_bytes;
                if (rc != QMX_OK) { rc = QMX_OK; ::qmx::clear_stale_error(); }    // (the half copy could not be tried: the int8 copy serves)
            }
        }
    } else {
        rc = QMX_OK;                                                    // (a trial that could not run decides nothing: the int8 copy stays)
        ::qmx::clear_stale_error();
    }
    (void)hipFree(d_tq);
    (void)hipGetLastError();
    return rc;
}

static int32_t segment_split_stats(qmx_segment *s) {
    if (s->dtype != QMX_DTYPE_F32 || (s->distance != QMX_DISTANCE_DOT && s->distance != QMX_DISTANCE_COSINE) || s->dim % 32 != 0 ||
        s->n < (1u << 18) || !s->fast_layout())
        return QMX_OK;
    uint32_t *d_stats = nullptr;
    QMX_HIP(hipMalloc((void **)&d_stats, 8));
    int32_t rc = QMX_OK;
    uint32_t h[2] = {0, 0};
    if (hipMemset(d_stats, 0, 8) != hipSuccess) rc = QMX_ERR_OTHER;
    if (rc == QMX_OK) rc = launch_split_row_stats(nullptr, s->d_rows, s->row_stride, s->n, s->dim, d_stats);
    if (rc == QMX_OK && hipMemcpy(h, d_stats, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = QMX_ERR_OTHER;
    (void)hipFree(d_stats);
    if (rc != QMX_OK) return rc;
    memcpy(&s->row_maxabs, &h[0], 4);
    float mss;
    memcpy(&mss, &h[1], 4);
    s->row_norm_max = sqrtf(mss);
    s->split_stats = s->row_maxabs > 0.f && s->row_maxabs < 3.0e38f && s->row_norm_max < 3.0e38f;   // (NaN / inf rows: the exact scan only)
    if (!s->split_stats) return QMX_OK;
    if (s->flags & QMX_SEG_AUTO_COPY) return segment_auto_copy(s);
    // an explicit flag: that copy; where it cannot be built (dims, memory, an element that is not finite) the other flags (if any) apply
    if ((s->flags & QMX_SEG_I8_COPY) && segment_i8_eligible(s) && segment_build_i8(s)) return QMX_OK;
    if ((s->flags & (QMX_SEG_SPLIT_COPY | QMX_SEG_HALF_COPY)) && segment_f16_eligible(s)) return segment_build_f16(s, (s->flags & QMX_SEG_HALF_COPY) != 0);
    return QMX_OK;
}

// SQ blocks large enough for the 128-query pass (scan_sqw.hip): the largest vector_offset, which its integer reject bound rests on
static int32_t segment_sq_stats(qmx_segment *s) {
    if (s->dtype != QMX_DTYPE_SQ_U8 || s->n < (1u << 18) || !s->d_row_offsets || !sq_mfma_ok(s->distance, s->scan_dim) || !(s->sq.multiplier > 0.f) ||
        s->scan_dim % 64 != 0)
        return QMX_OK;
    if (hipMalloc((void **)&s->d_sq_bi, (size_t)s->n * sizeof(int32_t)) != hipSuccess) {      // (no memory for the column: the 32-query kernel serves)
        (void)hipGetLastError();
        s->d_sq_bi = nullptr;
        return QMX_OK;
    }
    uint32_t *d_stats = nullptr;
    QMX_HIP(hipMalloc((void **)&d_stats, 8));
    int32_t rc = QMX_OK;
    uint32_t h[2] = {0u, 0u};
    if (hipMemset(d_stats, 0, 8) != hipSuccess) rc = QMX_ERR_OTHER;
    if (rc == QMX_OK) rc = launch_sqw_stats(nullptr, s->d_row_offsets, s->n, s->sq.multiplier, s->d_sq_bi, d_stats);
    if (rc == QMX_OK && hipMemcpy(h, d_stats, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = QMX_ERR_OTHER;
    (void)hipFree(d_stats);
    if (rc != QMX_OK) return rc;
    memcpy(&s->sq_off_absmax, &h[0], 4);
    s->sq_wide = h[1] == 0 && s->sq_off_absmax < 3.0e38f;
    return QMX_OK;
}

// 4-bit TurboQuant blocks large enough for the 128-query pass (scan_tq4w.hip): the ranges of the extras columns its integer reject bound rests on
static int32_t segment_tq_stats(qmx_segment *s) {
    if (s->dtype != QMX_DTYPE_TQ || s->tq_value_bits != 4 || s->n < (1u << 18) || s->d_tq_l1 || !s->d_tq_sf) return QMX_OK;
    uint32_t *d_stats = nullptr;
    QMX_HIP(hipMalloc((void **)&d_stats, 32));
    int32_t rc = QMX_OK;
    uint32_t h[8] = {0x7F800000u, 0u, 0x7F800000u, 0u, 0u, 0u, 0u, 0u};
    if (hipMemcpy(d_stats, h, 32, hipMemcpyHostToDevice) != hipSuccess) rc = QMX_ERR_OTHER;
    if (rc == QMX_OK) rc = launch_tq4w_stats(nullptr, s->d_tq_sf, s->d_tq_l2, s->d_rows, s->row_stride, (uint32_t)s->row_stride, s->n, d_stats);
    if (rc == QMX_OK && hipMemcpy(h, d_stats, 32, hipMemcpyDeviceToHost) != hipSuccess) rc = QMX_ERR_OTHER;
    (void)hipFree(d_stats);
    if (rc != QMX_OK) return rc;
    memcpy(&s->tq_sf_min, &h[0], 4);
    memcpy(&s->tq_sf_max, &h[1], 4);
    memcpy(&s->tq_l2_min, &h[2], 4);
    memcpy(&s->tq_l2_max, &h[5], 4);
    if (!s->d_tq_l2) s->tq_l2_min = s->tq_l2_max = 0.f;
    s->tq_c1 = h[4];
    s->tq_wide = h[3] == 0 && s->tq_sf_min > 0.f && s->tq_sf_max < 3.0e38f && s->tq_sf_min <= s->tq_sf_max;
    return QMX_OK;
}

// TurboQuantizer::new (turboquant/quantization.rs:127-158): padded dim (encoding.rs:194-201), the rotation's three permutation maps
// (rotation.rs:4-10,32-63 over permutation.rs: Fisher-Yates driven by Knuth's MMIX LCG, upper 32 bits mod bound) and its chunk decomposition
// (rotation.rs:222-233,264-280: decreasing powers of two, each WHT normalised by 1 / sqrt(size))
static int32_t tq_segment_setup(qmx_segment *s, const qmx_segment_desc *desc) {
    QMX_REQUIRE(desc->tq, QMX_ERR_BAD_ARG, "TQ segment needs qmx_tq_params");
    const qmx_tq_params &t = *desc->tq;
    QMX_REQUIRE(t.bits <= QMX_TQ_BITS1, QMX_ERR_BAD_ARG, "bad TQBits %u", t.bits);
    QMX_REQUIRE(!t.plus_mode || (t.ec_shift && t.ec_scale), QMX_ERR_BAD_ARG, "TQMode::Plus needs the storage's error correction (ec_shift / ec_scale)");
    QMX_REQUIRE(!t.plus_mode || (!is_device_ptr(t.ec_shift) && !is_device_ptr(t.ec_scale)), QMX_ERR_BAD_ARG, "ec_shift / ec_scale are host arrays");
    QMX_REQUIRE(!(t.bits == QMX_TQ_BITS1_5 && t.rotation_unpadded), QMX_ERR_BAD_ARG, "Bits1_5 requires TQRotation::Padded");
    auto next_multiple = [](uint64_t x, uint64_t m) { return (x + m - 1) / m * m; };
    const uint64_t dim = desc->dim;
    uint64_t padded = 0;
    switch (t.bits) {
        case QMX_TQ_BITS1: padded = next_multiple(dim, 8); s->tq_value_bits = 1; break;
        case QMX_TQ_BITS1_5: padded = next_multiple(dim * 3 / 2, 8); s->tq_value_bits = 1; break;
        case QMX_TQ_BITS2: padded = next_multiple(dim, 4); s->tq_value_bits = 2; break;
        default: padded = next_multiple(dim, 2); s->tq_value_bits = 4; break;
    }
    QMX_REQUIRE(padded <= 8192, QMX_ERR_NOT_SUPPORTED, "TurboQuant: padded dim %llu > 8192 (the rotation runs in LDS)", (unsigned long long)padded);
    s->tq_bits = t.bits;
    s->tq_invert = t.invert != 0;
    s->tq_padded_dim = (uint32_t)padded;
    s->tq_rot_dim = t.rotation_unpadded ? (uint32_t)dim : (uint32_t)padded;
    s->tq_code_bytes = (uint32_t)(padded * s->tq_value_bits / 8);
    s->row_bytes = s->tq_code_bytes + (desc->distance == QMX_DISTANCE_EUCLID ? 8 : 4) + (t.plus_mode ? 4 : 0);
    s->scan_dim = (s->tq_code_bytes + 15) & ~15u;        // bytes of a row of the device code block
    if (t.plus_mode) {   // ErrorCorrection::new (turboquant/quantization.rs:49-96): D'^2 as i16 weights, their scale, <M, M>
        const uint32_t pd = s->tq_padded_dim;
        std::vector<float> dps(pd);
        float mm = 0.0f, max_dps = 0.0f;
        for (uint32_t i = 0; i < pd; ++i) {
            mm += t.ec_shift[i] * t.ec_shift[i];
            const float sc = t.ec_scale[i];
            dps[i] = std::fabs(sc) > 1.1920929e-7f ? 1.0f / (sc * sc) : 0.0f;
            max_dps = std::max(max_dps, dps[i]);
        }
        const float QUANT_CAP = 32766.0f;
        s->tq_mm_const = mm;
        s->tq_weight_scale = max_dps > 1.1920929e-7f ? QUANT_CAP / max_dps : 1.0f;
        std::vector<int16_t> w(pd);
        for (uint32_t i = 0; i < pd; ++i) w[i] = (int16_t)std::min(std::max(std::round(dps[i] * s->tq_weight_scale), 0.0f), QUANT_CAP);
        QMX_HIP(hipMalloc((void **)&s->d_tq_shift, (size_t)pd * 4));
        QMX_HIP(hipMalloc((void **)&s->d_tq_scale, (size_t)pd * 4));
        QMX_HIP(hipMalloc((void **)&s->d_tq_weights, (size_t)pd * 2));
        QMX_HIP(hipMemcpy(s->d_tq_shift, t.ec_shift, (size_t)pd * 4, hipMemcpyHostToDevice));
        QMX_HIP(hipMemcpy(s->d_tq_scale, t.ec_scale, (size_t)pd * 4, hipMemcpyHostToDevice));
        QMX_HIP(hipMemcpy(s->d_tq_weights, w.data(), (size_t)pd * 2, hipMemcpyHostToDevice));
    }
    // the rotation tables
    const uint32_t rd = s->tq_rot_dim;
    static const uint64_t SEEDS[3] = {654605292835415893ull, 8636605637963351413ull, 1775280196666917949ull};
    std::vector<uint32_t> tables((size_t)6 * rd + 64);    // forward maps, chunk offsets / sizes, backward maps (last permutation first: apply_inverse's order)
    for (int p = 0; p < 3; ++p) {
        uint32_t *map = tables.data() + (size_t)p * rd;
        for (uint32_t i = 0; i < rd; ++i) map[i] = i;
        uint64_t state = SEEDS[p];
        for (uint32_t i = rd; i-- > 1;) {
            state = state * 6364136223846793005ull + 1442695040888963407ull;
            const uint32_t j = (uint32_t)((state >> 32) % ((uint64_t)i + 1));
            std::swap(map[i], map[j]);
        }
    }
    for (int p = 0; p < 3; ++p) {       // backward_maps[p][forward_maps[p][k]] = k (rotation.rs:47-53)
        const uint32_t *fwd = tables.data() + (size_t)p * rd;
        uint32_t *inv = tables.data() + (size_t)3 * rd + 64 + (size_t)(2 - p) * rd;
        for (uint32_t k = 0; k < rd; ++k) inv[fwd[k]] = k;
    }
    std::vector<double> norms;
    uint32_t nchunks = 0, off = 0;
    for (uint32_t rest = rd; rest;) {
        const uint32_t size = 1u << (31 - __builtin_clz(rest));
        rest ^= size;
        tables[(size_t)3 * rd + nchunks] = off;
        tables[(size_t)3 * rd + 32 + nchunks] = size;
        norms.push_back(1.0 / std::sqrt((double)size));
        off += size;
        ++nchunks;
    }
    s->tq_n_chunks = nchunks;
    QMX_HIP(hipMalloc((void **)&s->d_tq_tables, tables.size() * 4));
    QMX_HIP(hipMemcpy(s->d_tq_tables, tables.data(), tables.size() * 4, hipMemcpyHostToDevice));
    QMX_HIP(hipMalloc((void **)&s->d_tq_norms, std::max<size_t>(1, norms.size()) * 8));
    if (!norms.empty()) QMX_HIP(hipMemcpy(s->d_tq_norms, norms.data(), norms.size() * 8, hipMemcpyHostToDevice));
    if (desc->distance == QMX_DISTANCE_MANHATTAN) {   // what the L1 walk reads (tq_l1_policy.hpp)
        TqL1Dev d;
        memset(&d, 0, sizeof(d));
        d.inv.maps = s->d_tq_tables + (size_t)3 * rd + 64;
        d.inv.chunk_off = s->d_tq_tables + (size_t)3 * rd;
        d.inv.chunk_size = d.inv.chunk_off + 32;
        d.inv.chunk_norm = s->d_tq_norms;
        d.inv.n_chunks = nchunks; d.inv.rot_dim = rd; d.inv.padded_dim = s->tq_padded_dim; d.inv.dim = s->tq_padded_dim;
        d.shift = s->d_tq_shift; d.scale = s->d_tq_scale;
        d.value_bits = s->tq_value_bits; d.dim = (uint32_t)dim;
        QMX_HIP(hipMalloc(&s->d_tq_l1, sizeof(d)));
        QMX_HIP(hipMemcpy(s->d_tq_l1, &d, sizeof(d), hipMemcpyHostToDevice));
    }
    return QMX_OK;
}
TqRotationHost tq_rotation(const qmx_segment *s) {
    TqRotationHost h;
    h.d_maps = s->d_tq_tables;
    h.d_chunk_off = s->d_tq_tables + (size_t)3 * s->tq_rot_dim;
    h.d_chunk_size = h.d_chunk_off + 32;
    h.d_chunk_norm = s->d_tq_norms;
    h.n_chunks = s->tq_n_chunks; h.rot_dim = s->tq_rot_dim; h.padded_dim = s->tq_padded_dim; h.dim = s->dim;
    return h;
}

// HadamardRotation::apply_inverse: the same rounds over the backward maps
TqRotationHost tq_rotation_inverse(const qmx_segment *s) {
    TqRotationHost h = tq_rotation(s);
    h.d_maps = s->d_tq_tables + (size_t)3 * s->tq_rot_dim + 64;
    return h;
}

// turboquant/math.rs:3-15 (Abramowitz & Stegun 7.1.26)
static double tq_std_normal_cdf(double x) {
    const double y = x / 1.4142135623730951;
    const double a = fabs(y);
    const double t = 1.0 / (1.0 + 0.3275911 * a);
    const double poly = t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429))));
    const double r = 1.0 - poly * exp(-a * a);
    return 0.5 * (1.0 + (y >= 0.0 ? r : -r));
}

int32_t qmx_tq_fit_plus(int32_t device_id, uint32_t distance, uint32_t dim, const qmx_tq_params *params, const float *sample, uint64_t n_sample,
                        float *shift_out, float *scale_out) {
    QMX_REQUIRE(params && shift_out && scale_out && (n_sample == 0 || sample) && dim > 0, QMX_ERR_BAD_ARG, "bad argument");
    QMX_REQUIRE(distance <= QMX_DISTANCE_MANHATTAN, QMX_ERR_BAD_ARG, "bad distance %u", distance);
    QMX_REQUIRE(n_sample <= (1u << 20), QMX_ERR_BAD_ARG, "sample of %llu vectors (the reference takes 2 048 .. 8 192)", (unsigned long long)n_sample);
    QMX_TRY(check_device(device_id, nullptr));
    qmx_tq_params pre = *params;           // the pre-quantizer of the stats pass: TQMode::Normal, no error correction (:159-165)
    pre.plus_mode = 0; pre.ec_shift = nullptr; pre.ec_scale = nullptr;
    qmx_segment tmp;
    qmx_segment_desc d;
    memset(&d, 0, sizeof(d));
    d.dtype = QMX_DTYPE_TQ; d.distance = distance; d.dim = dim; d.tq = &pre; d.device_id = device_id;
    tmp.device = device_id; tmp.dtype = QMX_DTYPE_TQ; tmp.distance = distance; tmp.dim = dim;
    int32_t rc = tq_segment_setup(&tmp, &d);
    DevBuf bin, brot, bsh, bsc;
    do {
        if (rc != QMX_OK) break;
        const uint32_t pd = tmp.tq_padded_dim, n = (uint32_t)n_sample;
        // the outermost centroid and the two quantiles Phi(-+c_outer) (:172-184, quantile.rs:155-156)
        const float c_outer = tmp.tq_value_bits == 4 ? 2.733f : tmp.tq_value_bits == 2 ? 1.510f : 0.7978846f;
        const double p_outer = tq_std_normal_cdf((double)c_outer);
        float qp = (float)(2.0 * p_outer - 1.0);
        qp = qp < 0.0f ? 0.0f : qp > 0.99999f ? 0.99999f : qp;
        const double min_q = (1.0 - (double)qp) / 2.0, max_q = 1.0 - min_q;
        if ((rc = brot.reserve((size_t)std::max<uint32_t>(n, 1) * pd * 8)) != QMX_OK) break;
        if ((rc = bsh.reserve((size_t)pd * 4)) != QMX_OK || (rc = bsc.reserve((size_t)pd * 4)) != QMX_OK) break;
        const float *d_in = sample;
        if (n && !is_device_ptr(sample)) {
            if ((rc = bin.reserve((size_t)n * dim * 4)) != QMX_OK) break;
            if (hipMemcpy(bin.p, sample, (size_t)n * dim * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
            d_in = (const float *)bin.p;
        }
        if (n && (rc = launch_tq_rotate(nullptr, d_in, n, tq_rotation(&tmp), (double *)brot.p)) != QMX_OK) break;
        if ((rc = launch_tq_plus_fit(nullptr, (double *)brot.p, n, pd, distance, min_q, max_q, c_outer, (float *)bsh.p, (float *)bsc.p)) != QMX_OK) break;
        if (hipDeviceSynchronize() != hipSuccess) { rc = QMX_ERR_OTHER; break; }
        if (hipMemcpy(shift_out, bsh.p, (size_t)pd * 4, hipMemcpyDefault) != hipSuccess || hipMemcpy(scale_out, bsc.p, (size_t)pd * 4, hipMemcpyDefault) != hipSuccess) {
            rc = QMX_ERR_OTHER;
            break;
        }
    } while (0);
    bin.release(); brot.release(); bsh.release(); bsc.release();
    if (tmp.d_tq_tables) (void)hipFree(tmp.d_tq_tables);
    if (tmp.d_tq_l1) (void)hipFree(tmp.d_tq_l1);
    if (tmp.d_tq_norms) (void)hipFree(tmp.d_tq_norms);
    if (tmp.d_tq_shift) (void)hipFree(tmp.d_tq_shift);
    if (tmp.d_tq_scale) (void)hipFree(tmp.d_tq_scale);
    if (tmp.d_tq_weights) (void)hipFree(tmp.d_tq_weights);
    if (rc == QMX_ERR_OTHER) set_error("qmx_tq_fit_plus: HIP error");
    return rc;
}

int32_t qmx_tq_encode(int32_t device_id, uint32_t distance, uint32_t dim, const qmx_tq_params *params, const float *vectors, uint64_t n, void *out_rows) {
    QMX_REQUIRE(params && (n == 0 || (vectors && out_rows)) && dim > 0, QMX_ERR_BAD_ARG, "bad argument");
    QMX_REQUIRE(distance <= QMX_DISTANCE_MANHATTAN, QMX_ERR_BAD_ARG, "bad distance %u", distance);
    QMX_TRY(check_device(device_id, nullptr));
    if (n == 0) return QMX_OK;
    qmx_segment tmp;                       // parameter holder only: the rotation tables of TurboQuantizer::new
    qmx_segment_desc d;
    memset(&d, 0, sizeof(d));
    d.dtype = QMX_DTYPE_TQ; d.distance = distance; d.dim = dim; d.tq = params; d.device_id = device_id;
    tmp.device = device_id; tmp.dtype = QMX_DTYPE_TQ; tmp.distance = distance; tmp.dim = dim;
    int32_t rc = tq_segment_setup(&tmp, &d);
    DevBuf bin, brot, bout;
    do {
        if (rc != QMX_OK) break;
        const uint32_t row_bytes = (uint32_t)tmp.row_bytes;
        const uint64_t CH = 65536;         // vectors per pass (the f64 scratch is padded_dim * 8 bytes per vector)
        const bool in_dev = is_device_ptr(vectors), out_dev = is_device_ptr(out_rows);
        if ((rc = brot.reserve((size_t)std::min<uint64_t>(n, CH) * tmp.tq_padded_dim * 8)) != QMX_OK) break;
        if (!in_dev && (rc = bin.reserve((size_t)std::min<uint64_t>(n, CH) * dim * 4)) != QMX_OK) break;
        if (!out_dev && (rc = bout.reserve((size_t)std::min<uint64_t>(n, CH) * row_bytes)) != QMX_OK) break;
        for (uint64_t r0 = 0; r0 < n && rc == QMX_OK; r0 += CH) {
            const uint32_t cnt = (uint32_t)std::min<uint64_t>(CH, n - r0);
            const float *d_in = vectors + r0 * dim;
            if (!in_dev) {
                if (hipMemcpy(bin.p, vectors + r0 * dim, (size_t)cnt * dim * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
                d_in = (const float *)bin.p;
            }
            void *d_out = out_dev ? (void *)((char *)out_rows + r0 * row_bytes) : bout.p;
            if ((rc = launch_tq_rotate(nullptr, d_in, cnt, tq_rotation(&tmp), (double *)brot.p)) != QMX_OK) break;
            if ((rc = launch_tq_quantize(nullptr, (double *)brot.p, cnt, tmp.tq_padded_dim, tmp.tq_value_bits, distance, d_out, row_bytes, tmp.d_tq_shift,
                                         tmp.d_tq_scale)) != QMX_OK) break;
            if (hipDeviceSynchronize() != hipSuccess) { rc = QMX_ERR_OTHER; break; }
            if (!out_dev && hipMemcpy((char *)out_rows + r0 * row_bytes, bout.p, (size_t)cnt * row_bytes, hipMemcpyDeviceToHost) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
        }
    } while (0);
    bin.release(); brot.release(); bout.release();
    if (tmp.d_tq_tables) (void)hipFree(tmp.d_tq_tables);
    if (tmp.d_tq_l1) (void)hipFree(tmp.d_tq_l1);
    if (tmp.d_tq_norms) (void)hipFree(tmp.d_tq_norms);
    if (tmp.d_tq_shift) (void)hipFree(tmp.d_tq_shift);
    if (tmp.d_tq_scale) (void)hipFree(tmp.d_tq_scale);
    if (tmp.d_tq_weights) (void)hipFree(tmp.d_tq_weights);
    return rc;
}

int32_t qmx_segment_create(const qmx_segment_desc *desc, qmx_segment **out) {
    QMX_REQUIRE(desc && out, QMX_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    QMX_REQUIRE(desc->dtype <= QMX_DTYPE_TQ, QMX_ERR_BAD_ARG, "bad dtype %u", desc->dtype);
    QMX_REQUIRE(desc->distance <= QMX_DISTANCE_MANHATTAN, QMX_ERR_BAD_ARG, "bad distance %u", desc->distance);
    QMX_REQUIRE(desc->dim > 0, QMX_ERR_BAD_ARG, "dim must be > 0");
    QMX_REQUIRE(desc->n <= 0xFFFFFFFFull, QMX_ERR_BAD_ARG, "PointOffsetType is u32: n=%llu too large", (unsigned long long)desc->n);
    QMX_REQUIRE(desc->n == 0 || desc->data, QMX_ERR_BAD_ARG, "data is NULL");
    hipDeviceProp_t prop;
    QMX_TRY(check_device(desc->device_id, &prop));

    qmx_segment *s = new (std::nothrow) qmx_segment();
    QMX_REQUIRE(s, QMX_ERR_OUT_OF_MEMORY, "host allocation failed");
    s->device = desc->device_id;
    s->num_cus = prop.multiProcessorCount;
    s->dtype = desc->dtype;
    s->distance = desc->distance;
    s->dim = desc->dim;
    s->flags = desc->flags;
    s->n = desc->n;
    s->scan_dim = desc->dim;
    int32_t rc = QMX_OK;
    switch (desc->dtype) {
        case QMX_DTYPE_F32:
        case QMX_DTYPE_F16:
        case QMX_DTYPE_U8: s->row_bytes = (uint64_t)desc->dim * elem_bytes(desc->dtype); break;
        case QMX_DTYPE_SQ_U8:
            if (!desc->sq) { set_error("SQ segment needs qmx_sq_params"); rc = QMX_ERR_BAD_ARG; break; }
            s->sq = *desc->sq;
            if (s->sq.actual_dim != ((desc->dim + 15) / 16) * 16) {   // get_actual_dim, encoded_vectors_u8.rs:622-624
                set_error("actual_dim %u is not dim %u rounded up to 16", s->sq.actual_dim, desc->dim);
                rc = QMX_ERR_BAD_ARG;
                break;
            }
            s->scan_dim = s->sq.actual_dim;
            s->row_bytes = 4 + (uint64_t)s->sq.actual_dim;
            break;
        case QMX_DTYPE_PQ: {
            if (!desc->pq || !desc->pq->centroids) { set_error("PQ segment needs qmx_pq_params with centroids"); rc = QMX_ERR_BAD_ARG; break; }
            s->pq = *desc->pq;
            if (s->pq.chunk_size == 0 || s->pq.chunk_size > 256 || s->pq.n_centroids == 0 || s->pq.n_centroids > 256) {
                set_error("PQ: chunk_size %u must be 1..256 and n_centroids %u must be 1..256 (codes are u8)", s->pq.chunk_size, s->pq.n_centroids);
                rc = QMX_ERR_BAD_ARG;
                break;
            }
            s->pq_m = (desc->dim + s->pq.chunk_size - 1) / s->pq.chunk_size;   // get_vector_division, encoded_vectors_pq.rs:164-169
            s->scan_dim = s->pq_m;
            s->row_bytes = s->pq_m;
            const size_t cbytes = (size_t)s->pq.n_centroids * desc->dim * sizeof(float);
            hipError_t e = hipMalloc((void **)&s->d_centroids, cbytes);
            if (e == hipSuccess) e = hipMemcpy(s->d_centroids, desc->pq->centroids, cbytes, hipMemcpyDefault);
            if (e != hipSuccess) rc = hip_status(e, "PQ centroids upload", __FILE__, __LINE__);
            s->pq.centroids = nullptr;   // the caller's table is not referenced after create
            // score_internal's chunk terms, tabulated once (the HNSW build over a PQ segment scores stored <-> stored pairs with them)
            const size_t pbytes = (size_t)s->pq_m * s->pq.n_centroids * s->pq.n_centroids * sizeof(float);
            if (rc == QMX_OK && pbytes <= (256u << 20)) {
                e = hipMalloc((void **)&s->d_pq_pair, pbytes);
                if (e != hipSuccess) { rc = hip_status(e, "PQ pair table", __FILE__, __LINE__); break; }
                rc = launch_pq_pair_table(nullptr, desc->distance, desc->dim, s->pq, s->d_centroids, s->d_pq_pair);
                if (rc == QMX_OK && hipDeviceSynchronize() != hipSuccess) rc = QMX_ERR_OTHER;
            }
            break;
        }
        case QMX_DTYPE_BQ: {  // get_quantized_vector_size_from_params::<u128>(dim, encoding) (encoded_vectors_binary.rs:829-840, 412-419)
            s->bq_encoding = desc->bq ? desc->bq->encoding : (uint32_t)QMX_BQ_ONE_BIT;
            if (s->bq_encoding > QMX_BQ_ONE_AND_HALF_BITS) { set_error("bad BQ encoding %u", s->bq_encoding); rc = QMX_ERR_BAD_ARG; break; }
            {
                const uint32_t qe = desc->bq ? desc->bq->query_encoding : (uint32_t)QMX_BQ_QUERY_SAME_AS_STORAGE;
                if (qe > QMX_BQ_QUERY_SCALAR_8BITS) { set_error("bad BQ query encoding %u", qe); rc = QMX_ERR_BAD_ARG; break; }
                s->bq_query_bits = qe == QMX_BQ_QUERY_SCALAR_4BITS ? 4 : qe == QMX_BQ_QUERY_SCALAR_8BITS ? 8 : 1;
            }
            s->row_bytes = bq_row_bytes(desc->dim, s->bq_encoding);
            s->scan_dim = (uint32_t)s->row_bytes;
            if (desc->bq && desc->bq->mean && desc->bq->stddev && s->bq_encoding != QMX_BQ_ONE_BIT) {   // the stats encode the queries later
                const size_t b = (size_t)desc->dim * sizeof(float);
                hipError_t e = hipMalloc((void **)&s->d_bq_mean, b);
                if (e == hipSuccess) e = hipMalloc((void **)&s->d_bq_stddev, b);
                if (e == hipSuccess) e = hipMemcpy(s->d_bq_mean, desc->bq->mean, b, hipMemcpyDefault);
                if (e == hipSuccess) e = hipMemcpy(s->d_bq_stddev, desc->bq->stddev, b, hipMemcpyDefault);
                if (e != hipSuccess) rc = hip_status(e, "BQ vector stats upload", __FILE__, __LINE__);
            }
            break;
        }
        case QMX_DTYPE_TQ: rc = tq_segment_setup(s, desc); break;
        default:
            set_error("dtype %u not built yet", desc->dtype);
            rc = QMX_ERR_NOT_SUPPORTED;
    }
    if (rc == QMX_OK) rc = segment_upload(s, desc);
    if (rc == QMX_OK) rc = segment_split_stats(s);
    if (rc == QMX_OK) rc = segment_pq_rot(s);
    if (rc == QMX_OK) rc = segment_tq_stats(s);
    if (rc == QMX_OK) rc = segment_sq_stats(s);
    if (rc != QMX_OK) {
        segment_free(s);
        return rc;
    }
    *out = s;
    return QMX_OK;
}

// bytes per stored row of a storage file (reference row layout) and the header in front of the rows
static int32_t file_row_bytes(const qmx_segment_desc *desc, uint64_t *row_bytes_out, uint64_t *header_out) {
    uint64_t row_bytes = 0, header = 0;
    switch (desc->dtype) {
        case QMX_DTYPE_F32: case QMX_DTYPE_F16: case QMX_DTYPE_U8: row_bytes = (uint64_t)desc->dim * elem_bytes(desc->dtype); header = 4; break;
        case QMX_DTYPE_SQ_U8:
            QMX_REQUIRE(desc->sq, QMX_ERR_BAD_ARG, "SQ segment needs qmx_sq_params");
            row_bytes = 4ull + desc->sq->actual_dim;
            break;
        case QMX_DTYPE_PQ:
            QMX_REQUIRE(desc->pq && desc->pq->chunk_size, QMX_ERR_BAD_ARG, "PQ segment needs qmx_pq_params");
            row_bytes = ((uint64_t)desc->dim + desc->pq->chunk_size - 1) / desc->pq->chunk_size;
            break;
        case QMX_DTYPE_TQ: {    // TurboQuantizer::quantized_size_for (turboquant/encoding.rs:172-201)
            QMX_REQUIRE(desc->tq && desc->tq->bits <= QMX_TQ_BITS1, QMX_ERR_BAD_ARG, "TQ segment needs qmx_tq_params");
            const uint64_t d = desc->dim;
            const uint64_t padded = desc->tq->bits == QMX_TQ_BITS1 ? (d + 7) / 8 * 8 : desc->tq->bits == QMX_TQ_BITS1_5 ? (d * 3 / 2 + 7) / 8 * 8
                                  : desc->tq->bits == QMX_TQ_BITS2 ? (d + 3) / 4 * 4 : (d + 1) / 2 * 2;
            const uint64_t vb = desc->tq->bits == QMX_TQ_BITS4 ? 4 : desc->tq->bits == QMX_TQ_BITS2 ? 2 : 1;
            row_bytes = padded * vb / 8 + (desc->distance == QMX_DISTANCE_EUCLID ? 8 : 4) + (desc->tq->plus_mode ? 4 : 0);
            break;
        }
        default: row_bytes = bq_row_bytes(desc->dim, desc->bq ? desc->bq->encoding : 0u); break;   // BQ
    }
    *row_bytes_out = row_bytes;
    *header_out = header;
    return QMX_OK;
}

int32_t qmx_segment_create_from_files(const qmx_segment_desc *desc, const char *vectors_path, const char *deleted_path, qmx_segment **out) {
    QMX_REQUIRE(desc && vectors_path && out, QMX_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    QMX_REQUIRE(desc->dtype <= QMX_DTYPE_TQ && desc->dim > 0, QMX_ERR_BAD_ARG, "bad dtype / dim");
    QMX_TRY(check_device(desc->device_id, nullptr));
    uint64_t row_bytes = 0, header = 0;
    QMX_TRY(file_row_bytes(desc, &row_bytes, &header));
    FILE *f = fopen(vectors_path, "rb");
    QMX_REQUIRE(f, QMX_ERR_BAD_ARG, "cannot open %s", vectors_path);
    int32_t rc = QMX_OK;
    void *d_tmp = nullptr, *h_pin = nullptr;
    do {
        if (fseek(f, 0, SEEK_END) != 0) { set_error("cannot seek %s", vectors_path); rc = QMX_ERR_OTHER; break; }
        const uint64_t len = (uint64_t)ftell(f);
        rewind(f);
        if (header) {
            char magic[4] = {0, 0, 0, 0};
            if (len < header || fread(magic, 1, 4, f) != 4 || memcmp(magic, "data", 4) != 0) {   // VECTORS_HEADER
                set_error("%s does not start with the dense vector file header \"data\"", vectors_path);
                rc = QMX_ERR_BAD_ARG;
                break;
            }
        }
        const uint64_t in_file = (len - header) / row_bytes;      // num_vectors = (file_len - HEADER_SIZE) / dim / size_of::<T>()
        const uint64_t n = desc->n ? desc->n : in_file;
        if (n > in_file) { set_error("%s holds %llu rows, %llu asked for", vectors_path, (unsigned long long)in_file, (unsigned long long)n); rc = QMX_ERR_BAD_ARG; break; }
        const size_t total = (size_t)n * row_bytes;
        if (hipMalloc(&d_tmp, std::max<size_t>(total, 16)) != hipSuccess) { set_error("device allocation of %zu bytes failed", total); rc = QMX_ERR_OUT_OF_MEMORY; break; }
        const size_t chunk = 64u << 20;
        if (hipHostMalloc(&h_pin, chunk, hipHostMallocDefault) != hipSuccess) { set_error("pinned staging allocation failed"); rc = QMX_ERR_OUT_OF_MEMORY; break; }
        for (size_t off = 0; off < total && rc == QMX_OK; off += chunk) {
            const size_t want = std::min(chunk, total - off);
            if (fread(h_pin, 1, want, f) != want) { set_error("short read from %s", vectors_path); rc = QMX_ERR_OTHER; break; }
            if (hipMemcpy((char *)d_tmp + off, h_pin, want, hipMemcpyHostToDevice) != hipSuccess) { set_error("upload failed"); rc = QMX_ERR_OTHER; break; }
        }
        if (rc != QMX_OK) break;
        qmx_segment_desc d = *desc;
        d.n = n;
        d.data = d_tmp;
        d.row_stride_bytes = 0;
        d.flags = desc->flags & ~QMX_SEG_DATA_ON_DEVICE;          // copied (and re-packed to the 16-byte row pitch) into the segment's own block
        rc = qmx_segment_create(&d, out);
    } while (0);
    fclose(f);
    if (h_pin) (void)hipHostFree(h_pin);
    if (d_tmp) (void)hipFree(d_tmp);
    if (rc != QMX_OK || !deleted_path) return rc;
    // the "drop" file: header, padding to align_of::<usize>() = 8, then the bit words
    FILE *g = fopen(deleted_path, "rb");
    std::vector<uint64_t> words;
    if (!g) { set_error("cannot open %s", deleted_path); rc = QMX_ERR_BAD_ARG; }
    else {
        char magic[8];
        const uint64_t n = (*out)->n;
        words.resize((size_t)((n + 63) / 64));
        if (fread(magic, 1, 8, g) != 8 || memcmp(magic, "drop", 4) != 0) { set_error("%s does not start with the deleted-flags header \"drop\"", deleted_path); rc = QMX_ERR_BAD_ARG; }
        else if (!words.empty() && fread(words.data(), 8, words.size(), g) != words.size()) { set_error("%s is shorter than %llu flags", deleted_path, (unsigned long long)n); rc = QMX_ERR_BAD_ARG; }
        fclose(g);
        if (rc == QMX_OK) rc = qmx_segment_set_deleted(*out, nullptr, 0, words.data(), n);
    }
    if (rc != QMX_OK) {
        qmx_segment_destroy(*out);
        *out = nullptr;
    }
    return rc;
}

int32_t qmx_segment_create_chunked(const qmx_segment_desc *desc, const void *const *chunks, uint64_t rows_per_chunk, uint32_t n_chunks,
                                   qmx_segment **out) {
    QMX_REQUIRE(desc && out && (n_chunks == 0 || chunks), QMX_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    QMX_REQUIRE(desc->dtype <= QMX_DTYPE_TQ, QMX_ERR_BAD_ARG, "bad dtype %u", desc->dtype);
    QMX_REQUIRE(desc->distance <= QMX_DISTANCE_MANHATTAN && desc->dim > 0, QMX_ERR_BAD_ARG, "bad distance / dim");
    QMX_REQUIRE(desc->n <= 0xFFFFFFFFull, QMX_ERR_BAD_ARG, "PointOffsetType is u32: n=%llu too large", (unsigned long long)desc->n);
    QMX_REQUIRE(desc->n == 0 || (rows_per_chunk > 0 && (uint64_t)n_chunks * rows_per_chunk >= desc->n), QMX_ERR_BAD_ARG,
                "%u chunks of %llu rows cannot hold %llu rows", n_chunks, (unsigned long long)rows_per_chunk, (unsigned long long)desc->n);
    QMX_REQUIRE(!(desc->flags & QMX_SEG_DATA_ON_DEVICE), QMX_ERR_BAD_ARG, "chunks are copied into one block: QMX_SEG_DATA_ON_DEVICE does not apply");
    hipDeviceProp_t prop;
    QMX_TRY(check_device(desc->device_id, &prop));
    if (desc->dtype > QMX_DTYPE_U8) {
        // Quantized chunked (appendable) storages (vector_storage/quantized/quantized_chunked_mmap_storage/{read_only.rs:20, read_write.rs:18}): the chunks'
        // rows (reference row layout of the quantizer) are gathered into one device block, which then takes the ordinary route of qmx_segment_create -
        // SQ / TQ rows are split into their aligned code block + extras columns, PQ / BQ blocks are kept as they are (the segment owns the gathered block).
        uint64_t row_bytes = 0, header = 0;
        QMX_TRY(file_row_bytes(desc, &row_bytes, &header));
        const uint64_t src_stride = desc->row_stride_bytes ? desc->row_stride_bytes : row_bytes;
        QMX_REQUIRE(src_stride >= row_bytes, QMX_ERR_BAD_ARG, "row_stride_bytes %llu < row size %llu", (unsigned long long)src_stride, (unsigned long long)row_bytes);
        void *d_tmp = nullptr;
        QMX_HIP(hipMalloc(&d_tmp, (size_t)std::max<uint64_t>(1, desc->n) * row_bytes));
        hipError_t e = hipSuccess;
        for (uint32_t c = 0; e == hipSuccess && c < n_chunks && (uint64_t)c * rows_per_chunk < desc->n; ++c) {
            const uint64_t row0 = (uint64_t)c * rows_per_chunk, cnt = std::min<uint64_t>(rows_per_chunk, desc->n - row0);
            if (!chunks[c]) { e = hipErrorInvalidValue; break; }
            e = hipMemcpy2D((char *)d_tmp + row0 * row_bytes, row_bytes, chunks[c], src_stride, row_bytes, cnt, hipMemcpyDefault);
        }
        if (e != hipSuccess) {
            (void)hipFree(d_tmp);
            return hip_status(e, "chunk upload", __FILE__, __LINE__);
        }
        qmx_segment_desc d2 = *desc;
        d2.data = d_tmp;
        d2.row_stride_bytes = row_bytes;
        d2.flags |= QMX_SEG_DATA_ON_DEVICE;
        const int32_t rc = qmx_segment_create(&d2, out);
        if (rc != QMX_OK || !*out || (*out)->d_rows != d_tmp) (void)hipFree(d_tmp);     // (split into the segment's own blocks, or refused)
        else (*out)->owns_rows = true;                                                   // PQ / BQ: the gathered block IS the segment's block
        if (rc == QMX_OK && *out) (*out)->flags &= ~QMX_SEG_DATA_ON_DEVICE;
        return rc;
    }
    qmx_segment *s = new (std::nothrow) qmx_segment();
    QMX_REQUIRE(s, QMX_ERR_OUT_OF_MEMORY, "host allocation failed");
    s->device = desc->device_id;
    s->num_cus = prop.multiProcessorCount;
    s->dtype = desc->dtype; s->distance = desc->distance; s->dim = desc->dim; s->flags = desc->flags; s->n = desc->n;
    s->scan_dim = desc->dim;
    s->row_bytes = (uint64_t)desc->dim * elem_bytes(desc->dtype);
    const uint64_t src_stride = desc->row_stride_bytes ? desc->row_stride_bytes : s->row_bytes;
    s->row_stride = (s->row_bytes + 15) & ~15ull;
    const size_t bytes = (size_t)std::max<uint64_t>(1, s->n) * s->row_stride;
    hipError_t e = src_stride >= s->row_bytes ? hipMalloc(&s->d_rows, bytes) : hipErrorInvalidValue;
    s->owns_rows = e == hipSuccess;
    if (e == hipSuccess && s->row_stride != s->row_bytes) e = hipMemset(s->d_rows, 0, bytes);
    for (uint32_t c = 0; e == hipSuccess && c < n_chunks && (uint64_t)c * rows_per_chunk < s->n; ++c) {
        const uint64_t row0 = (uint64_t)c * rows_per_chunk, cnt = std::min<uint64_t>(rows_per_chunk, s->n - row0);
        if (!chunks[c]) { e = hipErrorInvalidValue; break; }
        e = hipMemcpy2D((char *)s->d_rows + row0 * s->row_stride, s->row_stride, chunks[c], src_stride, s->row_bytes, cnt, hipMemcpyDefault);
    }
    if (e != hipSuccess) {
        const int32_t rc = hip_status(e, "chunk upload", __FILE__, __LINE__);
        segment_free(s);
        return rc;
    }
    *out = s;
    return QMX_OK;
}

int32_t qmx_segment_destroy(qmx_segment *seg) {
    if (!seg) return QMX_OK;
    (void)hipSetDevice(seg->device);
    segment_free(seg);
    return QMX_OK;
}

int32_t qmx_segment_set_deleted(qmx_segment *seg, const uint64_t *point_deleted, uint64_t n_point_bits,
                                const uint64_t *vec_deleted, uint64_t n_vec_bits) {
    QMX_REQUIRE(seg, QMX_ERR_BAD_ARG, "NULL segment");
    QMX_HIP(hipSetDevice(seg->device));
    auto upload = [&](const uint64_t *src, uint64_t nbits, uint64_t **dst, uint64_t *dst_bits) -> int32_t {
        if (*dst) (void)hipFree(*dst);
        *dst = nullptr;
        *dst_bits = 0;
        if (!src) return QMX_OK;
        const size_t words = (size_t)((nbits + 63) / 64);
        QMX_HIP(hipMalloc((void **)dst, std::max<size_t>(words, 1) * 8));
        if (words) QMX_HIP(hipMemcpy(*dst, src, words * 8, hipMemcpyDefault));
        *dst_bits = nbits;
        return QMX_OK;
    };
    QMX_TRY(upload(point_deleted, n_point_bits, &seg->d_point_deleted, &seg->n_point_bits));
    QMX_TRY(upload(vec_deleted, n_vec_bits, &seg->d_vec_deleted, &seg->n_vec_bits));
    return QMX_OK;
}

int32_t qmx_segment_row_bytes(const qmx_segment *seg, uint64_t *out) {
    QMX_REQUIRE(seg && out, QMX_ERR_BAD_ARG, "NULL argument");
    *out = seg->row_bytes;
    return QMX_OK;
}

int32_t qmx_segment_get_info(const qmx_segment *seg, qmx_segment_info *out) {
    QMX_REQUIRE(seg && out, QMX_ERR_BAD_ARG, "NULL argument");
    memset(out, 0, sizeof(*out));
    out->derived_copy = !seg->d_rows_split ? 0u : seg->split_i8 ? QMX_SEG_I8_COPY : seg->split_half ? QMX_SEG_HALF_COPY : QMX_SEG_SPLIT_COPY;
    out->chosen_by_trial = seg->auto_choice ? 1u : 0u;
    out->derived_copy_bytes = seg->d_rows_split ? seg->copy_bytes : 0;
    out->i8_scale_balance = seg->split_i8 ? seg->i8_balance : 0.0f;
    out->trial_i8_ms = seg->auto_i8_ms;
    out->trial_half_ms = seg->auto_half_ms;
    out->trial_i8_verified_rows = seg->auto_i8_verified;
    out->trial_i8_fallback_queries = seg->auto_i8_fallback;
    return QMX_OK;
}

int32_t qmx_segment_read_rows(const qmx_segment *seg, const uint32_t *ids, uint32_t n, void *out_rows) {
    QMX_REQUIRE(seg && (n == 0 || (ids && out_rows)), QMX_ERR_BAD_ARG, "NULL argument");
    QMX_HIP(hipSetDevice(seg->device));
    if (seg->dtype == QMX_DTYPE_SQ_U8) {
        for (uint32_t i = 0; i < n; ++i) {
            QMX_REQUIRE(ids[i] < seg->n, QMX_ERR_OUT_OF_BOUNDS, "row %u out of range", ids[i]);
            char *dst = (char *)out_rows + (size_t)i * seg->row_bytes;
            QMX_HIP(hipMemcpy(dst, seg->d_row_offsets + ids[i], 4, hipMemcpyDefault));
            QMX_HIP(hipMemcpy(dst + 4, (const char *)seg->d_rows + (size_t)ids[i] * seg->row_stride, seg->sq.actual_dim, hipMemcpyDefault));
        }
        return QMX_OK;
    }
    if (seg->dtype == QMX_DTYPE_TQ) {
        const bool has_l2 = seg->d_tq_l2 != nullptr;
        for (uint32_t i = 0; i < n; ++i) {
            QMX_REQUIRE(ids[i] < seg->n, QMX_ERR_OUT_OF_BOUNDS, "row %u out of range", ids[i]);
            char *dst = (char *)out_rows + (size_t)i * seg->row_bytes;
            QMX_HIP(hipMemcpy(dst, (const char *)seg->d_rows + (size_t)ids[i] * seg->row_stride, seg->tq_code_bytes, hipMemcpyDefault));
            QMX_HIP(hipMemcpy(dst + seg->tq_code_bytes, seg->d_tq_sf + ids[i], 4, hipMemcpyDefault));
            if (has_l2) QMX_HIP(hipMemcpy(dst + seg->tq_code_bytes + 4, seg->d_tq_l2 + ids[i], 4, hipMemcpyDefault));
            if (seg->d_tq_xm) QMX_HIP(hipMemcpy(dst + seg->tq_code_bytes + (has_l2 ? 8 : 4), seg->d_tq_xm + ids[i], 4, hipMemcpyDefault));
        }
        return QMX_OK;
    }
    for (uint32_t i = 0; i < n; ++i) {
        QMX_REQUIRE(ids[i] < seg->n, QMX_ERR_OUT_OF_BOUNDS, "row %u out of range", ids[i]);
        QMX_HIP(hipMemcpy((char *)out_rows + (size_t)i * seg->row_bytes,
                          (const char *)seg->d_rows + (size_t)ids[i] * seg->row_stride, seg->row_bytes, hipMemcpyDefault));
    }
    return QMX_OK;
}

// ---------------------------------------------------------------------------------------------
// preprocess / casts / synth
// ---------------------------------------------------------------------------------------------
int32_t qmx_preprocess_f32(int32_t device_id, uint32_t distance, const float *in, uint64_t n, uint32_t dim, float *out) {
    QMX_REQUIRE(in && out && dim > 0, QMX_ERR_BAD_ARG, "bad argument");
    QMX_TRY(check_device(device_id, nullptr));
    const size_t bytes = (size_t)n * dim * sizeof(float);
    if (bytes == 0) return QMX_OK;
    const bool in_dev = is_device_ptr(in), out_dev = is_device_ptr(out);
    float *d_in = const_cast<float *>(in), *d_out = out;
    DevBuf bin, bout;
    if (!in_dev) {
        QMX_TRY(bin.reserve(bytes));
        QMX_HIP(hipMemcpy(bin.p, in, bytes, hipMemcpyHostToDevice));
        d_in = (float *)bin.p;
    }
    if (!out_dev) {
        QMX_TRY(bout.reserve(bytes));
        d_out = (float *)bout.p;
    }
    int32_t rc = QMX_OK;
    if (distance == QMX_DISTANCE_COSINE) rc = launch_cosine_preprocess_f32(nullptr, d_in, d_out, n, dim);
    else if (d_out != d_in) rc = hipMemcpy(d_out, d_in, bytes, hipMemcpyDeviceToDevice) == hipSuccess ? QMX_OK : QMX_ERR_OTHER;
    if (rc == QMX_OK && !out_dev) rc = hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost) == hipSuccess ? QMX_OK : QMX_ERR_OTHER;
    if (rc == QMX_OK && hipDeviceSynchronize() != hipSuccess) rc = QMX_ERR_OTHER;
    bin.release();
    bout.release();
    return rc;
}

int32_t qmx_cast_f32(int32_t device_id, uint32_t dst_dtype, const float *in, uint64_t count, void *out) {
    QMX_REQUIRE(in && out, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(dst_dtype <= QMX_DTYPE_U8, QMX_ERR_BAD_ARG, "bad dtype");
    QMX_TRY(check_device(device_id, nullptr));
    if (count == 0) return QMX_OK;
    const size_t in_bytes = (size_t)count * 4, out_bytes = (size_t)count * elem_bytes(dst_dtype);
    const bool in_dev = is_device_ptr(in), out_dev = is_device_ptr(out);
    DevBuf bin, bout;
    const float *d_in = in;
    void *d_out = out;
    if (!in_dev) {
        QMX_TRY(bin.reserve(in_bytes));
        QMX_HIP(hipMemcpy(bin.p, in, in_bytes, hipMemcpyHostToDevice));
        d_in = (const float *)bin.p;
    }
    if (!out_dev) {
        QMX_TRY(bout.reserve(out_bytes));
        d_out = bout.p;
    }
    int32_t rc = launch_cast_f32(nullptr, (int)dst_dtype, d_in, d_out, count);
    if (rc == QMX_OK && !out_dev) rc = hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost) == hipSuccess ? QMX_OK : QMX_ERR_OTHER;
    if (rc == QMX_OK && hipDeviceSynchronize() != hipSuccess) rc = QMX_ERR_OTHER;
    bin.release();
    bout.release();
    return rc;
}

int32_t qmx_sq_encode(int32_t device_id, uint32_t distance, const qmx_sq_params *params, const float *in, uint64_t n,
                      uint32_t dim, void *out_rows) {
    QMX_REQUIRE(params && (n == 0 || (in && out_rows)) && dim > 0, QMX_ERR_BAD_ARG, "bad argument");
    QMX_REQUIRE(distance <= QMX_DISTANCE_MANHATTAN, QMX_ERR_BAD_ARG, "bad distance");
    QMX_REQUIRE(params->actual_dim == ((dim + 15) / 16) * 16, QMX_ERR_BAD_ARG, "actual_dim must be dim rounded up to 16");
    QMX_REQUIRE(params->alpha != 0.0f, QMX_ERR_BAD_ARG, "alpha must be non-zero");
    QMX_TRY(check_device(device_id, nullptr));
    if (n == 0) return QMX_OK;
    const size_t in_bytes = (size_t)n * dim * 4, out_bytes = (size_t)n * (4 + (size_t)params->actual_dim);
    DevBuf bin, bout;
    const float *d_in = in;
    void *d_out = out_rows;
    int32_t rc = QMX_OK;
    do {
        if (!is_device_ptr(in)) {
            if ((rc = bin.reserve(in_bytes)) != QMX_OK) break;
            if (hipMemcpy(bin.p, in, in_bytes, hipMemcpyHostToDevice) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
            d_in = (const float *)bin.p;
        }
        const bool out_dev = is_device_ptr(out_rows);
        if (!out_dev) {
            if ((rc = bout.reserve(out_bytes)) != QMX_OK) break;
            d_out = bout.p;
        }
        if ((rc = launch_sq_encode(nullptr, (int)distance, *params, dim, d_in, n, nullptr, 0, nullptr, (uint8_t *)d_out, 0, 0)) != QMX_OK) break;
        if (!out_dev && hipMemcpy(out_rows, d_out, out_bytes, hipMemcpyDeviceToHost) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
        if (hipDeviceSynchronize() != hipSuccess) rc = QMX_ERR_OTHER;
    } while (0);
    bin.release();
    bout.release();
    return rc;
}

uint64_t qmx_bq_row_bytes(uint32_t dim, uint32_t encoding) { return bq_row_bytes(dim, encoding); }

int32_t qmx_bq_encode_ex(int32_t device_id, const qmx_bq_params *params, const float *in, uint64_t n, uint32_t dim, void *out_rows) {
    QMX_REQUIRE((n == 0 || (in && out_rows)) && dim > 0, QMX_ERR_BAD_ARG, "bad argument");
    const uint32_t encoding = params ? params->encoding : (uint32_t)QMX_BQ_ONE_BIT;
    QMX_REQUIRE(encoding <= QMX_BQ_ONE_AND_HALF_BITS, QMX_ERR_BAD_ARG, "bad BQ encoding %u", encoding);
    QMX_TRY(check_device(device_id, nullptr));
    if (n == 0) return QMX_OK;
    const size_t row_bytes = (size_t)bq_row_bytes(dim, encoding);
    const size_t in_bytes = (size_t)n * dim * 4, out_bytes = (size_t)n * row_bytes;
    const bool stats = params && params->mean && params->stddev && encoding != QMX_BQ_ONE_BIT;
    DevBuf bin, bout, bm, bs;
    const float *d_in = in, *d_mean = nullptr, *d_sd = nullptr;
    void *d_out = out_rows;
    int32_t rc = QMX_OK;
    do {
        if (!is_device_ptr(in)) {
            if ((rc = bin.reserve(in_bytes)) != QMX_OK) break;
            if (hipMemcpy(bin.p, in, in_bytes, hipMemcpyHostToDevice) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
            d_in = (const float *)bin.p;
        }
        if (stats) {
            if ((rc = bm.reserve((size_t)dim * 4)) != QMX_OK || (rc = bs.reserve((size_t)dim * 4)) != QMX_OK) break;
            if (hipMemcpy(bm.p, params->mean, (size_t)dim * 4, hipMemcpyDefault) != hipSuccess ||
                hipMemcpy(bs.p, params->stddev, (size_t)dim * 4, hipMemcpyDefault) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
            d_mean = (const float *)bm.p;
            d_sd = (const float *)bs.p;
        }
        const bool out_dev = is_device_ptr(out_rows);
        if (!out_dev) {
            if ((rc = bout.reserve(out_bytes)) != QMX_OK) break;
            d_out = bout.p;
        }
        if ((rc = launch_bq_encode(nullptr, d_in, n, dim, encoding, d_mean, d_sd, (uint8_t *)d_out, row_bytes)) != QMX_OK) break;
        if (!out_dev && hipMemcpy(out_rows, d_out, out_bytes, hipMemcpyDeviceToHost) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
        if (hipDeviceSynchronize() != hipSuccess) rc = QMX_ERR_OTHER;
    } while (0);
    bin.release();
    bout.release();
    bm.release();
    bs.release();
    return rc;
}

int32_t qmx_vector_stats(int32_t device_id, const float *vectors, uint64_t n, uint32_t dim, float *min_out, float *max_out, float *mean_out, float *stddev_out) {
    QMX_REQUIRE((n == 0 || vectors) && dim > 0 && mean_out && stddev_out, QMX_ERR_BAD_ARG, "bad argument");
    QMX_TRY(check_device(device_id, nullptr));
    DevBuf bin, bout;
    int32_t rc = QMX_OK;
    do {
        const float *d_in = vectors;
        if (n && !is_device_ptr(vectors)) {
            if ((rc = bin.reserve((size_t)n * dim * 4)) != QMX_OK) break;
            if (hipMemcpy(bin.p, vectors, (size_t)n * dim * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
            d_in = (const float *)bin.p;
        }
        if ((rc = bout.reserve((size_t)dim * 16)) != QMX_OK) break;
        float *o = (float *)bout.p;
        if ((rc = launch_vector_stats(nullptr, d_in, (uint64_t)dim * 4, n, dim, o, o + dim, o + 2 * (size_t)dim, o + 3 * (size_t)dim)) != QMX_OK) break;
        if (hipDeviceSynchronize() != hipSuccess) { rc = QMX_ERR_OTHER; break; }
        float *dst[4] = {min_out, max_out, mean_out, stddev_out};
        for (int k = 0; k < 4 && rc == QMX_OK; ++k)
            if (dst[k] && hipMemcpy(dst[k], o + (size_t)k * dim, (size_t)dim * 4, hipMemcpyDefault) != hipSuccess) rc = QMX_ERR_OTHER;
    } while (0);
    bin.release(); bout.release();
    if (rc == QMX_ERR_OTHER) set_error("qmx_vector_stats: HIP error");
    return rc;
}

int32_t qmx_bq_encode(int32_t device_id, const float *in, uint64_t n, uint32_t dim, void *out_rows) {
    return qmx_bq_encode_ex(device_id, nullptr, in, n, dim, out_rows);
}

int32_t qmx_pq_train(int32_t device_id, const float *sample, uint64_t n, uint32_t dim, uint32_t chunk_size, uint32_t n_centroids,
                     uint32_t max_iterations, float accuracy, uint32_t threads, float *out_centroids, uint32_t *out_iterations) {
    QMX_REQUIRE(out_centroids && (n == 0 || sample) && dim > 0, QMX_ERR_BAD_ARG, "bad argument");
    QMX_REQUIRE(chunk_size >= 1 && chunk_size <= 256 && n_centroids >= 1 && n_centroids <= 256, QMX_ERR_BAD_ARG, "chunk_size / n_centroids out of range");
    QMX_TRY(check_device(device_id, nullptr));
    const uint32_t m = (dim + chunk_size - 1) / chunk_size;
    const size_t cbytes = (size_t)n_centroids * dim * sizeof(float);
    if (n <= n_centroids) {   // not enough vectors: the points are the centroids, the rest zeros (encoded_vectors_pq.rs:354-362)
        std::vector<float> tmp((size_t)n_centroids * dim, 0.0f);
        if (n) QMX_HIP(hipMemcpy(tmp.data(), sample, (size_t)n * dim * 4, hipMemcpyDefault));
        QMX_HIP(hipMemcpy(out_centroids, tmp.data(), cbytes, hipMemcpyDefault));
        if (out_iterations) for (uint32_t c = 0; c < m; ++c) out_iterations[c] = 0;
        return QMX_OK;
    }
    DevBuf bin, bcen;
    const float *d_in = sample;
    float *d_cen = out_centroids;
    int32_t rc = QMX_OK;
    do {
        if (!is_device_ptr(sample)) {
            if ((rc = bin.reserve((size_t)n * dim * 4)) != QMX_OK) break;
            if (hipMemcpy(bin.p, sample, (size_t)n * dim * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
            d_in = (const float *)bin.p;
        }
        const bool out_dev = is_device_ptr(out_centroids);
        if (!out_dev) {
            if ((rc = bcen.reserve(cbytes)) != QMX_OK) break;
            d_cen = (float *)bcen.p;
        }
        if ((rc = launch_pq_train(nullptr, dim, chunk_size, n_centroids, d_in, n, max_iterations, accuracy, threads, d_cen, out_iterations)) != QMX_OK) break;
        if (!out_dev && hipMemcpy(out_centroids, d_cen, cbytes, hipMemcpyDeviceToHost) != hipSuccess) rc = QMX_ERR_OTHER;
    } while (0);
    bin.release();
    bcen.release();
    return rc;
}

int32_t qmx_sq_fit_min_max(int32_t device_id, uint32_t distance, const float *in, uint64_t n, uint32_t dim, qmx_sq_params *out) {
    QMX_REQUIRE(out && (n == 0 || in) && dim > 0, QMX_ERR_BAD_ARG, "bad argument");
    QMX_REQUIRE(distance <= QMX_DISTANCE_MANHATTAN, QMX_ERR_BAD_ARG, "bad distance");
    QMX_TRY(check_device(device_id, nullptr));
    DevBuf bin;
    const float *d_in = in;
    if (n && !is_device_ptr(in)) {
        QMX_TRY(bin.reserve((size_t)n * dim * 4));
        if (hipMemcpy(bin.p, in, (size_t)n * dim * 4, hipMemcpyHostToDevice) != hipSuccess) { bin.release(); return QMX_ERR_OTHER; }
        d_in = (const float *)bin.p;
    }
    float mn = 0.f, mx = 0.f;
    const int32_t rc = launch_minmax_f32(nullptr, d_in, n * dim, &mn, &mx);
    bin.release();
    QMX_TRY(rc);
    memset(out, 0, sizeof(*out));
    out->actual_dim = ((dim + 15) / 16) * 16;                 // get_actual_dim (:622-624)
    out->alpha = (mx - mn) / 127.0f;                          // alpha_offset_from_min_max (:523-527)
    out->offset = mn;
    out->invert = (distance == QMX_DISTANCE_EUCLID || distance == QMX_DISTANCE_MANHATTAN) ? 1 : 0;   // quantized_vectors.rs:232
    float m;
    if (distance == QMX_DISTANCE_DOT || distance == QMX_DISTANCE_COSINE) m = out->alpha * out->alpha;      // :210-221
    else if (distance == QMX_DISTANCE_MANHATTAN) m = out->alpha;
    else m = -2.0f * out->alpha * out->alpha;
    out->multiplier = out->invert ? -m : m;
    return QMX_OK;
}

static void sq_params_from_min_max(uint32_t distance, uint32_t dim, float mn, float mx, qmx_sq_params *out) {
    memset(out, 0, sizeof(*out));
    out->actual_dim = ((dim + 15) / 16) * 16;                 // get_actual_dim (:622-624)
    out->alpha = (mx - mn) / 127.0f;                          // alpha_offset_from_min_max (:523-527)
    out->offset = mn;
    out->invert = (distance == QMX_DISTANCE_EUCLID || distance == QMX_DISTANCE_MANHATTAN) ? 1 : 0;   // quantized_vectors.rs:232
    float m;
    if (distance == QMX_DISTANCE_DOT || distance == QMX_DISTANCE_COSINE) m = out->alpha * out->alpha;      // :210-221
    else if (distance == QMX_DISTANCE_MANHATTAN) m = out->alpha;
    else m = -2.0f * out->alpha * out->alpha;
    out->multiplier = out->invert ? -m : m;
}

int32_t qmx_sq_fit_quantile(int32_t device_id, uint32_t distance, const float *sample, uint64_t n_sample, uint32_t dim, uint64_t count,
                            float quantile, qmx_sq_params *out, int32_t *found) {
    QMX_REQUIRE(out && found && (n_sample == 0 || sample) && dim > 0, QMX_ERR_BAD_ARG, "bad argument");
    QMX_REQUIRE(distance <= QMX_DISTANCE_MANHATTAN, QMX_ERR_BAD_ARG, "bad distance");
    QMX_TRY(check_device(device_id, nullptr));
    *found = 0;
    if (count < 127 || quantile >= 1.0f) return QMX_OK;                                                    // quantile.rs:42-44
    const uint64_t len = n_sample * dim;
    if (len < 4) return QMX_OK;                                                                            // :54-56
    uint64_t cut = std::min<uint64_t>((len - 1) / 2, (uint64_t)((float)n_sample * (1.0f - quantile) / 2.0f));   // :58-62 (f32 arithmetic, truncating cast)
    cut = std::max<uint64_t>(cut, 1);
    if (len - 2 * cut - 1 < 2) return QMX_OK;                                                              // :70-72
    DevBuf bin, btmp;
    const float *d_in = sample;
    int32_t rc = QMX_OK;
    float mm[2] = {0.f, 0.f};
    do {
        if (!is_device_ptr(sample)) {
            if ((rc = bin.reserve((size_t)len * 4)) != QMX_OK) break;
            if (hipMemcpy(bin.p, sample, (size_t)len * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
            d_in = (const float *)bin.p;
        }
        if ((rc = btmp.reserve((size_t)len * 4)) != QMX_OK) break;
        rc = launch_order_statistics_f32(nullptr, d_in, (float *)btmp.p, len, cut + 1, len - cut - 1, mm);
    } while (0);
    bin.release();
    btmp.release();
    QMX_TRY(rc);
    sq_params_from_min_max(distance, dim, mm[0], mm[1], out);
    *found = 1;
    return QMX_OK;
}

int32_t qmx_pq_encode(int32_t device_id, const qmx_pq_params *params, const float *in, uint64_t n, uint32_t dim, uint8_t *out_codes) {
    QMX_REQUIRE(params && params->centroids && (n == 0 || (in && out_codes)) && dim > 0, QMX_ERR_BAD_ARG, "bad argument");
    QMX_REQUIRE(params->chunk_size >= 1 && params->chunk_size <= 256 && params->n_centroids >= 1 && params->n_centroids <= 256,
                QMX_ERR_BAD_ARG, "chunk_size / n_centroids out of range");
    QMX_TRY(check_device(device_id, nullptr));
    if (n == 0) return QMX_OK;
    const uint32_t m = (dim + params->chunk_size - 1) / params->chunk_size;
    const size_t in_bytes = (size_t)n * dim * 4, out_bytes = (size_t)n * m, cbytes = (size_t)params->n_centroids * dim * 4;
    DevBuf bin, bout, bc;
    const float *d_in = in, *d_c = params->centroids;
    uint8_t *d_out = out_codes;
    int32_t rc = QMX_OK;
    do {
        if (!is_device_ptr(in)) {
            if ((rc = bin.reserve(in_bytes)) != QMX_OK) break;
            if (hipMemcpy(bin.p, in, in_bytes, hipMemcpyHostToDevice) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
            d_in = (const float *)bin.p;
        }
        if (!is_device_ptr(params->centroids)) {
            if ((rc = bc.reserve(cbytes)) != QMX_OK) break;
            if (hipMemcpy(bc.p, params->centroids, cbytes, hipMemcpyHostToDevice) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
            d_c = (const float *)bc.p;
        }
        const bool out_dev = is_device_ptr(out_codes);
        if (!out_dev) {
            if ((rc = bout.reserve(out_bytes)) != QMX_OK) break;
            d_out = (uint8_t *)bout.p;
        }
        if ((rc = launch_pq_encode(nullptr, dim, *params, d_c, d_in, n, d_out)) != QMX_OK) break;
        if (!out_dev && hipMemcpy(out_codes, d_out, out_bytes, hipMemcpyDeviceToHost) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
        if (hipDeviceSynchronize() != hipSuccess) rc = QMX_ERR_OTHER;
    } while (0);
    bin.release(); bout.release(); bc.release();
    return rc;
}


}  // extern "C"

// api_search.hip — the C-ABI of include/qdrant_amd.h, brute-force top-k: the exact scans, the prefilters, the merge, the oversampled quantized search.
// (One of the api_*.hip translation units; what they share: api_internal.hpp.)
#include "api_internal.hpp"

extern "C" {



// triage aid (qmx_set_option("debug", 2)): synchronise after every stage of the split path and name it on stderr
static int32_t split_stage(qmx_query *q, const char *what) {
    if (option(OPT_DEBUG) < 2) return QMX_OK;
    hipError_t e = hipStreamSynchronize(q->stream);
    fprintf(stderr, "[qmx] split stage %-28s %s\n", what, e == hipSuccess ? "ok" : hipGetErrorString(e));
    fflush(stderr);
    return e == hipSuccess ? QMX_OK : QMX_ERR_OTHER;
}

// the verification pool of a search (kernels.hpp VerifyPool): SPLIT_VCAP entries per query of the batch, shared - behind qmx_query::sp_ver as
// [ids: cap][qsel: cap][off: nq][cnt: nq], exact scores in sp_vscores, the fill level in the plan block (byte 24: zeroed with it at the start of a search)
static int32_t verify_pool(qmx_query *q, unsigned char *plan, VerifyPool *vp) {
    const uint32_t cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((uint64_t)q->nq * SPLIT_VCAP, 262144), 1u << 26);
    QMX_TRY(q->sp_ver.reserve(((size_t)cap * 2 + (size_t)q->nq * 2) * 4));
    QMX_TRY(q->sp_vscores.reserve((size_t)cap * 4));
    uint32_t *b = (uint32_t *)q->sp_ver.p;
    vp->ids = b;
    vp->qsel = b + cap;
    vp->off = b + (size_t)2 * cap;
    vp->cnt = vp->off + q->nq;
    vp->used = (uint32_t *)(plan + 24);
    vp->cap = cap;
    const int64_t mx = option(OPT_VERIFY_MAX_PER_QUERY);
    vp->max_per_query = mx > 0 ? (uint32_t)std::min<int64_t>(mx, cap) : cap;
    return QMX_OK;
}
// |approximate - exact| <= band * |q| * max |row|, worst case, every term at its bound:
//   one product of f16-rounded operands (HALF copy): each operand within 2^-11 of its value -> (2^-10 + 2^-22) sum |q_i r_i| <= ... |q| |r|
//   three products of f16 pairs: x - (h + l) within 2^-22 |x|, the dropped l.l term 2^-22                    -> 3 * 2^-22
//   f32 accumulation of the matrix cores over dim terms: dim * 2^-23 * sum |terms| (a round-off of 2^-23 per addition covers
//   truncating adders as well), f16 subnormal flush of tiny elements: < 2^-27 sqrt(dim)
// both rounded up generously; the exact side carries no error (the survivors are re-scored by the reference-order kernel).
static inline float split_rel_band(bool half, uint32_t dim) {
    const float acc = (float)dim * 1.1920929e-7f;                      // dim * 2^-23
    return (half ? 9.765625e-4f + 9.5367432e-7f : 1.9073486e-6f) + acc;  // 2^-10 + 2^-20 | 2^-19
}

// ids of a strided sample of the candidates (rows 0, step, 2 step, ...): a sample that sees the whole block, whatever its order
__global__ void sample_ids_kernel(uint32_t *ids, uint32_t n, uint64_t step) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ids[i] = (uint32_t)((uint64_t)i * step);
}

// ---- PQ top-k of 4 and more queries over a large block: the 6-bit prefilter + exact verification (pq_prefilter.hip).  Same contract as the f32
// prefilter below: the lists are the exact scan's, a query whose lists overflow takes the exact scan alone. ----
constexpr uint32_t PQF_TILE = 256;          // queries per pass (64 four-query groups; the regroup kernel's histogram)
constexpr uint32_t PQF_WCAP = 512;          // candidates one wave may list per pass (expected: tens)
static int32_t pq_prefilter_enqueue(qmx_query *q, uint32_t top, uint64_t n_cand, qmx_scored_point *d_out, uint32_t *d_counts,
                                    const volatile uint8_t *is_stopped, qmx_counters *counters, bool timed) {
    const qmx_segment *s = q->seg;
    const SplitPlanLayout pl(q->nq);
    const uint32_t m = s->pq_m, m_pad = (m + 31) / 32 * 32;
    const uint32_t tile_max = std::min<uint32_t>(PQF_TILE, q->nq);
    const uint32_t grid_max = pq_prefilter_grid(s->num_cus, tile_max, nullptr);
    QMX_TRY(q->gthr.reserve((size_t)std::max<uint32_t>(q->nq_padded, PQF_TILE) * sizeof(uint64_t)));
    QMX_TRY(q->pq_table.reserve(pq_prefilter_table_bytes(m, tile_max) + (size_t)(PQF_TILE + 4) * 4));
    QMX_TRY(q->sp_f32.reserve(1024 * sizeof(float)));
    QMX_TRY(q->sp_cand.reserve((size_t)tile_max * SPLIT_CAND_CAP * sizeof(uint64_t)));
    QMX_TRY(q->sp_cnt.reserve((size_t)SPLIT_QT_MAX * 4));
    {   // (the grid of a smaller last tile may be larger than the first tile's: size for the worst over tile sizes 1..tile_max)
        uint32_t g = grid_max;
        for (uint32_t t = 4; t <= tile_max; t += 4) g = std::max(g, pq_prefilter_grid(s->num_cus, t, nullptr));
        QMX_TRY(q->sp_wl.reserve(pq_prefilter_wlists_bytes(g, PQF_WCAP)));
    }
    QMX_TRY(q->sp_plan.reserve(pl.bytes));
    unsigned char *plan = (unsigned char *)q->sp_plan.p;
    VerifyPool vp;
    QMX_TRY(verify_pool(q, plan, &vp));
    float *band = (float *)q->sp_f32.p;                       // [PQF_TILE] in units of the integer score
    q->last_counters = qmx_counters{};
    q->last_split = false;
    // the sample (as for the f32 prefilter): its k-th best exact score per query is a lower bound of the final k-th best
    const int sshift = (int)std::min<int64_t>(std::max<int64_t>(option(OPT_PRESCAN_SHIFT) - 2, 1), 20);
    const uint64_t S = std::min<uint64_t>(n_cand, std::max<uint64_t>(n_cand >> sshift, 8192));
    if (q->sp_sample_n != S || q->sp_sample_of != n_cand) {
        QMX_TRY(q->sp_sample.reserve((size_t)S * 4));
        ::qmx::clear_stale_error();
        hipLaunchKernelGGL(sample_ids_kernel, dim3((uint32_t)((S + 255) / 256)), dim3(256), 0, q->stream, (uint32_t *)q->sp_sample.p, (uint32_t)S, n_cand / S);
        QMX_HIP(hipGetLastError());
        q->sp_sample_n = S;
        q->sp_sample_of = n_cand;
    }
    const uint32_t *d_sample = (const uint32_t *)q->sp_sample.p;
    QMX_HIP(hipMemsetAsync(plan, 0, pl.zero_bytes, q->stream));
    uint32_t n_tiles = 0, launches = 0;
    for (uint32_t tile0 = 0; tile0 < q->nq; tile0 += PQF_TILE, ++n_tiles) {
        const uint32_t nq_tile = std::min<uint32_t>(PQF_TILE, q->nq - tile0);
        if (is_stopped && *is_stopped) {
            set_error("search cancelled");
            return QMX_ERR_CANCELLED;
        }
        uint64_t *gthr = (uint64_t *)q->gthr.p + tile0;
        ScanArgs a;
        fill_args(q, tile0, nq_tile, a);
        a.n_cand = n_cand;
        a.top = top;
        // 1. exact scores of the sample (the exact kernel's score mode over an id list) -> k-th best per query
        QMX_TRY(q->scores.reserve((size_t)nq_tile * S * sizeof(float)));
        const uint32_t SQT = tile_qt(s, q);
        for (uint32_t st0 = 0; st0 < nq_tile; st0 += SQT) {
            const uint32_t nq_sub = std::min<uint32_t>(SQT, nq_tile - st0);
            ScanArgs pre;
            fill_args(q, tile0 + st0, nq_sub, pre);
            pre.ids = d_sample;
            pre.n_cand = S;
            pre.top = 1;
            pre.scores = (float *)q->scores.p + (size_t)st0 * S;
            pre.scores_stride = S;
            uint32_t pgrid = 0;
            QMX_TRY(launch_scan(q, (int)pow2_ceil(nq_sub), SCAN_SCORES, pre, &pgrid));
            ++launches;
        }
        QMX_TRY(launch_custom_topk(q->stream, (const float *)q->scores.p, S, d_sample, a.del, nq_tile, top, d_out + (size_t)tile0 * top, d_counts + tile0, gthr));
        // 2. the 6-bit tables of the tile's query groups, thresholds and bands in units of the integer score
        int32_t *thr = (int32_t *)((unsigned char *)q->pq_table.p + pq_prefilter_table_bytes(m, tile_max));
        QMX_TRY(launch_pq_lut8(q->stream, a.queries, q->q_stride, nq_tile, m, s->pq.n_centroids, gthr, q->pq_table.p, thr, band));
        QMX_HIP(hipMemsetAsync(q->sp_cnt.p, 0, (size_t)SPLIT_QT_MAX * 4, q->stream));
        // 3. the approximate scan of the whole block over the rotated copy
        uint32_t grid = 0;
        size_t slot = 0;
        if (timed) QMX_TRY(timing_begin(q, &slot));
        QMX_TRY(launch_pq_prefilter(q->stream, a, s->d_pq_rot, q->pq_table.p, thr, nq_tile, s->num_cus, q->sp_wl.p, PQF_WCAP, &grid));
        q->last_kernel = last_noted_kernel();
        if (timed) QMX_TRY(timing_end(q, slot));
        // 4. per-wave lists -> per-query lists (deleted rows dropped), then the rows worth an exact score
        int *tile_ovf = (int *)(plan + pl.tile_ovf) + n_tiles;
        QMX_TRY(launch_regroup_lists(q->stream, a.del, (const unsigned char *)q->sp_wl.p + pq_prefilter_wlists_counts_bytes(grid), (const uint32_t *)q->sp_wl.p, PQF_WCAP,
                                     grid * 16, (uint64_t *)q->sp_cand.p, (uint32_t *)q->sp_cnt.p, SPLIT_CAND_CAP, tile_ovf));
        QMX_TRY(launch_split_select(q->stream, (const uint64_t *)q->sp_cand.p, (const uint32_t *)q->sp_cnt.p, SPLIT_CAND_CAP, band, nq_tile, top, vp, tile0,
                                    tile_ovf, (uint32_t *)(plan + pl.ovf_q) + tile0, (SplitStats *)plan));
        launches += 6;
    }
    const void *pf_kernel = q->last_kernel;
    // 5. exact scores of the survivors (pq_pair_kernel: score_point_sse's order), sorted by (score, lower id first)
    PairSel sel{vp.qsel, 0, nullptr, vp.used};
    QMX_TRY(score_pairs_device(q, sel, vp.ids, vp.cap, (float *)q->sp_vscores.p, false));
    QMX_TRY(launch_sort_scored(q->stream, (const float *)q->sp_vscores.p, vp.ids, vp.cnt, 0, q->nq, top, d_out, d_counts, vp.off));
    // 6. the exact scan of the queries whose lists overflowed, and of those only (the kernels start, read the count and return when it is zero)
    uint32_t *ovf_list = (uint32_t *)(plan + pl.list);
    QMX_TRY(launch_split_plan(q->stream, (const uint32_t *)(plan + pl.ovf_q), q->nq, (const uint64_t *)q->gthr.p, ovf_list, (uint64_t *)(plan + pl.gthr_packed),
                              pl.list_cap, (uint32_t *)(plan + pl.count), (int *)(plan + pl.run16), (int *)(plan + pl.run64), pl.n_run64, (SplitStats *)plan, nullptr, 0,
                              nullptr));
    {
        ScanArgs a;
        fill_args(q, 0, q->nq, a);
        a.n_cand = n_cand;
        a.top = top;
        a.q_map = ovf_list;
        a.run_if = (const int *)(plan + pl.count);
        const uint64_t want = (n_cand + 1023) / 1024, cap = std::max<uint64_t>(1, ((uint64_t)s->num_cus * 2 + q->nq - 1) / q->nq);
        uint32_t slabs = (uint32_t)std::max<uint64_t>(1, std::min(want, cap));
        QMX_TRY(q->partial.reserve((size_t)slabs * q->nq * top * sizeof(uint64_t)));
        a.partial = (uint64_t *)q->partial.p;
        a.partial_qt = q->nq;
        QMX_TRY(launch_scan_pq(q->stream, SCAN_TOPK, a, s->num_cus, &slabs));
        QMX_TRY(launch_merge_keys(q->stream, (const uint64_t *)q->partial.p, slabs, q->nq, q->nq, top, d_out, d_counts, top, 0, nullptr, a.run_if, ovf_list,
                                  slabs * q->nq));
        launches += 5;
    }
    q->last_kernel = pf_kernel;
    q->last_split = true;
    q->last_pq = true;
    {
        qmx_counters &c = q->last_counters;
        c.vectors_scored = (uint64_t)q->nq * n_cand;
        // the rotated copy once per four-query group (all but the first find it in L2) + the sample's rows per query
        c.bytes_read = (uint64_t)((q->nq + 3) / 4) * n_cand * m_pad + (uint64_t)q->nq * S * s->row_bytes;
        c.kernel_launches = launches;
        c.prefilter_queries = q->nq;
        q->last_row_bytes = s->row_bytes;
        q->last_n_cand = n_cand;
        if (counters) *counters = c;
    }
    return QMX_OK;
}

// ---- 4-bit TurboQuant / scalar-int8 top-k of 33 and more queries over a large block: 128 queries per pass of the codes (scan_tq4w.hip, scan_sqw.hip).  The pass's scores are exact;
// what it shares with the prefilters is the plumbing: a sample's k-th best score admits the candidates, per-wave lists are regrouped per query, the k best
// keys of a query (band 0: every tie of the k-th score with them) are re-scored by the pair kernel and sorted; a query whose lists overflowed (masses of
// equal scores, a sample that is all deleted) takes the 32-query scan - conditional launches that read their flag and return. ----
constexpr uint32_t TQW_FQT = 32;
static int32_t wide_exact_enqueue(qmx_query *q, uint32_t top, uint64_t n_cand, qmx_scored_point *d_out, uint32_t *d_counts, const volatile uint8_t *is_stopped,
                                  qmx_counters *counters, bool timed) {
    const qmx_segment *s = q->seg;
    const bool sq = s->dtype == QMX_DTYPE_SQ_U8;
    // TurboQuant, option tq_wide_high_digit: the pass multiplies the queries' HIGH digits only; its scores are within band[q] of the exact ones, the selection
    // keeps what an exact score >= T could hide behind (approximate >= max(T - band, A_k - 2 band)) and the pair kernel re-scores that
    const bool tq_high = !sq && option(OPT_TQ_WIDE_HIGH_DIGIT) != 0;
    const SplitPlanLayout pl(q->nq, TQW_FQT);
    const uint32_t grid_cap = (uint32_t)s->num_cus * 8;
    QMX_TRY(q->gthr.reserve((size_t)std::max<uint32_t>(q->nq_padded, SPLIT_QT) * sizeof(uint64_t)));
    QMX_TRY(q->sp_bq.reserve(sq ? sqw_query_bytes(s->scan_dim) : tq4w_query_bytes(s->scan_dim)));
    QMX_TRY(q->sp_f32.reserve(1024 * sizeof(float)));
    QMX_TRY(q->sp_cand.reserve((size_t)SPLIT_QT * SPLIT_CAND_CAP * sizeof(uint64_t)));
    QMX_TRY(q->sp_cnt.reserve((size_t)SPLIT_QT_MAX * 4));
    QMX_TRY(q->sp_wl.reserve(sq ? sqw_wlists_bytes(s->num_cus) : tq4w_wlists_bytes(s->num_cus)));
    QMX_TRY(q->sp_plan.reserve(pl.bytes));
    QMX_TRY(q->sp_fq.reserve((size_t)pl.list_cap * q->q_stride));
    QMX_TRY(q->partial.reserve((size_t)grid_cap * TQW_FQT * std::min<uint32_t>(top, MAX_TOP_FAST) * sizeof(uint64_t)));
    unsigned char *plan = (unsigned char *)q->sp_plan.p;
    VerifyPool vp;
    QMX_TRY(verify_pool(q, plan, &vp));
    float *qinfo = (float *)q->sp_f32.p, *band = qinfo + 512;      // band: zero - the pass's scores are the exact ones - or infinite: a query without a usable bound
    int32_t *thr_i = (int32_t *)(qinfo + 768);      // [256]: the bounds on the whole sum, then on the high sum alone
    q->last_counters = qmx_counters{};
    q->last_split = false;
    // the sample: every 128-th row (prescan_shift - 3; at least 8 192 rows): its k-th best score leaves ~128 k candidates per query to the pass - the pass's
    // epilogue pays per candidate (10 M x 768, 128 queries: 2.07 / 2.00 / 2.01 ms with every 256-th / 128-th / 64-th row, whose own scores cost more).
    // SQ: every 256-th (its epilogue is one integer add and compare per pair; the sample's scores and their selection are 0.19 ms of the search at 1 / 128)
    const int sshift = (int)std::min<int64_t>(std::max<int64_t>(option(OPT_PRESCAN_SHIFT) - (sq ? 2 : 3), 1), 20);
    const uint64_t S = std::min<uint64_t>(n_cand, std::max<uint64_t>(n_cand >> sshift, 8192));
    if (q->sp_sample_n != S || q->sp_sample_of != n_cand) {
        QMX_TRY(q->sp_sample.reserve((size_t)S * 4));
        ::qmx::clear_stale_error();
        hipLaunchKernelGGL(sample_ids_kernel, dim3((uint32_t)((S + 255) / 256)), dim3(256), 0, q->stream, (uint32_t *)q->sp_sample.p, (uint32_t)S, n_cand / S);
        QMX_HIP(hipGetLastError());
        q->sp_sample_n = S;
        q->sp_sample_of = n_cand;
    }
    const uint32_t *d_sample = (const uint32_t *)q->sp_sample.p;
    QMX_HIP(hipMemsetAsync(plan, 0, pl.zero_bytes, q->stream));
    uint32_t n_tiles = 0, launches = 2;
    const void *wide_kernel = nullptr;
    for (uint32_t tile0 = 0; tile0 < q->nq; tile0 += SPLIT_QT, ++n_tiles) {
        const uint32_t nq_tile = std::min<uint32_t>(SPLIT_QT, q->nq - tile0);
        if (is_stopped && *is_stopped) {
            set_error("search cancelled");
            return QMX_ERR_CANCELLED;
        }
        uint64_t *gthr = (uint64_t *)q->gthr.p + tile0;
        ScanArgs a;
        fill_args(q, tile0, nq_tile, a);
        a.n_cand = n_cand;
        a.top = top;
        // 1. exact scores of the sample -> the k-th best of each query = a lower bound of its final k-th best
        QMX_TRY(q->scores.reserve((size_t)nq_tile * S * sizeof(float)));
        QMX_TRY(score_matrix_enqueue(q, tile0, nq_tile, d_sample, S, (float *)q->scores.p, S, &launches));
        QMX_TRY(launch_custom_topk(q->stream, (const float *)q->scores.p, S, d_sample, a.del, nq_tile, top, d_out + (size_t)tile0 * top, d_counts + tile0, gthr));
        // 2. the queries' codes / digits as operand images, the integer reject bounds
        if (sq) QMX_TRY(launch_sqw_pack(q->stream, a, gthr, s->sq_off_absmax, q->sp_bq.p, thr_i, qinfo, band, (uint32_t *)q->sp_cnt.p, SPLIT_QT_MAX));
        else QMX_TRY(launch_tq4w_pack(q->stream, a, gthr, s->tq_sf_min, s->tq_sf_max, s->tq_l2_min, s->tq_l2_max, s->tq_c1, tq_high ? 1 : 0, q->sp_bq.p, thr_i, qinfo, band,
                                      (uint32_t *)q->sp_cnt.p, SPLIT_QT_MAX));
        // 3. the pass
        uint32_t grid = 0;
        size_t slot = 0;
        if (timed) QMX_TRY(timing_begin(q, &slot));
        if (sq) QMX_TRY(launch_scan_sqw(q->stream, a, q->sp_bq.p, thr_i, s->d_sq_bi, qinfo, s->num_cus, q->sp_wl.p, &grid));
        else QMX_TRY(launch_scan_tq4w(q->stream, a, q->sp_bq.p, thr_i, qinfo, tq_high ? band : nullptr, s->num_cus, q->sp_wl.p, &grid));
        wide_kernel = last_noted_kernel();
        if (timed) QMX_TRY(timing_end(q, slot));
        // 4. per-wave lists -> per-query lists (deleted rows dropped), then the k best keys of each query
        int *tile_ovf = (int *)(plan + pl.tile_ovf) + n_tiles;
        QMX_TRY(launch_regroup_lists(q->stream, a.del, (const unsigned char *)q->sp_wl.p + (sq ? sqw_wlists_counts_bytes(s->num_cus) : tq4w_wlists_counts_bytes(s->num_cus)), (const uint32_t *)q->sp_wl.p,
                                     sq ? sqw_wcap() : tq4w_wcap(),
                                     grid * 8, (uint64_t *)q->sp_cand.p, (uint32_t *)q->sp_cnt.p, SPLIT_CAND_CAP, tile_ovf));
        QMX_TRY(launch_split_select(q->stream, (const uint64_t *)q->sp_cand.p, (const uint32_t *)q->sp_cnt.p, SPLIT_CAND_CAP, band, nq_tile, top, vp, tile0, tile_ovf,
                                    (uint32_t *)(plan + pl.ovf_q) + tile0, (SplitStats *)plan, tq_high ? qinfo + 3 * SPLIT_QT : nullptr, tq_high));
        launches += 6;
    }
    // 5. the selected rows through the pair kernel (the same bits), sorted by (score, lower id first)
    PairSel sel{vp.qsel, 0, nullptr, vp.used};
    QMX_TRY(score_pairs_device(q, sel, vp.ids, vp.cap, (float *)q->sp_vscores.p, false));
    QMX_TRY(launch_sort_scored(q->stream, (const float *)q->sp_vscores.p, vp.ids, vp.cnt, 0, q->nq, top, d_out, d_counts, vp.off));
    // 6. the 32-query scan of the queries whose lists overflowed, packed: one 16-query pass when 1..16 of them, passes of 32 otherwise
    uint32_t *ovf_list = (uint32_t *)(plan + pl.list);
    uint64_t *gthr_packed = (uint64_t *)(plan + pl.gthr_packed);
    QMX_TRY(launch_split_plan(q->stream, (const uint32_t *)(plan + pl.ovf_q), q->nq, (const uint64_t *)q->gthr.p, ovf_list, gthr_packed, pl.list_cap,
                              (uint32_t *)(plan + pl.count), (int *)(plan + pl.run16), (int *)(plan + pl.run64), pl.n_run64, (SplitStats *)plan, q->d_queries,
                              q->q_stride, q->sp_fq.p));
    for (uint32_t pass = 0; pass <= pl.n_run64; ++pass) {
        if (pass && q->nq <= 16) break;
        const uint32_t p0 = pass ? (pass - 1) * TQW_FQT : 0;
        const uint32_t nq_sub = pass ? std::min<uint32_t>(TQW_FQT, q->nq - p0) : std::min<uint32_t>(16, q->nq);
        const int *run_if = pass ? (const int *)(plan + pl.run64) + (pass - 1) : (const int *)(plan + pl.run16);
        ScanArgs a;
        fill_args(q, 0, nq_sub, a);
        a.queries = (const char *)q->sp_fq.p + (size_t)p0 * q->q_stride;
        a.n_cand = n_cand;
        a.top = top;
        a.partial = (uint64_t *)q->partial.p;
        a.gthr = gthr_packed + p0;
        a.run_if = run_if;
        const int fqt = (int)std::max<uint32_t>(16, pow2_ceil(nq_sub));
        a.partial_qt = (uint32_t)fqt;
        uint32_t grid = grid_cap;
        if (sq) QMX_TRY(launch_scan_sq_mfma(q->stream, fqt, SCAN_TOPK, a, s->num_cus, &grid));
        else QMX_TRY(launch_scan_tq_mfma(q->stream, fqt, SCAN_TOPK, a, s->num_cus, &grid));
        QMX_TRY(launch_merge_keys(q->stream, (const uint64_t *)q->partial.p, grid, (uint32_t)fqt, nq_sub, top, d_out, d_counts, top, 0, nullptr, run_if, ovf_list + p0));
        launches += 2;
    }
    q->last_kernel = wide_kernel;
    q->last_split = true;
    q->last_pq = false;
    q->last_fqt = TQW_FQT;
    {
        qmx_counters &c = q->last_counters;
        c.vectors_scored = (uint64_t)q->nq * n_cand;
        c.bytes_read = (uint64_t)n_tiles * n_cand * s->row_bytes + (uint64_t)((q->nq + MAX_QT_MFMA - 1) / MAX_QT_MFMA) * S * s->row_bytes;
        c.kernel_launches = launches;
        c.prefilter_queries = q->nq;
        q->last_row_bytes = s->row_bytes;
        q->last_n_cand = n_cand;
        if (counters) *counters = c;
    }
    return QMX_OK;
}

int32_t search_enqueue(qmx_query *q, uint32_t top, const uint32_t *d_ids, uint64_t n_ids,
                              qmx_scored_point *d_out, uint32_t *d_counts, const volatile uint8_t *is_stopped,
                              qmx_counters *counters, bool timed) {
    const qmx_segment *s = q->seg;
    const uint64_t n_cand = d_ids ? n_ids : s->scan_rows();
    if (tq_l1(s)) {     // the score matrix (tiles of queries: at most 256 MiB of scores at a time), then one block per query selects its k best live candidates
        q->last_counters = qmx_counters{};
        q->last_split = false;
        ScanArgs a;
        fill_args(q, 0, q->nq, a);
        const uint32_t qtile = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(q->nq, (1ull << 26) / std::max<uint64_t>(n_cand, 1)));
        QMX_TRY(q->scores.reserve((size_t)qtile * std::max<uint64_t>(n_cand, 1) * sizeof(float)));
        for (uint32_t q0 = 0; q0 < q->nq; q0 += qtile) {
            if (is_stopped && *is_stopped) {
                set_error("search cancelled");
                return QMX_ERR_CANCELLED;
            }
            const uint32_t nq_tile = std::min<uint32_t>(qtile, q->nq - q0);
            QMX_TRY(tq_l1_scores_device(q, q0, nq_tile, d_ids, n_cand, (float *)q->scores.p, n_cand, nullptr));
            QMX_TRY(launch_custom_topk(q->stream, (const float *)q->scores.p, n_cand, d_ids, a.del, nq_tile, top, d_out + (size_t)q0 * top, d_counts + q0));
            if (counters) counters->kernel_launches += 1 + 3 * (uint32_t)((n_cand + 65535) / 65536);
        }
        if (counters) {
            counters->vectors_scored += (uint64_t)q->nq * n_cand;
            counters->bytes_read += (uint64_t)((q->nq + qtile - 1) / qtile) * n_cand * s->row_bytes;
        }
        return QMX_OK;
    }
    if (s->dtype == QMX_DTYPE_PQ && s->d_pq_rot && !d_ids && top <= MAX_TOP_FAST && n_cand >= (1u << 18) && !option(OPT_NO_PQ_PREFILTER) &&
        q->nq >= (uint32_t)std::max<int64_t>(1, option(OPT_PQ_PREFILTER_MIN_QUERIES)))
        return pq_prefilter_enqueue(q, top, n_cand, d_out, d_counts, is_stopped, counters, timed);
    if (s->dtype == QMX_DTYPE_TQ && s->tq_wide && !d_ids && top <= MAX_TOP_FAST && n_cand >= (1u << 18) && mfma_scan_ok(s) && option(OPT_TQ_WIDE_MIN_QUERIES) > 0 &&
        q->nq >= (uint32_t)option(OPT_TQ_WIDE_MIN_QUERIES) && (size_t)MAX_QT_MFMA * q->q_stride <= 150 * 1024) {
        ScanArgs probe;
        fill_args(q, 0, std::min<uint32_t>(q->nq, SPLIT_QT), probe);
        probe.n_cand = n_cand;
        probe.top = top;
        if (tq4w_shape_ok(probe)) return wide_exact_enqueue(q, top, n_cand, d_out, d_counts, is_stopped, counters, timed);
    }
    if (s->dtype == QMX_DTYPE_SQ_U8 && s->sq_wide && !d_ids && top <= MAX_TOP_FAST && n_cand >= (1u << 18) && mfma_scan_ok(s) && option(OPT_SQ_WIDE_MIN_QUERIES) > 0 &&
        q->nq >= (uint32_t)option(OPT_SQ_WIDE_MIN_QUERIES)) {
        ScanArgs probe;
        fill_args(q, 0, std::min<uint32_t>(q->nq, SPLIT_QT), probe);
        probe.n_cand = n_cand;
        probe.top = top;
        if (sqw_shape_ok(probe)) return wide_exact_enqueue(q, top, n_cand, d_out, d_counts, is_stopped, counters, timed);
    }
    // partial lists: one per block; bound the grid by what the buffer holds
    const uint32_t grid_cap = (uint32_t)s->num_cus * 8;
    // f32 dot / cosine rows of 256, 512 or 768 floats, whole block: 64 queries per pass (scan_mfma16.hip); everything else 32 / 16
    const bool q64 = s->dtype == QMX_DTYPE_F32 && mfma_scan_ok(s) && q->nq > MAX_QT_MFMA && mfma16_dim_ok(64, s->dim) &&
                     !option(OPT_NO_MFMA16);
    // ... and rows of 1024 .. 1536 floats 32 per pass: that kernel keeps the queries in registers, not in an LDS tile (tile_qt's limit)
    const bool q32 = s->dtype == QMX_DTYPE_F32 && mfma_scan_ok(s) && q->nq > MAX_QT && s->dim > 768 && mfma16_dim_ok(32, s->dim) &&
                     !option(OPT_NO_MFMA16);
    // more than 64 queries over a large f32 dot / cosine block: 128 per pass through the f16-split matrix-core prefilter, the survivors
    // re-scored exactly (scan_split.hip); the result is the exact scan's, bit for bit
    // ... and with a derived copy of the block (QMX_SEG_HALF_COPY / QMX_SEG_SPLIT_COPY) that path serves EVERY batch size: it streams 2 (4) bytes
    // per element instead of 4 and is HBM-bound whatever the number of queries (10 M x 768: 3.0 ms per pass against 4.4 ms for the f32 stream)
    // (rows of up to 768 floats: conditional exact passes of 64 queries; up to 2 048 floats - 1 024, 1 536: the 32-query shape - only over a derived copy)
    const uint32_t split_fqt = split_fallback_qt(s->dim);
    const bool split_dims = s->dtype == QMX_DTYPE_F32 && mfma_scan_ok(s) && (mfma16_dim_ok(64, s->dim) || (s->d_rows_split && split_fqt != 0)) && !option(OPT_NO_MFMA16);
    const bool split = split_dims && (q64 || (s->d_rows_split && q->nq >= (uint32_t)std::max<int64_t>(1, option(OPT_SPLIT_MIN_QUERIES)))) && s->split_stats &&
                       !d_ids && top <= MAX_TOP_FAST && n_cand >= (1u << 18) && s->dim % 128 == 0 && s->row_stride % 16 == 0 && !option(OPT_NO_SPLIT_SCAN);
    // the 256-query shape halves the bytes streamed per query; a batch that does not fill it is served by the 128-query shape (less matrix work)
    const uint32_t split_qt = (split && s->d_rows_split && s->split_half && q->nq > SPLIT_QT && !option(OPT_NO_SPLIT256)) ? SPLIT_QT_MAX : SPLIT_QT;
    const uint32_t TQ = split ? split_qt : q64 ? MAX_QT_TOPK : q32 ? MAX_QT_MFMA : tile_qt(s, q);
    const uint32_t ptop_max = std::min<uint32_t>(top, MAX_TOP_FAST);
    const uint32_t n_pass = (top + MAX_TOP_FAST - 1) / MAX_TOP_FAST;
    QMX_TRY(q->partial.reserve((size_t)grid_cap * std::min<uint32_t>(TQ, MAX_QT_TOPK) * ptop_max * sizeof(uint64_t)));
    if (n_pass > 1) QMX_TRY(q->bounds.reserve((size_t)TQ * sizeof(uint64_t)));
    QMX_TRY(q->gthr.reserve((size_t)std::max<uint32_t>(q->nq_padded, SPLIT_QT_MAX) * sizeof(uint64_t)));
    // ---- split passes first (their verification and, if ever needed, the exact fallback run once for all of them afterwards) ----
    std::vector<std::pair<uint32_t, uint32_t>> split_tiles;      // (tile0, nq_tile)
    float *sp_qnorm = nullptr, *sp_thr = nullptr, *sp_band = nullptr, *sp_scales = nullptr, *sp_qmax = nullptr;
    const SplitPlanLayout pl(q->nq, split_fqt ? split_fqt : SPLIT_FQT);
    unsigned char *plan = nullptr;
    VerifyPool vp{};
    q->last_counters = qmx_counters{};
    q->last_split = false;
    if (split) {
        QMX_TRY(q->sp_bq.reserve(split_query_bytes(s->dim)));
        QMX_TRY(q->sp_f32.reserve(1280 * sizeof(float)));
        QMX_TRY(q->sp_cand.reserve((size_t)split_qt * SPLIT_CAND_CAP * sizeof(uint64_t)));
        QMX_TRY(q->sp_cnt.reserve((size_t)SPLIT_QT_MAX * 4));
        if (s->d_rows_split) QMX_TRY(q->sp_wl.reserve(split_wlists_bytes(s->num_cus)));
        QMX_TRY(q->sp_plan.reserve(pl.bytes));
        QMX_TRY(verify_pool(q, (unsigned char *)q->sp_plan.p, &vp));
        QMX_TRY(q->sp_fq.reserve((size_t)pl.list_cap * q->q_stride));
        if (s->split_i8) {
            QMX_TRY(q->sp_probe.reserve((size_t)q->nq * (split_i8_probe() + 1) * 4));
            QMX_TRY(q->sp_pscores.reserve((size_t)q->nq * split_i8_probe() * 4));
            // (no memset of the probe counts: the gather of a tile reads the counts of the tiles up to it - its own, written by the probe kernel in front
            // of it, and the earlier ones', emptied by their bound kernels)
        }
        plan = (unsigned char *)q->sp_plan.p;
        float *f = (float *)q->sp_f32.p;
        sp_qnorm = f; sp_thr = f + 256; sp_band = f + 512; sp_scales = f + 768; sp_qmax = f + 1024;
        // the sample: every (n_cand / S)-th row, S = n_cand / 256 (at least 8192): its k-th best leaves ~256 k candidates per query to the
        // main pass, at 1 / 256 of the pass's row traffic for the sample's exact scores (measured on C2: 1/128 .. 1/512 are equally good)
        // ("prescan_shift" - 2: the option of the exact scans' prefix pre-scan, 10 by default, moves this sample with it)
        // (with the derived copy: one more halving - 8 192 rows of a 10 M block are one tile per row stream of the sample scan, and the
        // refine step after the first sixteenth of the block owns the threshold anyway: 26 us of the step, measured)
        const int sshift = (int)std::min<int64_t>(std::max<int64_t>(option(OPT_PRESCAN_SHIFT) + (s->d_rows_split ? 1 : -2), 1), 20);
        // (a 2 048-row sample lets the four query tiles of a 128-query batch run side by side - 22 us instead of 63 for the pre-scan - but its weaker threshold
        // triples the candidates of the first launch: regroup, refine and select together give the 40 us back, measured; 8 192 stays)
        const uint64_t S = std::min<uint64_t>(n_cand, std::max<uint64_t>(n_cand >> sshift, 8192));
        if (q->sp_sample_n != S || q->sp_sample_of != n_cand) {
            QMX_TRY(q->sp_sample.reserve((size_t)S * 4));
            ::qmx::clear_stale_error();
            hipLaunchKernelGGL(sample_ids_kernel, dim3((uint32_t)((S + 255) / 256)), dim3(256), 0, q->stream, (uint32_t *)q->sp_sample.p, (uint32_t)S, n_cand / S);
            QMX_HIP(hipGetLastError());
            q->sp_sample_n = S;
            q->sp_sample_of = n_cand;
        }
        QMX_HIP(hipMemsetAsync(plan, 0, pl.zero_bytes, q->stream));
    }
    for (uint32_t tile0 = 0; tile0 < q->nq; tile0 += TQ) {
        const uint32_t nq_tile = std::min<uint32_t>(TQ, q->nq - tile0);
        if (split && (nq_tile > MAX_QT_TOPK || s->d_rows_split)) {
            if (is_stopped && *is_stopped) {
                set_error("search cancelled");
                return QMX_ERR_CANCELLED;
            }
            const uint32_t S = (uint32_t)q->sp_sample_n;
            const uint32_t *d_sample = (const uint32_t *)q->sp_sample.p;
            uint64_t *gthr = (uint64_t *)q->gthr.p + tile0;
            ScanArgs a;
            fill_args(q, tile0, nq_tile, a);
            a.n_cand = n_cand;
            a.top = top;
            // 1. exact scores of the sample -> the k-th best of each query = a lower bound of its final k-th best
            QMX_TRY(q->scores.reserve((size_t)nq_tile * S * sizeof(float)));
            QMX_TRY(score_matrix_enqueue(q, tile0, nq_tile, d_sample, S, (float *)q->scores.p, S, nullptr));
            QMX_TRY(launch_custom_topk(q->stream, (const float *)q->scores.p, S, d_sample, a.del, nq_tile, top, d_out + (size_t)tile0 * top, d_counts + tile0, gthr));
            QMX_TRY(split_stage(q, "prescan"));
            if (s->split_i8) {
                // 2'. the int8 copy: codes, scales, worst-case bands, thresholds from the sample's exact k-th best (sp_f32: qnorm -> T_exact, qmax -> the scales)
                float *sp_texact = sp_qnorm, *sp_qscale = sp_qmax;
                QMX_TRY(launch_split_i8_pack(q->stream, (const float *)q->enc.p + (size_t)tile0 * s->dim, nq_tile, s->dim, s->d_i8_scale, gthr, s->d_i8_stats,
                                             s->row_norm_max, q->sp_bq.p, sp_qscale, sp_band, sp_thr, sp_texact, (uint32_t *)q->sp_cnt.p, SPLIT_QT_MAX));
                QMX_TRY(split_stage(q, "int8 pack"));
                // 3'. the strided sixteenth, then the rest; after each launch the exact scores of the k best candidates so far renew the bound
                const uint32_t np = split_i8_probe();
                uint32_t *probe_ids = (uint32_t *)q->sp_probe.p, *probe_cnt = probe_ids + (size_t)q->nq * np;
                int *tile_ovf = (int *)(plan + pl.tile_ovf) + split_tiles.size();
                // (which tiles the first launch takes: every 16th.  Its candidates are admitted on the SAMPLE's bound - a hundred times those of the main launch
                // per tile -, so the first launch is bound by its candidate lists, not by its stream; strides of 8 .. 64 measured the same step time,
                // profiles/r5_i8_sample_stride.md)
                const uint32_t sstride = 16;
                for (uint32_t ph = 1; ph <= 2; ++ph) {
                    const uint32_t phase = ph | (sstride << 8);
                    size_t slot = 0;
                    if (timed) QMX_TRY(timing_begin(q, &slot));
                    QMX_TRY(launch_scan_i8copy(q->stream, a, q->sp_bq.p, sp_qscale, sp_thr, s->num_cus, s->d_rows_split, q->sp_wl.p, phase));
                    q->last_kernel = last_noted_kernel();
                    if (timed) QMX_TRY(timing_end(q, slot));
                    QMX_TRY(launch_split_regroup(q->stream, a, q->sp_wl.p, s->num_cus, (uint64_t *)q->sp_cand.p, (uint32_t *)q->sp_cnt.p, SPLIT_CAND_CAP, tile_ovf,
                                                 phase, SPLIT_QT));
                    QMX_TRY(launch_split_i8_probe(q->stream, (const uint64_t *)q->sp_cand.p, (const uint32_t *)q->sp_cnt.p, SPLIT_CAND_CAP, sp_band, nq_tile, top, tile_ovf,
                                                  probe_ids + (size_t)tile0 * np, probe_cnt + tile0));
                    const void *scan_kernel = q->last_kernel;
                    PairSel psel{nullptr, np, probe_cnt};
                    QMX_TRY(score_pairs_device(q, psel, probe_ids, (uint64_t)(tile0 + nq_tile) * np, (float *)q->sp_pscores.p, false));
                    q->last_kernel = scan_kernel;
                    QMX_TRY(launch_split_i8_bound(q->stream, (const float *)q->sp_pscores.p + (size_t)tile0 * np, probe_cnt + tile0, nq_tile, top, sp_band, sp_qscale,
                                                  sp_thr, sp_texact));
                }
                QMX_TRY(split_stage(q, "int8 scan"));
                // 4'. the rows worth an exact score: approximate score >= T_exact - band
                QMX_TRY(launch_split_select(q->stream, (const uint64_t *)q->sp_cand.p, (const uint32_t *)q->sp_cnt.p, SPLIT_CAND_CAP, sp_band, nq_tile, top, vp, tile0,
                                            tile_ovf, (uint32_t *)(plan + pl.ovf_q) + tile0, (SplitStats *)plan, sp_texact));
                QMX_TRY(split_stage(q, "select"));
                split_tiles.push_back({tile0, nq_tile});
                if (counters) counters->kernel_launches += 13;
                continue;
            }
            // 2. the batch's queries split into f16 pairs; thresholds and bands in accumulator / score units
            const float row_scale = split_row_scale(s->row_maxabs);
            const int half = s->split_half ? 1 : 0;
            const uint32_t tqt = nq_tile > SPLIT_QT ? SPLIT_QT_MAX : SPLIT_QT;      // the shape of THIS tile (a remainder of <= 128 queries takes the 128 shape)
            QMX_TRY(launch_split_pack_queries(q->stream, (const float *)q->enc.p + (size_t)tile0 * s->dim, nq_tile, s->dim, sp_qmax, sp_qnorm, q->sp_bq.p, half, tqt));
            QMX_TRY(launch_split_thresholds(q->stream, gthr, sp_qnorm, sp_qmax, nq_tile, split_rel_band(half, s->dim), s->row_norm_max, row_scale, sp_scales, sp_thr,
                                            sp_band, tqt, (uint32_t *)q->sp_cnt.p, SPLIT_QT_MAX));
            QMX_TRY(split_stage(q, "pack + thresholds"));
            // 3. the approximate scan of the whole block
            // over a derived copy in two launches: the strided sixteenth of the tiles first, whose k-th best approximate score tightens the
            // threshold of the other fifteen (sp_refine_kernel): ~16 k candidates per query instead of ~10 k x 16 from the sample's threshold alone
            for (uint32_t phase = s->d_rows_split ? 1 : 0; phase <= (s->d_rows_split ? 2u : 0u); ++phase) {
                size_t slot = 0;
                if (timed) QMX_TRY(timing_begin(q, &slot));
                QMX_TRY(launch_scan_f32_split(q->stream, a, q->sp_bq.p, row_scale, sp_scales, sp_thr, (uint64_t *)q->sp_cand.p, (uint32_t *)q->sp_cnt.p,
                                              SPLIT_CAND_CAP, s->num_cus, s->d_rows_split, half, q->sp_wl.p, phase, tqt));
                q->last_kernel = last_noted_kernel();
                if (timed) QMX_TRY(timing_end(q, slot));
                if (s->d_rows_split)
                    QMX_TRY(launch_split_regroup(q->stream, a, q->sp_wl.p, s->num_cus, (uint64_t *)q->sp_cand.p, (uint32_t *)q->sp_cnt.p, SPLIT_CAND_CAP,
                                                 (int *)(plan + pl.tile_ovf) + split_tiles.size(), phase, tqt));
                if (phase == 1)
                    QMX_TRY(launch_split_refine(q->stream, (const uint64_t *)q->sp_cand.p, (const uint32_t *)q->sp_cnt.p, SPLIT_CAND_CAP, sp_band, nq_tile, top,
                                                sp_scales, sp_thr));
            }
            QMX_TRY(split_stage(q, "split kernel"));
            // 4. the rows worth an exact score
            QMX_TRY(launch_split_select(q->stream, (const uint64_t *)q->sp_cand.p, (const uint32_t *)q->sp_cnt.p, SPLIT_CAND_CAP, sp_band, nq_tile, top, vp, tile0,
                                        (const int *)(plan + pl.tile_ovf) + split_tiles.size(), (uint32_t *)(plan + pl.ovf_q) + tile0, (SplitStats *)plan));
            QMX_TRY(split_stage(q, "select"));
            split_tiles.push_back({tile0, nq_tile});
            if (counters) counters->kernel_launches += 8;
            continue;
        }
        const int qt = (int)pow2_ceil(nq_tile);
        for (uint32_t pass = 0; pass < n_pass; ++pass) {
            if (is_stopped && *is_stopped) {
                set_error("search cancelled");
                return QMX_ERR_CANCELLED;
            }
            const uint32_t off = pass * MAX_TOP_FAST;
            const uint32_t ptop = std::min<uint32_t>(MAX_TOP_FAST, top - off);
            ScanArgs a;
            fill_args(q, tile0, nq_tile, a);
            a.ids = d_ids;
            a.n_cand = n_cand;
            a.top = ptop;
            a.partial = (uint64_t *)q->partial.p;
            a.partial_qt = (uint32_t)qt;
            a.key_bound = pass ? (const uint64_t *)q->bounds.p : nullptr;
            // The chain-major scan (scan_mfma16.hip) keeps one top list per wave and query: 512 lists per query on the chip, each of
            // which would learn its reject threshold from its own 1 / 512 of the rows (~k ln(n / 512 k) insertions per list, each
            // a wave-serial event the other waves of the block wait for at the next barrier).  A pre-scan of the first 1 / 1024 of the
            // block gives every list the k-th best score of that prefix as a starting threshold: a lower bound of the final k-th
            // best score, so nothing that belongs to the result is rejected (ties pass), and only ~1024 k rows per query beat it.
            // (Running the pre-scan as a top-k pass of the chain-major kernel itself was insertion-bound: 0.2 ms instead of 0.06.)
            const bool m16 = s->dtype == QMX_DTYPE_F32 && mfma_scan_ok(s) && mfma16_scan_ok(qt, SCAN_TOPK, a);
            const bool sqm = (s->dtype == QMX_DTYPE_SQ_U8 || s->dtype == QMX_DTYPE_TQ ? qt >= 4 : s->dtype == QMX_DTYPE_F16 && qt >= 8) && mfma_scan_ok(s);   // scan_sq_mfma.hip starts from the bound too
            const bool m4 = s->dtype == QMX_DTYPE_F32 && qt >= 8 && mfma_scan_ok(s);                                        // scan_mfma.hip (4x4x1) as well
            const bool bqk = s->dtype == QMX_DTYPE_BQ && qt >= 4;   // bq_rows_kernel: integer scores, selection-bound without a starting threshold
            if (pass == 0 && n_cand >= (1u << 18) && (m16 || sqm || m4 || bqk) && !option(OPT_NO_PRESCAN)) {
                const int pre_shift = (int)std::min<int64_t>(std::max<int64_t>(option(OPT_PRESCAN_SHIFT), 1), 20);  // tuning: measured 5..10 on C2, the main pass does not care, the pre-scan itself gets cheaper
                const uint64_t pre_n = std::max<uint64_t>(n_cand >> pre_shift, 1u << 13) & ~(uint64_t)15;
                {
                    // score matrix of the prefix (the score-mode kernels, <= tile_qt queries per launch), one block per query selects its
                    // k best live candidates, the k-th becomes the bound
                    QMX_TRY(q->scores.reserve((size_t)nq_tile * pre_n * sizeof(float)));
                    QMX_TRY(score_matrix_enqueue(q, tile0, nq_tile, d_ids, pre_n, (float *)q->scores.p, pre_n, nullptr));
// (the bound at the tile's own offset: the bounds of earlier split tiles are read again by the plan of their exact passes)
                    QMX_TRY(launch_custom_topk(q->stream, (const float *)q->scores.p, pre_n, d_ids, a.del, nq_tile, ptop, d_out + (size_t)tile0 * top,
                                               d_counts + tile0, (uint64_t *)q->gthr.p + tile0));
                }
                a.gthr = (const uint64_t *)q->gthr.p + tile0;
                if (counters) counters->kernel_launches += 2;
            }
            uint32_t grid = grid_cap;
            size_t slot = 0;
            if (timed) QMX_TRY(timing_begin(q, &slot));
            QMX_TRY(launch_scan(q, qt, SCAN_TOPK, a, &grid));
            q->last_kernel = last_noted_kernel();
            if (timed) QMX_TRY(timing_end(q, slot));
            QMX_TRY(launch_merge_keys(q->stream, (const uint64_t *)q->partial.p, grid, (uint32_t)qt, nq_tile, ptop,
                                      d_out + (size_t)tile0 * top, d_counts + tile0, top, off,
                                      n_pass > 1 ? (uint64_t *)q->bounds.p : nullptr));
            if (counters) counters->kernel_launches += 2;
        }
    }
    if (!split_tiles.empty()) {
        // 5. exact scores of the survivors (the gather kernel of qmx_rescore: the reference's bits), sorted by (score, lower id first)
        const void *split_kernel = q->last_kernel;
        const uint32_t first = split_tiles.front().first, last = split_tiles.back().first + split_tiles.back().second;
        // (split tiles are a prefix of the batch - the remainder tile, if any, comes last -: the sort walks queries 0 .. last)
        QMX_REQUIRE(first == 0, QMX_ERR_OTHER, "split tiles must start at query 0");
        PairSel sel{vp.qsel, 0, nullptr, vp.used};
        QMX_TRY(score_pairs_device(q, sel, vp.ids, vp.cap, (float *)q->sp_vscores.p, false));
        QMX_TRY(split_stage(q, "verify gather"));
        QMX_TRY(launch_sort_scored(q->stream, (const float *)q->sp_vscores.p, vp.ids, vp.cnt, 0, last, top, d_out, d_counts, vp.off));
        QMX_TRY(split_stage(q, "verify sort"));
        // 6. the exact scan of the queries whose lists overflowed (masses of near-equal scores, a sample that is all deleted), and of those only:
        // packed, one 16-query pass when 1..16 of them, passes of 64 otherwise.  The kernels start, read their flag and return when it is clear.
        uint32_t *ovf_list = (uint32_t *)(plan + pl.list);
        uint64_t *gthr_packed = (uint64_t *)(plan + pl.gthr_packed);
        const uint32_t FQT = split_fqt ? split_fqt : SPLIT_FQT;
        const uint32_t n_run64 = (last + FQT - 1) / FQT, n_slots = n_run64 * FQT;
        QMX_TRY(launch_split_plan(q->stream, (const uint32_t *)(plan + pl.ovf_q), last, (const uint64_t *)q->gthr.p, ovf_list, gthr_packed, n_slots,
                                  (uint32_t *)(plan + pl.count), (int *)(plan + pl.run16), (int *)(plan + pl.run64), n_run64, (SplitStats *)plan, q->d_queries,
                                  q->q_stride, q->sp_fq.p));
        for (uint32_t pass = 0; pass <= n_run64; ++pass) {      // pass 0: the 16-query shape; pass p >= 1: packed queries 64 (p - 1) ..
            if (pass && last <= 16) break;
            const uint32_t p0 = pass ? (pass - 1) * FQT : 0;
            const uint32_t nq_sub = pass ? std::min<uint32_t>(FQT, last - p0) : std::min<uint32_t>(16, last);
            const int *run_if = pass ? (const int *)(plan + pl.run64) + (pass - 1) : (const int *)(plan + pl.run16);
            ScanArgs a;
            fill_args(q, 0, nq_sub, a);
            a.queries = (const char *)q->sp_fq.p + (size_t)p0 * q->q_stride;
            a.n_cand = n_cand;
            a.top = top;
            a.partial = (uint64_t *)q->partial.p;
            a.gthr = gthr_packed + p0;
            a.run_if = run_if;
            const int fqt = (int)std::max<uint32_t>(16, pow2_ceil(nq_sub));
            a.partial_qt = (uint32_t)fqt;
            uint32_t grid = grid_cap;
            QMX_REQUIRE(mfma16_scan_ok(fqt, SCAN_TOPK, a), QMX_ERR_OTHER, "split fallback shape");
            QMX_TRY(launch_scan_f32_mfma16(q->stream, fqt, a, s->num_cus, &grid));
            QMX_TRY(launch_merge_keys(q->stream, (const uint64_t *)q->partial.p, grid, (uint32_t)fqt, nq_sub, top, d_out, d_counts, top, 0, nullptr, run_if,
                                      ovf_list + p0));
        }
        QMX_TRY(split_stage(q, "fallback (conditional)"));
        q->last_kernel = split_kernel;      // (the fallback launches above are not what ran)
        q->last_split = true;
        q->last_pq = false;
        q->last_fqt = FQT;
    }
    {
        // what the host knows at enqueue; the prefilter's own share (candidates, verified rows, exact passes of overflowed queries) is on the device
        // until the stream is synchronised: fold_split_counters
        qmx_counters &c = q->last_counters;
        uint32_t split_q = 0;
        uint64_t bytes = 0;
        for (auto &t : split_tiles) {
            split_q += t.second;
            // one pass over the derived copy (2 or 4 bytes per element; the f32 rows themselves when there is none) + the sample's exact scores
            bytes += n_cand * (uint64_t)s->dim * (s->split_i8 ? 1 : s->d_rows_split && s->split_half ? 2 : 4);
            bytes += (uint64_t)((t.second + tile_qt(s, q) - 1) / tile_qt(s, q)) * q->sp_sample_n * s->row_bytes;
        }
        const uint32_t rest = q->nq - split_q;
        bytes += (uint64_t)((rest + TQ - 1) / TQ) * n_cand * s->row_bytes * n_pass;
        c.vectors_scored = (uint64_t)q->nq * n_cand * n_pass;
        c.bytes_read = bytes;
        c.kernel_launches = counters ? counters->kernel_launches : 0;
        c.prefilter_queries = split_q;
        q->last_row_bytes = s->row_bytes;
        q->last_n_cand = n_cand;
        if (counters) {
            const uint64_t launches = counters->kernel_launches;
            *counters = c;
            counters->kernel_launches = launches;
        }
    }
    return QMX_OK;
}

// after the stream is synchronised: the device's share of the last search's counters (prefilter candidates, exactly re-scored rows, the queries
// that took the exact scan) -> c, bytes_read completed with the rows those steps read
int32_t fold_split_counters(qmx_query *q, qmx_counters *c) {
    if (!q->last_split || !q->sp_plan.p) return QMX_OK;
    SplitStats st;
    QMX_HIP(hipMemcpy(&st, q->sp_plan.p, sizeof(st), hipMemcpyDeviceToHost));
    c->prefilter_candidates = st.candidates;
    c->verified_rows = st.verified;
    c->fallback_queries = st.fallback_queries;
    c->bytes_read += st.verified * q->last_row_bytes;
    if (st.fallback_queries) {
        const uint32_t f = st.fallback_queries;
        const uint64_t passes = q->last_pq ? f : f <= 16 ? 1 : (f + q->last_fqt - 1) / q->last_fqt;     // (the exact PQ kernel streams the codes once per query)
        c->bytes_read += passes * q->last_n_cand * q->last_row_bytes;
    }
    return QMX_OK;
}

int32_t qmx_search_topk(qmx_query *q, uint32_t top, const uint32_t *ids, uint64_t n_ids, qmx_scored_point *out,
                        uint32_t *out_counts, const volatile uint8_t *is_stopped, qmx_counters *counters) {
    QMX_REQUIRE(q && out && out_counts, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(top >= 1, QMX_ERR_BAD_ARG, "top must be > 0 (FixedLengthPriorityQueue::new panics on 0)");
    QMX_REQUIRE(top <= MAX_TOP, QMX_ERR_NOT_SUPPORTED, "top %u > %u not supported yet", top, MAX_TOP);
    QMX_HIP(hipSetDevice(q->seg->device));
    if (counters) memset(counters, 0, sizeof(*counters));
    if (q->nq == 0) return QMX_OK;
    const void *d_ids = nullptr;
    if (ids) {
        if (n_ids == 0) {  // empty candidate list: every queue stays empty
            if (is_device_ptr(out_counts)) {
                QMX_HIP(hipMemsetAsync(out_counts, 0, (size_t)q->nq * 4, q->stream));
                QMX_HIP(hipStreamSynchronize(q->stream));
            } else {
                for (uint32_t i = 0; i < q->nq; ++i) out_counts[i] = 0;
            }
            return QMX_OK;
        }
        QMX_TRY(stage_in(q, q->ids, ids, (size_t)n_ids * 4, &d_ids));
    }
    const bool out_dev = is_device_ptr(out);
    const bool cnt_dev = is_device_ptr(out_counts);
    qmx_scored_point *d_out = out;
    uint32_t *d_counts = out_counts;
    if (!out_dev) {
        QMX_TRY(q->out.reserve((size_t)q->nq * top * sizeof(qmx_scored_point)));
        d_out = (qmx_scored_point *)q->out.p;
    }
    if (!cnt_dev) {
        QMX_TRY(q->counts.reserve((size_t)q->nq * sizeof(uint32_t)));
        d_counts = (uint32_t *)q->counts.p;
    }
    const bool timed = q->timing || (q->seg->flags & QMX_SEG_TIME_KERNELS) != 0;
    QMX_TRY(search_enqueue(q, top, (const uint32_t *)d_ids, n_ids, d_out, d_counts, is_stopped, counters, timed));
    if (!out_dev) QMX_TRY(copy_out(q->stream, out, d_out, (size_t)q->nq * top * sizeof(qmx_scored_point)));
    if (!cnt_dev) QMX_TRY(copy_out(q->stream, out_counts, d_counts, (size_t)q->nq * sizeof(uint32_t)));
    QMX_TRY(check_err_flag(q));  // synchronises the stream
    if (counters) QMX_TRY(fold_split_counters(q, counters));
    if (timed) {
        const float before = q->timing_ms;
        QMX_TRY(timing_fold(q));
        if (counters) counters->kernel_ms = q->timing_ms - before;
    }
    return QMX_OK;
}

int32_t qmx_query_last_counters(qmx_query *q, qmx_counters *out) {
    QMX_REQUIRE(q && out, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_HIP(hipSetDevice(q->device));
    QMX_HIP(hipStreamSynchronize(q->stream));
    *out = q->last_counters;
    return fold_split_counters(q, out);
}

int32_t qmx_search_topk_async(qmx_query *q, uint32_t top, const uint32_t *ids, uint64_t n_ids,
                              qmx_scored_point *out_dev, uint32_t *out_counts_dev) {
    QMX_REQUIRE(q && out_dev && out_counts_dev, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(top >= 1 && top <= MAX_TOP, QMX_ERR_NOT_SUPPORTED, "top %u not in 1..%u", top, MAX_TOP);
    QMX_REQUIRE(!ids || is_device_ptr(ids), QMX_ERR_BAD_ARG, "async search needs device ids");
    QMX_HIP(hipSetDevice(q->seg->device));
    if (q->nq == 0) return QMX_OK;
    const bool timed = q->timing || (q->seg->flags & QMX_SEG_TIME_KERNELS) != 0;
    return search_enqueue(q, top, ids, n_ids, out_dev, out_counts_dev, nullptr, nullptr, timed);
}

int32_t qmx_merge_topk(int32_t device_id, const qmx_scored_point *lists, const uint32_t *list_counts, uint32_t n_lists,
                       uint32_t nq, uint32_t k, qmx_scored_point *out, uint32_t *out_counts) {
    QMX_REQUIRE(lists && out && out_counts, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(k >= 1 && k <= MAX_TOP, QMX_ERR_NOT_SUPPORTED, "k %u not in 1..%u", k, MAX_TOP);
    QMX_TRY(check_device(device_id, nullptr));
    if (nq == 0) return QMX_OK;
    const size_t lbytes = (size_t)n_lists * nq * k * sizeof(qmx_scored_point);
    const size_t cbytes = (size_t)n_lists * nq * sizeof(uint32_t);
    const size_t obytes = (size_t)nq * k * sizeof(qmx_scored_point);
    DevBuf bl, bc, bo, boc;
    const qmx_scored_point *d_lists = lists;
    const uint32_t *d_lc = list_counts;
    qmx_scored_point *d_out = out;
    uint32_t *d_oc = out_counts;
    int32_t rc = QMX_OK;
    do {
        if (!is_device_ptr(lists)) {
            if ((rc = bl.reserve(lbytes)) != QMX_OK) break;
            if (hipMemcpy(bl.p, lists, lbytes, hipMemcpyHostToDevice) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
            d_lists = (const qmx_scored_point *)bl.p;
        }
        if (list_counts && !is_device_ptr(list_counts)) {
            if ((rc = bc.reserve(cbytes)) != QMX_OK) break;
            if (hipMemcpy(bc.p, list_counts, cbytes, hipMemcpyHostToDevice) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
            d_lc = (const uint32_t *)bc.p;
        }
        const bool od = is_device_ptr(out), ocd = is_device_ptr(out_counts);
        if (!od) { if ((rc = bo.reserve(obytes)) != QMX_OK) break; d_out = (qmx_scored_point *)bo.p; }
        if (!ocd) { if ((rc = boc.reserve((size_t)nq * 4)) != QMX_OK) break; d_oc = (uint32_t *)boc.p; }
        if ((rc = launch_merge_points(nullptr, d_lists, d_lc, nullptr, n_lists, nq, k, d_out, d_oc)) != QMX_OK) break;
        if (!od && hipMemcpy(out, d_out, obytes, hipMemcpyDeviceToHost) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
        if (!ocd && hipMemcpy(out_counts, d_oc, (size_t)nq * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
        if (hipDeviceSynchronize() != hipSuccess) rc = QMX_ERR_OTHER;
    } while (0);
    bl.release(); bc.release(); bo.release(); boc.release();
    return rc;
}

int32_t qmx_merge_topk_async(int32_t device_id, void *hip_stream, const qmx_scored_point *lists_dev,
                             const uint32_t *list_counts_dev, const uint32_t *list_idx_base_dev, uint32_t n_lists,
                             uint32_t nq, uint32_t k, qmx_scored_point *out_dev, uint32_t *out_counts_dev) {
    QMX_REQUIRE(lists_dev && out_dev && out_counts_dev, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(k >= 1 && k <= MAX_TOP, QMX_ERR_NOT_SUPPORTED, "k %u not in 1..%u", k, MAX_TOP);
    QMX_HIP(hipSetDevice(device_id));
    if (nq == 0) return QMX_OK;
    return launch_merge_points((hipStream_t)hip_stream, lists_dev, list_counts_dev, list_idx_base_dev, n_lists, nq, k,
                               out_dev, out_counts_dev);
}


uint64_t qmx_topk_record_bytes(uint32_t nq, uint32_t k) {
    return ((uint64_t)nq * k * sizeof(qmx_scored_point) + (uint64_t)nq * sizeof(uint32_t) + 7) & ~7ull;
}

int32_t qmx_merge_topk_packed_async(int32_t device_id, void *hip_stream, const void *records_dev, const uint32_t *list_idx_base_dev, uint32_t n_lists,
                                    uint32_t nq, uint32_t k, qmx_scored_point *out_dev, uint32_t *out_counts_dev) {
    QMX_REQUIRE(records_dev && out_dev && out_counts_dev, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(k >= 1 && k <= MAX_TOP, QMX_ERR_NOT_SUPPORTED, "k %u not in 1..%u", k, MAX_TOP);
    QMX_REQUIRE(((uintptr_t)records_dev & 7) == 0, QMX_ERR_BAD_ARG, "records must be 8-byte aligned");
    QMX_HIP(hipSetDevice(device_id));
    if (nq == 0) return QMX_OK;
    const uint64_t rec = qmx_topk_record_bytes(nq, k);
    const qmx_scored_point *lists = (const qmx_scored_point *)records_dev;
    const uint32_t *counts = (const uint32_t *)((const char *)records_dev + (uint64_t)nq * k * sizeof(qmx_scored_point));
    return launch_merge_points((hipStream_t)hip_stream, lists, counts, list_idx_base_dev, n_lists, nq, k, out_dev, out_counts_dev,
                               rec / sizeof(qmx_scored_point), rec / sizeof(uint32_t));
}

int32_t qmx_search_quantized(const qmx_hnsw *g, qmx_query *quantized, qmx_query *raw, const qmx_search_params *p, const uint32_t *ids,
                             uint64_t n_ids, qmx_scored_point *out, uint32_t *out_counts, const volatile uint8_t *is_stopped,
                             qmx_counters *counters) {
    QMX_REQUIRE(quantized && p && out && out_counts, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(p->top >= 1, QMX_ERR_BAD_ARG, "top must be > 0");
    const bool rescore = p->rescore != 0;
    QMX_REQUIRE(!rescore || raw, QMX_ERR_BAD_ARG, "rescoring needs the original-vector query batch");
    QMX_REQUIRE(!raw || (raw->nq == quantized->nq && raw->device == quantized->device), QMX_ERR_BAD_ARG, "the two query batches must match");
    // get_oversampled_top (vector_index_search_common.rs:27-46): (oversampling * top as f64) as usize when > 1.0
    // (never clamped: the reference never searches fewer candidates than oversampling asks for; what does not fit fails loudly)
    const uint32_t top_limit = g ? HNSW_MAX_EF : MAX_TOP;
    QMX_REQUIRE(p->top <= top_limit, QMX_ERR_NOT_SUPPORTED, "top %u > %u not supported", p->top, top_limit);
    uint32_t otop = p->top;
    if (p->oversampling > 1.0f) {
        const double o = (double)p->oversampling * (double)p->top;
        QMX_REQUIRE(o <= (double)top_limit, QMX_ERR_NOT_SUPPORTED, "oversampled top %.0f > %u not supported", o, top_limit);
        otop = (uint32_t)o;
    }
    QMX_HIP(hipSetDevice(quantized->device));
    if (counters) memset(counters, 0, sizeof(*counters));
    const uint32_t nq = quantized->nq;
    if (nq == 0) return QMX_OK;
    QMX_TRY(quantized->cand.reserve((size_t)nq * otop * sizeof(qmx_scored_point)));
    QMX_TRY(quantized->cand_cnt.reserve((size_t)nq * 4));
    QMX_TRY(quantized->cand_ids.reserve((size_t)nq * otop * 4));
    qmx_scored_point *d_cand = (qmx_scored_point *)quantized->cand.p;
    uint32_t *d_cnt = (uint32_t *)quantized->cand_cnt.p, *d_ids = (uint32_t *)quantized->cand_ids.p;
    // stage 1: the quantized (or raw, when the caller passes the raw batch as `quantized`) search with the oversampled top
    if (g) {
        const uint32_t ef = std::max(p->hnsw_ef, otop);     // graph_layers.rs:549
        QMX_TRY(hnsw_search_sync(g, quantized, otop, ef, d_cand, d_cnt, is_stopped, counters, p->acorn != 0));   // SearchAlgorithm of the request
    } else {
        QMX_TRY(qmx_search_topk(quantized, otop, ids, n_ids, d_cand, d_cnt, is_stopped, counters));
    }
    const bool out_dev = is_device_ptr(out), cnt_dev = is_device_ptr(out_counts);
    if (!rescore) {   // search_result.truncate(top)
        qmx_scored_point *d_out = out;
        uint32_t *d_oc = out_counts;
        if (!out_dev) { QMX_TRY(quantized->out.reserve((size_t)nq * p->top * sizeof(qmx_scored_point))); d_out = (qmx_scored_point *)quantized->out.p; }
        if (!cnt_dev) { QMX_TRY(quantized->counts.reserve((size_t)nq * 4)); d_oc = (uint32_t *)quantized->counts.p; }
        QMX_TRY(launch_split_candidates(quantized->stream, d_cand, d_cnt, otop, nq, nullptr, p->top, d_out, d_oc));
        if (!out_dev) QMX_TRY(copy_out(quantized->stream, out, d_out, (size_t)nq * p->top * sizeof(qmx_scored_point)));
        if (!cnt_dev) QMX_TRY(copy_out(quantized->stream, out_counts, d_oc, (size_t)nq * 4));
        QMX_HIP(hipStreamSynchronize(quantized->stream));
        return QMX_OK;
    }
    // stage 2: postprocess_search_result (:48-91): re-score the candidates with the original vectors, sort, truncate
    QMX_TRY(launch_split_candidates(quantized->stream, d_cand, d_cnt, otop, nq, d_ids, 0, nullptr, nullptr));
    QMX_HIP(hipStreamSynchronize(quantized->stream));
    QMX_TRY(qmx_rescore(raw, d_ids, d_cnt, otop, std::min(p->top, otop), out, out_counts));
    if (counters) {
        counters->bytes_read += (uint64_t)nq * otop * raw->seg->row_bytes;
        counters->kernel_launches += 2;
    }
    return QMX_OK;
}

}  // extern "C"

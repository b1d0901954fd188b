// api_sharded.hip — the C-ABI of include/qdrant_amd.h, one query batch against N segments from one host thread.
// (One of the api_*.hip translation units; what they share: api_internal.hpp.)
#include "api_internal.hpp"

extern "C" {

// ---------------------------------------------------------------------------------------------
// one query batch against N segments, possibly on N devices, from ONE host thread: the fan-out of SegmentsSearcher::search
// (lib/collection/src/collection_manager/segments_searcher.rs:250-285) + the merge of the per-segment lists (BatchResultAggregator,
// lib/shard/src/search_result_aggregator.rs:50-121) behind one call.  Every segment's local stage is enqueued on its own batch's stream
// (the devices run concurrently), its Q x top x 8 B list travels to the first batch's device (a peer copy over xGMI when it lives
// elsewhere: the one exchange step of SURVEY 8e), the k-way merge runs there behind the N "list arrived" events.
// ---------------------------------------------------------------------------------------------
static int32_t sharded_enqueue(qmx_query *const *queries, const qmx_hnsw *const *graphs, uint32_t n_segments, uint32_t top, uint32_t ef, const uint32_t *id_bases,
                               qmx_scored_point *d_out, uint32_t *d_counts, const volatile uint8_t *is_stopped, qmx_counters *counters) {
    qmx_query *root = queries[0];
    const uint32_t nq = root->nq;
    const size_t lbytes = (size_t)nq * top * sizeof(qmx_scored_point), cbytes = (size_t)nq * 4;
    for (uint32_t i = 0; i < n_segments; ++i) {
        QMX_REQUIRE(queries[i] && queries[i]->nq == nq, QMX_ERR_BAD_ARG, "segment %u: every batch must hold the same %u queries", i, nq);
        for (uint32_t j = 0; j < i; ++j) QMX_REQUIRE(queries[j] != queries[i], QMX_ERR_BAD_ARG, "segments %u and %u share one query batch (one qmx_query per (batch, segment))", j, i);
        if (graphs) {
            QMX_REQUIRE(graphs[i], QMX_ERR_BAD_ARG, "segment %u: NULL graph", i);
            QMX_TRY(hnsw_check(graphs[i], queries[i], top, ef));
        }
    }
    // (the merge kernel's limit, checked before anything is enqueued: the async form would return with every segment's scan in flight)
    QMX_REQUIRE((uint64_t)n_segments * top <= 16384, QMX_ERR_NOT_SUPPORTED, "merge of %u segments x top %u exceeds 16384 entries per query", n_segments, top);
    if (counters) memset(counters, 0, sizeof(*counters));
    QMX_HIP(hipSetDevice(root->device));
    QMX_TRY(root->sh_lists.reserve(n_segments * (lbytes + cbytes) + (size_t)n_segments * 4));
    unsigned char *gl = (unsigned char *)root->sh_lists.p;
    qmx_scored_point *g_lists = (qmx_scored_point *)gl;
    uint32_t *g_counts = (uint32_t *)(gl + n_segments * lbytes);
    uint32_t *g_bases = g_counts + (size_t)n_segments * nq;
    {   // id bases: segment-local offsets + base = the caller's id space (0 when NULL)
        std::vector<uint32_t> &hb = root->sh_bases_host;
        hb.assign(n_segments, 0u);
        if (id_bases) for (uint32_t i = 0; i < n_segments; ++i) hb[i] = id_bases[i];
        QMX_HIP(hipMemcpyAsync(g_bases, hb.data(), (size_t)n_segments * 4, hipMemcpyHostToDevice, root->stream));
    }
    for (uint32_t i = 0; i < n_segments; ++i) {
        qmx_query *q = queries[i];
        if (is_stopped && *is_stopped) {
            set_error("search cancelled");
            return QMX_ERR_CANCELLED;
        }
        QMX_HIP(hipSetDevice(q->device));
        QMX_TRY(q->out.reserve(lbytes));
        QMX_TRY(q->counts.reserve(cbytes));
        const bool timed = q->timing || (q->seg->flags & QMX_SEG_TIME_KERNELS) != 0;
        qmx_counters local{};
        if (graphs) {
            if (graphs[i]->n_points == 0) QMX_HIP(hipMemsetAsync(q->counts.p, 0, cbytes, q->stream));
            else QMX_TRY(hnsw_enqueue(graphs[i], q, top, ef, (qmx_scored_point *)q->out.p, (uint32_t *)q->counts.p, nullptr, timed));
            local.kernel_launches = 1;
        } else {
            QMX_TRY(search_enqueue(q, top, nullptr, 0, (qmx_scored_point *)q->out.p, (uint32_t *)q->counts.p, is_stopped, &local, timed));
        }
        if (counters) {
            counters->vectors_scored += local.vectors_scored;
            counters->bytes_read += local.bytes_read;
            counters->kernel_launches += local.kernel_launches + 1;
            counters->prefilter_queries += local.prefilter_queries;
        }
        // the list travels on the producing stream (ordered behind the scan without an event), then "arrived" is recorded for the merge.  The shared
        // lists may still be read by the merge of the PREVIOUS call (async calls pipelined without a sync in between: that merge waits for its slowest
        // segment, a fast segment's stream is long past it): the copy into them waits for that merge first
        if (q != root && root->sh_merged) QMX_HIP(hipStreamWaitEvent(q->stream, root->sh_merged, 0));
        if (q->device != root->device) {
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, root->device, q->device);
            if (can) {   // direct xGMI writes instead of a staged copy; "already enabled" is not an error
                QMX_HIP(hipSetDevice(root->device));
                hipError_t e = hipDeviceEnablePeerAccess(q->device, 0);
                if (e != hipSuccess) (void)hipGetLastError();
                QMX_HIP(hipSetDevice(q->device));
            }
            QMX_HIP(hipMemcpyPeerAsync((unsigned char *)g_lists + i * lbytes, root->device, q->out.p, q->device, lbytes, q->stream));
            QMX_HIP(hipMemcpyPeerAsync(g_counts + (size_t)i * nq, root->device, q->counts.p, q->device, cbytes, q->stream));
        } else {
            QMX_HIP(hipMemcpyAsync((unsigned char *)g_lists + i * lbytes, q->out.p, lbytes, hipMemcpyDeviceToDevice, q->stream));
            QMX_HIP(hipMemcpyAsync(g_counts + (size_t)i * nq, q->counts.p, cbytes, hipMemcpyDeviceToDevice, q->stream));
        }
        if (q != root) {
            if (!q->sh_done) QMX_HIP(hipEventCreateWithFlags(&q->sh_done, hipEventDisableTiming));
            QMX_HIP(hipEventRecord(q->sh_done, q->stream));
        }
    }
    QMX_HIP(hipSetDevice(root->device));
    for (uint32_t i = 1; i < n_segments; ++i) QMX_HIP(hipStreamWaitEvent(root->stream, queries[i]->sh_done, 0));
    QMX_TRY(launch_merge_points(root->stream, g_lists, g_counts, g_bases, n_segments, nq, top, d_out, d_counts));
    if (!root->sh_merged) QMX_HIP(hipEventCreateWithFlags(&root->sh_merged, hipEventDisableTiming));
    QMX_HIP(hipEventRecord(root->sh_merged, root->stream));
    return QMX_OK;
}

static int32_t sharded_sync(qmx_query *const *queries, const qmx_hnsw *const *graphs, uint32_t n_segments, uint32_t top, uint32_t ef, const uint32_t *id_bases,
                            qmx_scored_point *out, uint32_t *out_counts, const volatile uint8_t *is_stopped, qmx_counters *counters) {
    QMX_REQUIRE(queries && n_segments >= 1 && queries[0] && out && out_counts, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(top >= 1 && top <= MAX_TOP, QMX_ERR_NOT_SUPPORTED, "top %u not in 1..%u", top, MAX_TOP);
    qmx_query *root = queries[0];
    if (root->nq == 0) return QMX_OK;
    QMX_HIP(hipSetDevice(root->device));
    const bool out_dev = is_device_ptr(out), cnt_dev = is_device_ptr(out_counts);
    const size_t lbytes = (size_t)root->nq * top * sizeof(qmx_scored_point), cbytes = (size_t)root->nq * 4;
    qmx_scored_point *d_out = out;
    uint32_t *d_counts = out_counts;
    if (!out_dev) { QMX_TRY(root->sh_out.reserve(lbytes + cbytes)); d_out = (qmx_scored_point *)root->sh_out.p; }
    if (!cnt_dev) { QMX_TRY(root->sh_out.reserve(lbytes + cbytes)); d_counts = (uint32_t *)((unsigned char *)root->sh_out.p + lbytes); }
    QMX_TRY(sharded_enqueue(queries, graphs, n_segments, top, ef, id_bases, d_out, d_counts, is_stopped, counters));
    if (!out_dev) QMX_TRY(copy_out(root->stream, out, d_out, lbytes));
    if (!cnt_dev) QMX_TRY(copy_out(root->stream, out_counts, d_counts, cbytes));
    // the root's stream is behind every segment's stream (the merge waited for their events): one wait completes the call; the other batches'
    // error flags are read behind their own (already finished) streams
    int32_t rc = QMX_OK;
    for (uint32_t i = 0; i < n_segments; ++i) {
        QMX_HIP(hipSetDevice(queries[i]->device));
        const int32_t r = check_err_flag(queries[i]);
        if (r != QMX_OK && rc == QMX_OK) rc = r;
        if (counters && !graphs) {
            qmx_counters c{};
            (void)fold_split_counters(queries[i], &c);
            counters->prefilter_candidates += c.prefilter_candidates;
            counters->verified_rows += c.verified_rows;
            counters->fallback_queries += c.fallback_queries;
            counters->bytes_read += c.bytes_read;
        }
        const bool timed = queries[i]->timing || (queries[i]->seg->flags & QMX_SEG_TIME_KERNELS) != 0;
        if (timed) {
            const float before = queries[i]->timing_ms;
            QMX_TRY(timing_fold(queries[i]));
            if (counters) counters->kernel_ms += queries[i]->timing_ms - before;
        }
    }
    QMX_HIP(hipSetDevice(root->device));
    return rc;
}

int32_t qmx_sharded_search_topk(qmx_query *const *queries, uint32_t n_segments, uint32_t top, const uint32_t *id_bases, qmx_scored_point *out,
                                uint32_t *out_counts, const volatile uint8_t *is_stopped, qmx_counters *counters) {
    return sharded_sync(queries, nullptr, n_segments, top, 0, id_bases, out, out_counts, is_stopped, counters);
}
int32_t qmx_sharded_hnsw_search(const qmx_hnsw *const *graphs, qmx_query *const *queries, uint32_t n_segments, uint32_t top, uint32_t ef,
                                const uint32_t *id_bases, qmx_scored_point *out, uint32_t *out_counts, const volatile uint8_t *is_stopped,
                                qmx_counters *counters) {
    QMX_REQUIRE(graphs, QMX_ERR_BAD_ARG, "NULL argument");
    return sharded_sync(queries, graphs, n_segments, top, ef, id_bases, out, out_counts, is_stopped, counters);
}
int32_t qmx_sharded_search_topk_async(qmx_query *const *queries, uint32_t n_segments, uint32_t top, const uint32_t *id_bases, qmx_scored_point *out_dev,
                                      uint32_t *out_counts_dev) {
    QMX_REQUIRE(queries && n_segments >= 1 && queries[0] && out_dev && out_counts_dev, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(top >= 1 && top <= MAX_TOP, QMX_ERR_NOT_SUPPORTED, "top %u not in 1..%u", top, MAX_TOP);
    if (queries[0]->nq == 0) return QMX_OK;
    return sharded_enqueue(queries, nullptr, n_segments, top, 0, id_bases, out_dev, out_counts_dev, nullptr, nullptr);
}
int32_t qmx_sharded_query_update(qmx_query *const *queries, uint32_t n_segments, const float *batch) {
    QMX_REQUIRE(queries && n_segments >= 1 && batch, QMX_ERR_BAD_ARG, "NULL argument");
    for (uint32_t i = 0; i < n_segments; ++i) {
        QMX_REQUIRE(queries[i], QMX_ERR_BAD_ARG, "segment %u: NULL batch", i);
        QMX_TRY(qmx_query_update(queries[i], batch));
    }
    return QMX_OK;
}

}  // extern "C"

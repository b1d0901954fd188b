// api_hnsw.hip — the C-ABI of include/qdrant_amd.h, HNSW graphs: create / import, the walks, the device build.
// (One of the api_*.hip translation units; what they share: api_internal.hpp.)
#include "api_internal.hpp"
#include <algorithm>
#include <atomic>

extern "C" {


int32_t qmx_hnsw_destroy(qmx_hnsw *g) {
    if (!g) return QMX_OK;
    (void)hipSetDevice(g->device);
    void *ptrs[] = {g->d_reindex, g->d_neighbors, g->d_ep_ids, g->d_ep_levels, g->d_xp_ids, g->d_xp_levels, g->d_level_offsets, g->d_offsets, g->d_l0, g->d_l0x};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete g;
    return QMX_OK;
}

static int32_t upload_bytes(void **dst, const void *src, uint64_t count, size_t elem) {
    *dst = nullptr;
    const size_t bytes = std::max<size_t>((size_t)count * elem, elem);
    QMX_HIP(hipMalloc(dst, bytes));
    if (count) QMX_HIP(hipMemcpy(*dst, src, (size_t)count * elem, hipMemcpyDefault));
    return QMX_OK;
}
static int32_t upload_array(uint32_t **dst, const uint32_t *src, uint64_t count) { return upload_bytes((void **)dst, src, count, 4); }
static int32_t upload_array(uint64_t **dst, const uint64_t *src, uint64_t count) { return upload_bytes((void **)dst, src, count, 8); }

int32_t qmx_hnsw_create(const qmx_hnsw_desc *d, qmx_hnsw **out) {
    QMX_REQUIRE(d && out, QMX_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    QMX_REQUIRE(d->m >= 1 && d->m0 >= 1, QMX_ERR_BAD_ARG, "m / m0 must be > 0");
    QMX_REQUIRE(d->n_points == 0 || (d->n_levels >= 1 && d->reindex && d->level_offsets && d->offsets), QMX_ERR_BAD_ARG,
                "graph arrays missing");
    QMX_REQUIRE(d->n_neighbors == 0 || d->neighbors, QMX_ERR_BAD_ARG, "neighbors missing");
    QMX_REQUIRE(d->n_entry_points == 0 || (d->entry_point_ids && d->entry_point_levels), QMX_ERR_BAD_ARG, "entry points missing");
    QMX_REQUIRE(d->n_extra_entry_points == 0 || (d->extra_entry_point_ids && d->extra_entry_point_levels), QMX_ERR_BAD_ARG,
                "extra entry points missing");
    QMX_REQUIRE(d->n_points == 0 || d->n_offsets >= (uint64_t)d->n_points + 1, QMX_ERR_BAD_ARG,
                "offsets must hold n_points + 1 entries at least (level 0 has a slot per point)");
    // structural checks on host-visible arrays (a corrupt links file must not crash the GPU); they run before the device is
    // touched, so a bad file is reported as such on any host
    if (d->n_points && !is_device_ptr(d->level_offsets)) {
        QMX_REQUIRE(d->level_offsets[0] == 0 && d->level_offsets[d->n_levels] + 1 == d->n_offsets, QMX_ERR_BAD_ARG,
                    "level_offsets do not span the offsets array");
        for (uint32_t l = 0; l < d->n_levels; ++l)
            QMX_REQUIRE(d->level_offsets[l] <= d->level_offsets[l + 1], QMX_ERR_BAD_ARG, "level_offsets must be non-decreasing");
        QMX_REQUIRE(d->level_offsets[d->n_levels > 1 ? 1 : d->n_levels] == d->n_points, QMX_ERR_BAD_ARG,
                    "level 0 must have one slot per point");
    }
    if (d->n_points && !is_device_ptr(d->offsets)) {
        for (uint64_t i = 0; i + 1 < d->n_offsets; ++i)
            QMX_REQUIRE(d->offsets[i] <= d->offsets[i + 1], QMX_ERR_BAD_ARG, "offsets must be non-decreasing");
        QMX_REQUIRE(d->offsets[d->n_offsets - 1] <= d->n_neighbors, QMX_ERR_BAD_ARG, "offsets run past the neighbors array");
    }
    // every link stored on level l >= 1 must point at a node that HAS a slot on level l (the walk indexes offsets[] with
    // level_offsets[l] + reindex[id]); the same for an entry point and the level it claims.  The kernel bounds the slot as well.
    const bool host_graph = d->n_points && !is_device_ptr(d->level_offsets) && !is_device_ptr(d->offsets) && !is_device_ptr(d->reindex) &&
                            (d->n_neighbors == 0 || !is_device_ptr(d->neighbors));
    if (host_graph) {
        for (uint32_t i = 0; i < d->n_points; ++i)
            QMX_REQUIRE(d->reindex[i] < d->n_points, QMX_ERR_OUT_OF_BOUNDS, "reindex entry out of range");
        for (uint32_t l = 1; l < d->n_levels; ++l) {
            const uint64_t size_l = d->level_offsets[l + 1] - d->level_offsets[l];
            for (uint64_t slot = d->level_offsets[l]; slot < d->level_offsets[l + 1]; ++slot)
                for (uint64_t j = d->offsets[slot]; j < d->offsets[slot + 1]; ++j) {
                    const uint32_t id = d->neighbors[j];
                    QMX_REQUIRE(id < d->n_points && d->reindex[id] < size_l, QMX_ERR_OUT_OF_BOUNDS,
                                "link %u on level %u points at a node that is not on that level", id, l);
                }
        }
        auto ep_ok = [&](uint32_t id, uint32_t lv) {
            if (id >= d->n_points) return false;
            const uint32_t l = std::min<uint32_t>(lv, d->n_levels - 1);
            return l == 0 || (uint64_t)d->reindex[id] < d->level_offsets[l + 1] - d->level_offsets[l];
        };
        for (uint32_t i = 0; i < d->n_entry_points && !is_device_ptr(d->entry_point_ids) && !is_device_ptr(d->entry_point_levels); ++i)
            QMX_REQUIRE(ep_ok(d->entry_point_ids[i], d->entry_point_levels[i]), QMX_ERR_OUT_OF_BOUNDS, "entry point %u is not on its level %u",
                        d->entry_point_ids[i], d->entry_point_levels[i]);
        for (uint32_t i = 0; i < d->n_extra_entry_points && !is_device_ptr(d->extra_entry_point_ids) && !is_device_ptr(d->extra_entry_point_levels); ++i)
            QMX_REQUIRE(ep_ok(d->extra_entry_point_ids[i], d->extra_entry_point_levels[i]), QMX_ERR_OUT_OF_BOUNDS,
                        "extra entry point %u is not on its level %u", d->extra_entry_point_ids[i], d->extra_entry_point_levels[i]);
    }
    for (uint32_t i = 0; i < d->n_entry_points && !is_device_ptr(d->entry_point_ids); ++i)
        QMX_REQUIRE(d->entry_point_ids[i] < d->n_points, QMX_ERR_OUT_OF_BOUNDS, "entry point %u out of range", d->entry_point_ids[i]);
    for (uint32_t i = 0; i < d->n_extra_entry_points && !is_device_ptr(d->extra_entry_point_ids); ++i)
        QMX_REQUIRE(d->extra_entry_point_ids[i] < d->n_points, QMX_ERR_OUT_OF_BOUNDS, "extra entry point out of range");
    QMX_TRY(check_device(d->device_id, nullptr));
    qmx_hnsw *g = new (std::nothrow) qmx_hnsw();
    QMX_REQUIRE(g, QMX_ERR_OUT_OF_MEMORY, "host allocation failed");
    g->device = d->device_id;
    g->m = d->m; g->m0 = d->m0; g->n_points = d->n_points; g->n_levels = d->n_levels;
    g->n_ep = d->n_entry_points; g->n_xp = d->n_extra_entry_points;
    g->n_offsets = d->n_offsets; g->n_neighbors = d->n_neighbors;
    int32_t rc = QMX_OK;
    do {
        if ((rc = upload_array(&g->d_reindex, d->reindex, d->n_points)) != QMX_OK) break;
        if ((rc = upload_array(&g->d_level_offsets, d->level_offsets, d->n_points ? (uint64_t)d->n_levels + 1 : 0)) != QMX_OK) break;
        if ((rc = upload_array(&g->d_offsets, d->offsets, d->n_points ? d->n_offsets : 0)) != QMX_OK) break;
        if ((rc = upload_array(&g->d_neighbors, d->neighbors, d->n_neighbors)) != QMX_OK) break;
        if ((rc = upload_array(&g->d_ep_ids, d->entry_point_ids, d->n_entry_points)) != QMX_OK) break;
        if ((rc = upload_array(&g->d_ep_levels, d->entry_point_levels, d->n_entry_points)) != QMX_OK) break;
        if ((rc = upload_array(&g->d_xp_ids, d->extra_entry_point_ids, d->n_extra_entry_points)) != QMX_OK) break;
        if ((rc = upload_array(&g->d_xp_levels, d->extra_entry_point_levels, d->n_extra_entry_points)) != QMX_OK) break;
    } while (0);
    // packed level-0 table (one round trip per hop instead of two); lists longer than m0 or 63 keep the CSR path
    if (rc == QMX_OK && d->n_points && d->m0 <= 63) {
        const uint32_t stride = d->m0 + 1;
        bool fits = true;
        if (!is_device_ptr(d->offsets))
            for (uint64_t i = 0; i < d->n_points && fits; ++i) fits = d->offsets[i + 1] - d->offsets[i] <= d->m0;
        else fits = false;   // device-side arrays are not inspected
        if (fits && hipMalloc((void **)&g->d_l0, (size_t)d->n_points * stride * 4) == hipSuccess) {
            g->l0_stride = stride;
            rc = launch_hnsw_pack_level0(nullptr, g->d_offsets, g->d_neighbors, d->n_points, stride, g->d_l0);
            if (rc == QMX_OK && hipDeviceSynchronize() != hipSuccess) rc = QMX_ERR_OTHER;
        } else {
            (void)hipGetLastError();
            g->d_l0 = nullptr;
        }
    }
    if (rc != QMX_OK) {
        qmx_hnsw_destroy(g);
        return rc;
    }
    *out = g;
    return QMX_OK;
}

int32_t qmx_hnsw_create_from_plain_file(const void *bytes, uint64_t n_bytes, const qmx_hnsw_desc *desc, qmx_hnsw **out) {
    QMX_REQUIRE(bytes && desc && out, QMX_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    QMX_REQUIRE(!is_device_ptr(bytes), QMX_ERR_BAD_ARG, "the links file must be host memory (mmap it)");
    QMX_REQUIRE(n_bytes >= 64, QMX_ERR_BAD_ARG, "links file shorter than its 64-byte header");
    const uint8_t *b = (const uint8_t *)bytes;
    uint64_t hdr[5];
    memcpy(hdr, b, sizeof(hdr));
    const uint64_t point_count = hdr[0], levels_count = hdr[1], total_neighbors = hdr[2], total_offsets = hdr[3], pad = hdr[4];
    QMX_REQUIRE(levels_count != 0xFFFFFFFFFFFFFF01ull && levels_count != 0xFFFFFFFFFFFFFF02ull, QMX_ERR_NOT_SUPPORTED,
                "compressed graph links (header version %llx): re-serialize as Plain first", (unsigned long long)levels_count);
    QMX_REQUIRE(point_count <= 0xFFFFFFFFull && levels_count <= 64 && (pad == 0 || pad == 4), QMX_ERR_BAD_ARG, "not a plain links header");
    // section sizes, with overflow-safe bounds (every count is checked against the file size first)
    QMX_REQUIRE(total_neighbors <= n_bytes / 4 && total_offsets <= n_bytes / 8, QMX_ERR_BAD_ARG, "links header counts exceed the file size");
    const uint64_t off_levels = 64, off_reindex = off_levels + levels_count * 8, off_neigh = off_reindex + point_count * 4,
                   off_offsets = off_neigh + total_neighbors * 4 + pad, end = off_offsets + total_offsets * 8;
    QMX_REQUIRE(end <= n_bytes && off_offsets % 8 == 0, QMX_ERR_BAD_ARG, "links file truncated or misaligned (%llu > %llu)",
                (unsigned long long)end, (unsigned long long)n_bytes);
    QMX_REQUIRE(point_count == 0 || total_offsets >= 1, QMX_ERR_BAD_ARG, "empty offsets section");
    std::vector<uint64_t> level_offsets((size_t)levels_count + 1);
    memcpy(level_offsets.data(), b + off_levels, (size_t)levels_count * 8);
    level_offsets[(size_t)levels_count] = total_offsets ? total_offsets - 1 : 0;
    // the sections are only 4-byte aligned inside an arbitrary buffer: copy what needs 8
    std::vector<uint64_t> offsets((size_t)total_offsets);
    memcpy(offsets.data(), b + off_offsets, (size_t)total_offsets * 8);
    std::vector<uint32_t> reindex((size_t)point_count), neighbors((size_t)total_neighbors);
    memcpy(reindex.data(), b + off_reindex, (size_t)point_count * 4);
    memcpy(neighbors.data(), b + off_neigh, (size_t)total_neighbors * 4);
    qmx_hnsw_desc d = *desc;
    d.n_points = (uint32_t)point_count;
    d.n_levels = (uint32_t)levels_count;
    d.reindex = reindex.data();
    d.level_offsets = level_offsets.data();
    d.offsets = offsets.data();
    d.n_offsets = total_offsets;
    d.neighbors = neighbors.data();
    d.n_neighbors = total_neighbors;
    for (uint64_t i = 0; i < total_neighbors; ++i)
        QMX_REQUIRE(neighbors[i] < point_count, QMX_ERR_OUT_OF_BOUNDS, "link %u out of range", neighbors[i]);
    for (uint64_t i = 0; i < point_count; ++i)
        QMX_REQUIRE(reindex[i] < point_count, QMX_ERR_OUT_OF_BOUNDS, "reindex entry out of range");
    return qmx_hnsw_create(&d, out);
}


// ---------------------------------------------------------------------------------------------
// HNSW build on device (hnsw_build.hpp)
// ---------------------------------------------------------------------------------------------
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static void fill_args_segment(const qmx_segment *s, ScanArgs &a) {
    memset(&a, 0, sizeof(a));
    a.rows = s->d_rows;
    a.n_rows = s->n;
    a.row_stride = s->row_stride;
    a.dim = s->scan_dim;
    const uint32_t eb = elem_bytes(s->dtype);
    const uint32_t full = s->dtype <= QMX_DTYPE_U8 ? s->scan_dim - s->scan_dim % 32 : s->scan_dim;   // as fill_args
    a.nseg = full * eb / 128;
    a.rem_pieces = (full * eb % 128) / 16;
    a.tail_start = full;
    a.del = s->deleted_view();
    a.flags = s->flags;
    if (s->dtype == QMX_DTYPE_SQ_U8) {
        a.sq_multiplier = s->sq.multiplier;
        a.row_offsets = s->d_row_offsets;
        // get_shift (encoded_vectors_u8.rs:116-134)
        float shift = (s->distance == QMX_DISTANCE_DOT || s->distance == QMX_DISTANCE_COSINE)
                          ? (float)s->sq.actual_dim * s->sq.offset * s->sq.offset : 0.0f;
        a.sq_shift = s->sq.invert ? -shift : shift;
    }
    if (s->dtype == QMX_DTYPE_PQ) {
        a.pq_m = s->pq_m;
        a.pq_ncent = s->pq.n_centroids;
        a.pq_pair = s->d_pq_pair;
        a.pq_invert = s->pq.invert;
        a.pq_centroids = s->d_centroids;      // the codebook itself (the table-free build, pq.hip)
        a.pq_dim = s->dim;
        a.pq_chunk = s->pq.chunk_size;
        a.pq_kind = (s->distance == QMX_DISTANCE_DOT || s->distance == QMX_DISTANCE_COSINE) ? 0u : s->distance == QMX_DISTANCE_MANHATTAN ? 1u : 2u;
    }
    if (s->dtype == QMX_DTYPE_BQ) {   // as fill_args; stored <-> stored scores are one-bit
        a.bq_dim = s->dim;
        a.bq_flip = (s->flags & QMX_SEG_BQ_TOGGLE_INVERT) ? 1 : 0;
        a.bq_qbits = 1;
    }
    if (s->dtype == QMX_DTYPE_TQ) {   // as fill_args + the layout of the entries the build makes per batch + score_symmetric's inputs
        uint32_t pieces;
        a.tq_sf = s->d_tq_sf;
        a.tq_l2 = s->d_tq_l2;
        a.tq_bits = s->tq_value_bits;
        a.tq_invert = s->tq_invert ? 1 : 0;
        a.tq_planes = (s->tq_value_bits == 1 && s->d_tq_shift) ? 16 : 8;
        tq_entry_layout(s, &pieces, &a.tq_qbytes_off, &a.aux_off);
        a.q_stride = lds_tile_stride(a.aux_off + QUERY_AUX_BYTES);
        a.bq_qbits = pieces;
        a.tq_code_bytes = s->tq_code_bytes;
        a.tq_ec = TqEc{s->d_tq_weights, s->d_tq_xm, s->tq_weight_scale, s->tq_mm_const};
    }
}

static int32_t launch_hnsw_build_any(const qmx_segment *seg, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu) {
    if (a.mv_offsets) {   // multi-vector points (qmx_multi_hnsw_build)
        if (seg->dtype == QMX_DTYPE_SQ_U8) return launch_hnsw_build_maxsim_sq(nullptr, (int)seg->distance, a, h, phase, grid, per_cu);
        if (seg->dtype == QMX_DTYPE_BQ) return launch_hnsw_build_maxsim_bq(nullptr, a, h, phase, grid, per_cu);
        if (seg->dtype == QMX_DTYPE_PQ) return launch_hnsw_build_maxsim_pq(nullptr, a, h, phase, grid, per_cu);
        if (seg->dtype == QMX_DTYPE_TQ) return launch_hnsw_build_maxsim_tq(nullptr, a, h, phase, grid, per_cu);
        return launch_hnsw_build_maxsim_dense(nullptr, (int)seg->dtype, (int)seg->distance, a, h, phase, grid, per_cu);
    }
    if (seg->dtype == QMX_DTYPE_SQ_U8) return launch_hnsw_build_sq(nullptr, (int)seg->distance, a, h, phase, grid, per_cu);
    if (seg->dtype == QMX_DTYPE_BQ) return launch_hnsw_build_bq(nullptr, a, h, phase, grid, per_cu);
    if (seg->dtype == QMX_DTYPE_PQ) return launch_hnsw_build_pq(nullptr, a, h, phase, grid, per_cu);
    if (tq_l1(seg)) return launch_hnsw_build_tq_l1(nullptr, a, h, phase, grid, per_cu, seg->tq_rot_dim, seg->tq_padded_dim);
    if (seg->dtype == QMX_DTYPE_TQ) return launch_hnsw_build_tq(nullptr, a, h, phase, grid, per_cu);
    return launch_hnsw_build_dense(nullptr, (int)seg->dtype, (int)seg->distance, a, h, phase, grid, per_cu);
}

int32_t qmx_hnsw_get_info(const qmx_hnsw *g, qmx_hnsw_info *out) {
    QMX_REQUIRE(g && out, QMX_ERR_BAD_ARG, "NULL argument");
    out->m = g->m; out->m0 = g->m0; out->n_points = g->n_points; out->n_levels = g->n_levels;
    out->n_offsets = g->n_offsets; out->n_neighbors = g->n_neighbors;
    out->n_entry_points = g->n_ep; out->n_extra_entry_points = g->n_xp;
    return QMX_OK;
}

int32_t qmx_hnsw_export_plain(const qmx_hnsw *g, uint32_t *reindex, uint64_t *level_offsets, uint64_t *offsets, uint32_t *neighbors,
                              uint32_t *ep_ids, uint32_t *ep_levels, uint32_t *xp_ids, uint32_t *xp_levels) {
    QMX_REQUIRE(g, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(g->n_points == 0 || !g->h_offsets.empty(), QMX_ERR_NOT_SUPPORTED, "only graphs built by qmx_hnsw_build keep a host copy to export");
    auto cp = [](void *dst, const void *src, size_t bytes) { if (dst && bytes) memcpy(dst, src, bytes); };
    cp(reindex, g->h_reindex.data(), g->h_reindex.size() * 4);
    cp(level_offsets, g->h_level_offsets.data(), g->h_level_offsets.size() * 8);
    cp(offsets, g->h_offsets.data(), g->h_offsets.size() * 8);
    cp(neighbors, g->h_neighbors.data(), g->h_neighbors.size() * 4);
    cp(ep_ids, g->h_ep_ids.data(), g->h_ep_ids.size() * 4);
    cp(ep_levels, g->h_ep_levels.data(), g->h_ep_levels.size() * 4);
    cp(xp_ids, g->h_xp_ids.data(), g->h_xp_ids.size() * 4);
    cp(xp_levels, g->h_xp_levels.data(), g->h_xp_levels.size() * 4);
    return QMX_OK;
}

int32_t qmx_hnsw_build(const qmx_segment *seg, const qmx_hnsw_build_params *bp, qmx_hnsw **out) {
    return qmx_hnsw_build_quantized(seg, nullptr, bp, out);
}

static int32_t hnsw_build_impl(const qmx_segment *seg, const qmx_segment *original, const qmx_hnsw_build_params *bp, qmx_hnsw **out, const MultiBuild *mb);

int32_t qmx_hnsw_build_quantized(const qmx_segment *seg, const qmx_segment *original, const qmx_hnsw_build_params *bp, qmx_hnsw **out) {
    return hnsw_build_impl(seg, original, bp, out, nullptr);
}

// Build fan-out over independent segments (gpu_devices_manager.rs:120-143 + hnsw/build.rs:53: one device locked per segment build, builds share nothing):
// one host thread per segment, each driving its segment's device; the thread's own error text travels back with its status.
static int32_t sharded_hnsw_build_body(const qmx_segment *const *segments, const qmx_segment *const *originals, uint32_t n_segments,
                                       const qmx_hnsw_build_params *bp, qmx_hnsw **out_graphs, int32_t *out_status);
int32_t qmx_sharded_hnsw_build(const qmx_segment *const *segments, const qmx_segment *const *originals, uint32_t n_segments,
                               const qmx_hnsw_build_params *bp, qmx_hnsw **out_graphs, int32_t *out_status) {
    try {
        return sharded_hnsw_build_body(segments, originals, n_segments, bp, out_graphs, out_status);
    } catch (...) {         // nothing C++ crosses the C ABI
        set_error("qmx_sharded_hnsw_build: out of host memory");
        return QMX_ERR_OUT_OF_MEMORY;
    }
}
static int32_t sharded_hnsw_build_body(const qmx_segment *const *segments, const qmx_segment *const *originals, uint32_t n_segments,
                                       const qmx_hnsw_build_params *bp, qmx_hnsw **out_graphs, int32_t *out_status) {
    QMX_REQUIRE(segments && bp && out_graphs && n_segments >= 1, QMX_ERR_BAD_ARG, "NULL argument");
    for (uint32_t i = 0; i < n_segments; ++i) {
        out_graphs[i] = nullptr;
        if (out_status) out_status[i] = QMX_OK;
    }
    for (uint32_t i = 0; i < n_segments; ++i) QMX_REQUIRE(segments[i], QMX_ERR_BAD_ARG, "segment %u: NULL", i);
    std::vector<int32_t> rcs(n_segments, QMX_OK);
    std::vector<std::string> errs(n_segments);
    auto work = [&](uint32_t i) {
        try {
            rcs[i] = hnsw_build_impl(segments[i], originals ? originals[i] : nullptr, bp, &out_graphs[i], nullptr);
            if (rcs[i] != QMX_OK) errs[i] = last_error_text();          // (thread-local: copied out before the thread ends)
        } catch (...) {     // (bad_alloc of a host vector inside the build: a status, not a terminate)
            rcs[i] = QMX_ERR_OUT_OF_MEMORY;
        }
    };
    if (n_segments == 1) {
        work(0);
    } else {
        // One worker thread per DEVICE x QMX_BUILDS_PER_DEVICE (the reference locks one GPU of its pool per segment build, gpu_devices_manager.rs:120-143;
        // here a device takes up to two builds at once: each holds its full scratch - visited bitmaps, selection buffers - so the count is bounded).
        // Workers draw segments of their device from a shared cursor.  No C++ exception leaves this extern "C" function: a thread that cannot be
        // started leaves its share to the calling thread, which runs whatever is left inline after joining the rest.
        constexpr uint32_t QMX_BUILDS_PER_DEVICE = 2;
        std::vector<int> devs;
        for (uint32_t i = 0; i < n_segments; ++i)
            if (std::find(devs.begin(), devs.end(), segments[i]->device) == devs.end()) devs.push_back(segments[i]->device);
        std::vector<std::atomic<uint32_t>> cursor(devs.size());
        for (auto &c : cursor) c.store(0);
        auto drain = [&](size_t d) {      // the next unbuilt segment of device d, until none is left
            for (;;) {
                const uint32_t start = cursor[d].fetch_add(1);
                uint32_t seen = 0, pick = n_segments;
                for (uint32_t i = 0; i < n_segments; ++i)
                    if (segments[i]->device == devs[d] && seen++ == start) { pick = i; break; }
                if (pick == n_segments) return;
                work(pick);
            }
        };
        std::vector<std::thread> pool;
        try {
            pool.reserve(devs.size() * QMX_BUILDS_PER_DEVICE);
            for (size_t d = 0; d < devs.size(); ++d)
                for (uint32_t k = 0; k < QMX_BUILDS_PER_DEVICE; ++k) pool.emplace_back(drain, d);
        } catch (...) {
            // (std::system_error from the thread constructor, bad_alloc: keep what started)
        }
        for (auto &t : pool)
            if (t.joinable()) t.join();
        for (size_t d = 0; d < devs.size(); ++d) drain(d);      // whatever no worker took (all of it if no thread could be started)
    }
    int32_t first = QMX_OK;
    for (uint32_t i = 0; i < n_segments; ++i) {
        if (out_status) out_status[i] = rcs[i];
        if (rcs[i] != QMX_OK && first == QMX_OK) {
            first = rcs[i];
            set_error("segment %u: %s", i, errs[i].c_str());
        }
    }
    return first;
}

int32_t qmx_multi_hnsw_build_quantized(const qmx_segment *inner, const qmx_segment *original_inner, const uint64_t *point_offsets, uint32_t n_points,
                                       const uint64_t *point_deleted, uint64_t n_deleted_bits, const qmx_hnsw_build_params *bp, qmx_hnsw **out) {
    QMX_REQUIRE(inner && point_offsets && bp && out, QMX_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    QMX_REQUIRE(inner->dtype == QMX_DTYPE_F32 || inner->dtype == QMX_DTYPE_F16 || inner->dtype == QMX_DTYPE_SQ_U8 || inner->dtype == QMX_DTYPE_BQ ||
                    inner->dtype == QMX_DTYPE_PQ || (inner->dtype == QMX_DTYPE_TQ && !tq_l1(inner)),
                QMX_ERR_NOT_SUPPORTED, "device HNSW build over multi-vectors: inner dtype %u not supported (f32, f16, SQ, BQ, PQ, TurboQuant except over Manhattan)",
                inner->dtype);
    QMX_REQUIRE(!is_device_ptr(point_offsets) && !is_device_ptr(point_deleted), QMX_ERR_BAD_ARG, "point_offsets and point_deleted are host arrays");
    for (uint32_t p = 0; p < n_points; ++p)
        QMX_REQUIRE(point_offsets[p] <= point_offsets[p + 1], QMX_ERR_BAD_ARG, "point_offsets is not ascending at %u", p);
    QMX_REQUIRE(point_offsets[n_points] <= inner->n, QMX_ERR_OUT_OF_BOUNDS, "point_offsets reach past the %llu inner rows of the segment",
                (unsigned long long)inner->n);
    const MultiBuild mb{point_offsets, n_points, (point_deleted && n_deleted_bits) ? point_deleted : nullptr, point_deleted ? n_deleted_bits : 0};
    return hnsw_build_impl(inner, (inner->dtype == QMX_DTYPE_PQ || inner->dtype == QMX_DTYPE_TQ) ? original_inner : nullptr, bp, out, &mb);
}
int32_t qmx_multi_hnsw_build(const qmx_segment *inner, const uint64_t *point_offsets, uint32_t n_points, const uint64_t *point_deleted,
                             uint64_t n_deleted_bits, const qmx_hnsw_build_params *bp, qmx_hnsw **out) {
    return qmx_multi_hnsw_build_quantized(inner, nullptr, point_offsets, n_points, point_deleted, n_deleted_bits, bp, out);
}

static int32_t hnsw_build_impl(const qmx_segment *seg, const qmx_segment *original, const qmx_hnsw_build_params *bp, qmx_hnsw **out, const MultiBuild *mb) {
    QMX_REQUIRE(seg && bp && out, QMX_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    QMX_REQUIRE(seg->dtype <= QMX_DTYPE_BQ || seg->dtype == QMX_DTYPE_TQ, QMX_ERR_NOT_SUPPORTED, "device HNSW build: dtype %u not supported", seg->dtype);
    const bool from_original = seg->dtype == QMX_DTYPE_PQ || seg->dtype == QMX_DTYPE_TQ;
    if (from_original) {   // point_scorer.rs:197-212: the insertion searches score through the query (PQ: LUT) of the ORIGINAL vector
        QMX_REQUIRE(original, QMX_ERR_NOT_SUPPORTED,
                    "a PQ / TurboQuant segment cannot score a stored row as a query (encode_internal_vector -> None): pass the original f32 segment to qmx_hnsw_build_quantized");
        QMX_REQUIRE(original->dtype == QMX_DTYPE_F32 && original->dim == seg->dim && original->n >= seg->n && original->device == seg->device,
                    QMX_ERR_BAD_ARG, "the original segment must be f32, of the same dim, on the same device and hold every row of the quantized segment");
        QMX_REQUIRE(seg->dtype != QMX_DTYPE_PQ || seg->d_pq_pair ||
                        (!mb && !option(OPT_HNSW_PQ_TABLE_BUILD) && pq_direct_walk_ok(seg->dim, seg->pq_m, seg->pq.chunk_size, seg->pq.n_centroids)),
                    QMX_ERR_NOT_SUPPORTED,
                    "PQ build: the centroid pair table (m x n_centroids^2 floats) exceeds 256 MB");
    }
    QMX_REQUIRE(seg->dtype == QMX_DTYPE_SQ_U8 || seg->dtype == QMX_DTYPE_PQ || seg->fast_layout(), QMX_ERR_NOT_SUPPORTED,
                "adopted device block is not 16-byte aligned");
    QMX_REQUIRE(bp->m >= 1 && bp->m0 >= bp->m && bp->m0 <= HNSW_BUILD_MAX_M0, QMX_ERR_BAD_ARG, "need 1 <= m <= m0 <= %u", HNSW_BUILD_MAX_M0);
    QMX_REQUIRE(bp->ef_construct >= 1 && bp->ef_construct <= HNSW_MAX_EF, QMX_ERR_NOT_SUPPORTED, "ef_construct %u not in 1..%u",
                bp->ef_construct, HNSW_MAX_EF);
    QMX_REQUIRE(seg->n <= 0xFFFFFFFFull, QMX_ERR_BAD_ARG, "too many rows");
    QMX_HIP(hipSetDevice(seg->device));
    const uint32_t n = mb ? mb->n_points : (uint32_t)seg->n, m = bp->m, m0 = bp->m0;
    const uint32_t max_batch = bp->max_batch ? bp->max_batch : 16384;

    // ---- levels (graph_layers_builder.rs:388-396), the same draw as the CPU oracle ----
    std::vector<uint8_t> level(std::max<uint32_t>(n, 1));
    std::vector<uint32_t> up_off(std::max<uint32_t>(n, 1));
    const double level_factor = 1.0 / log((double)(m > 2 ? m : 2));
    uint64_t n_up = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t r = splitmix64(bp->seed ^ (0xA0761D6478BD642Full * ((uint64_t)i + 1)));
        const double u = ((double)(r >> 11) + 0.5) * (1.0 / 9007199254740992.0);
        double lv = round(-log(u) * level_factor);
        if (lv > (double)(HNSW_BUILD_MAX_LEVELS - 1)) lv = HNSW_BUILD_MAX_LEVELS - 1;
        level[i] = (uint8_t)lv;
        up_off[i] = (uint32_t)n_up;
        n_up += level[i];
    }
    QMX_REQUIRE(n_up <= 0xFFFFFFFFull, QMX_ERR_BAD_ARG, "too many upper-level lists");
    // deleted flags on the host: deleted points are never indexed and never entry points
    std::vector<uint64_t> pdel, vdel;
    if (mb) {
        if (mb->h_deleted) pdel.assign(mb->h_deleted, mb->h_deleted + (mb->n_deleted_bits + 63) / 64);
    } else
    if (seg->d_point_deleted) { pdel.resize((seg->n_point_bits + 63) / 64); QMX_HIP(hipMemcpy(pdel.data(), seg->d_point_deleted, pdel.size() * 8, hipMemcpyDeviceToHost)); }
    if (!mb && seg->d_vec_deleted) { vdel.resize((seg->n_vec_bits + 63) / 64); QMX_HIP(hipMemcpy(vdel.data(), seg->d_vec_deleted, vdel.size() * 8, hipMemcpyDeviceToHost)); }
    const uint64_t n_point_bits = mb ? mb->n_deleted_bits : seg->n_point_bits;
    auto live = [&](uint32_t id) {
        const bool vd = (!vdel.empty() && id < seg->n_vec_bits) ? ((vdel[id >> 6] >> (id & 63)) & 1) : false;
        const bool pd = !pdel.empty() ? (id < n_point_bits ? ((pdel[id >> 6] >> (id & 63)) & 1) : true) : false;
        return !vd && !pd;
    };

    // ---- device state ----
    DevBuf b_level, b_upoff, b_links0, b_cnt0, b_linksU, b_cntU, b_lock, b_vis, b_log, b_sel, b_sels, b_selc, b_normf, b_normi, b_bq, b_bqsrc, b_rot, b_next, b_mvoff,
        b_mvdel;
    auto release_all = [&]() {
        for (DevBuf *b : {&b_level, &b_upoff, &b_links0, &b_cnt0, &b_linksU, &b_cntU, &b_lock, &b_vis, &b_log, &b_sel, &b_sels, &b_selc, &b_normf, &b_normi,
                          &b_bq, &b_bqsrc, &b_rot, &b_next, &b_mvoff, &b_mvdel}) b->release();
    };
    int32_t rc = QMX_OK;
    qmx_hnsw *g = nullptr;
    do {
#define QB(expr) if ((rc = (expr)) != QMX_OK) break
#define QH(expr) if ((expr) != hipSuccess) { rc = hip_status(hipGetLastError(), #expr, __FILE__, __LINE__); if (rc == QMX_OK) rc = QMX_ERR_OTHER; break; }
        const size_t nn = std::max<uint32_t>(n, 1);
        QB(b_level.reserve(nn)); QB(b_upoff.reserve(nn * 4));
        QB(b_links0.reserve(nn * m0 * 4)); QB(b_cnt0.reserve(nn * 4));
        QB(b_linksU.reserve(std::max<uint64_t>(n_up, 1) * m * 4)); QB(b_cntU.reserve(std::max<uint64_t>(n_up, 1) * 4));
        QB(b_lock.reserve(nn * 4));
        QH(hipMemcpy(b_level.p, level.data(), nn, hipMemcpyHostToDevice));
        QH(hipMemcpy(b_upoff.p, up_off.data(), nn * 4, hipMemcpyHostToDevice));
        QH(hipMemset(b_cnt0.p, 0, nn * 4));
        QH(hipMemset(b_cntU.p, 0, std::max<uint64_t>(n_up, 1) * 4));
        QH(hipMemset(b_lock.p, 0, nn * 4));
        QB(b_sel.reserve((size_t)max_batch * HNSW_BUILD_MAX_LEVELS * m0 * 4));
        QB(b_sels.reserve((size_t)max_batch * HNSW_BUILD_MAX_LEVELS * m0 * 4));
        QB(b_selc.reserve((size_t)max_batch * HNSW_BUILD_MAX_LEVELS * 4));

        ScanArgs a;
        fill_args_segment(seg, a);
        if (mb) {   // the graph's points are multi-vectors: offsets into the inner rows, deletion per POINT (the inner rows carry no flags of their own)
            QB(b_mvoff.reserve((size_t)(n + 1) * 8));
            QH(hipMemcpy(b_mvoff.p, mb->h_offsets, (size_t)(n + 1) * 8, hipMemcpyHostToDevice));
            a.mv_offsets = (const uint64_t *)b_mvoff.p;
            DeletedView dv;
            memset(&dv, 0, sizeof(dv));
            dv.n_rows = n;
            if (!pdel.empty()) {
                QB(b_mvdel.reserve(pdel.size() * 8));
                QH(hipMemcpy(b_mvdel.p, pdel.data(), pdel.size() * 8, hipMemcpyHostToDevice));
                dv.point_deleted = (const uint64_t *)b_mvdel.p;
                dv.n_point_bits = mb->n_deleted_bits;
            }
            a.del = dv;
        }
        HnswBuildArgs h;
        memset(&h, 0, sizeof(h));
        h.g.links0 = (uint32_t *)b_links0.p; h.g.cnt0 = (uint32_t *)b_cnt0.p; h.g.linksU = (uint32_t *)b_linksU.p; h.g.cntU = (uint32_t *)b_cntU.p;
        h.g.up_off = (const uint32_t *)b_upoff.p; h.g.m = m; h.g.m0 = m0;
        h.level = (const uint8_t *)b_level.p;
        h.n_points = n;
        h.ef_construct = bp->ef_construct;
        h.sel_ids = (uint32_t *)b_sel.p; h.sel_scores = (float *)b_sels.p; h.sel_cnt = (uint32_t *)b_selc.p;
        h.lock = (uint32_t *)b_lock.p;
        // bytes of a row as it lies in HBM: the SQ block holds the codes only (the vector_offset column is separate)
        const uint64_t dev_row_bytes = seg->dtype == QMX_DTYPE_SQ_U8 ? (uint64_t)seg->sq.actual_dim : seg->row_bytes;
        h.row_bytes = (uint32_t)dev_row_bytes;
        h.lds_query_bytes = (uint32_t)((dev_row_bytes + 127) / 128 * 128 + 128);
        uint64_t lut_stride = 0;
        uint64_t max_entries = max_batch;      // query entries a batch may need: one per point - or, multi-vector points, one per inner vector
        bool pq_direct_build = false;
        if (seg->dtype == QMX_DTYPE_PQ) {   // query entries = LUTs of the batch's original vectors, read through L2 (as the PQ walk does)
            lut_stride = ((uint64_t)seg->pq_m * seg->pq.n_centroids * sizeof(float) + 15) & ~15ull;
            if (mb) {   // at least the longest point, at most 1 GiB of LUTs (the insertion loop shortens a batch that would need more)
                uint64_t longest = 1;
                for (uint32_t p = 0; p < n; ++p) longest = std::max<uint64_t>(longest, mb->h_offsets[p + 1] - mb->h_offsets[p]);
                max_entries = std::max<uint64_t>(longest, std::min<uint64_t>(mb->h_offsets[n] ? mb->h_offsets[n] : 1, (1ull << 30) / lut_stride));
                a.q_stride = (uint32_t)lut_stride;
            }
            // the table-free build (pq.hip HopPQDirectBuild + HopPQInternalDirect; the default where the codebook allows, option hnsw_pq_table_build for the
            // other): the entries are the preprocessed original vectors themselves, staged in LDS per insertion - no LUTs are made, and none are reserved
            // (max_entries x lut_stride is 1.6 GB at m = 96: several builds on one device would multiply it for nothing)
            pq_direct_build = !mb && !option(OPT_HNSW_PQ_TABLE_BUILD) && pq_direct_walk_ok(seg->dim, seg->pq_m, seg->pq.chunk_size, seg->pq.n_centroids);
            if (!pq_direct_build) QB(b_bq.reserve((size_t)max_entries * lut_stride));
            QB(b_bqsrc.reserve((size_t)max_entries * seg->dim * sizeof(float)));
            h.batch_queries = (const unsigned char *)b_bq.p;
            h.batch_q_stride = lut_stride;
            h.lds_query_bytes = 0;
            if (pq_direct_build) {
                h.batch_queries = (const unsigned char *)b_bqsrc.p;
                h.batch_q_stride = (uint64_t)seg->dim * 4;
                h.lds_query_bytes = seg->dim * 4;
                // (round 5 tried an 8-bit LUT image per new point behind its vector, to prefilter the hops of the insertion searches: the same graph,
                // 25.6 s against 24.9 s at 2 M x 1536 points - gone from the code since round 6, profiles/r5_walk_variants_pq_hop_prefilter.jsonl)
            }
        }
        if (seg->dtype == QMX_DTYPE_TQ) {   // query entries = precompute_query of the batch's original vectors, staged in LDS per insertion
            if (mb) {   // one entry per inner vector of the batch: at least the longest point, at most 256 MiB of rotated vectors
                uint64_t longest = 1;
                for (uint32_t p = 0; p < n; ++p) longest = std::max<uint64_t>(longest, mb->h_offsets[p + 1] - mb->h_offsets[p]);
                max_entries = std::max<uint64_t>(longest, std::min<uint64_t>(mb->h_offsets[n] ? mb->h_offsets[n] : 1,
                                                                             (1ull << 28) / ((uint64_t)seg->tq_padded_dim * sizeof(double))));
            }
            QB(b_bq.reserve((size_t)max_entries * a.q_stride));
            QB(b_bqsrc.reserve((size_t)max_entries * seg->dim * sizeof(float)));
            QB(b_rot.reserve((size_t)max_entries * seg->tq_padded_dim * sizeof(double)));
            h.batch_queries = (const unsigned char *)b_bq.p;
            h.batch_q_stride = a.q_stride;
            h.lds_query_bytes = a.q_stride;
            if (tq_l1(seg)) {   // over Manhattan the entry is the original vector as given (quantization.rs:532-535), its hop scratch behind it in LDS (tq_l1_policy.hpp)
                a.tq_l1 = seg->d_tq_l1;
                a.q_stride = tq_l1_query_bytes(seg->dim);
                QB(b_bq.reserve((size_t)max_batch * a.q_stride));
                h.batch_queries = (const unsigned char *)b_bq.p;
                h.batch_q_stride = a.q_stride;
                h.lds_query_bytes = tq_l1_lds_bytes(seg->dim, seg->tq_rot_dim);
            }
        }
        if (mb) h.lds_query_bytes = 0;      // nothing staged: the inner rows of the new point are read where they lie
        if (seg->dtype == QMX_DTYPE_U8 && seg->distance == QMX_DISTANCE_COSINE && seg->dim >= 32) {   // the per-pair cosine's query norm of a stored row
            QB(b_normf.reserve(nn * 4)); QB(b_normi.reserve(nn * 4));
            QB(launch_u8_row_norms(nullptr, seg->d_rows, seg->row_stride, n, seg->dim, seg->flags, (float *)b_normf.p, (int32_t *)b_normi.p));
            a.row_norms_f = (const float *)b_normf.p;
            a.row_norms_i = (const int32_t *)b_normi.p;
        }
        if (h.lds_query_bytes > HNSW_LDS_QUERY_MAX) {
            set_error("rows of %llu bytes do not fit the LDS query slot", (unsigned long long)dev_row_bytes);
            rc = QMX_ERR_NOT_SUPPORTED;
            break;
        }
        h.log_cap = 16384;
        h.vis_words = ((uint64_t)n + 31) / 32;
        if (h.vis_words == 0) h.vis_words = 1;
        int per_cu1 = 1, per_cu2 = 1;
        QB(launch_hnsw_build_any(seg, a, h, 1, 0, &per_cu1));
        QB(launch_hnsw_build_any(seg, a, h, 2, 0, &per_cu2));
        uint64_t slots1 = std::min<uint64_t>({(uint64_t)seg->num_cus * per_cu1, (uint64_t)HNSW_SLOT_CAP, (uint64_t)max_batch});
        slots1 = std::max<uint64_t>(1, std::min<uint64_t>(slots1, HNSW_VIS_BUDGET / (h.vis_words * 4)));
        const uint64_t slots2 = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)seg->num_cus * per_cu2, max_batch));
        QB(b_next.reserve(8));
        h.next = (uint32_t *)b_next.p;
        QB(b_vis.reserve((size_t)slots1 * h.vis_words * 4));
        QB(b_log.reserve((size_t)slots1 * h.log_cap * 4));
        QH(hipMemset(b_vis.p, 0, (size_t)slots1 * h.vis_words * 4));
        h.visited = (uint32_t *)b_vis.p;
        h.vis_log = (uint32_t *)b_log.p;

        // ---- insertion loop ----
        // EntryPoints (entry_points.rs:46-94) kept on the host: the live point of the highest level seen first is the
        // entry; the `entry_points_num` highest others are the extra entries
        bool have_ep = false;
        uint32_t ep_id = 0, ep_level = 0, inserted = 0;
        std::vector<std::pair<uint32_t, uint32_t>> extra;   // (level, id)
        auto note_point = [&](uint32_t id) {
            const uint32_t lv = level[id];
            if (!have_ep) { have_ep = true; ep_id = id; ep_level = lv; return; }
            std::pair<uint32_t, uint32_t> other(lv, id);
            if (lv > ep_level) { other = {ep_level, ep_id}; ep_id = id; ep_level = lv; }
            if (bp->entry_points_num == 0) return;
            if (extra.size() < bp->entry_points_num) { extra.push_back(other); return; }
            size_t lo = 0;
            for (size_t i = 1; i < extra.size(); ++i) if (extra[i].first < extra[lo].first) lo = i;
            if (extra[lo].first < other.first) extra[lo] = other;
        };
        uint32_t next = 0;
        while (next < n && rc == QMX_OK) {
            if (!have_ep) {                      // the first live point: nothing to link to
                if (live(next)) { note_point(next); ++inserted; }
                ++next;
                continue;
            }
            uint32_t count = std::min<uint32_t>({max_batch, std::max<uint32_t>(1, inserted / 32), n - next});
            // a point above the current top level ends its batch: the next batch starts from it
            for (uint32_t i = 0; i < count; ++i)
                if (level[next + i] > ep_level && live(next + i)) { count = i + 1; break; }
            if (mb && from_original)      // the batch's inner vectors must fit the entries (a single point always does)
                while (count > 1 && mb->h_offsets[next + count] - mb->h_offsets[next] > max_entries) --count;
            h.first = next; h.count = count; h.ep_id = ep_id; h.ep_level = ep_level;
            if (seg->dtype == QMX_DTYPE_PQ) {
                // quantized_vectors.raw_scorer(original vector): Metric::preprocess (quantized_query_scorer.rs:39-41; identity for a row
                // normalised at insert, up to the reference's 1e-6 rule), then EncodedVectorsPQ::encode_query for every point of the batch
                // (multi-vector points: for every inner vector of the batch's points, in storage order)
                const uint64_t r0 = mb ? mb->h_offsets[next] : next, nr = mb ? mb->h_offsets[next + count] - r0 : count;
                float *src = (float *)b_bqsrc.p;
                if (nr) {
                    QH(hipMemcpy2DAsync(src, (size_t)seg->dim * 4, (const char *)original->d_rows + r0 * original->row_stride, original->row_stride,
                                        (size_t)seg->dim * 4, nr, hipMemcpyDeviceToDevice, nullptr));
                    if (seg->distance == QMX_DISTANCE_COSINE) QB(launch_cosine_preprocess_f32(nullptr, src, src, nr, seg->dim));
                    if (!pq_direct_build) QB(launch_pq_lut(nullptr, seg->distance, seg->dim, seg->pq, seg->d_centroids, src, (uint32_t)nr, (float *)b_bq.p));
                }
            }
            if (tq_l1(seg)) {   // EncodedVectorsTQ over Manhattan: no preprocessing, no rotation - the rows themselves, zero padded to whole 16 bytes
                QH(hipMemsetAsync(b_bq.p, 0, (size_t)count * a.q_stride, nullptr));
                QH(hipMemcpy2DAsync(b_bq.p, a.q_stride, (const char *)original->d_rows + (uint64_t)next * original->row_stride, original->row_stride,
                                    (size_t)seg->dim * 4, count, hipMemcpyDeviceToDevice, nullptr));
            } else if (seg->dtype == QMX_DTYPE_TQ) {   // the same for EncodedVectorsTQ: preprocess, rotate, TurboQuantizer::precompute_query
                const uint64_t r0 = mb ? mb->h_offsets[next] : next, nr = mb ? mb->h_offsets[next + count] - r0 : count;      // (multi-vector points: every inner vector)
                float *src = (float *)b_bqsrc.p;
                if (nr) {
                    QH(hipMemcpy2DAsync(src, (size_t)seg->dim * 4, (const char *)original->d_rows + r0 * original->row_stride, original->row_stride,
                                        (size_t)seg->dim * 4, nr, hipMemcpyDeviceToDevice, nullptr));
                    if (seg->distance == QMX_DISTANCE_COSINE) QB(launch_cosine_preprocess_f32(nullptr, src, src, nr, seg->dim));
                    QB(launch_tq_rotate(nullptr, src, (uint32_t)nr, tq_rotation(seg), (double *)b_rot.p));
                    QB(launch_tq_query_encode(nullptr, (double *)b_rot.p, (uint32_t)nr, seg->tq_padded_dim, seg->tq_value_bits, seg->distance == QMX_DISTANCE_EUCLID ? 1 : 0,
                                              b_bq.p, a.q_stride, a.aux_off, seg->d_tq_shift, seg->d_tq_scale, a.tq_qbytes_off));
                }
            }
            const uint32_t grid1 = (uint32_t)std::min<uint64_t>(slots1, count), grid2 = (uint32_t)std::min<uint64_t>(slots2, count);
            if (h.next) {
                const uint32_t start[2] = {grid1, grid2};
                QH(hipMemcpyAsync(h.next, start, sizeof(start), hipMemcpyHostToDevice, nullptr));      // (pageable source: the copy is staged before the call returns)
            }
            QB(launch_hnsw_build_any(seg, a, h, 1, grid1, &per_cu1));
            QB(launch_hnsw_build_any(seg, a, h, 2, grid2, &per_cu2));
            for (uint32_t i = 0; i < count; ++i)
                if (live(next + i)) { note_point(next + i); ++inserted; }
            next += count;
        }
        if (rc != QMX_OK) break;
        QH(hipDeviceSynchronize());

        // ---- export: fixed-capacity lists -> plain GraphLinks arrays (graph_links/serializer.rs:52-209) ----
        std::vector<uint32_t> links0((size_t)nn * m0), cnt0(nn), linksU(std::max<uint64_t>(n_up, 1) * m), cntU(std::max<uint64_t>(n_up, 1));
        QH(hipMemcpy(links0.data(), b_links0.p, links0.size() * 4, hipMemcpyDeviceToHost));
        QH(hipMemcpy(cnt0.data(), b_cnt0.p, cnt0.size() * 4, hipMemcpyDeviceToHost));
        QH(hipMemcpy(linksU.data(), b_linksU.p, linksU.size() * 4, hipMemcpyDeviceToHost));
        QH(hipMemcpy(cntU.data(), b_cntU.p, cntU.size() * 4, hipMemcpyDeviceToHost));
        release_all();
        uint32_t maxl = 0;
        for (uint32_t i = 0; i < n; ++i) maxl = std::max<uint32_t>(maxl, level[i]);
        const uint32_t L = n ? maxl + 1 : 0;
        g = new (std::nothrow) qmx_hnsw();
        if (!g) { rc = QMX_ERR_OUT_OF_MEMORY; break; }
        std::vector<uint64_t> count_ge(L + 1, 0);
        for (uint32_t i = 0; i < n; ++i) for (uint32_t l = 0; l <= level[i]; ++l) count_ge[l]++;
        // back_index: points by descending level, ties by id
        std::vector<uint32_t> back(nn);
        {
            std::vector<uint64_t> start(L + 1, 0);
            uint64_t acc = 0;
            for (int32_t l = (int32_t)L - 1; l >= 0; --l) { start[l] = acc; acc += count_ge[l] - (l + 1 < (int32_t)L ? count_ge[l + 1] : 0); }
            for (uint32_t i = 0; i < n; ++i) back[start[level[i]]++] = i;
        }
        g->h_reindex.resize(n);
        for (uint32_t i = 0; i < n; ++i) g->h_reindex[back[i]] = i;
        uint64_t total_slots = 0;
        for (uint32_t l = 0; l < L; ++l) total_slots += count_ge[l];
        g->h_level_offsets.assign(L + 1, 0);
        g->h_offsets.assign(total_slots + 1, 0);
        uint64_t nnb = 0;
        for (uint32_t i = 0; i < n; ++i) { nnb += cnt0[i]; for (uint32_t l = 1; l <= level[i]; ++l) nnb += cntU[up_off[i] + l - 1]; }
        g->h_neighbors.resize(nnb);
        uint64_t off = 0, slot = 0;
        for (uint32_t l = 0; l < L; ++l) {
            g->h_level_offsets[l] = slot;
            for (uint64_t j = 0; j < count_ge[l]; ++j) {
                const uint32_t id = l == 0 ? (uint32_t)j : back[j];
                g->h_offsets[slot++] = off;
                const uint32_t len = l == 0 ? cnt0[id] : cntU[up_off[id] + l - 1];
                const uint32_t *src = l == 0 ? &links0[(size_t)id * m0] : &linksU[((size_t)up_off[id] + l - 1) * m];
                memcpy(g->h_neighbors.data() + off, src, (size_t)len * 4);
                off += len;
            }
        }
        g->h_level_offsets[L] = slot;
        g->h_offsets[slot] = off;
        if (have_ep) { g->h_ep_ids.push_back(ep_id); g->h_ep_levels.push_back(ep_level); }
        for (auto &e : extra) { g->h_xp_ids.push_back(e.second); g->h_xp_levels.push_back(e.first); }
        qmx_hnsw_desc d;
        memset(&d, 0, sizeof(d));
        d.m = m; d.m0 = m0; d.n_points = n; d.n_levels = L;
        d.reindex = g->h_reindex.data(); d.level_offsets = g->h_level_offsets.data(); d.offsets = g->h_offsets.data();
        d.n_offsets = g->h_offsets.size(); d.neighbors = g->h_neighbors.data(); d.n_neighbors = g->h_neighbors.size();
        d.entry_point_ids = g->h_ep_ids.data(); d.entry_point_levels = g->h_ep_levels.data(); d.n_entry_points = (uint32_t)g->h_ep_ids.size();
        d.extra_entry_point_ids = g->h_xp_ids.data(); d.extra_entry_point_levels = g->h_xp_levels.data();
        d.n_extra_entry_points = (uint32_t)g->h_xp_ids.size();
        d.device_id = seg->device;
        qmx_hnsw *dev = nullptr;
        QB(qmx_hnsw_create(&d, &dev));
        // move the device arrays into g (which owns the host copy)
        g->device = dev->device; g->m = dev->m; g->m0 = dev->m0; g->n_points = dev->n_points; g->n_levels = dev->n_levels;
        g->n_ep = dev->n_ep; g->n_xp = dev->n_xp; g->n_offsets = dev->n_offsets; g->n_neighbors = dev->n_neighbors;
        g->d_reindex = dev->d_reindex; g->d_neighbors = dev->d_neighbors; g->d_ep_ids = dev->d_ep_ids; g->d_ep_levels = dev->d_ep_levels;
        g->d_xp_ids = dev->d_xp_ids; g->d_xp_levels = dev->d_xp_levels; g->d_level_offsets = dev->d_level_offsets; g->d_offsets = dev->d_offsets;
        g->d_l0 = dev->d_l0; g->l0_stride = dev->l0_stride;
        delete dev;
#undef QB
#undef QH
    } while (0);
    release_all();
    if (rc != QMX_OK) {
        if (g) qmx_hnsw_destroy(g);
        return rc;
    }
    *out = g;
    return QMX_OK;
}



static int32_t launch_hnsw(const qmx_query *q, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    const qmx_segment *s = q->seg;
    if (a.cq_desc && a.mv_offsets) {
        if (s->dtype <= QMX_DTYPE_U8) {
            QMX_REQUIRE(s->fast_layout(), QMX_ERR_NOT_SUPPORTED, "adopted device block is not 16-byte aligned");
            return launch_hnsw_custom_maxsim_dense(q->stream, (int)s->dtype, (int)s->distance, a, h, grid, per_cu);
        }
        if (s->dtype == QMX_DTYPE_SQ_U8) return launch_hnsw_custom_maxsim_sq(q->stream, (int)s->distance, a, h, grid, per_cu);
        set_error("custom walk over multi-vector points: inner rows of dtype %u are not built (dense and SQ are)", s->dtype);
        return QMX_ERR_NOT_SUPPORTED;
    }
    if (a.cq_desc) {
        if (s->dtype <= QMX_DTYPE_U8) {
            QMX_REQUIRE(s->fast_layout(), QMX_ERR_NOT_SUPPORTED, "adopted device block is not 16-byte aligned");
            return launch_hnsw_custom_dense(q->stream, (int)s->dtype, (int)s->distance, a, h, grid, per_cu);
        }
        if (s->dtype == QMX_DTYPE_SQ_U8) return launch_hnsw_custom_sq(q->stream, (int)s->distance, a, h, grid, per_cu);
        if (s->dtype == QMX_DTYPE_PQ) return launch_hnsw_custom_pq(q->stream, a, h, grid, per_cu);
        if (s->dtype == QMX_DTYPE_BQ) return launch_hnsw_custom_bq(q->stream, a, h, grid, per_cu);
        if (tq_l1(s)) return launch_hnsw_custom_tq_l1(q->stream, a, h, grid, per_cu, s->tq_rot_dim);
        if (s->dtype == QMX_DTYPE_TQ) return launch_hnsw_custom_tq(q->stream, a, h, grid, per_cu);
        set_error("dtype %u not built yet", s->dtype);
        return QMX_ERR_NOT_SUPPORTED;
    }
    if (a.mv_offsets) {
        if (s->dtype <= QMX_DTYPE_U8) {
            QMX_REQUIRE(s->fast_layout(), QMX_ERR_NOT_SUPPORTED, "adopted device block is not 16-byte aligned");
            return launch_hnsw_maxsim_dense(q->stream, (int)s->dtype, (int)s->distance, a, h, grid, per_cu);
        }
        if (s->dtype == QMX_DTYPE_SQ_U8) return launch_hnsw_maxsim_sq(q->stream, (int)s->distance, a, h, grid, per_cu);
        if (s->dtype == QMX_DTYPE_BQ) return launch_hnsw_maxsim_bq(q->stream, a, h, grid, per_cu);
        if (s->dtype == QMX_DTYPE_PQ) return launch_hnsw_maxsim_pq(q->stream, a, h, grid, per_cu);
        if (s->dtype == QMX_DTYPE_TQ) return launch_hnsw_maxsim_tq(q->stream, a, h, grid, per_cu);      // (over Manhattan: refused by hnsw_enqueue)
        set_error("MaxSim walk: inner rows of dtype %u are not built", s->dtype);
        return QMX_ERR_NOT_SUPPORTED;
    }
    if (s->dtype <= QMX_DTYPE_U8) {
        QMX_REQUIRE(s->fast_layout(), QMX_ERR_NOT_SUPPORTED, "adopted device block is not 16-byte aligned");
        return launch_hnsw_dense(q->stream, (int)s->dtype, (int)s->distance, a, h, grid, per_cu);
    }
    if (s->dtype == QMX_DTYPE_SQ_U8) return launch_hnsw_sq(q->stream, (int)s->distance, a, h, grid, per_cu);
    if (s->dtype == QMX_DTYPE_PQ && a.queries == q->enc.p && !a.cq_desc && !a.mv_offsets) return launch_hnsw_pq_direct(q->stream, a, h, grid, per_cu);
    if (s->dtype == QMX_DTYPE_PQ) {
        return launch_hnsw_pq(q->stream, a, h, grid, per_cu);
    }
    if (s->dtype == QMX_DTYPE_BQ) return launch_hnsw_bq(q->stream, a, h, grid, per_cu);
    if (tq_l1(s)) return launch_hnsw_tq_l1(q->stream, a, h, grid, per_cu, s->tq_rot_dim);
    if (s->dtype == QMX_DTYPE_TQ) return launch_hnsw_tq(q->stream, a, h, grid, per_cu);
    set_error("dtype %u not built yet", s->dtype);
    return QMX_ERR_NOT_SUPPORTED;
}


int32_t hnsw_enqueue(const qmx_hnsw *g, qmx_query *q, uint32_t top, uint32_t ef, qmx_scored_point *d_out,
                            uint32_t *d_counts, uint32_t *d_scored, bool timed, bool acorn, const MultiWalk *mw,
                            const ExpandedOut *xo, const CustomWalk *cw, const PopTrace *pt) {
    const qmx_segment *s = q->seg;
    QMX_REQUIRE(!tq_l1(s) || !mw, QMX_ERR_NOT_SUPPORTED, "multi-vector walks through a TurboQuant storage over Manhattan are not built");
    ScanArgs a;
    fill_args(q, 0, q->nq, a);
    if (tq_l1(s)) {   // the walk scores against the query as given (tq_l1_policy.hpp): f32 entries of dim floats, 16-byte padded
        const uint32_t qs = tq_l1_query_bytes(s->dim);
        QMX_TRY(q->tq_rot.reserve((size_t)q->nq * qs));
        QMX_HIP(hipMemsetAsync(q->tq_rot.p, 0, (size_t)q->nq * qs, q->stream));
        QMX_HIP(hipMemcpy2DAsync(q->tq_rot.p, qs, q->enc.p, (size_t)s->dim * 4, (size_t)s->dim * 4, q->nq, hipMemcpyDeviceToDevice, q->stream));
        a.queries = q->tq_rot.p;
        a.q_stride = qs;
    }
    // PQ: the LUT-free walk (pq.hip HopPQDirect) - the entry of a search is its preprocessed vector (q->enc keeps them), staged in LDS
    const bool pq_direct = s->dtype == QMX_DTYPE_PQ && !cw && !mw && option(OPT_HNSW_PQ_DIRECT_WALK) > 0 &&
                           s->d_centroids &&
                           pq_direct_walk_ok(s->dim, s->pq_m, s->pq.chunk_size, s->pq.n_centroids);
    if (pq_direct) {
        a.queries = q->enc.p;
        a.q_stride = s->dim * 4;
    }
    const uint32_t n_searches = cw ? cw->n_queries : mw ? mw->n_queries : q->nq;
    if (cw) {
        a.cq_desc = cw->d_desc;
        a.cq_coefs = cw->d_coefs;
    }
    if (mw) {
        a.mv_offsets = mw->d_offsets;
        a.mv_qfirst = mw->d_qfirst;
        a.del = mw->del;
    }
    HnswArgs h;
    memset(&h, 0, sizeof(h));
    h.reindex = g->d_reindex; h.level_offsets = g->d_level_offsets; h.offsets = g->d_offsets; h.neighbors = g->d_neighbors;
    h.l0 = g->d_l0; h.l0_stride = g->l0_stride;
    // SQ segments, plain walk over a packed level 0: the link rows bring the linked rows' vector_offset along (scan_quant.hip RowSQX): the table is made once
    // per (graph, segment) - 10 M points x 64 dwords = 2.56 GB, one gather pass - and costs the walk one more line per hop instead of a random request per row
    if (s->dtype == QMX_DTYPE_SQ_U8 && g->d_l0 && !acorn && !xo && !mw && !cw && !(option(OPT_HNSW_REFERENCE_HEAP_ORDER) > 0) && s->d_row_offsets && g->n_points <= s->n) {
        std::lock_guard<std::mutex> lock(g->l0x_mu);
        // (the table belongs to the first SQ segment walked over this graph: another segment - another quantization of the same points - keeps the column;
        // nothing is ever freed under a running kernel)
        if (!g->d_l0x && !g->l0x_failed) {
            const size_t bytes = (size_t)g->n_points * (2 * (g->l0_stride - 1)) * 4;
            if (hipMalloc((void **)&g->d_l0x, bytes) != hipSuccess) {
                (void)hipGetLastError();
                g->d_l0x = nullptr;
                g->l0x_failed = true;
            } else {
                QMX_TRY(launch_hnsw_pack_level0_aux(q->stream, g->d_l0, g->n_points, g->l0_stride, s->d_row_offsets, s->n, g->d_l0x));
                QMX_HIP(hipStreamSynchronize(q->stream));      // (other threads' streams read it once the lock is gone)
                g->l0x_segment_uid = s->uid;
            }
        }
        if (g->d_l0x && g->l0x_segment_uid == s->uid) {
            h.l0 = g->d_l0x;
            h.l0_aux_off = g->l0_stride - 1;               // = m0: the row is [m0 link slots][m0 offsets]
            h.l0_stride = 2 * (g->l0_stride - 1);
        }
    }
    h.n_offsets = g->n_offsets; h.n_neighbors = g->n_neighbors;
    h.n_points = g->n_points; h.n_levels = g->n_levels; h.m = g->m; h.m0 = g->m0;
    h.ep_ids = g->d_ep_ids; h.ep_levels = g->d_ep_levels; h.n_ep = g->n_ep;
    h.xp_ids = g->d_xp_ids; h.xp_levels = g->d_xp_levels; h.n_xp = g->n_xp;
    h.ef = ef; h.top = top; h.nq = n_searches;
    h.out = d_out; h.out_counts = d_counts; h.out_scored = d_scored;
    if (xo) { h.expanded = xo->d_ids; h.expanded_cnt = xo->d_cnt; h.xcap = xo->xcap; }
    if (pt) {
        QMX_REQUIRE(!xo && !acorn && !mw && !cw, QMX_ERR_NOT_SUPPORTED, "the pop trace is the plain walk's");
        h.pops = pt->d_pops; h.pop_cnt = pt->d_cnt; h.pop_cap = pt->cap;
    }
    // option hnsw_reference_heap_order: the plain walk keeps the reference's two binary heaps (hnsw.hpp RefHeaps); other walks are unaffected
    h.ref_heaps = (option(OPT_HNSW_REFERENCE_HEAP_ORDER) > 0 && !acorn && !xo && !mw && !cw && !tq_l1(s)) ? 1 : 0;
    h.ref_cap = HNSW_REF_CAND_CAP;
    h.lds_query_bytes = q->q_stride <= HNSW_LDS_QUERY_MAX ? q->q_stride : 0;
    if (tq_l1(s)) {
        h.lds_query_bytes = tq_l1_lds_bytes(s->dim, s->tq_rot_dim);
        QMX_REQUIRE(h.lds_query_bytes <= HNSW_LDS_QUERY_MAX, QMX_ERR_NOT_SUPPORTED, "TurboQuant over Manhattan, the walk: %u bytes of LDS per search", h.lds_query_bytes);
    }
    if (mw) {   // [16-byte header][the multi-query's inner vectors, when the longest fits a modest share of the LDS; else every search reads its own through L2]
        const uint64_t need = 16 + (uint64_t)std::max<uint32_t>(mw->max_tokens, 1) * q->q_stride;
        h.lds_query_bytes = need <= 64 * 1024 ? (uint32_t)need : 16;
    }
    // A PQ LUT of more than half the LDS leaves one search per CU; the walk is a chain of dependent memory round trips,
    // so many searches per CU with the LUT read through L2 win (measured: tools/bench_hnsw.py)
    if (s->dtype == QMX_DTYPE_PQ && q->q_stride > 16 * 1024 && !mw) h.lds_query_bytes = 0;      // (a multi-query's LUTs are always staged)
    if (pq_direct) h.lds_query_bytes = a.q_stride;
    if (cw) {   // [32-byte header][the examples' entries]: staged when they fit a modest share of the LDS, read through L2 otherwise (PQ LUTs always)
        const uint64_t need = 32 + (uint64_t)std::max<uint32_t>(cw->max_examples, 1) * q->q_stride;
        h.lds_query_bytes = (need <= 48 * 1024 && h.lds_query_bytes != 0) ? (uint32_t)need : 32;
        if (tq_l1(s)) {   // TurboQuant over Manhattan: [header][the examples, always staged][the hop scratch][64 scores per example] (hnsw.hpp HopCustom::hop)
            const uint64_t ne = std::max<uint32_t>(cw->max_examples, 1);
            const uint64_t need_l1 = 32 + ne * a.q_stride + (tq_l1_lds_bytes(s->dim, s->tq_rot_dim) - tq_l1_query_bytes(s->dim)) + ne * 256;
            QMX_REQUIRE(need_l1 <= HNSW_LDS_QUERY_MAX, QMX_ERR_NOT_SUPPORTED, "a custom query of %u examples over TurboQuant / Manhattan needs %llu bytes of LDS",
                        cw->max_examples, (unsigned long long)need_l1);
            h.lds_query_bytes = (uint32_t)need_l1;
        }
        if (cw->lds_bytes) {      // multi-vector examples: always staged (the MaxSim policy reads its tokens from LDS)
            QMX_REQUIRE(cw->lds_bytes <= HNSW_LDS_QUERY_MAX, QMX_ERR_NOT_SUPPORTED, "a custom query of %u bytes of example tokens does not fit the LDS", cw->lds_bytes);
            h.lds_query_bytes = cw->lds_bytes;
        }
    }
    if (std::max(top, ef) > HNSW_MAX_EF_REG && !mw && !cw && !tq_l1(s)) {   // a list this long lives in LDS behind the query entry, which is then always staged (a PQ LUT too)
        const size_t beam = ((size_t)std::max(top, ef) * 9 + 15) / 16 * 16;
        QMX_REQUIRE((size_t)a.q_stride + beam + 2048 <= 160 * 1024, QMX_ERR_NOT_SUPPORTED,
                    "hnsw max(top, ef) = %u: the query entry (%u bytes) and the list do not fit the LDS together", std::max(top, ef), a.q_stride);
        h.lds_query_bytes = a.q_stride;
    }
    // the visited set in LDS (hnsw.hpp LdsVisited) where the search has room for it beside its query entry; the bitmap below stays allocated for what the
    // table's buckets cannot hold
    h.vis_lds = (!acorn && !h.ref_heaps && !option(OPT_HNSW_NO_LDS_VISITED) && g->n_points <= HNSW_VIS_LDS_MAX_POINTS && h.lds_query_bytes <= 32 * 1024) ? HNSW_VIS_LDS_BYTES : 0;
    // the PQ walk through per-search LUTs (HopPQ, not the LUT-free walk): the LUTs' 8-bit images for the hop prefilter (pq.hip HopPQ::prefilter), built
    // here from the batch's f32 LUTs.  Its 24 KiB per search take the LDS the visited table would: that walk keeps the bitmap
    if (s->dtype == QMX_DTYPE_PQ && !pq_direct && !acorn && !mw && !cw && !h.ref_heaps && !option(OPT_HNSW_NO_PQ_PREFILTER) && a.queries == q->d_queries &&
        s->pq_m <= 128 && s->pq.n_centroids <= 256 && q->q_stride > 16 * 1024 && h.lds_query_bytes == 0) {
        const uint32_t st8 = pq_walk_lut8_stride(s->pq_m);
        QMX_TRY(q->hnsw_pq8.reserve((size_t)n_searches * st8));
        QMX_TRY(launch_pq_walk_lut8(q->stream, q->d_queries, q->q_stride, n_searches, s->pq_m, s->pq.n_centroids, q->hnsw_pq8.p));
        h.pq8 = (const unsigned char *)q->hnsw_pq8.p;
        h.pq8_stride = st8;
        h.vis_lds = 0;
    }
    // ... and the LUT-free walk (HopPQDirect) prefilters its hops on the image of the EXACT-ORDER LUT - the entries it recomputes; a batch whose LUTs came
    // from the matrix cores (<= 1e-5 off that order) gets exact-order LUTs made for the image alone
    if (pq_direct && !acorn && !xo && !h.ref_heaps && !option(OPT_HNSW_NO_PQ_PREFILTER) && s->pq_m <= 128 && s->pq.n_centroids <= 256) {
        const uint32_t st8 = pq_walk_lut8_stride(s->pq_m);
        const size_t lut_bytes = (size_t)s->pq_m * s->pq.n_centroids * sizeof(float);
        const void *luts = q->d_queries;
        uint32_t lstride = q->q_stride;
        const bool dotlike = s->distance == QMX_DISTANCE_DOT || s->distance == QMX_DISTANCE_COSINE;
        if (s->pq.lut_mfma && dotlike) {
            qmx_pq_params exact = s->pq;
            exact.lut_mfma = 0;
            QMX_TRY(q->hnsw_lutx.reserve((size_t)n_searches * lut_bytes));
            QMX_TRY(launch_pq_lut(q->stream, s->distance, s->dim, exact, s->d_centroids, (const float *)q->enc.p, n_searches, (float *)q->hnsw_lutx.p));
            luts = q->hnsw_lutx.p;
            lstride = (uint32_t)lut_bytes;
        }
        QMX_TRY(q->hnsw_pq8.reserve((size_t)n_searches * st8));
        QMX_TRY(launch_pq_walk_lut8(q->stream, luts, lstride, n_searches, s->pq_m, s->pq.n_centroids, q->hnsw_pq8.p));
        h.pq8 = (const unsigned char *)q->hnsw_pq8.p;
        h.pq8_stride = st8;
        h.vis_lds = 0;
    }
    h.log_cap = HNSW_LOG_CAP;
    {   // tests: force the whole-bitmap clear path
        const int64_t v = option(OPT_HNSW_LOG_CAP);
        if (v >= 1 && v <= (int64_t)HNSW_LOG_CAP) h.log_cap = (uint32_t)v;
    }
    h.vis_words = ((uint64_t)g->n_points + 31) / 32;
    if (h.vis_words == 0) h.vis_words = 1;
    h.acorn = acorn ? 1 : 0;
    h.hop_cap = 64;
    if (acorn) {   // two visited lists; every explored node may add m0 points to one scoring batch
        QMX_REQUIRE(g->m0 >= 1 && g->m0 <= 128, QMX_ERR_NOT_SUPPORTED, "ACORN walk: m0 = %u not in 1..128", g->m0);
        h.vis_words *= 2;
        h.hop_cap = std::min<uint32_t>((g->m0 * (g->m0 + 1) + 63) / 64 * 64, 4160);    // (m0 > 64: the kernel scores what it holds before the buffer could overflow)
    }
    int per_cu = 1;
    QMX_TRY(launch_hnsw(q, a, h, 0, &per_cu));
    if (s->dtype == QMX_DTYPE_PQ && h.lds_query_bytes == 0 && option(OPT_HNSW_PQ_PER_CU) > 0) per_cu = (int)std::min<int64_t>(per_cu, option(OPT_HNSW_PQ_PER_CU));
    // whole waves per SIMD: with 9 searches per CU one SIMD carries three waves and the others two, and the slowest SIMD sets the pace
    // (measured on the SQ walk at 10 M points: 5.76 ms with 8 per CU, 6.09 with 9, 6.23 with 7: profiles/r5_sq_walk_visited.md)
    if (per_cu >= 8) per_cu -= per_cu % 4;
    if (option(OPT_HNSW_PER_CU) > 0) per_cu = (int)std::min<int64_t>(per_cu, option(OPT_HNSW_PER_CU));
    uint64_t slots = std::min<uint64_t>({(uint64_t)n_searches, (uint64_t)s->num_cus * per_cu, (uint64_t)HNSW_SLOT_CAP});
    const uint64_t by_budget = std::max<uint64_t>(1, HNSW_VIS_BUDGET / (h.vis_words * 4));
    slots = std::max<uint64_t>(1, std::min(slots, by_budget));
    if (xo) {      // search_with_vectors: a slot's stack of evicted candidates that tie with the bound (hnsw.hpp; the reference keeps every one of them in `candidates`)
        h.ev_cap = HNSW_EV_SPILL_CAP;
        QMX_TRY(q->hnsw_refc.reserve((size_t)slots * h.ev_cap * sizeof(uint64_t)));
        h.ev_spill = (uint64_t *)q->hnsw_refc.p;
    }
    if (h.ref_heaps) {
        slots = std::min<uint64_t>(slots, HNSW_REF_SLOT_CAP);
        QMX_TRY(q->hnsw_refc.reserve((size_t)slots * h.ref_cap * sizeof(uint2)));
        h.ref_cands = (uint2 *)q->hnsw_refc.p;
    }
    if (q->hnsw_slots < slots || q->hnsw_vis_words != h.vis_words) {
        // (re)allocate for the largest slot count this handle can use, zero once: the kernel returns the bitmaps all-zero
        const uint64_t want = std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)std::max<uint32_t>(std::max(q->nq, n_searches), 1), (uint64_t)s->num_cus * per_cu,
                                                                       (uint64_t)HNSW_SLOT_CAP, by_budget}));
        QMX_HIP(hipStreamSynchronize(q->stream));
        q->hnsw_slots = 0;
        QMX_TRY(q->hnsw_vis.reserve((size_t)want * h.vis_words * 4));
        QMX_TRY(q->hnsw_log.reserve((size_t)want * HNSW_LOG_CAP * 4));
        QMX_HIP(hipMemsetAsync(q->hnsw_vis.p, 0, (size_t)want * h.vis_words * 4, q->stream));
        q->hnsw_slots = (uint32_t)want;
        q->hnsw_vis_words = h.vis_words;
    }
    h.visited = (uint32_t *)q->hnsw_vis.p;
    h.vis_log = (uint32_t *)q->hnsw_log.p;
    // the slots draw their searches from a counter that starts behind the first `slots` (hnsw.hpp)
    QMX_TRY(q->hnsw_next.reserve(32));         // [u32 next search][pad][u64 hop candidates offered to the PQ prefilter][u64 scored exactly]
    QMX_HIP(hipMemsetD32Async((hipDeviceptr_t)q->hnsw_next.p, (int)slots, 1, q->stream));
    QMX_HIP(hipMemsetAsync((char *)q->hnsw_next.p + 8, 0, 16, q->stream));
    h.next_query = (uint32_t *)q->hnsw_next.p;
    h.pq_stats = h.pq8 ? (unsigned long long *)((char *)q->hnsw_next.p + 8) : nullptr;
    size_t slot = 0;
    if (timed) QMX_TRY(timing_begin(q, &slot));
    QMX_TRY(launch_hnsw(q, a, h, (uint32_t)slots, &per_cu));
    q->last_kernel = last_noted_kernel();
    if (timed) QMX_TRY(timing_end(q, slot));
    return QMX_OK;
}

int32_t hnsw_check(const qmx_hnsw *g, const qmx_query *q, uint32_t top, uint32_t ef) {
    QMX_REQUIRE(g->device == q->seg->device, QMX_ERR_BAD_ARG, "graph lives on device %d, the segment on %d", g->device, q->seg->device);
    QMX_REQUIRE((uint64_t)g->n_points <= q->seg->n, QMX_ERR_OUT_OF_BOUNDS, "graph has %u points, the segment %llu rows", g->n_points,
                (unsigned long long)q->seg->n);
    QMX_REQUIRE(top >= 1, QMX_ERR_BAD_ARG, "top must be > 0");
    QMX_REQUIRE(std::max(top, ef) <= HNSW_MAX_EF, QMX_ERR_NOT_SUPPORTED, "max(top, ef) = %u > %u not supported yet", std::max(top, ef),
                HNSW_MAX_EF);
    return QMX_OK;
}

int32_t hnsw_search_sync(const qmx_hnsw *g, qmx_query *q, uint32_t top, uint32_t ef, qmx_scored_point *out, uint32_t *out_counts,
                                const volatile uint8_t *is_stopped, qmx_counters *counters, bool acorn) {
    QMX_REQUIRE(g && q && out && out_counts, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_TRY(hnsw_check(g, q, top, ef));
    QMX_HIP(hipSetDevice(q->device));
    if (counters) memset(counters, 0, sizeof(*counters));
    if (q->nq == 0) return QMX_OK;
    if (is_stopped && *is_stopped) {
        set_error("search cancelled");
        return QMX_ERR_CANCELLED;
    }
    if (g->n_points == 0) {   // get_entry_point() -> None -> empty result (graph_layers.rs:539-542)
        if (is_device_ptr(out_counts)) QMX_HIP(hipMemset(out_counts, 0, (size_t)q->nq * 4));
        else memset(out_counts, 0, (size_t)q->nq * 4);
        return QMX_OK;
    }
    const bool out_dev = is_device_ptr(out), cnt_dev = is_device_ptr(out_counts);
    qmx_scored_point *d_out = out;
    uint32_t *d_counts = out_counts;
    if (!out_dev) { QMX_TRY(q->out.reserve((size_t)q->nq * top * sizeof(qmx_scored_point))); d_out = (qmx_scored_point *)q->out.p; }
    if (!cnt_dev) { QMX_TRY(q->counts.reserve((size_t)q->nq * 4)); d_counts = (uint32_t *)q->counts.p; }
    QMX_TRY(q->hnsw_scored.reserve((size_t)q->nq * 4));
    const bool timed = q->timing || (q->seg->flags & QMX_SEG_TIME_KERNELS) != 0;
    QMX_TRY(hnsw_enqueue(g, q, top, ef, d_out, d_counts, (uint32_t *)q->hnsw_scored.p, timed, acorn));
    if (!out_dev) QMX_TRY(copy_out(q->stream, out, d_out, (size_t)q->nq * top * sizeof(qmx_scored_point)));
    if (!cnt_dev) QMX_TRY(copy_out(q->stream, out_counts, d_counts, (size_t)q->nq * 4));
    std::vector<uint32_t> scored(counters ? q->nq : 0);
    unsigned long long pq_stats[2] = {0, 0};
    if (counters) QMX_HIP(hipMemcpyAsync(scored.data(), q->hnsw_scored.p, (size_t)q->nq * 4, hipMemcpyDeviceToHost, q->stream));
    if (counters && q->hnsw_next.p) QMX_HIP(hipMemcpyAsync(pq_stats, (char *)q->hnsw_next.p + 8, 16, hipMemcpyDeviceToHost, q->stream));
    QMX_TRY(check_err_flag(q));   // synchronises the stream
    if (counters) {
        uint64_t total = 0;
        for (uint32_t v : scored) total += v;
        counters->vectors_scored = total;
        // the PQ walk's hop prefilter (hnsw.hpp HopPQ::prefilter): level-0 hop candidates that met the 8-bit bound / those that survived it and were scored exactly
        counters->prefilter_candidates = pq_stats[0];
        counters->verified_rows = pq_stats[1];
        counters->bytes_read = total * q->seg->row_bytes;
        counters->kernel_launches = 1;
    }
    if (timed) {
        const float before = q->timing_ms;
        QMX_TRY(timing_fold(q));
        if (counters) counters->kernel_ms = q->timing_ms - before;
    }
    return QMX_OK;
}

int32_t qmx_hnsw_search(const qmx_hnsw *g, qmx_query *q, uint32_t top, uint32_t ef, qmx_scored_point *out, uint32_t *out_counts,
                        const volatile uint8_t *is_stopped, qmx_counters *counters) {
    return hnsw_search_sync(g, q, top, ef, out, out_counts, is_stopped, counters, false);
}
int32_t qmx_hnsw_search_acorn(const qmx_hnsw *g, qmx_query *q, uint32_t top, uint32_t ef, qmx_scored_point *out, uint32_t *out_counts,
                              const volatile uint8_t *is_stopped, qmx_counters *counters) {
    return hnsw_search_sync(g, q, top, ef, out, out_counts, is_stopped, counters, true);
}

int32_t qmx_hnsw_search_traced(const qmx_hnsw *g, qmx_query *q, uint32_t top, uint32_t ef, qmx_scored_point *out, uint32_t *out_counts,
                               qmx_scored_point *pops, uint32_t pop_cap, uint32_t *pop_counts) {
    QMX_REQUIRE(g && q && out && out_counts && pops && pop_counts && pop_cap >= 1, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(!is_device_ptr(out) && !is_device_ptr(out_counts) && !is_device_ptr(pops) && !is_device_ptr(pop_counts), QMX_ERR_BAD_ARG,
                "qmx_hnsw_search_traced writes host arrays");
    QMX_TRY(hnsw_check(g, q, top, ef));
    QMX_HIP(hipSetDevice(q->device));
    if (q->nq == 0) return QMX_OK;
    memset(out_counts, 0, (size_t)q->nq * 4);
    memset(pop_counts, 0, (size_t)q->nq * 4);
    if (g->n_points == 0) return QMX_OK;
    QMX_TRY(q->out.reserve((size_t)q->nq * top * sizeof(qmx_scored_point)));
    QMX_TRY(q->counts.reserve((size_t)q->nq * 4));
    QMX_TRY(q->cand.reserve((size_t)q->nq * pop_cap * sizeof(qmx_scored_point)));
    QMX_TRY(q->xcnt.reserve((size_t)q->nq * 4));
    PopTrace pt{(qmx_scored_point *)q->cand.p, (uint32_t *)q->xcnt.p, pop_cap};
    QMX_TRY(hnsw_enqueue(g, q, top, ef, (qmx_scored_point *)q->out.p, (uint32_t *)q->counts.p, nullptr, false, false, nullptr, nullptr, nullptr, &pt));
    QMX_TRY(copy_out(q->stream, out, q->out.p, (size_t)q->nq * top * sizeof(qmx_scored_point)));
    QMX_TRY(copy_out(q->stream, out_counts, q->counts.p, (size_t)q->nq * 4));
    QMX_TRY(copy_out(q->stream, pops, q->cand.p, (size_t)q->nq * pop_cap * sizeof(qmx_scored_point)));
    QMX_TRY(copy_out(q->stream, pop_counts, q->xcnt.p, (size_t)q->nq * 4));
    return check_err_flag(q);   // synchronises the stream
}

int32_t qmx_hnsw_search_async(const qmx_hnsw *g, qmx_query *q, uint32_t top, uint32_t ef, qmx_scored_point *out_dev,
                              uint32_t *out_counts_dev, uint32_t *out_scored_dev) {
    QMX_REQUIRE(g && q && out_dev && out_counts_dev, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_TRY(hnsw_check(g, q, top, ef));
    QMX_HIP(hipSetDevice(q->device));
    if (q->nq == 0) return QMX_OK;
    if (g->n_points == 0) {
        QMX_HIP(hipMemsetAsync(out_counts_dev, 0, (size_t)q->nq * 4, q->stream));
        return QMX_OK;
    }
    const bool timed = q->timing || (q->seg->flags & QMX_SEG_TIME_KERNELS) != 0;
    return hnsw_enqueue(g, q, top, ef, out_dev, out_counts_dev, out_scored_dev, timed);
}

int32_t qmx_hnsw_search_with_vectors(const qmx_hnsw *g, qmx_query *links, qmx_query *base, uint32_t top, uint32_t ef, qmx_scored_point *out,
                                     uint32_t *out_counts, const volatile uint8_t *is_stopped, qmx_counters *counters) {
    QMX_REQUIRE(g && links && base && out && out_counts, QMX_ERR_BAD_ARG, "NULL argument");
    QMX_REQUIRE(base->nq == links->nq && base->device == links->device, QMX_ERR_BAD_ARG, "the two query batches must match");
    QMX_REQUIRE(base->seg->n >= g->n_points, QMX_ERR_OUT_OF_BOUNDS, "graph has %u points, the base-vector segment %llu rows", g->n_points,
                (unsigned long long)base->seg->n);
    QMX_TRY(hnsw_check(g, links, top, ef));
    QMX_REQUIRE(top <= MAX_TOP, QMX_ERR_NOT_SUPPORTED, "top %u > %u", top, MAX_TOP);
    QMX_HIP(hipSetDevice(links->device));
    if (counters) memset(counters, 0, sizeof(*counters));
    const uint32_t nq = links->nq;
    if (nq == 0) return QMX_OK;
    if (is_stopped && *is_stopped) {
        set_error("search cancelled");
        return QMX_ERR_CANCELLED;
    }
    if (g->n_points == 0) {
        if (is_device_ptr(out_counts)) QMX_HIP(hipMemset(out_counts, 0, (size_t)nq * 4));
        else memset(out_counts, 0, (size_t)nq * 4);
        return QMX_OK;
    }
    const uint32_t beam_ef = std::max(top, ef);
    // a search pops about ef..2 ef candidates: the list is sized for 32 ef (or every point); a search that pops more (a graph the walk wanders through:
    // a filter that leaves few points, a bad entry) makes the batch run again with a list as long as the largest count it reported - every point at most
    uint32_t xcap = (uint32_t)std::min<uint64_t>(g->n_points, (uint64_t)32 * beam_ef + 256);
    QMX_TRY(links->cand.reserve((size_t)nq * beam_ef * sizeof(qmx_scored_point)));
    QMX_TRY(links->cand_cnt.reserve((size_t)nq * 4));
    QMX_TRY(links->hnsw_scored.reserve((size_t)nq * 4));
    QMX_TRY(links->xcnt.reserve((size_t)nq * 4));
    const bool timed = links->timing || (links->seg->flags & QMX_SEG_TIME_KERNELS) != 0;
    std::vector<uint32_t> cnt(nq), sc(nq);
    uint64_t popped = 0, scored = 0;
    uint32_t walks = 0;
    while (true) {
        QMX_TRY(links->cand_ids.reserve((size_t)nq * xcap * 4));
        ExpandedOut xo{(uint32_t *)links->cand_ids.p, (uint32_t *)links->xcnt.p, xcap};
        QMX_TRY(hnsw_enqueue(g, links, std::min(top, beam_ef), ef, (qmx_scored_point *)links->cand.p, (uint32_t *)links->cand_cnt.p,
                             (uint32_t *)links->hnsw_scored.p, timed, false, nullptr, &xo));
        ++walks;
        QMX_HIP(hipMemcpyAsync(cnt.data(), links->xcnt.p, (size_t)nq * 4, hipMemcpyDeviceToHost, links->stream));
        QMX_HIP(hipMemcpyAsync(sc.data(), links->hnsw_scored.p, (size_t)nq * 4, hipMemcpyDeviceToHost, links->stream));
        QMX_TRY(check_err_flag(links));     // synchronises
        uint32_t worst = 0;
        popped = scored = 0;
        for (uint32_t i = 0; i < nq; ++i) {
            worst = std::max(worst, cnt[i]);
            popped += cnt[i];
            scored += sc[i];
        }
        if (worst <= xcap) break;
        // (the walk is deterministic: the second run pops the same candidates, now all listed; a count above n_points + 1 cannot be)
        QMX_REQUIRE(xcap < g->n_points + 1u, QMX_ERR_OTHER, "a search popped %u candidates of a graph of %u points", worst, g->n_points);
        xcap = (uint32_t)std::min<uint64_t>((uint64_t)g->n_points + 1, worst);
        if (is_stopped && *is_stopped) {
            set_error("search cancelled");
            return QMX_ERR_CANCELLED;
        }
    }
    // base_search_context: FixedLengthPriorityQueue(ef) over the base scores of the popped candidates, into_iter_sorted().take(top)
    QMX_TRY(qmx_rescore(base, (const uint32_t *)links->cand_ids.p, (const uint32_t *)links->xcnt.p, xcap, top, out, out_counts));
    if (counters) {
        counters->vectors_scored = scored + popped;
        counters->bytes_read = scored * links->seg->row_bytes + popped * base->seg->row_bytes;
        counters->kernel_launches = 2 + walks;
        if (timed) { const float before = links->timing_ms; QMX_TRY(timing_fold(links)); counters->kernel_ms = links->timing_ms - before; }
    }
    return QMX_OK;
}

int32_t qmx_multi_hnsw_search(const qmx_hnsw *g, qmx_query *inner, const uint32_t *query_first, uint32_t n_queries, const uint64_t *point_offsets,
                              uint32_t n_points, const uint64_t *point_deleted, uint64_t n_deleted_bits, uint32_t top, uint32_t ef,
                              qmx_scored_point *out, uint32_t *out_counts, qmx_counters *counters) {
    QMX_REQUIRE(g && inner && query_first && point_offsets && out && out_counts, QMX_ERR_BAD_ARG, "NULL argument");
    const qmx_segment *s = inner->seg;
    QMX_REQUIRE(g->device == s->device, QMX_ERR_BAD_ARG, "graph lives on device %d, the segment on %d", g->device, s->device);
    QMX_REQUIRE(g->n_points <= n_points, QMX_ERR_OUT_OF_BOUNDS, "graph has %u points, the multi-vector storage %u", g->n_points, n_points);
    QMX_REQUIRE(top >= 1, QMX_ERR_BAD_ARG, "top must be > 0");
    QMX_REQUIRE(std::max(top, ef) <= HNSW_MAX_EF, QMX_ERR_NOT_SUPPORTED, "max(top, ef) = %u > %u not supported yet", std::max(top, ef), HNSW_MAX_EF);
    QMX_REQUIRE(!is_device_ptr(query_first) && !is_device_ptr(point_offsets), QMX_ERR_BAD_ARG, "query_first and point_offsets are host arrays");
    QMX_HIP(hipSetDevice(inner->device));
    if (counters) memset(counters, 0, sizeof(*counters));
    if (n_queries == 0) return QMX_OK;
    QMX_REQUIRE(query_first[n_queries] <= inner->nq, QMX_ERR_OUT_OF_BOUNDS, "multi-queries reach past the %u inner query vectors of the batch", inner->nq);
    uint32_t max_tokens = 0;
    for (uint32_t j = 0; j < n_queries; ++j) {
        QMX_REQUIRE(query_first[j] <= query_first[j + 1], QMX_ERR_BAD_ARG, "query_first is not ascending at %u", j);
        max_tokens = std::max(max_tokens, query_first[j + 1] - query_first[j]);
    }
    for (uint32_t p = 0; p < n_points; ++p)
        QMX_REQUIRE(point_offsets[p] <= point_offsets[p + 1], QMX_ERR_BAD_ARG, "point_offsets is not ascending at %u", p);
    QMX_REQUIRE(point_offsets[n_points] <= s->n, QMX_ERR_OUT_OF_BOUNDS, "point_offsets reach past the %llu inner rows of the segment",
                (unsigned long long)s->n);
    const bool out_dev = is_device_ptr(out), cnt_dev = is_device_ptr(out_counts);
    if (g->n_points == 0) {   // get_entry_point() -> None
        if (cnt_dev) QMX_HIP(hipMemset(out_counts, 0, (size_t)n_queries * 4));
        else memset(out_counts, 0, (size_t)n_queries * 4);
        return QMX_OK;
    }
    QMX_TRY(inner->mv_qfirst.reserve((size_t)(n_queries + 1) * 4));
    QMX_TRY(inner->mv_offsets.reserve((size_t)(n_points + 1) * 8));
    QMX_HIP(hipMemcpyAsync(inner->mv_qfirst.p, query_first, (size_t)(n_queries + 1) * 4, hipMemcpyHostToDevice, inner->stream));
    QMX_HIP(hipMemcpyAsync(inner->mv_offsets.p, point_offsets, (size_t)(n_points + 1) * 8, hipMemcpyHostToDevice, inner->stream));
    MultiWalk mw;
    memset(&mw, 0, sizeof(mw));
    mw.d_qfirst = (const uint32_t *)inner->mv_qfirst.p;
    mw.d_offsets = (const uint64_t *)inner->mv_offsets.p;
    mw.n_queries = n_queries;
    mw.max_tokens = max_tokens;
    mw.del.n_rows = n_points;      // deletion is per POINT (the id tracker's bitslice over multi-vector points), not per inner row
    if (point_deleted && n_deleted_bits) {
        const void *d_bits = nullptr;
        QMX_TRY(stage_in(inner, inner->mv_deleted, point_deleted, (size_t)((n_deleted_bits + 63) / 64) * 8, &d_bits));
        mw.del.point_deleted = (const uint64_t *)d_bits;
        mw.del.n_point_bits = n_deleted_bits;
    }
    if (inner->has_filter) { mw.del.allowed = (const uint64_t *)inner->filter.p; mw.del.n_allowed_bits = inner->n_filter_bits; }
    qmx_scored_point *d_out = out;
    uint32_t *d_counts = out_counts;
    if (!out_dev) { QMX_TRY(inner->out.reserve((size_t)n_queries * top * sizeof(qmx_scored_point))); d_out = (qmx_scored_point *)inner->out.p; }
    if (!cnt_dev) { QMX_TRY(inner->counts.reserve((size_t)n_queries * 4)); d_counts = (uint32_t *)inner->counts.p; }
    QMX_TRY(inner->hnsw_scored.reserve((size_t)n_queries * 4));
    const bool timed = inner->timing || (s->flags & QMX_SEG_TIME_KERNELS) != 0;
    QMX_TRY(hnsw_enqueue(g, inner, top, ef, d_out, d_counts, (uint32_t *)inner->hnsw_scored.p, timed, false, &mw));
    if (!out_dev) QMX_TRY(copy_out(inner->stream, out, d_out, (size_t)n_queries * top * sizeof(qmx_scored_point)));
    if (!cnt_dev) QMX_TRY(copy_out(inner->stream, out_counts, d_counts, (size_t)n_queries * 4));
    QMX_TRY(check_err_flag(inner));    // synchronises (the staged partitions may go away)
    if (counters) {
        std::vector<uint32_t> sc(n_queries);
        QMX_HIP(hipMemcpy(sc.data(), inner->hnsw_scored.p, (size_t)n_queries * 4, hipMemcpyDeviceToHost));
        uint64_t total = 0;
        for (uint32_t v : sc) total += v;
        counters->vectors_scored = total;            // POINTS scored (each costs |query| x |point| inner scores)
        counters->kernel_launches = 1;
        if (timed) { const float before = inner->timing_ms; QMX_TRY(timing_fold(inner)); counters->kernel_ms = inner->timing_ms - before; }
    }
    return QMX_OK;
}

}  // extern "C"

// api_internal.hpp — what the translation units of the C-ABI (api_*.hip) share: the handle structs behind the opaque pointers of include/qdrant_amd.h,
// their small helpers, the constants of the brute-force paths, and the declarations of the internal functions one family calls in another.
// Host-side logic only: no CPU scoring path exists in these files - every score is produced by a gfx950 kernel or the call fails.
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <ctype.h>
#include <cxxabi.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <new>
#include <vector>

#include "kernels.hpp"
#include "tq_rotate.hpp"

namespace qmx {
// api_core.hip
const std::string &last_error_text();                       // the calling thread's last error text (qmx_last_error)
const void *last_noted_kernel();                             // the kernel the calling thread launched last (QMX_NOTE_KERNEL)
bool is_device_ptr(const void *p);
uint32_t elem_bytes(uint32_t dtype);
int32_t check_device(int32_t device_id, hipDeviceProp_t *prop_out);
// growable device scratch
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int32_t reserve(size_t bytes) {
        if (bytes <= cap) return QMX_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max<size_t>(bytes, 4096);
        QMX_HIP(hipMalloc(&p, want));
        cap = want;
        return QMX_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
}  // namespace qmx

using namespace qmx;

// ---------------------------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------------------------
inline uint64_t next_segment_uid() {
    static std::atomic<uint64_t> counter{0};
    return ++counter;
}
struct qmx_segment {
    const uint64_t uid = next_segment_uid();      // never reused: what a cache keyed by a segment compares (a freed segment's address may come back)
    int device = 0;
    int num_cus = 256;
    uint32_t dtype = 0, distance = 0, dim = 0, flags = 0;
    uint64_t n = 0;
    uint64_t row_bytes = 0;    // reference row layout
    uint64_t row_stride = 0;   // bytes between rows in d_rows
    uint32_t scan_dim = 0;     // elements the metric consumes per row
    void *d_rows = nullptr;
    bool owns_rows = false;
    uint64_t *d_point_deleted = nullptr;
    uint64_t n_point_bits = 0;
    uint64_t *d_vec_deleted = nullptr;
    uint64_t n_vec_bits = 0;
    qmx_sq_params sq{};
    qmx_pq_params pq{};
    uint32_t bq_encoding = 0;            // qmx_bq_encoding
    uint32_t bq_query_bits = 1;          // QueryEncoding: 1 = SameAsStorage, 4 / 8 = Scalar4bits / Scalar8bits
    float *d_bq_mean = nullptr, *d_bq_stddev = nullptr;   // VectorStats of the 2-bit / 1.5-bit encodings (device copies), or null
    float *d_centroids = nullptr;
    float *d_pq_pair = nullptr;       // [m][ncent][ncent] chunk distances between centroids (score_internal terms; built when <= 256 MB)
    uint32_t pq_m = 0;
    void *d_pq_rot = nullptr;         // PQ blocks of 2^18 rows and more, m <= 96: the rotated copy of the codes the 6-bit prefilter scans (pq_prefilter.hip)
    float *d_row_offsets = nullptr;   // SQ: vector_offset column (rows hold the 16-byte aligned code block)
    bool sq_wide = false;             // SQ block the 128-query pass serves (scan_sqw.hip): every vector_offset finite, the largest magnitude below
    float sq_off_absmax = 0.f;
    int32_t *d_sq_bi = nullptr;       // ... and its column ceil(vector_offset / multiplier) + 1
    // TurboQuant (scan_tq.hip): parameters, the extras columns and the rotation tables
    uint32_t tq_bits = 0, tq_value_bits = 0, tq_padded_dim = 0, tq_rot_dim = 0, tq_code_bytes = 0, tq_n_chunks = 0;
    bool tq_invert = false;
    float *d_tq_sf = nullptr, *d_tq_l2 = nullptr, *d_tq_xm = nullptr;   // extras columns (xm: TQ+ only)
    bool tq_wide = false;              // 4-bit block the 128-query pass serves (scan_tq4w.hip): its extras columns are positive finite numbers, ranges below
    float tq_sf_min = 0.f, tq_sf_max = 0.f, tq_l2_min = 0.f, tq_l2_max = 0.f;
    uint32_t tq_c1 = 0;                // ... and the largest sum of |codebook bytes| over a row's codes
    float *d_tq_shift = nullptr, *d_tq_scale = nullptr;                   // TQ+ ErrorCorrection (device copies), or null
    int16_t *d_tq_weights = nullptr;                                      // ... d_prime_sq_i16
    float tq_weight_scale = 1.0f, tq_mm_const = 0.0f;
    uint32_t *d_tq_tables = nullptr;   // [3][rot_dim] maps, then chunk offsets and sizes
    void *d_tq_l1 = nullptr;           // Manhattan: the TqL1Dev of the walk (tq_rotate.hpp)
    double *d_tq_norms = nullptr;      // [n_chunks]
    // f32 dot / cosine blocks large enough for the split prefilter (scan_split.hip): max |x| and max row norm, taken once at create
    bool split_stats = false;
    float row_maxabs = 0.f, row_norm_max = 0.f;
    void *d_rows_split = nullptr;     // QMX_SEG_SPLIT_COPY / QMX_SEG_HALF_COPY: the block as f16 pairs / f16 high parts in the matrix cores' LDS layout
    bool split_half = false;          // ... which of the two
    bool split_i8 = false;            // QMX_SEG_I8_COPY: d_rows_split holds int8 codes instead (scan_split.hip, "The INT8 copy")
    float *d_i8_scale = nullptr;      // ... the columns' scales [dim]
    uint32_t *d_i8_stats = nullptr;   // ... {C1, C2^2, -, -}: the worst row's sum |c| and sum c^2
    float i8_balance = 0.0f;          // ... the range ratio G the scales were balanced to (0 = every column at its floor max |x| / 127)
    uint64_t copy_bytes = 0;          // bytes of d_rows_split
    bool auto_choice = false;         // QMX_SEG_AUTO_COPY: the copy was chosen by the trial of segment_auto_copy; what it measured:
    float auto_i8_ms = 0.0f, auto_half_ms = 0.0f, auto_i8_verified = 0.0f;
    uint32_t auto_i8_fallback = 0;

    bool fast_layout() const {
        if (dtype == QMX_DTYPE_BQ || dtype == QMX_DTYPE_TQ) return row_stride % 16 == 0 && ((uintptr_t)d_rows % 16) == 0;
        if (dtype <= QMX_DTYPE_U8) {
            const uint64_t eb = dtype == QMX_DTYPE_F32 ? 4 : dtype == QMX_DTYPE_F16 ? 2 : 1;
            return dim < 32 ? (row_stride % eb == 0 && ((uintptr_t)d_rows % eb) == 0)
                            : (row_stride % 16 == 0 && ((uintptr_t)d_rows % 16) == 0);
        }
        return true;
    }
    DeletedView deleted_view() const {
        DeletedView v;
        v.point_deleted = d_point_deleted;
        v.n_point_bits = n_point_bits;
        v.vec_deleted = d_vec_deleted;
        v.n_vec_bits = n_vec_bits;
        v.n_rows = n;
        v.allowed = nullptr;
        v.n_allowed_bits = 0;
        return v;
    }
    // rows the brute-force stream visits: iter_zeros(point_deleted) ends at the bitslice length
    uint64_t scan_rows() const { return d_point_deleted ? std::min<uint64_t>(n, n_point_bits) : n; }
};

struct qmx_query {
    const qmx_segment *seg = nullptr;
    int device = 0;            // copy of seg->device: destroy must not touch a segment that died first
    uint32_t nq = 0;
    uint32_t nq_padded = 0;
    uint32_t q_stride = 0;     // bytes
    uint32_t bq_bits = 1;      // BQ: bit planes per query value (1 for SameAsStorage and for internal queries = stored rows)
    uint32_t tq_qbytes_off = 0; // TurboQuant 1-bit: where the i8 form of the query sits inside an entry
    uint32_t aux_off = 0;      // bytes
    void *d_queries = nullptr; // [nq_padded][q_stride]
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    // HIP-event pairs around the scoring kernels (timing mode): recorded without synchronising,
    // summed by timing_collect()
    struct EvPair { hipEvent_t a = nullptr, b = nullptr; };
    std::vector<EvPair> evs;
    size_t ev_used = 0;
    float timing_ms = 0.f;
    uint32_t timing_launches = 0;
    DevBuf partial, out, counts, ids, scores, misc, enc, bounds, gthr;
    DevBuf mv_qfirst, mv_offsets, mv_deleted;        // multi-vector MaxSim: query ranges, point offsets, point-level deleted bits
    DevBuf cq_multi;          // custom queries over multi-vector points: the combined scores (cq_scores holds the per-example MaxSim rows)
    DevBuf cq_sims, cq_scores, cq_desc, cq_coefs;   // custom queries: example similarities, combined scores, descriptors, feedback coefficients
    uint32_t n_cq_coefs = 0;
    DevBuf cand, cand_cnt, cand_ids;   // qmx_search_quantized: oversampled candidates of the quantized stage
    // split prefilter (scan_split.hip): split queries, per-query norms / thresholds / bands, scales, candidate and verification buffers, flag
    DevBuf sp_bq, sp_f32, sp_cand, sp_cnt, sp_ver, sp_vscores, sp_sample, sp_wl, xcnt, tq_rot;
    DevBuf sh_lists, sh_out;   // qmx_sharded_*: the segments' lists gathered on this (the first) batch's device, the merged lists of a host-output call
    std::vector<uint32_t> sh_bases_host;
    hipEvent_t sh_done = nullptr;   // "this segment's list arrived on the merging device"
    hipEvent_t sh_merged = nullptr; // (root batch) "the merge of the previous sharded call has read the shared lists": the segments' next copies into them wait for it
    DevBuf pq_table;           // PQ prefilter: the 6-bit tables of the tile's query groups, their integer thresholds behind them
    DevBuf sp_probe, sp_pscores;   // the int8 copy's passes: [nq][64] probe ids + [nq] counts, their exact scores
    DevBuf sp_plan, sp_fq;     // ... the per-query overflow flags + the plan of the conditional exact passes (SplitPlanLayout), the overflowed queries packed
    // counters of the last search enqueued on this batch: the host's share is known at enqueue, the prefilter's share sits in sp_plan until
    // the stream is synchronised (qmx_query_last_counters / the synchronous entry points fold it in)
    qmx_counters last_counters{};
    bool last_split = false, last_pq = false;
    uint32_t last_fqt = 64;          // queries per conditional exact pass of the last prefilter search (fold_split_counters)
    uint64_t last_row_bytes = 0, last_n_cand = 0;
    uint64_t sp_sample_n = 0, sp_sample_of = 0;
    DevBuf filter;             // payload-filter allow bitmap of this query batch (qmx_query_set_filter)
    uint64_t n_filter_bits = 0;
    bool has_filter = false;
    DevBuf hnsw_lutx;          // exact-order LUTs made for the hop prefilter's image of a LUT-free walk whose batch LUTs came from the matrix cores
    DevBuf hnsw_pq8;           // the 8-bit LUT images of the batch for the PQ walk's hop prefilter (HnswArgs::pq8)
    DevBuf hnsw_next;          // the walk's work counter (HnswArgs::next_query)
    DevBuf hnsw_refc;          // option hnsw_reference_heap_order: the per-slot `candidates` heaps
    DevBuf hnsw_vis, hnsw_log, hnsw_scored;   // HNSW scratch: per-slot visited bitmaps (kept all-zero between launches) + logs
    uint32_t hnsw_slots = 0;
    uint64_t hnsw_vis_words = 0;
    int *d_err = nullptr;
    uint32_t partial_grid_cap = 0;
    bool timing = false;
    const void *last_kernel = nullptr;   // host handle of the last top-k scan / graph walk kernel launched for this batch
};

// f32 dot / cosine rows of >= 32 elements scan 8..32 queries per pass on the f32 matrix cores (scan_mfma.hip)
static bool mfma_scan_ok(const qmx_segment *s) {
    if (s->dtype == QMX_DTYPE_SQ_U8) return sq_mfma_ok(s->distance, s->scan_dim) && !option(OPT_NO_MFMA_SCAN);
    if (s->dtype == QMX_DTYPE_TQ) return !option(OPT_NO_MFMA_SCAN);   // scan_sq_mfma.hip TqOps / Tq1Ops
    return (s->dtype == QMX_DTYPE_F32 || s->dtype == QMX_DTYPE_F16) && (s->distance == QMX_DISTANCE_DOT || s->distance == QMX_DISTANCE_COSINE) && s->dim >= 32 &&
           s->fast_layout() && !option(OPT_NO_MFMA_SCAN);
}
constexpr uint32_t MAX_QT_MFMA = 32;
constexpr uint32_t MAX_QT_TOPK = 64;   // the chain-major f32 top-k scan (scan_mfma16.hip) takes 64 queries per pass of the block
// queries scored per pass of the stored block
static bool bq_mfma_ok(const qmx_query *q);
static uint32_t tile_qt(const qmx_segment *s, const qmx_query *q) {
    if (s->dtype == QMX_DTYPE_BQ) return bq_mfma_ok(q) && (size_t)MAX_QT_MFMA * q->q_stride <= 150 * 1024 ? MAX_QT_MFMA : MAX_QT;
    if (s->dtype == QMX_DTYPE_SQ_U8) return mfma_scan_ok(s) ? MAX_QT_MFMA : MAX_QT;
    if (s->dtype == QMX_DTYPE_TQ) return mfma_scan_ok(s) && (size_t)MAX_QT_MFMA * q->q_stride <= 150 * 1024 ? MAX_QT_MFMA : 4;      // (the VALU kernels of TurboQuant are built for 1, 2 and 4 queries)
    return mfma_scan_ok(s) && (size_t)MAX_QT_MFMA * (((size_t)s->dim * 4 + 127) / 128 * 128 + QUERY_AUX_BYTES) <= 150 * 1024 ? MAX_QT_MFMA : MAX_QT;
}

// BQ rows against scalar-encoded queries (4 / 8 bit planes): 4 queries and more go to the int8 matrix cores (scan_sq_mfma.hip BqOps); the entries carry
// the byte form of the values for it (query_alloc)
static bool bq_mfma_ok(const qmx_query *q) {
    return q->seg->dtype == QMX_DTYPE_BQ && q->tq_qbytes_off != 0 && !option(OPT_NO_MFMA_SCAN) && (size_t)MAX_QT * q->q_stride <= 150 * 1024;
}

// Entries of a query tile that scan_sq_mfma.hip keeps in LDS: its B operand is one ds_read_b128 per lane at (query n) * stride + (16-byte piece kg),
// 16 lanes per LDS cycle over 64 banks - a stride of 64 bytes mod 256 (what 128-byte aligned bodies + the 64-byte aux block give) puts queries n and
// n + 4 on the same banks (SQ_LDS_BANK_CONFLICT 74 % of the LDS cycles of the 1-bit scan); 16 bytes mod 256 spreads the 16 queries of a group over
// the 16 slots of a bank row.
static uint32_t lds_tile_stride(uint32_t bytes) { return bytes + (16u + 256u - bytes % 256u) % 256u; }

// A TurboQuant query entry: `pieces` 16-byte query pieces per 16-byte row piece (scan_tq.hip; 1-bit storage under TQ+: 16 bit planes), zero padded
// to whole 64-byte row steps (the matrix-core scan, scan_sq_mfma.hip TqOps, reads whole steps); behind the bit planes of a 1-bit storage the
// same query as i8 bytes (8 per row byte; 16 with the two halves of a 16-bit TQ+ query); then the aux block.
static void tq_entry_layout(const qmx_segment *seg, uint32_t *pieces, uint32_t *qbytes_off, uint32_t *aux_off) {
    *pieces = seg->tq_value_bits == 4 ? 4 : (seg->tq_value_bits == 1 && seg->d_tq_shift) ? 16 : 8;
    const uint32_t body = (seg->scan_dim + 63) & ~63u;
    *aux_off = body * *pieces;
    *qbytes_off = 0;
    if (seg->tq_value_bits == 1) {
        *qbytes_off = *aux_off;
        *aux_off += body * (*pieces == 16 ? 16 : 8);
    }
}

// stage a possibly-host buffer on the query's stream; returns a device pointer
static int32_t stage_in(qmx_query *q, DevBuf &buf, const void *src, size_t bytes, const void **dev_out) {
    if (bytes == 0 || !src) {
        *dev_out = nullptr;
        return QMX_OK;
    }
    if (is_device_ptr(src)) {
        *dev_out = src;
        return QMX_OK;
    }
    QMX_TRY(buf.reserve(bytes));
    QMX_HIP(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, q->stream));
    *dev_out = buf.p;
    return QMX_OK;
}

static int32_t copy_out(hipStream_t st, void *dst, const void *src_dev, size_t bytes) {
    if (bytes == 0) return QMX_OK;
    QMX_HIP(hipMemcpyAsync(dst, src_dev, bytes, is_device_ptr(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    return QMX_OK;
}

static int32_t check_err_flag(qmx_query *q) {
    int flag = 0;
    QMX_HIP(hipMemcpyAsync(&flag, q->d_err, sizeof(int), hipMemcpyDeviceToHost, q->stream));
    QMX_HIP(hipStreamSynchronize(q->stream));
    if (flag) {
        QMX_HIP(hipMemsetAsync(q->d_err, 0, sizeof(int), q->stream));
        if (flag == 2) {
            set_error("a search outgrew its scratch: the `candidates` heap of hnsw_reference_heap_order (%u entries) or search_with_vectors' stack of evicted candidates "
                      "that tie with the bound (%u)", HNSW_REF_CAND_CAP, HNSW_EV_SPILL_CAP);
            return QMX_ERR_NOT_SUPPORTED;
        }
        set_error("point offset out of range for this segment (the reference panics here)");
        return QMX_ERR_OUT_OF_BOUNDS;
    }
    return QMX_OK;
}

// ---- kernel timing: event pairs on the query's stream, no host synchronisation while recording ----
static int32_t timing_begin(qmx_query *q, size_t *slot) {
    if (q->ev_used == q->evs.size()) {
        qmx_query::EvPair p;
        QMX_HIP(hipEventCreate(&p.a));
        QMX_HIP(hipEventCreate(&p.b));
        q->evs.push_back(p);
    }
    *slot = q->ev_used++;
    QMX_HIP(hipEventRecord(q->evs[*slot].a, q->stream));
    return QMX_OK;
}
static int32_t timing_end(qmx_query *q, size_t slot) {
    QMX_HIP(hipEventRecord(q->evs[slot].b, q->stream));
    return QMX_OK;
}
// stream must be idle (caller synchronised): folds the recorded pairs into timing_ms
static int32_t timing_fold(qmx_query *q) {
    for (size_t i = 0; i < q->ev_used; ++i) {
        float ms = 0.f;
        QMX_HIP(hipEventElapsedTime(&ms, q->evs[i].a, q->evs[i].b));
        q->timing_ms += ms;
        q->timing_launches++;
    }
    q->ev_used = 0;
    return QMX_OK;
}

static uint32_t pow2_ceil(uint32_t x) {
    uint32_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

// ---- constants and plain structs of the families, used across them ----
// queries per conditional exact pass behind the f32 prefilters: the 64-query shape of the chain-major scan where it takes the row length (dim <= 768),
// the 32-query shape beyond (dim <= 2048), 0: no prefilter for this row length
static inline uint32_t split_fallback_qt(uint32_t dim) { return mfma16_dim_ok(64, dim) ? 64u : mfma16_dim_ok(32, dim) ? 32u : 0u; }

// ---------------------------------------------------------------------------------------------
// brute-force top-k
// ---------------------------------------------------------------------------------------------
constexpr uint32_t MAX_TOP = 65536;  // top > MAX_TOP_FAST runs in passes of MAX_TOP_FAST, each bounded by the last key of the one before (a top of 65536: 1024 passes)

constexpr uint32_t SPLIT_QT = 128;          // queries per pass of the split prefilter (scan_split.hip) ...
constexpr uint32_t SPLIT_QT_MAX = 256;      // ... and of its 256-query shape over a half copy (batches of more than 128 queries)
constexpr uint32_t SPLIT_CAND_CAP = 131072; // candidate keys per query and pass (expected: ~1000 k; heavy-tailed rows under the int8 band: tens of thousands)
constexpr uint32_t SPLIT_VCAP = 16384;      // rows that get an exact score, per query of the batch ON AVERAGE: the batch shares one pool (verify_pool) from which a query
                                            // takes what it needs (expected: ~k; the one-product mode's band holds ~60 on iid rows, the int8 band ~100 on Gaussian rows,
                                            // thousands - with a long tail over the queries - where a few coordinates dominate): 16384 rows x 3 KiB are 50 MB of
                                            // gathers - a 128-query batch that fills the pool gathers a fifth of a block pass; beyond that the exact scan is cheaper
constexpr uint32_t SPLIT_FQT = 64;          // queries per conditional exact pass behind the prefilter (one 16-query pass instead when 1..16 overflowed)

// device block behind qmx_query::sp_plan: what the prefilter of one search did and which of its queries take the exact scan after all
struct SplitPlanLayout {
    size_t count, run16, run64, tile_ovf, ovf_q, zero_bytes, list, gthr_packed, bytes;   // byte offsets (SplitStats sits at 0)
    uint32_t n_run64, list_cap;
    explicit SplitPlanLayout(uint32_t nq, uint32_t fqt = SPLIT_FQT) {      // fqt: queries per conditional exact pass (64; 32 for rows the 64-query shape does not take)
        n_run64 = (nq + fqt - 1) / fqt;
        list_cap = n_run64 * fqt;
        count = 32; run16 = 36; run64 = 40;
        tile_ovf = run64 + (size_t)n_run64 * 4;
        ovf_q = tile_ovf + ((size_t)nq / 128 + 1) * 4;
        zero_bytes = ovf_q + (size_t)nq * 4;                       // everything up to here starts a search as zeros
        list = (zero_bytes + 7) / 8 * 8;
        gthr_packed = (list + (size_t)list_cap * 4 + 7) / 8 * 8;
        bytes = gthr_packed + (size_t)list_cap * 8;
    }
};

// ---------------------------------------------------------------------------------------------
// HNSW search on device
// ---------------------------------------------------------------------------------------------
struct qmx_hnsw {
    int device = 0;
    uint32_t m = 0, m0 = 0, n_points = 0, n_levels = 0, n_ep = 0, n_xp = 0;
    uint64_t n_offsets = 0, n_neighbors = 0;
    uint32_t *d_reindex = nullptr, *d_neighbors = nullptr, *d_ep_ids = nullptr, *d_ep_levels = nullptr, *d_xp_ids = nullptr,
             *d_xp_levels = nullptr;
    uint64_t *d_level_offsets = nullptr, *d_offsets = nullptr;
    uint32_t *d_l0 = nullptr;     // packed level 0 [n_points][l0_stride]: count, links (built when every list fits 63 links)
    uint32_t l0_stride = 0;
    // the same table with the SQ vector_offset of every linked row behind the links ([n_points][2 m0]), made at the first plain walk of an SQ segment
    // over this graph and kept for that segment (its uid): searches share the graph across threads, hence the lock
    mutable std::mutex l0x_mu;
    mutable uint32_t *d_l0x = nullptr;
    mutable uint64_t l0x_segment_uid = 0;
    mutable bool l0x_failed = false;     // no memory for it: the walk keeps the offsets column
    // host copy of the plain arrays (graphs built by qmx_hnsw_build; empty otherwise) for qmx_hnsw_export_plain
    std::vector<uint32_t> h_reindex, h_neighbors, h_ep_ids, h_ep_levels, h_xp_ids, h_xp_levels;
    std::vector<uint64_t> h_level_offsets, h_offsets;
};

constexpr uint32_t HNSW_SLOT_CAP = 4096;
constexpr uint32_t HNSW_LOG_CAP = 16384;                    // words logged per search before falling back to a full clear
constexpr uint64_t HNSW_VIS_BUDGET = 8ull << 30;            // bytes of visited bitmaps per query handle

// the points of a build over multi-vectors (qmx_multi_hnsw_build): point p = inner rows [offsets[p], offsets[p + 1]) of the segment, deleted flags per POINT
struct MultiBuild {
    const uint64_t *h_offsets;
    uint32_t n_points;
    const uint64_t *h_deleted;
    uint64_t n_deleted_bits;
};

// the MaxSim walk over multi-vector points (qmx_multi_hnsw_search): device arrays of the query / point partitions and the POINT-level deleted view
// search_on_level_with_vectors: where the walk lists the candidates it pops
struct ExpandedOut {
    uint32_t *d_ids, *d_cnt;
    uint32_t xcap;
};

// qmx_hnsw_search_traced: where the plain walk lists the candidates it pops and expands, with their scores
struct PopTrace {
    qmx_scored_point *d_pops;
    uint32_t *d_cnt;
    uint32_t cap;
};

struct MultiWalk {
    const uint32_t *d_qfirst;
    const uint64_t *d_offsets;
    uint32_t n_queries, max_tokens;
    DeletedView del;
};

// a custom query (Recommend / Discover / Context / Feedback) as the walk's scorer (qmx_custom_hnsw_search): the descriptors on the device, the
// largest number of examples one query has
struct CustomWalk {
    const qmx_custom_query *d_desc;
    const float *d_coefs;
    uint32_t n_queries, max_examples;
    uint32_t lds_bytes = 0;      // multi-vector examples: bytes of the largest staged query block (header + offset table + the examples' tokens)
};

// EncodedVectorsTQ over Distance::Manhattan: no integer kernel, every score dequantises and rotates the row back (tq_l1.hip)
static bool tq_l1(const qmx_segment *s) { return s->dtype == QMX_DTYPE_TQ && s->distance == QMX_DISTANCE_MANHATTAN; }

// ---- internal functions one family calls in another (defined in the file named) ----
extern "C" {
int32_t hnsw_enqueue(const qmx_hnsw *g, qmx_query *q, uint32_t top, uint32_t ef, qmx_scored_point *d_out,
                            uint32_t *d_counts, uint32_t *d_scored, bool timed, bool acorn = false, const MultiWalk *mw = nullptr,
                            const ExpandedOut *xo = nullptr, const CustomWalk *cw = nullptr, const PopTrace *pt = nullptr);
int32_t hnsw_check(const qmx_hnsw *g, const qmx_query *q, uint32_t top, uint32_t ef);
int32_t hnsw_search_sync(const qmx_hnsw *g, qmx_query *q, uint32_t top, uint32_t ef, qmx_scored_point *out, uint32_t *out_counts,
                                const volatile uint8_t *is_stopped, qmx_counters *counters, bool acorn);
void fill_args(const qmx_query *q, uint32_t tile0, uint32_t nq_tile, ScanArgs &a);
int32_t launch_scan(const qmx_query *q, int qt, ScanMode mode, const ScanArgs &a, uint32_t *grid);
int32_t tq_l1_scores_device(qmx_query *q, uint32_t q0, uint32_t nq, const uint32_t *d_ids, uint64_t n, float *d_scores, uint64_t stride, const PairSel *sel);
int32_t score_matrix_enqueue(const qmx_query *q, uint32_t tile0, uint32_t nq_tile, const uint32_t *d_ids, uint64_t n, float *d_scores, uint64_t stride,
                                    uint32_t *launches);
int32_t score_ids_device(qmx_query *q, const uint32_t *d_ids, uint64_t n, float *d_scores, qmx_counters *counters);
int32_t score_pairs_device(qmx_query *q, const PairSel &sel, const uint32_t *d_ids, uint64_t n_items, float *d_scores,
                                  bool timed);
int32_t search_enqueue(qmx_query *q, uint32_t top, const uint32_t *d_ids, uint64_t n_ids,
                              qmx_scored_point *d_out, uint32_t *d_counts, const volatile uint8_t *is_stopped,
                              qmx_counters *counters, bool timed);
int32_t fold_split_counters(qmx_query *q, qmx_counters *c);
TqRotationHost tq_rotation(const qmx_segment *s);
TqRotationHost tq_rotation_inverse(const qmx_segment *s);
}

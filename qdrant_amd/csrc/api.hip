// api.hip — the C-ABI of include/qdrant_amd.h on top of the HIP kernels.
// Host-side logic only: handle lifetime, argument checks, staging of host buffers, stream order.
// No CPU scoring path exists here: every score is produced by a gfx950 kernel or the call fails.
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

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int32_t hip_status(hipError_t e, const char *what, const char *file, int line) {
    set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    (void)hipGetLastError();
    switch (e) {
        case hipErrorOutOfMemory: return QMX_ERR_OUT_OF_MEMORY;
        case hipErrorNoDevice:
        case hipErrorInvalidDevice:
        case hipErrorNoBinaryForGpu:
        case hipErrorInsufficientDriver: return QMX_ERR_NO_DEVICE;
        case hipErrorNotReady: return QMX_ERR_NOT_READY;
        case hipErrorInvalidValue: return QMX_ERR_BAD_ARG;
        default: return QMX_ERR_OTHER;
    }
}

// ---- kernel-path options: the environment is read once, here, at load time ----
static const char *const g_option_names[OPT_COUNT] = {"no_mfma_scan", "no_mfma16", "no_mfma16_q64", "no_prescan", "prescan_shift", "hnsw_no_packed_l0",
                                                     "hnsw_pq_lds_lut", "hnsw_log_cap", "bq_lanes8", "mfma_no_nt", "mfma_no_fast", "no_pq_tiled", "no_split_scan", "split_min_queries", "no_split256", "no_pq_pair", "no_pq_prefilter", "pq_prefilter_min_queries", "hnsw_pq_per_cu", "tq_rotate_block", "no_topk_small", "verify_max_per_query", "no_hnsw_pq_block",
                                                     "hnsw_pq_block_waves", "debug"};
struct OptionTable {
    std::atomic<int64_t> v[OPT_COUNT];
    int64_t initial[OPT_COUNT];
    OptionTable() {
        for (int i = 0; i < OPT_COUNT; ++i) {
            char env[64] = "QMX_";
            size_t k = 4;
            for (const char *c = g_option_names[i]; *c && k + 1 < sizeof(env); ++c) env[k++] = (char)toupper((unsigned char)*c);
            env[k] = 0;
            const char *e = getenv(env);
            int64_t val = 0;
            if (e) { val = (*e >= '0' && *e <= '9') ? atoll(e) : 1; }   // "QMX_X=" / "QMX_X=yes" count as set
            if (i == OPT_PRESCAN_SHIFT && !e) val = 10;
            if (i == OPT_SPLIT_MIN_QUERIES && !e) val = 1;
            if (i == OPT_PQ_PREFILTER_MIN_QUERIES && !e) val = 4;
            initial[i] = val;
            v[i].store(val, std::memory_order_relaxed);
        }
    }
};
static OptionTable g_options;
int64_t option(Option o) { return g_options.v[o].load(std::memory_order_relaxed); }

static thread_local const void *g_last_kernel = nullptr;
void note_kernel(const void *host_function) { g_last_kernel = host_function; }

void clear_stale_error() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        const bool debug = option(OPT_DEBUG) != 0;
        if (debug) fprintf(stderr, "[qmx] dropped stale HIP error %d (%s) before a kernel launch\n", (int)e, hipGetErrorString(e));
    }
}

static bool is_device_ptr(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

static uint32_t elem_bytes(uint32_t dtype) {
    switch (dtype) {
        case QMX_DTYPE_F32: return 4;
        case QMX_DTYPE_F16: return 2;
        default: return 1;
    }
}

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

static int32_t check_device(int32_t device_id, hipDeviceProp_t *prop_out) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        (void)hipGetLastError();
        set_error("no HIP device visible (%s); libqdrant_amd has no CPU fallback", e == hipSuccess ? "count=0" : hipGetErrorString(e));
        return QMX_ERR_NO_DEVICE;
    }
    QMX_REQUIRE(device_id >= 0 && device_id < count, QMX_ERR_NO_DEVICE, "device %d out of range (have %d)", device_id, count);
    hipDeviceProp_t prop;
    QMX_HIP(hipGetDeviceProperties(&prop, device_id));
    QMX_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, QMX_ERR_NO_DEVICE,
                "device %d is %s; this library ships gfx950 (MI355X) code objects only", device_id, prop.gcnArchName);
    QMX_HIP(hipSetDevice(device_id));
    if (prop_out) *prop_out = prop;
    return QMX_OK;
}

}  // namespace qmx

using namespace qmx;

// ---------------------------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------------------------
struct qmx_segment {
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
    // TurboQuant (scan_tq.hip): parameters, the extras columns and the rotation tables
    uint32_t tq_bits = 0, tq_value_bits = 0, tq_padded_dim = 0, tq_rot_dim = 0, tq_code_bytes = 0, tq_n_chunks = 0;
    bool tq_invert = false;
    float *d_tq_sf = nullptr, *d_tq_l2 = nullptr, *d_tq_xm = nullptr;   // extras columns (xm: TQ+ only)
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
    uint64_t last_row_bytes = 0, last_n_cand = 0;
    uint64_t sp_sample_n = 0, sp_sample_of = 0;
    DevBuf filter;             // payload-filter allow bitmap of this query batch (qmx_query_set_filter)
    uint64_t n_filter_bits = 0;
    bool has_filter = false;
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

// ---------------------------------------------------------------------------------------------
// library / device
// ---------------------------------------------------------------------------------------------
extern "C" {

uint32_t qmx_abi_version(void) { return 5; }

static int option_index(const char *name) {
    if (!name) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strcmp(name, g_option_names[i]) == 0) return i;
    return -1;
}
int32_t qmx_set_option(const char *name, int64_t value) {
    const int i = option_index(name);
    QMX_REQUIRE(i >= 0, QMX_ERR_BAD_ARG, "unknown option '%s'", name ? name : "(null)");
    g_options.v[i].store(value < 0 ? g_options.initial[i] : value, std::memory_order_relaxed);
    return QMX_OK;
}
int32_t qmx_get_option(const char *name, int64_t *out_value) {
    const int i = option_index(name);
    QMX_REQUIRE(i >= 0 && out_value, QMX_ERR_BAD_ARG, "unknown option '%s'", name ? name : "(null)");
    *out_value = option((Option)i);
    return QMX_OK;
}

int32_t qmx_last_error(char *buf, size_t buf_len) {
    if (!buf || buf_len == 0) return QMX_ERR_BAD_ARG;
    snprintf(buf, buf_len, "%s", g_last_error.c_str());
    return QMX_OK;
}

int32_t qmx_device_count(int32_t *out_count) {
    QMX_REQUIRE(out_count, QMX_ERR_BAD_ARG, "out_count is NULL");
    *out_count = 0;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return QMX_ERR_NO_DEVICE;
    }
    int ok = 0;
    for (int i = 0; i < count; ++i) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    *out_count = ok;
    return QMX_OK;
}

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
static bool segment_i8_eligible(const qmx_segment *s) { return s->split_stats && split_i8_dim_ok(s->dim) && mfma16_dim_ok(64, s->dim); }    // (dims the prefilter path serves: search_enqueue)
static bool segment_f16_eligible(const qmx_segment *s) { return s->split_stats && s->dim % 128 == 0; }
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
    *ms_out = best;
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
            const bool half_wins = rc == QMX_OK && s->d_rows_split && s->auto_half_ms < s->auto_i8_ms;
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
static TqRotationHost tq_rotation(const qmx_segment *s) {
    TqRotationHost h;
    h.d_maps = s->d_tq_tables;
    h.d_chunk_off = s->d_tq_tables + (size_t)3 * s->tq_rot_dim;
    h.d_chunk_size = h.d_chunk_off + 32;
    h.d_chunk_norm = s->d_tq_norms;
    h.n_chunks = s->tq_n_chunks; h.rot_dim = s->tq_rot_dim; h.padded_dim = s->tq_padded_dim; h.dim = s->dim;
    return h;
}

// HadamardRotation::apply_inverse: the same rounds over the backward maps
static TqRotationHost tq_rotation_inverse(const qmx_segment *s) {
    TqRotationHost h = tq_rotation(s);
    h.d_maps = s->d_tq_tables + (size_t)3 * s->tq_rot_dim + 64;
    return h;
}
// EncodedVectorsTQ over Distance::Manhattan: no integer kernel, every score dequantises and rotates the row back (tq_l1.hip)
static bool tq_l1(const qmx_segment *s) { return s->dtype == QMX_DTYPE_TQ && s->distance == QMX_DISTANCE_MANHATTAN; }

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
static void fill_args(const qmx_query *q, uint32_t tile0, uint32_t nq_tile, ScanArgs &a) {
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

static int32_t launch_scan(const qmx_query *q, int qt, ScanMode mode, const ScanArgs &a, uint32_t *grid) {
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
static int32_t tq_l1_scores_device(qmx_query *q, uint32_t q0, uint32_t nq, const uint32_t *d_ids, uint64_t n, float *d_scores, uint64_t stride, const PairSel *sel) {
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
static int32_t score_matrix_enqueue(const qmx_query *q, uint32_t tile0, uint32_t nq_tile, const uint32_t *d_ids, uint64_t n, float *d_scores, uint64_t stride,
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
static int32_t score_ids_device(qmx_query *q, const uint32_t *d_ids, uint64_t n, float *d_scores, qmx_counters *counters) {
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
// brute-force top-k
// ---------------------------------------------------------------------------------------------
constexpr uint32_t MAX_TOP = 65536;  // top > MAX_TOP_FAST runs in passes of MAX_TOP_FAST, each bounded by the last key of the one before (a top of 65536: 1024 passes)

static int32_t score_pairs_device(qmx_query *q, const PairSel &sel, const uint32_t *d_ids, uint64_t n_items, float *d_scores, bool timed);

// triage aid (qmx_set_option("debug", 2)): synchronise after every stage of the split path and name it on stderr
static int32_t split_stage(qmx_query *q, const char *what) {
    if (option(OPT_DEBUG) < 2) return QMX_OK;
    hipError_t e = hipStreamSynchronize(q->stream);
    fprintf(stderr, "[qmx] split stage %-28s %s\n", what, e == hipSuccess ? "ok" : hipGetErrorString(e));
    fflush(stderr);
    return e == hipSuccess ? QMX_OK : QMX_ERR_OTHER;
}

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
    explicit SplitPlanLayout(uint32_t nq) {
        n_run64 = (nq + SPLIT_FQT - 1) / SPLIT_FQT;
        list_cap = n_run64 * SPLIT_FQT;
        count = 32; run16 = 36; run64 = 40;
        tile_ovf = run64 + (size_t)n_run64 * 4;
        ovf_q = tile_ovf + ((size_t)nq / 128 + 1) * 4;
        zero_bytes = ovf_q + (size_t)nq * 4;                       // everything up to here starts a search as zeros
        list = (zero_bytes + 7) / 8 * 8;
        gthr_packed = (list + (size_t)list_cap * 4 + 7) / 8 * 8;
        bytes = gthr_packed + (size_t)list_cap * 8;
    }
};
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
        q->last_kernel = g_last_kernel;
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
        QMX_TRY(launch_merge_keys(q->stream, (const uint64_t *)q->partial.p, slabs, q->nq, q->nq, top, d_out, d_counts, top, 0, nullptr, a.run_if, ovf_list));
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

static int32_t search_enqueue(qmx_query *q, uint32_t top, const uint32_t *d_ids, uint64_t n_ids,
                              qmx_scored_point *d_out, uint32_t *d_counts, const volatile uint8_t *is_stopped,
                              qmx_counters *counters, bool timed) {
    const qmx_segment *s = q->seg;
    const uint64_t n_cand = d_ids ? n_ids : s->scan_rows();
    if (tq_l1(s)) {     // the score matrix (tiles of queries: at most 2^31 scores at a time), then one block per query selects its k best live candidates
        q->last_counters = qmx_counters{};
        q->last_split = false;
        ScanArgs a;
        fill_args(q, 0, q->nq, a);
        const uint32_t qtile = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(q->nq, (1ull << 31) / std::max<uint64_t>(n_cand, 1)));
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
    // partial lists: one per block; bound the grid by what the buffer holds
    const uint32_t grid_cap = (uint32_t)s->num_cus * 8;
    // f32 dot / cosine rows of 256, 512 or 768 floats, whole block: 64 queries per pass (scan_mfma16.hip); everything else 32 / 16
    const bool q64 = s->dtype == QMX_DTYPE_F32 && mfma_scan_ok(s) && q->nq > MAX_QT_MFMA && mfma16_dim_ok(64, s->dim) &&
                     !option(OPT_NO_MFMA16) && !option(OPT_NO_MFMA16_Q64);
    // ... and rows of 1024 .. 1536 floats 32 per pass: that kernel keeps the queries in registers, not in an LDS tile (tile_qt's limit)
    const bool q32 = s->dtype == QMX_DTYPE_F32 && mfma_scan_ok(s) && q->nq > MAX_QT && s->dim > 768 && mfma16_dim_ok(32, s->dim) &&
                     !option(OPT_NO_MFMA16);
    // more than 64 queries over a large f32 dot / cosine block: 128 per pass through the f16-split matrix-core prefilter, the survivors
    // re-scored exactly (scan_split.hip); the result is the exact scan's, bit for bit
    // ... and with a derived copy of the block (QMX_SEG_HALF_COPY / QMX_SEG_SPLIT_COPY) that path serves EVERY batch size: it streams 2 (4) bytes
    // per element instead of 4 and is HBM-bound whatever the number of queries (10 M x 768: 3.0 ms per pass against 4.4 ms for the f32 stream)
    const bool split_dims = s->dtype == QMX_DTYPE_F32 && mfma_scan_ok(s) && mfma16_dim_ok(64, s->dim) && !option(OPT_NO_MFMA16);
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
    const SplitPlanLayout pl(q->nq);
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
                for (uint32_t phase = 1; phase <= 2; ++phase) {
                    size_t slot = 0;
                    if (timed) QMX_TRY(timing_begin(q, &slot));
                    QMX_TRY(launch_scan_i8copy(q->stream, a, q->sp_bq.p, sp_qscale, sp_thr, s->num_cus, s->d_rows_split, q->sp_wl.p, phase));
                    q->last_kernel = g_last_kernel;
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
                q->last_kernel = g_last_kernel;
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
            q->last_kernel = g_last_kernel;
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
        const uint32_t n_run64 = (last + SPLIT_FQT - 1) / SPLIT_FQT, n_slots = n_run64 * SPLIT_FQT;
        QMX_TRY(launch_split_plan(q->stream, (const uint32_t *)(plan + pl.ovf_q), last, (const uint64_t *)q->gthr.p, ovf_list, gthr_packed, n_slots,
                                  (uint32_t *)(plan + pl.count), (int *)(plan + pl.run16), (int *)(plan + pl.run64), n_run64, (SplitStats *)plan, q->d_queries,
                                  q->q_stride, q->sp_fq.p));
        for (uint32_t pass = 0; pass <= n_run64; ++pass) {      // pass 0: the 16-query shape; pass p >= 1: packed queries 64 (p - 1) ..
            if (pass && last <= 16) break;
            const uint32_t p0 = pass ? (pass - 1) * SPLIT_FQT : 0;
            const uint32_t nq_sub = pass ? std::min<uint32_t>(SPLIT_FQT, last - p0) : std::min<uint32_t>(16, last);
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
static int32_t fold_split_counters(qmx_query *q, qmx_counters *c) {
    if (!q->last_split || !q->sp_plan.p) return QMX_OK;
    SplitStats st;
    QMX_HIP(hipMemcpy(&st, q->sp_plan.p, sizeof(st), hipMemcpyDeviceToHost));
    c->prefilter_candidates = st.candidates;
    c->verified_rows = st.verified;
    c->fallback_queries = st.fallback_queries;
    c->bytes_read += st.verified * q->last_row_bytes;
    if (st.fallback_queries) {
        const uint32_t f = st.fallback_queries;
        const uint64_t passes = q->last_pq ? f : f <= 16 ? 1 : (f + SPLIT_FQT - 1) / SPLIT_FQT;     // (the exact PQ kernel streams the codes once per query)
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
    // host copy of the plain arrays (graphs built by qmx_hnsw_build; empty otherwise) for qmx_hnsw_export_plain
    std::vector<uint32_t> h_reindex, h_neighbors, h_ep_ids, h_ep_levels, h_xp_ids, h_xp_levels;
    std::vector<uint64_t> h_level_offsets, h_offsets;
};

int32_t qmx_hnsw_destroy(qmx_hnsw *g) {
    if (!g) return QMX_OK;
    (void)hipSetDevice(g->device);
    void *ptrs[] = {g->d_reindex, g->d_neighbors, g->d_ep_ids, g->d_ep_levels, g->d_xp_ids, g->d_xp_levels, g->d_level_offsets, g->d_offsets, g->d_l0};
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
    if (rc == QMX_OK && d->n_points && d->m0 <= 63 && !option(OPT_HNSW_NO_PACKED_L0)) {
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

constexpr uint32_t HNSW_SLOT_CAP = 4096;
constexpr uint32_t HNSW_LOG_CAP = 16384;                    // words logged per search before falling back to a full clear
constexpr uint64_t HNSW_VIS_BUDGET = 8ull << 30;            // bytes of visited bitmaps per query handle

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
        return launch_hnsw_build_maxsim_dense(nullptr, (int)seg->dtype, (int)seg->distance, a, h, phase, grid, per_cu);
    }
    if (seg->dtype == QMX_DTYPE_SQ_U8) return launch_hnsw_build_sq(nullptr, (int)seg->distance, a, h, phase, grid, per_cu);
    if (seg->dtype == QMX_DTYPE_BQ) return launch_hnsw_build_bq(nullptr, a, h, phase, grid, per_cu);
    if (seg->dtype == QMX_DTYPE_PQ) return launch_hnsw_build_pq(nullptr, a, h, phase, grid, per_cu);
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

// the points of a build over multi-vectors (qmx_multi_hnsw_build): point p = inner rows [offsets[p], offsets[p + 1]) of the segment, deleted flags per POINT
struct MultiBuild {
    const uint64_t *h_offsets;
    uint32_t n_points;
    const uint64_t *h_deleted;
    uint64_t n_deleted_bits;
};
static int32_t hnsw_build_impl(const qmx_segment *seg, const qmx_segment *original, const qmx_hnsw_build_params *bp, qmx_hnsw **out, const MultiBuild *mb);

int32_t qmx_hnsw_build_quantized(const qmx_segment *seg, const qmx_segment *original, const qmx_hnsw_build_params *bp, qmx_hnsw **out) {
    return hnsw_build_impl(seg, original, bp, out, nullptr);
}

// Build fan-out over independent segments (gpu_devices_manager.rs:120-143 + hnsw/build.rs:53: one device locked per segment build, builds share nothing):
// one host thread per segment, each driving its segment's device; the thread's own error text travels back with its status.
int32_t qmx_sharded_hnsw_build(const qmx_segment *const *segments, const qmx_segment *const *originals, uint32_t n_segments,
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
        rcs[i] = hnsw_build_impl(segments[i], originals ? originals[i] : nullptr, bp, &out_graphs[i], nullptr);
        if (rcs[i] != QMX_OK) errs[i] = g_last_error;          // (thread-local: copied out before the thread ends)
    };
    if (n_segments == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        pool.reserve(n_segments);
        for (uint32_t i = 0; i < n_segments; ++i) pool.emplace_back(work, i);
        for (auto &t : pool) t.join();
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

int32_t qmx_multi_hnsw_build(const qmx_segment *inner, const uint64_t *point_offsets, uint32_t n_points, const uint64_t *point_deleted,
                             uint64_t n_deleted_bits, const qmx_hnsw_build_params *bp, qmx_hnsw **out) {
    QMX_REQUIRE(inner && point_offsets && bp && out, QMX_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    QMX_REQUIRE(inner->dtype == QMX_DTYPE_F32 || inner->dtype == QMX_DTYPE_F16 || inner->dtype == QMX_DTYPE_SQ_U8 || inner->dtype == QMX_DTYPE_BQ,
                QMX_ERR_NOT_SUPPORTED, "device HNSW build over multi-vectors: inner dtype %u not supported (f32, f16, SQ, BQ)", inner->dtype);
    QMX_REQUIRE(!is_device_ptr(point_offsets) && !is_device_ptr(point_deleted), QMX_ERR_BAD_ARG, "point_offsets and point_deleted are host arrays");
    for (uint32_t p = 0; p < n_points; ++p)
        QMX_REQUIRE(point_offsets[p] <= point_offsets[p + 1], QMX_ERR_BAD_ARG, "point_offsets is not ascending at %u", p);
    QMX_REQUIRE(point_offsets[n_points] <= inner->n, QMX_ERR_OUT_OF_BOUNDS, "point_offsets reach past the %llu inner rows of the segment",
                (unsigned long long)inner->n);
    const MultiBuild mb{point_offsets, n_points, (point_deleted && n_deleted_bits) ? point_deleted : nullptr, point_deleted ? n_deleted_bits : 0};
    return hnsw_build_impl(inner, nullptr, bp, out, &mb);
}

static int32_t hnsw_build_impl(const qmx_segment *seg, const qmx_segment *original, const qmx_hnsw_build_params *bp, qmx_hnsw **out, const MultiBuild *mb) {
    QMX_REQUIRE(seg && bp && out, QMX_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    QMX_REQUIRE(seg->dtype <= QMX_DTYPE_BQ || seg->dtype == QMX_DTYPE_TQ, QMX_ERR_NOT_SUPPORTED, "device HNSW build: dtype %u not supported", seg->dtype);
    QMX_REQUIRE(!tq_l1(seg), QMX_ERR_NOT_SUPPORTED, "device HNSW build through a TurboQuant storage over Manhattan is not built (build over the original vectors)");
    const bool from_original = seg->dtype == QMX_DTYPE_PQ || seg->dtype == QMX_DTYPE_TQ;
    if (from_original) {   // point_scorer.rs:197-212: the insertion searches score through the query (PQ: LUT) of the ORIGINAL vector
        QMX_REQUIRE(original, QMX_ERR_NOT_SUPPORTED,
                    "a PQ / TurboQuant segment cannot score a stored row as a query (encode_internal_vector -> None): pass the original f32 segment to qmx_hnsw_build_quantized");
        QMX_REQUIRE(original->dtype == QMX_DTYPE_F32 && original->dim == seg->dim && original->n >= seg->n && original->device == seg->device,
                    QMX_ERR_BAD_ARG, "the original segment must be f32, of the same dim, on the same device and hold every row of the quantized segment");
        QMX_REQUIRE(seg->dtype != QMX_DTYPE_PQ || seg->d_pq_pair, QMX_ERR_NOT_SUPPORTED, "PQ build: the centroid pair table (m x n_centroids^2 floats) exceeds 256 MB");
    }
    QMX_REQUIRE(seg->dtype == QMX_DTYPE_SQ_U8 || seg->dtype == QMX_DTYPE_PQ || seg->fast_layout(), QMX_ERR_NOT_SUPPORTED,
                "adopted device block is not 16-byte aligned");
    QMX_REQUIRE(bp->m >= 1 && bp->m0 >= bp->m && bp->m0 <= HNSW_BUILD_MAX_M0, QMX_ERR_BAD_ARG, "need 1 <= m <= m0 <= %u", HNSW_BUILD_MAX_M0);
    QMX_REQUIRE(bp->ef_construct >= 1 && bp->ef_construct <= HNSW_MAX_EF_REG, QMX_ERR_NOT_SUPPORTED, "ef_construct %u not in 1..%u",
                bp->ef_construct, HNSW_MAX_EF_REG);
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
    DevBuf b_level, b_upoff, b_links0, b_cnt0, b_linksU, b_cntU, b_lock, b_vis, b_log, b_sel, b_sels, b_selc, b_normf, b_normi, b_bq, b_bqsrc, b_rot, b_mvoff,
        b_mvdel;
    auto release_all = [&]() {
        for (DevBuf *b : {&b_level, &b_upoff, &b_links0, &b_cnt0, &b_linksU, &b_cntU, &b_lock, &b_vis, &b_log, &b_sel, &b_sels, &b_selc, &b_normf, &b_normi,
                          &b_bq, &b_bqsrc, &b_rot, &b_mvoff, &b_mvdel}) b->release();
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
        if (seg->dtype == QMX_DTYPE_PQ) {   // query entries = LUTs of the batch's original vectors, read through L2 (as the PQ walk does)
            lut_stride = ((uint64_t)seg->pq_m * seg->pq.n_centroids * sizeof(float) + 15) & ~15ull;
            QB(b_bq.reserve((size_t)max_batch * lut_stride));
            QB(b_bqsrc.reserve((size_t)max_batch * seg->dim * sizeof(float)));
            h.batch_queries = (const unsigned char *)b_bq.p;
            h.batch_q_stride = lut_stride;
            h.lds_query_bytes = 0;
        }
        if (seg->dtype == QMX_DTYPE_TQ) {   // query entries = precompute_query of the batch's original vectors, staged in LDS per insertion
            QB(b_bq.reserve((size_t)max_batch * a.q_stride));
            QB(b_bqsrc.reserve((size_t)max_batch * seg->dim * sizeof(float)));
            QB(b_rot.reserve((size_t)max_batch * seg->tq_padded_dim * sizeof(double)));
            h.batch_queries = (const unsigned char *)b_bq.p;
            h.batch_q_stride = a.q_stride;
            h.lds_query_bytes = a.q_stride;
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
            h.first = next; h.count = count; h.ep_id = ep_id; h.ep_level = ep_level;
            if (seg->dtype == QMX_DTYPE_PQ) {
                // quantized_vectors.raw_scorer(original vector): Metric::preprocess (quantized_query_scorer.rs:39-41; identity for a row
                // normalised at insert, up to the reference's 1e-6 rule), then EncodedVectorsPQ::encode_query for every point of the batch
                float *src = (float *)b_bqsrc.p;
                QH(hipMemcpy2DAsync(src, (size_t)seg->dim * 4, (const char *)original->d_rows + (uint64_t)next * original->row_stride, original->row_stride,
                                    (size_t)seg->dim * 4, count, hipMemcpyDeviceToDevice, nullptr));
                if (seg->distance == QMX_DISTANCE_COSINE) QB(launch_cosine_preprocess_f32(nullptr, src, src, count, seg->dim));
                QB(launch_pq_lut(nullptr, seg->distance, seg->dim, seg->pq, seg->d_centroids, src, count, (float *)b_bq.p));
            }
            if (seg->dtype == QMX_DTYPE_TQ) {   // the same for EncodedVectorsTQ: preprocess, rotate, TurboQuantizer::precompute_query
                float *src = (float *)b_bqsrc.p;
                QH(hipMemcpy2DAsync(src, (size_t)seg->dim * 4, (const char *)original->d_rows + (uint64_t)next * original->row_stride, original->row_stride,
                                    (size_t)seg->dim * 4, count, hipMemcpyDeviceToDevice, nullptr));
                if (seg->distance == QMX_DISTANCE_COSINE) QB(launch_cosine_preprocess_f32(nullptr, src, src, count, seg->dim));
                QB(launch_tq_rotate(nullptr, src, count, tq_rotation(seg), (double *)b_rot.p));
                QB(launch_tq_query_encode(nullptr, (double *)b_rot.p, count, seg->tq_padded_dim, seg->tq_value_bits, seg->distance == QMX_DISTANCE_EUCLID ? 1 : 0,
                                          b_bq.p, a.q_stride, a.aux_off, seg->d_tq_shift, seg->d_tq_scale, a.tq_qbytes_off));
            }
            QB(launch_hnsw_build_any(seg, a, h, 1, (uint32_t)std::min<uint64_t>(slots1, count), &per_cu1));
            QB(launch_hnsw_build_any(seg, a, h, 2, (uint32_t)std::min<uint64_t>(slots2, count), &per_cu2));
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

// the MaxSim walk over multi-vector points (qmx_multi_hnsw_search): device arrays of the query / point partitions and the POINT-level deleted view
// search_on_level_with_vectors: where the walk lists the candidates it pops
struct ExpandedOut {
    uint32_t *d_ids, *d_cnt;
    uint32_t xcap;
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
        set_error("MaxSim walk: inner rows of dtype %u are not built (dense, SQ and BQ are)", s->dtype);
        return QMX_ERR_NOT_SUPPORTED;
    }
    if (s->dtype <= QMX_DTYPE_U8) {
        QMX_REQUIRE(s->fast_layout(), QMX_ERR_NOT_SUPPORTED, "adopted device block is not 16-byte aligned");
        return launch_hnsw_dense(q->stream, (int)s->dtype, (int)s->distance, a, h, grid, per_cu);
    }
    if (s->dtype == QMX_DTYPE_SQ_U8) return launch_hnsw_sq(q->stream, (int)s->distance, a, h, grid, per_cu);
    if (s->dtype == QMX_DTYPE_PQ) {
        // a LUT too large to stage once per wave (the old kernel then gathers it through L2): one block per search, the LUT in LDS (hnsw_pq_block.hip)
        if (q->q_stride > 16 * 1024 && !option(OPT_NO_HNSW_PQ_BLOCK) && !option(OPT_HNSW_PQ_LDS_LUT) && pq_block_walk_ok(a, h)) {
            const int64_t wv = option(OPT_HNSW_PQ_BLOCK_WAVES);
            return launch_hnsw_pq_block(q->stream, a, h, grid, per_cu, wv > 0 ? (int)wv : 8);
        }
        return launch_hnsw_pq(q->stream, a, h, grid, per_cu);
    }
    if (s->dtype == QMX_DTYPE_BQ) return launch_hnsw_bq(q->stream, a, h, grid, per_cu);
    if (tq_l1(s)) return launch_hnsw_tq_l1(q->stream, a, h, grid, per_cu, s->tq_rot_dim);
    if (s->dtype == QMX_DTYPE_TQ) return launch_hnsw_tq(q->stream, a, h, grid, per_cu);
    set_error("dtype %u not built yet", s->dtype);
    return QMX_ERR_NOT_SUPPORTED;
}


static int32_t hnsw_enqueue(const qmx_hnsw *g, qmx_query *q, uint32_t top, uint32_t ef, qmx_scored_point *d_out,
                            uint32_t *d_counts, uint32_t *d_scored, bool timed, bool acorn = false, const MultiWalk *mw = nullptr,
                            const ExpandedOut *xo = nullptr, const CustomWalk *cw = nullptr) {
    const qmx_segment *s = q->seg;
    QMX_REQUIRE(!tq_l1(s) || (!mw && !cw), QMX_ERR_NOT_SUPPORTED, "custom / multi-vector walks through a TurboQuant storage over Manhattan are not built");
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
    h.n_offsets = g->n_offsets; h.n_neighbors = g->n_neighbors;
    h.n_points = g->n_points; h.n_levels = g->n_levels; h.m = g->m; h.m0 = g->m0;
    h.ep_ids = g->d_ep_ids; h.ep_levels = g->d_ep_levels; h.n_ep = g->n_ep;
    h.xp_ids = g->d_xp_ids; h.xp_levels = g->d_xp_levels; h.n_xp = g->n_xp;
    h.ef = ef; h.top = top; h.nq = n_searches;
    h.out = d_out; h.out_counts = d_counts; h.out_scored = d_scored;
    if (xo) { h.expanded = xo->d_ids; h.expanded_cnt = xo->d_cnt; h.xcap = xo->xcap; }
    h.lds_query_bytes = q->q_stride <= HNSW_LDS_QUERY_MAX ? q->q_stride : 0;
    if (tq_l1(s)) {
        h.lds_query_bytes = tq_l1_lds_bytes(s->dim, s->tq_rot_dim);
        QMX_REQUIRE(h.lds_query_bytes <= HNSW_LDS_QUERY_MAX, QMX_ERR_NOT_SUPPORTED, "TurboQuant over Manhattan, the walk: %u bytes of LDS per search", h.lds_query_bytes);
    }
    if (mw) {   // [16-byte header][the multi-query's inner vectors]
        const uint64_t need = 16 + (uint64_t)std::max<uint32_t>(mw->max_tokens, 1) * q->q_stride;
        QMX_REQUIRE(need <= HNSW_LDS_QUERY_MAX, QMX_ERR_NOT_SUPPORTED, "a multi-query of %u inner vectors x %u bytes does not fit the LDS", mw->max_tokens,
                    q->q_stride);
        h.lds_query_bytes = (uint32_t)need;
    }
    // A PQ LUT of more than half the LDS leaves one search per CU; the walk is a chain of dependent memory round trips,
    // so many searches per CU with the LUT read through L2 win (measured: tools/bench_hnsw.py)
    if (s->dtype == QMX_DTYPE_PQ && q->q_stride > 16 * 1024 && !option(OPT_HNSW_PQ_LDS_LUT)) h.lds_query_bytes = 0;
    if (cw) {   // [32-byte header][the examples' entries]: staged when they fit a modest share of the LDS, read through L2 otherwise (PQ LUTs always)
        const uint64_t need = 32 + (uint64_t)std::max<uint32_t>(cw->max_examples, 1) * q->q_stride;
        h.lds_query_bytes = (need <= 48 * 1024 && h.lds_query_bytes != 0) ? (uint32_t)need : 32;
        if (cw->lds_bytes) {      // multi-vector examples: always staged (the MaxSim policy reads its tokens from LDS)
            QMX_REQUIRE(cw->lds_bytes <= HNSW_LDS_QUERY_MAX, QMX_ERR_NOT_SUPPORTED, "a custom query of %u bytes of example tokens does not fit the LDS", cw->lds_bytes);
            h.lds_query_bytes = cw->lds_bytes;
        }
    }
    if (std::max(top, ef) > HNSW_MAX_EF_REG && !mw && !cw && !tq_l1(s)) {   // a list this long lives in LDS behind the query entry, which is then always staged (a PQ LUT too)
        const size_t beam = ((size_t)std::max(top, ef) * 9 + 15) / 16 * 16;
        QMX_REQUIRE((size_t)q->q_stride + beam + 2048 <= 160 * 1024, QMX_ERR_NOT_SUPPORTED,
                    "hnsw max(top, ef) = %u: the query entry (%u bytes) and the list do not fit the LDS together", std::max(top, ef), q->q_stride);
        h.lds_query_bytes = q->q_stride;
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
    uint64_t slots = std::min<uint64_t>({(uint64_t)n_searches, (uint64_t)s->num_cus * per_cu, (uint64_t)HNSW_SLOT_CAP});
    const uint64_t by_budget = std::max<uint64_t>(1, HNSW_VIS_BUDGET / (h.vis_words * 4));
    slots = std::max<uint64_t>(1, std::min(slots, by_budget));
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
    size_t slot = 0;
    if (timed) QMX_TRY(timing_begin(q, &slot));
    QMX_TRY(launch_hnsw(q, a, h, (uint32_t)slots, &per_cu));
    q->last_kernel = g_last_kernel;
    if (timed) QMX_TRY(timing_end(q, slot));
    return QMX_OK;
}

static int32_t hnsw_check(const qmx_hnsw *g, const qmx_query *q, uint32_t top, uint32_t ef) {
    QMX_REQUIRE(g->device == q->seg->device, QMX_ERR_BAD_ARG, "graph lives on device %d, the segment on %d", g->device, q->seg->device);
    QMX_REQUIRE((uint64_t)g->n_points <= q->seg->n, QMX_ERR_OUT_OF_BOUNDS, "graph has %u points, the segment %llu rows", g->n_points,
                (unsigned long long)q->seg->n);
    QMX_REQUIRE(top >= 1, QMX_ERR_BAD_ARG, "top must be > 0");
    QMX_REQUIRE(std::max(top, ef) <= HNSW_MAX_EF, QMX_ERR_NOT_SUPPORTED, "max(top, ef) = %u > %u not supported yet", std::max(top, ef),
                HNSW_MAX_EF);
    return QMX_OK;
}

static int32_t hnsw_search_sync(const qmx_hnsw *g, qmx_query *q, uint32_t top, uint32_t ef, qmx_scored_point *out, uint32_t *out_counts,
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
    if (counters) QMX_HIP(hipMemcpyAsync(scored.data(), q->hnsw_scored.p, (size_t)q->nq * 4, hipMemcpyDeviceToHost, q->stream));
    QMX_TRY(check_err_flag(q));   // synchronises the stream
    if (counters) {
        uint64_t total = 0;
        for (uint32_t v : scored) total += v;
        counters->vectors_scored = total;
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

// ---------------------------------------------------------------------------------------------
// pair scoring (ragged score_points, rescoring, score_internal)
// ---------------------------------------------------------------------------------------------
static int32_t score_pairs_device(qmx_query *q, const PairSel &sel, const uint32_t *d_ids, uint64_t n_items, float *d_scores,
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
    // a search pops about ef..2 ef candidates; the list is sized for 32 ef (or every point) and a search that pops more is reported
    const uint32_t xcap = (uint32_t)std::min<uint64_t>(g->n_points, (uint64_t)32 * beam_ef + 256);
    QMX_TRY(links->cand.reserve((size_t)nq * beam_ef * sizeof(qmx_scored_point)));
    QMX_TRY(links->cand_cnt.reserve((size_t)nq * 4));
    QMX_TRY(links->cand_ids.reserve((size_t)nq * xcap * 4));
    QMX_TRY(links->hnsw_scored.reserve((size_t)nq * 4));
    QMX_TRY(links->xcnt.reserve((size_t)nq * 4));
    ExpandedOut xo{(uint32_t *)links->cand_ids.p, (uint32_t *)links->xcnt.p, xcap};
    const bool timed = links->timing || (links->seg->flags & QMX_SEG_TIME_KERNELS) != 0;
    QMX_TRY(hnsw_enqueue(g, links, std::min(top, beam_ef), ef, (qmx_scored_point *)links->cand.p, (uint32_t *)links->cand_cnt.p,
                         (uint32_t *)links->hnsw_scored.p, timed, false, nullptr, &xo));
    std::vector<uint32_t> cnt(nq), sc(nq);
    QMX_HIP(hipMemcpyAsync(cnt.data(), links->xcnt.p, (size_t)nq * 4, hipMemcpyDeviceToHost, links->stream));
    QMX_HIP(hipMemcpyAsync(sc.data(), links->hnsw_scored.p, (size_t)nq * 4, hipMemcpyDeviceToHost, links->stream));
    QMX_TRY(check_err_flag(links));     // synchronises
    uint64_t popped = 0, scored = 0;
    for (uint32_t i = 0; i < nq; ++i) {
        QMX_REQUIRE(cnt[i] <= xcap, QMX_ERR_NOT_SUPPORTED, "search %u popped %u candidates, more than the %u the base-scoring list holds", i, cnt[i], xcap);
        popped += cnt[i];
        scored += sc[i];
    }
    // base_search_context: FixedLengthPriorityQueue(ef) over the base scores of the popped candidates, into_iter_sorted().take(top)
    QMX_TRY(qmx_rescore(base, (const uint32_t *)links->cand_ids.p, (const uint32_t *)links->xcnt.p, xcap, top, out, out_counts));
    if (counters) {
        counters->vectors_scored = scored + popped;
        counters->bytes_read = scored * links->seg->row_bytes + popped * base->seg->row_bytes;
        counters->kernel_launches = 3;
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

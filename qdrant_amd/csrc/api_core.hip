// api_core.hip — the C-ABI of include/qdrant_amd.h, library / device: errors, options, device checks.
// (One of the api_*.hip translation units; what they share: api_internal.hpp.)
#include "api_internal.hpp"

namespace qmx {

static thread_local std::string g_last_error;
const std::string &last_error_text() { return g_last_error; }

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
static const char *const g_option_names[OPT_COUNT] = {"no_mfma_scan", "no_mfma16", "no_prescan", "prescan_shift", "hnsw_log_cap", "no_split_scan", "split_min_queries", "no_split256", "no_pq_pair", "no_pq_prefilter", "pq_prefilter_min_queries", "hnsw_pq_per_cu", "tq_rotate_block", "no_topk_small", "verify_max_per_query", "hnsw_pq_direct_walk", "hnsw_pq_table_build", "hnsw_no_pq_prefilter", "hnsw_no_lds_visited", "pq_lut_no_lds", "hnsw_per_cu", "hnsw_reference_heap_order", "tq_wide_min_queries", "tq_wide_high_digit", "sq_wide_min_queries", "i8_resident", "debug"};
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
            if (i == OPT_TQ_WIDE_MIN_QUERIES && !e) val = 33;
            if (i == OPT_SQ_WIDE_MIN_QUERIES && !e) val = 33;
            if (i == OPT_I8_RESIDENT && !e) val = 1;
            initial[i] = val;
            v[i].store(val, std::memory_order_relaxed);
        }
    }
};
static OptionTable g_options;
int64_t option(Option o) { return g_options.v[o].load(std::memory_order_relaxed); }

static thread_local const void *g_last_kernel = nullptr;
const void *last_noted_kernel() { return g_last_kernel; }
void note_kernel(const void *host_function) { g_last_kernel = host_function; }

void clear_stale_error() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        const bool debug = option(OPT_DEBUG) != 0;
        if (debug) fprintf(stderr, "[qmx] dropped stale HIP error %d (%s) before a kernel launch\n", (int)e, hipGetErrorString(e));
    }
}

bool is_device_ptr(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

uint32_t elem_bytes(uint32_t dtype) {
    switch (dtype) {
        case QMX_DTYPE_F32: return 4;
        case QMX_DTYPE_F16: return 2;
        default: return 1;
    }
}



int32_t check_device(int32_t device_id, hipDeviceProp_t *prop_out) {
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


extern "C" {

// ---------------------------------------------------------------------------------------------
// library / device
// ---------------------------------------------------------------------------------------------

uint32_t qmx_abi_version(void) { return 8; }   // 8: qmx_merge_topk_packed_async / qmx_topk_record_bytes, the pruned option list of round 6

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

}  // extern "C"

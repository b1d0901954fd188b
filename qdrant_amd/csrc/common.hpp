// common.hpp — shared host/device helpers for libqdrant_amd.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <atomic>
#include <string>

#include "../../include/qdrant_amd.h"

namespace qmx {

// ---- error plumbing ------------------------------------------------------------------------
void set_error(const char *fmt, ...);
int32_t hip_status(hipError_t e, const char *what, const char *file, int line);
// hipGetLastError() reports the last error of ANY earlier runtime call of this thread (also one made
// by another library sharing the runtime); drop such a stale error before a launch so that the
// check after the launch speaks about this launch only.  QMX_DEBUG=1 prints what was dropped.
void clear_stale_error();

// ---- kernel-path options ---------------------------------------------------------------------
// Every option selects between kernels that produce the SAME results (parity tests run both sides); none changes a score.
// Read once from the environment (QMX_<NAME>) when the library is loaded, changed afterwards only through
// qmx_set_option (include/qdrant_amd.h).  Knobs that break results exist only in -DQMX_TUNING builds.
enum Option {
    OPT_NO_MFMA_SCAN = 0,     // 8+ query tiles keep the VALU scan
    OPT_NO_MFMA16,            // no chain-major 16x16x4 scan (scan_mfma16.hip)
    OPT_NO_PRESCAN,           // no threshold pre-scan in front of the top-k scans
    OPT_PRESCAN_SHIFT,        // pre-scan over n >> shift rows (default 10)
    OPT_HNSW_LOG_CAP,         // visited-word log entries per search before the whole-bitmap clear
    OPT_NO_SPLIT_SCAN,        // f32 scans of more than 64 queries keep the exact chain-major kernel (no f16-split prefilter + verification)
    OPT_SPLIT_MIN_QUERIES,    // with a derived copy of the block: batches of at least this many queries take the prefilter (default 1: all)
    OPT_NO_SPLIT256,          // batches of more than 128 queries over a half copy: keep the 128-query shape of the prefilter
    OPT_NO_PQ_PAIR,           // PQ score_internal recomputes the centroid distances instead of reading the pair table
    OPT_NO_PQ_PREFILTER,      // PQ top-k scans of 4+ queries keep the exact one-LUT-per-block kernel (no 6-bit prefilter + verification)
    OPT_PQ_PREFILTER_MIN_QUERIES,   // ... from this many queries on (default 4)
    OPT_HNSW_PQ_PER_CU,       // PQ walk with the LUT read through L2: at most this many concurrent searches per CU (0 = what fits)
    OPT_TQ_ROTATE_BLOCK,      // TurboQuant rotation by the one-block-per-vector kernel (not one wave per vector)
    OPT_NO_TOPK_SMALL,        // top-k of short score rows by the insertion kernel (not the rank-sort of the pruned row)
    OPT_VERIFY_MAX_PER_QUERY, // prefilters: a query with more rows inside its band than this takes the exact scan, whatever room the batch's pool has (0 = no such limit)
    OPT_HNSW_PQ_DIRECT_WALK,  // the PQ walk recomputes LUT entries from the codebook (pq.hip HopPQDirect: a twentieth of the HBM traffic; 1.3 x the time on a 2 M-row graph, the same at 10 M) instead of gathering per-search LUTs
    OPT_HNSW_PQ_TABLE_BUILD,  // the PQ build scores through per-insertion LUTs and the centroid pair table (round 2's build: 100 TB of table sectors per 2 M points) instead of
                              // recomputing both kinds of entries from the codebook (pq.hip HopPQDirectBuild + HopPQInternalDirect, the default where the codebook allows)
    OPT_HNSW_NO_PQ_PREFILTER, // the PQ walk scores every hop candidate exactly (rounds 1-4) instead of dropping, on an 8-bit upper bound, those the beam cannot take
    OPT_HNSW_NO_LDS_VISITED,  // the walk's visited set lives in the per-slot HBM bitmap only (rounds 1-4), not in the 16 KiB LDS table in front of it
    OPT_PQ_LUT_NO_LDS,        // the MFMA LUT build reads its operands from global memory per instruction (round 1's pq_lut_mfma_kernel) instead of staging both in LDS
    OPT_HNSW_PER_CU,          // cap on the searches resident per CU of any walk (0 = what the occupancy allows)
    OPT_HNSW_REFERENCE_HEAP_ORDER,   // the plain HNSW walk keeps `nearest` / `candidates` as the reference's two binary heaps (std sift order, one lane): the reference's lists among equal scores; slow, a verification mode
    OPT_TQ_WIDE_MIN_QUERIES,  // 4-bit TurboQuant top-k scans of at least this many queries take the 128-query pass (scan_tq4w.hip: codes decoded once per tile; default 33, 0 = never)
    OPT_TQ_WIDE_HIGH_DIGIT,   // the 128-query TurboQuant pass multiplies the queries' high digits alone (half the matrix work; the pass's scores carry a band, the
                              // survivors are re-scored exactly - the same lists) instead of both digits (exact scores in the pass).  Faster on rows whose scores
                              // spread wide against the band (iid unit rows: 2.41 against 2.55 ms per 128 queries), much slower where they crowd (clustered: 5.8 / 3.1)
    OPT_SQ_WIDE_MIN_QUERIES,  // scalar-int8 top-k scans of at least this many queries take the 128-query pass (scan_sqw.hip; default 33, 0 = never)
    OPT_I8_RESIDENT,          // 1 (default): the int8-copy prefilter over rows of up to 1 024 coordinates keeps the queries' whole operand image resident in LDS
                              // (scan_i8copy_kernel_res: no stage barrier, 97 KiB of LDS at 768 coordinates); 0 = the staged kernel of rounds 3 - 6 (scan_i8copy_kernel:
                              // 144 KiB), which longer rows take anyway
    OPT_DEBUG,                // log dropped stale HIP errors
    OPT_COUNT
};
int64_t option(Option o);
// hipFuncSetAttribute (the opt-in to more than 64 KiB of dynamic LDS) is per DEVICE: a call site remembers on which devices it ran, so that a host thread
// serving segments on several GPUs sets it on each of them (a `static thread_local bool` set it on the thread's first device only)
struct DeviceOnce {
    std::atomic<uint64_t> done{0};
    int dev = 0;
    bool need() {
        if (hipGetDevice(&dev) != hipSuccess) { dev = 0; return true; }
        return dev >= 64 || !(done.load(std::memory_order_acquire) & (1ull << dev));
    }
    void mark() { if (dev < 64) done.fetch_or(1ull << dev, std::memory_order_release); }
};
// the host-side handle of the scoring kernel this thread launched last (qmx_query_last_kernel reports its symbol)
void note_kernel(const void *host_function);
#define QMX_NOTE_KERNEL(fn) ::qmx::note_kernel(reinterpret_cast<const void *>(fn))

#define QMX_HIP(expr)                                                        \
    do {                                                                     \
        hipError_t e__ = (expr);                                             \
        if (e__ != hipSuccess) return ::qmx::hip_status(e__, #expr, __FILE__, __LINE__); \
    } while (0)

#define QMX_TRY(expr)                           \
    do {                                        \
        int32_t s__ = (expr);                   \
        if (s__ != QMX_OK) return s__;          \
    } while (0)

#define QMX_REQUIRE(cond, code, ...)            \
    do {                                        \
        if (!(cond)) {                          \
            ::qmx::set_error(__VA_ARGS__);      \
            return (code);                      \
        }                                       \
    } while (0)

// ---- device helpers --------------------------------------------------------------------------
#if defined(__HIPCC__)

constexpr int WAVE = 64;

// DPP controls (cdna4 ISA: quad_perm 0x00-0xFF, row_half_mirror 0x141, row_mirror 0x140)
constexpr int DPP_QUAD_XOR1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;  // lane i <-> 7-i inside each group of 8
constexpr int DPP_QUAD_BCAST0 = 0x00;       // quad_perm:[0,0,0,0]
constexpr int DPP_QUAD_BCAST1 = 0x55;
constexpr int DPP_QUAD_BCAST2 = 0xAA;
constexpr int DPP_QUAD_BCAST3 = 0xFF;

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane_uniform) {
    uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, lane_uniform);
    uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane_uniform);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_up1_u64(uint64_t v) {
    uint32_t lo = __shfl_up((int)(uint32_t)v, 1, 64);
    uint32_t hi = __shfl_up((int)(uint32_t)(v >> 32), 1, 64);
    return ((uint64_t)hi << 32) | lo;
}

// Total order of ScoredPointOffset for the bounded queue, as one u64 (bigger key = better):
//   high 32 bits: f32 score mapped monotonically to u32, NaN greatest (OrderedFloat,
//                 lib/common/common/src/types.rs:21-25);
//   low 32 bits : ~idx, so that among equal scores the LOWER offset wins — the element a linear
//                 scan pushes first survives in FixedLengthPriorityQueue::push (strict `<`,
//                 fixed_length_priority_queue.rs:53-57).
// key 0 is never produced by a real point (idx 0xFFFFFFFF is not a valid offset): "empty slot".
__device__ __forceinline__ uint32_t score_to_ord(float s) {
    uint32_t b = __float_as_uint(s);
    if (s != s) return 0xFFFFFFFFu;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_to_score(uint32_t o) {
    uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(b);
}
__device__ __forceinline__ uint64_t make_key(float s, uint32_t idx) {
    return ((uint64_t)score_to_ord(s) << 32) | (uint32_t)(~idx);
}
__device__ __forceinline__ uint32_t key_idx(uint64_t k) { return ~(uint32_t)k; }
__device__ __forceinline__ float key_score(uint64_t k) { return ord_to_score((uint32_t)(k >> 32)); }

// BitSlice<u64, Lsb0> (lib/common/common/src/bitvec.rs:6-7)
__device__ __forceinline__ bool bit_get(const uint64_t *bits, uint64_t i) {
    return (bits[i >> 6] >> (i & 63)) & 1ull;
}
// NotDeletedChecker::check (lib/segment/src/vector_storage/raw_scorer.rs:596-603)
// + the optional payload filter of ScorerFilters (hnsw_index/point_scorer.rs:78-85, 160-181: `filters.check_vector(id)`
//   = not deleted AND filter_context.check(id)), given as an allow bitmap evaluated by the caller's payload index
struct DeletedView {
    const uint64_t *point_deleted;
    uint64_t n_point_bits;
    const uint64_t *vec_deleted;
    uint64_t n_vec_bits;
    uint64_t n_rows;
    const uint64_t *allowed;     // nullptr = no payload filter; ids past n_allowed_bits are rejected
    uint64_t n_allowed_bits;
    __device__ __forceinline__ bool live(uint32_t id) const {
        bool vdel = (vec_deleted && id < n_vec_bits) ? bit_get(vec_deleted, id) : false;
        bool pdel = point_deleted ? (id < n_point_bits ? bit_get(point_deleted, id) : true)
                                  : !(id < n_rows);
        bool ok = allowed ? (id < n_allowed_bits && bit_get(allowed, id)) : true;
        return !vdel && !pdel && ok;
    }
};

// One register-resident bounded list per wave: lane i holds the i-th best key (descending).
// insert() is wave-uniform; returns nothing, the k-th best is readlane(list, top-1).
__device__ __forceinline__ void wave_list_insert(uint64_t &list, uint64_t nk, int lane) {
    uint64_t gt = __ballot(list > nk);
    int p = __popcll(gt);
    uint64_t up = shfl_up1_u64(list);
    list = lane < p ? list : (lane == p ? nk : up);
}

#endif  // __HIPCC__
}  // namespace qmx

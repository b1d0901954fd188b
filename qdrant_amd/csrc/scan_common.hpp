// scan_common.hpp — the wave64 row-streaming kernel shared by every stored element type.
//
// Replaces the reference's innermost loops
//   BatchFilteredSearcher::peek_top_iter   lib/segment/src/index/hnsw_index/point_scorer.rs:423-472
//   RawScorerImpl::score_points            lib/segment/src/vector_storage/raw_scorer.rs:561-564
//   MetricQueryScorer::score_stored_batch  lib/segment/src/vector_storage/query_scorer/metric_query_scorer.rs:81-92
// for a device-resident copy of the stored block.
//
// Mapping (HBM-bound, VALU only — no MFMA: intensity is 0.5*Q flop/B):
//   * 8 lanes own one row per step: each lane loads one 16-byte piece of a 128-byte row segment,
//     so one wave-level `global_load_dwordx4` touches 8 rows x one full 128-B cache line each
//     (8 lines per instruction, exactly like a contiguous 1-KiB access) and every fetched byte
//     is used once.  R row-groups per lane keep R x UNROLL loads in flight per wave.
//   * the query tile (QT preprocessed queries) sits in LDS; the 8 row-groups of a wave read the
//     same 128 bytes (LDS broadcast), one ds_read_b128 per query per segment, reused for R rows.
//   * the lane->piece map is chosen so that a lane's NACC accumulators are exactly the SIMD lanes
//     of the reference's AVX registers; the cross-lane sum is three DPP steps in the reference's
//     own hsum order  =>  scores are bit-identical to the x86 reference, not just within 1e-5.
//   * per-wave top-k lives in registers (lane i = i-th best, u64 keys); a row only reaches the
//     insert path when it beats the wave's current k-th best (one v_cmp + ballot per row group).
#pragma once
#include "kernels.hpp"

namespace qmx {

// lane position inside its group of 8 -> which 16-byte piece of the 128-byte segment it loads.
// t = 4*h + r  ->  piece = 2*r + h : a quad (4 consecutive lanes) holds one half (h) of every AVX
// register r = 0..3, so "register a+b, c+d, (a+b)+(c+d)" are quad_perm DPPs and "high half + low
// half" is row_half_mirror.
__device__ __forceinline__ int lane_piece(int t) { return 2 * (t & 3) + (t >> 2); }

// A policy whose query entry holds QPIECES 16-byte pieces per 16-byte row piece (binary quantization with a scalar-encoded
// query: QPIECES bit planes per u128 word, encoded_vectors_binary.rs:721-756) says so with `static constexpr int QPIECES = n`
// and takes them in `mac_pieces`; everything else has one query piece per row piece.
template <class P, class = void> struct query_pieces { static constexpr int value = 1; };
template <class P> struct query_pieces<P, decltype((void)P::QPIECES)> { static constexpr int value = P::QPIECES; };

// a policy with `typedef ... dec_t`, `decode(v, dec)` and `mac_decoded(acc, q, dec)` has its row pieces decoded once per step (scan_tq.hip)
template <class P, class = void> struct has_row_decode { static constexpr bool value = false; };
template <class P> struct has_row_decode<P, decltype((void)sizeof(typename P::dec_t))> { static constexpr bool value = true; };

// one 128-byte segment step: the lane's 16-byte row pieces (R rows) against every query of the tile; the lane's query piece(s)
// sit at byte q_off (x QPIECES) of each query entry
template <class P, int QT, int R>
__device__ __forceinline__ void scan_step(typename P::acc_t (&acc)[QT][R][P::NACC],
                                          typename P::acc_t (&raux)[R][P::NRAUX > 0 ? P::NRAUX : 1],
                                          const uint4 (&v)[R], const unsigned char *q_base, uint32_t q_off, uint32_t q_stride,
                                          bool lane_on) {
    constexpr int QP = query_pieces<P>::value;
    if (P::NRAUX > 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) P::row_aux(raux[r], v[r]);
    }
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        if constexpr (QP == 1) {
            uint4 qv = *reinterpret_cast<const uint4 *>(q_base + q_off + (uint32_t)q * q_stride);
            if (!lane_on) qv = make_uint4(0, 0, 0, 0);   // partial segment: the scalar-tail elements sit right behind the body
#pragma unroll
            for (int r = 0; r < R; ++r) P::mac(acc[q][r], qv, v[r]);
        } else if constexpr (!has_row_decode<P>::value) {
            uint4 qv[QP];
#pragma unroll
            for (int k = 0; k < QP; ++k) {
                qv[k] = *reinterpret_cast<const uint4 *>(q_base + q_off * QP + k * 16 + (uint32_t)q * q_stride);
                if (!lane_on) qv[k] = make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) P::mac_pieces(acc[q][r], qv, v[r]);
        }
    }
    if constexpr (has_row_decode<P>::value) {
        // rows whose pieces are DECODED before they meet a query (TurboQuant codes -> codebook bytes): once per row piece, not once per query
        typename P::dec_t dec[R];
#pragma unroll
        for (int r = 0; r < R; ++r) P::decode(v[r], dec[r]);
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            uint4 qv[QP];
#pragma unroll
            for (int k = 0; k < QP; ++k) {
                qv[k] = *reinterpret_cast<const uint4 *>(q_base + q_off * QP + k * 16 + (uint32_t)q * q_stride);
                if (!lane_on) qv[k] = make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) P::mac_decoded(acc[q][r], qv, dec[r]);
        }
    }
}

// a 16-byte piece of a stored row, streamed once: nontemporal, so the lines do not push the query tile / partial lists out of L2
// (not for row types whose rows end inside a line as a rule - a policy says so with `static constexpr bool TEMPORAL_ROWS = true` -:
// such a row shares its last line with the next row's first load, which must still find it in cache; measured on 192-byte BQ
// rows: 0.34 ms temporal, 0.49 ms nontemporal.  The choice is per policy, not per launch: a run-time branch around the load kind
// costs the f32 scan half its bandwidth.)
template <class P, class = void> struct rows_temporal { static constexpr bool value = false; };
template <class P> struct rows_temporal<P, decltype((void)P::TEMPORAL_ROWS)> { static constexpr bool value = P::TEMPORAL_ROWS; };
template <bool NT>
__device__ __forceinline__ uint4 load_row_piece(const unsigned char *p) {
    if (!NT) return *reinterpret_cast<const uint4 *>(p);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}

template <class P, int QT, int R, int UNROLL, bool HAS_IDS, int MODE>
__global__ __launch_bounds__(SCAN_BLOCK) void scan_kernel(const ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = SCAN_BLOCK / WAVE;
    constexpr int NRA = P::NRAUX > 0 ? P::NRAUX : 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- stage the query tile in LDS (once per block) ----
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.queries);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const uint32_t n16 = (uint32_t)QT * a.q_stride / 16;
        for (uint32_t i = tid; i < n16; i += SCAN_BLOCK) dst[i] = src[i];
    }
    __syncthreads();

    const int t = lane & 7;
    const int g = lane >> 3;
    const int piece = lane_piece(t);
    const int piece_off = piece * 16;
    const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);

    uint64_t list[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) list[q] = 0;

    const uint32_t gw = blockIdx.x * NW + wave;
    const uint32_t tw = gridDim.x * NW;
    constexpr uint32_t TILE = 8 * R;
    const uint64_t n_tiles = (a.n_cand + TILE - 1) / TILE;
    const uint32_t nseg = a.nseg;
    const bool piece_in_rem = piece < (int)a.rem_pieces;

    for (uint64_t tile = gw; tile < n_tiles; tile += tw) {
        uint32_t rid[R];
        bool valid[R];
        uint64_t cand[R];
        const unsigned char *rp[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint64_t c = tile * TILE + (uint32_t)(r * 8 + g);
            valid[r] = c < a.n_cand;
            cand[r] = c;
            uint64_t cc = valid[r] ? c : 0;
            uint32_t id = HAS_IDS ? a.ids[cc] : (uint32_t)cc;
            if (HAS_IDS && id >= a.n_rows) {
                if (valid[r]) *a.err_flag = 1;
                id = 0;
                valid[r] = false;
            }
            rid[r] = id;
            rp[r] = rows + (uint64_t)id * a.row_stride + piece_off;
        }

        typename P::acc_t acc[QT][R][P::NACC];
        typename P::acc_t raux[R][NRA];
#pragma unroll
        for (int q = 0; q < QT; ++q)
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int k = 0; k < P::NACC; ++k) acc[q][r][k] = 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < NRA; ++k) raux[r][k] = 0;

#pragma unroll UNROLL
        for (uint32_t s = 0; s < nseg; ++s) {
            uint4 v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = load_row_piece<!rows_temporal<P>::value>(rp[r] + (uint64_t)s * 128);
            scan_step<P, QT, R>(acc, raux, v, smem, s * 128 + piece_off, a.q_stride, true);
        }
        if (a.rem_pieces) {
            // last, partial segment of the SIMD body (element types narrower than f32): pieces past
            // it are zero on both sides (the query tile is zero-padded), rows are never read past
            uint4 v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                v[r] = make_uint4(0, 0, 0, 0);
                if (piece_in_rem) v[r] = *reinterpret_cast<const uint4 *>(rp[r] + (uint64_t)nseg * 128);
            }
            scan_step<P, QT, R>(acc, raux, v, smem, nseg * 128 + piece_off, a.q_stride, piece_in_rem);
        }

#pragma unroll
        for (int q = 0; q < QT; ++q) {
            if (q < (int)a.nq) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float score = P::finish(acc[q][r], raux[r], smem + (uint32_t)q * a.q_stride,
                                                  rows + (uint64_t)rid[r] * a.row_stride, rid[r], a);
                    if (MODE == SCAN_SCORES) {
                        if (valid[r] && t == 0) a.scores[(uint64_t)q * a.scores_stride + cand[r]] = score;
                    } else {
                        const uint64_t key = make_key(score, rid[r]);
                        const uint64_t thr = readlane_u64(list[q], (int)a.top - 1);
                        bool c = valid[r] && (t == 0) && (key > thr);
                        if (__ballot(c)) {
                            c = c && a.del.live(rid[r]) && (!a.key_bound || key < a.key_bound[q]);
                            uint64_t m = __ballot(c);
                            while (m) {
                                const int src = __builtin_ctzll(m);
                                m &= m - 1;
                                const uint64_t nk = readlane_u64(key, src);
                                if (nk > readlane_u64(list[q], (int)a.top - 1))
                                    wave_list_insert(list[q], nk, lane);
                            }
                        }
                    }
                }
            }
        }
    }

    if (MODE == SCAN_SCORES) return;

    // ---- block merge: 8 wave lists -> 1 list per query, then one global write per block ----
    __syncthreads();  // everyone is done reading the query tile
    uint64_t *lds_keys = reinterpret_cast<uint64_t *>(smem);
    const uint32_t top = a.top;
#pragma unroll
    for (int q = 0; q < QT; ++q)
        if (lane < (int)top) lds_keys[((uint32_t)wave * QT + q) * top + lane] = list[q];
    __syncthreads();
    for (uint32_t q = wave; q < a.nq; q += NW) {
        uint64_t merged = 0;
        for (int sw = 0; sw < NW; ++sw) {
            const uint64_t key = lane < (int)top ? lds_keys[((uint32_t)sw * QT + q) * top + lane] : 0;
            uint64_t m = __ballot(key > readlane_u64(merged, (int)top - 1));
            while (m) {
                const int src = __builtin_ctzll(m);
                m &= m - 1;
                const uint64_t nk = readlane_u64(key, src);
                if (nk > readlane_u64(merged, (int)top - 1)) wave_list_insert(merged, nk, lane);
            }
        }
        if (lane < (int)top) a.partial[((uint64_t)blockIdx.x * QT + q) * top + lane] = merged;
    }
}

// Launch helper: picks the grid from the occupancy of this instantiation.
template <class P, int QT, int R, int UNROLL, bool HAS_IDS, int MODE>
int32_t launch_scan_inst(hipStream_t st, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    constexpr int NW = SCAN_BLOCK / WAVE;
    size_t lds = (size_t)QT * a.q_stride;
    if (MODE == SCAN_TOPK) {
        size_t lk = (size_t)NW * QT * a.top * sizeof(uint64_t);
        if (lk > lds) lds = lk;
    }
    lds = (lds + 15) & ~(size_t)15;
    QMX_REQUIRE(lds <= 160 * 1024, QMX_ERR_NOT_SUPPORTED, "query tile needs %zu B of LDS (> 160 KiB)", lds);
    auto kfn = scan_kernel<P, QT, R, UNROLL, HAS_IDS, MODE>;
    static thread_local DeviceOnce attr_once;
    if (attr_once.need()) {
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.mark();
    }
    int per_cu = 0;
    QMX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, SCAN_BLOCK, lds));
    if (per_cu < 1) per_cu = 1;
    const uint64_t n_tiles = (a.n_cand + 8 * R - 1) / (8 * R);
    uint64_t want = (n_tiles + NW - 1) / NW;
    uint64_t cap = (uint64_t)num_cus * per_cu;
    uint32_t grid = (uint32_t)(want < cap ? want : cap);
    if (grid < 1) grid = 1;
    if (grid_out) {
        if (*grid_out && MODE == SCAN_TOPK && grid > *grid_out) grid = *grid_out;  // caller's partial buffer bound
        *grid_out = grid;
    }
    ::qmx::clear_stale_error();
    QMX_NOTE_KERNEL(kfn);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(SCAN_BLOCK), lds, st, a);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

template <class P, int QT, int R, int U>
int32_t launch_qt(hipStream_t st, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid) {
    const bool ids = a.ids != nullptr;
    if (mode == SCAN_TOPK) {
        return ids ? launch_scan_inst<P, QT, R, U, true, SCAN_TOPK>(st, a, num_cus, grid)
                   : launch_scan_inst<P, QT, R, U, false, SCAN_TOPK>(st, a, num_cus, grid);
    }
    return ids ? launch_scan_inst<P, QT, R, U, true, SCAN_SCORES>(st, a, num_cus, grid)
               : launch_scan_inst<P, QT, R, U, false, SCAN_SCORES>(st, a, num_cus, grid);
}

// Policies whose batches of 4 and more queries run on the matrix cores (TurboQuant: scan_sq_mfma.hip) say MAX_VALU_QT = 4: their 8- and 16-query
// tiled VALU kernels (the largest instantiations of this file's kernel, and the slowest to compile) are not built; api_*.hip tiles by 4 for them
// when the matrix-core route is switched off.
template <class P, class = void>
struct max_valu_qt { static constexpr int value = 16; };
template <class P>
struct max_valu_qt<P, decltype((void)P::MAX_VALU_QT)> { static constexpr int value = P::MAX_VALU_QT; };

template <class P>
int32_t launch_policy(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid) {
    switch (qt) {
        case 1: return launch_qt<P, 1, 4, 4>(st, mode, a, num_cus, grid);
        case 2: return launch_qt<P, 2, 4, 2>(st, mode, a, num_cus, grid);
        case 4: return launch_qt<P, 4, 2, 4>(st, mode, a, num_cus, grid);
        case 8:
            if constexpr (max_valu_qt<P>::value >= 8) return launch_qt<P, 8, 2, 2>(st, mode, a, num_cus, grid);
            break;
        case 16:
            if constexpr (max_valu_qt<P>::value >= 16) return launch_qt<P, 16, P::R16, 2>(st, mode, a, num_cus, grid);
            break;
    }
    set_error("unsupported query tile %d", qt);
    return QMX_ERR_BAD_ARG;
}

// 8-lane reductions of the per-lane partials of one row (DPP only, no LDS)
__device__ __forceinline__ float reduce8_f32(float x) {
    x += dpp_f32<DPP_QUAD_XOR1>(x);
    x += dpp_f32<DPP_QUAD_XOR2>(x);
    return x + dpp_f32<DPP_ROW_HALF_MIRROR>(x);
}
__device__ __forceinline__ uint32_t reduce4_u32(uint32_t x) {   // the 4 lanes of a quad
    x += (uint32_t)dpp_i32<DPP_QUAD_XOR1>((int)x);
    return x + (uint32_t)dpp_i32<DPP_QUAD_XOR2>((int)x);
}
__device__ __forceinline__ uint32_t reduce8_u32(uint32_t x) {
    x = reduce4_u32(x);
    return x + (uint32_t)dpp_i32<DPP_ROW_HALF_MIRROR>((int)x);
}

// ------------------------------------------------------------------------------------------
// Small rows (below the reference's AVX threshold, spaces/simple.rs:15): one lane per candidate,
// the policy S restates the reference's SSE / scalar leaf in a single thread, so scores stay
// bit-identical.  blockIdx.y = query; rows this short are a few cache lines, re-reading them per
// query costs nothing that matters.
// ------------------------------------------------------------------------------------------
constexpr int SMALL_BLOCK = 256;

template <class S, bool HAS_IDS, int MODE>
__global__ __launch_bounds__(SMALL_BLOCK) void scan_small_kernel(const ScanArgs a) {
    constexpr int NW = SMALL_BLOCK / WAVE;
    __shared__ uint64_t sh[NW][WAVE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t q = blockIdx.y;
    const unsigned char *qp = reinterpret_cast<const unsigned char *>(a.queries) + (uint64_t)q * a.q_stride;
    const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
    const int top = (int)a.top;
    uint64_t list = 0;
    for (uint64_t base = ((uint64_t)blockIdx.x * NW + wave) * WAVE; base < a.n_cand;
         base += (uint64_t)gridDim.x * NW * WAVE) {
        const uint64_t c = base + lane;
        bool valid = c < a.n_cand;
        uint32_t id = HAS_IDS ? a.ids[valid ? c : 0] : (uint32_t)c;
        if (HAS_IDS && valid && id >= a.n_rows) {
            *a.err_flag = 1;
            valid = false;
        }
        if (!valid) id = 0;
        const float score = a.n_rows ? S::score(qp, rows + (uint64_t)id * a.row_stride, id, a) : 0.0f;
        if (MODE == SCAN_SCORES) {
            if (valid) a.scores[(uint64_t)q * a.scores_stride + c] = score;
        } else {
            const uint64_t key = make_key(score, id);
            bool cnd = valid && key > readlane_u64(list, top - 1);
            if (__ballot(cnd)) {
                cnd = cnd && a.del.live(id) && (!a.key_bound || key < a.key_bound[q]);
                uint64_t m = __ballot(cnd);
                while (m) {
                    const int src = __builtin_ctzll(m);
                    m &= m - 1;
                    const uint64_t nk = readlane_u64(key, src);
                    if (nk > readlane_u64(list, top - 1)) wave_list_insert(list, nk, lane);
                }
            }
        }
    }
    if (MODE == SCAN_SCORES) return;
    sh[wave][lane] = list;
    __syncthreads();
    if (wave == 0) {
        uint64_t merged = sh[0][lane];
        for (int w = 1; w < NW; ++w) {
            const uint64_t key = sh[w][lane];
            uint64_t m = __ballot(key > readlane_u64(merged, top - 1));
            while (m) {
                const int src = __builtin_ctzll(m);
                m &= m - 1;
                const uint64_t nk = readlane_u64(key, src);
                if (nk > readlane_u64(merged, top - 1)) wave_list_insert(merged, nk, lane);
            }
        }
        if (lane < top) a.partial[((uint64_t)blockIdx.x * a.partial_qt + q) * top + lane] = merged;
    }
}

template <class S, bool HAS_IDS, int MODE>
int32_t launch_small_inst(hipStream_t st, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    uint64_t want = (a.n_cand + SMALL_BLOCK - 1) / SMALL_BLOCK;
    uint64_t cap = (uint64_t)num_cus * 8;
    uint32_t grid = (uint32_t)(want < cap ? want : cap);
    if (grid < 1) grid = 1;
    if (grid_out) {
        if (*grid_out && MODE == SCAN_TOPK && grid > *grid_out) grid = *grid_out;
        *grid_out = grid;
    }
    ::qmx::clear_stale_error();
    QMX_NOTE_KERNEL((scan_small_kernel<S, HAS_IDS, MODE>));
    hipLaunchKernelGGL((scan_small_kernel<S, HAS_IDS, MODE>), dim3(grid, a.nq), dim3(SMALL_BLOCK), 0, st, a);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
template <class S>
int32_t launch_small(hipStream_t st, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid) {
    const bool ids = a.ids != nullptr;
    if (mode == SCAN_TOPK)
        return ids ? launch_small_inst<S, true, SCAN_TOPK>(st, a, num_cus, grid)
                   : launch_small_inst<S, false, SCAN_TOPK>(st, a, num_cus, grid);
    return ids ? launch_small_inst<S, true, SCAN_SCORES>(st, a, num_cus, grid)
               : launch_small_inst<S, false, SCAN_SCORES>(st, a, num_cus, grid);
}

// ------------------------------------------------------------------------------------------
// Pair scoring: item j = (query qsel[j], stored row ids[j]) -> scores[j].  Serves the ragged
// RawScorer::score_points of HNSW hops batched across searches (graph_layers.rs:125-139,305-313),
// the rescoring pass (vector_index_search_common.rs:73-90) and score_internal.  One 8-lane group
// per item with the same lane policies as the scan (so the same bits); the query piece comes
// from the query tile in HBM/L2 instead of LDS.
// ------------------------------------------------------------------------------------------
constexpr int PAIR_BLOCK = 256;

// score of one stored row against one query by the 8 lanes of a group (t = lane & 7); the result is
// valid in every lane of the group.  `qp` = the query's tile entry (elements, zero padding, aux) in
// LDS or in global memory.  Shared by pair_kernel and the HNSW hop scorer, so both produce the
// bits of the scan.
// (The step loops below run in batches of UNROLL steps whose loads are ALL issued before the first is used, whatever the row length: a plain
// `#pragma unroll UNROLL` over a runtime trip count leaves the remainder - every step of a row shorter than UNROLL steps, the last two of an SQ row of
// 768 bytes at UNROLL 4 - as a loop of dependent round trips.  Steps past the row re-read its last step - a valid address, the value unused - so the
// loads stay straight-line code.)
template <class P, int UNROLL = 8>      // UNROLL row pieces (and their query pieces) in flight per lane
__device__ __forceinline__ float group_score(const ScanArgs &a, const unsigned char *qp, uint32_t id, int t) {
    constexpr int NRA = P::NRAUX > 0 ? P::NRAUX : 1;
    const int piece = lane_piece(t);
    const int piece_off = piece * 16;
    const bool piece_in_rem = piece < (int)a.rem_pieces;
    const unsigned char *row = reinterpret_cast<const unsigned char *>(a.rows) + (uint64_t)id * a.row_stride;
    const unsigned char *rp = row + piece_off;
    typename P::acc_t acc[1][1][P::NACC];
    typename P::acc_t raux[1][NRA];
#pragma unroll
    for (int k = 0; k < P::NACC; ++k) acc[0][0][k] = 0;
#pragma unroll
    for (int k = 0; k < NRA; ++k) raux[0][k] = 0;
    const uint32_t nseg = a.nseg;
    for (uint32_t s0 = 0; s0 < nseg; s0 += UNROLL) {
        uint4 v[UNROLL][1];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t s = s0 + (uint32_t)u < nseg ? s0 + (uint32_t)u : nseg - 1;
            v[u][0] = *reinterpret_cast<const uint4 *>(rp + (uint64_t)s * 128);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            if (s0 + (uint32_t)u < nseg) scan_step<P, 1, 1>(acc, raux, v[u], qp, (s0 + (uint32_t)u) * 128 + piece_off, 0, true);
    }
    if (a.rem_pieces) {
        uint4 v[1];
        v[0] = make_uint4(0, 0, 0, 0);
        if (piece_in_rem) v[0] = *reinterpret_cast<const uint4 *>(rp + (uint64_t)a.nseg * 128);
        scan_step<P, 1, 1>(acc, raux, v, qp, a.nseg * 128 + piece_off, 0, piece_in_rem);
    }
    return P::finish(acc[0][0], raux[0], qp, row, id, a);
}

// R rows of one 8-lane group against one query in a single pass: R x UNROLL row pieces in flight per lane instead of
// UNROLL (the HNSW hop scores <= m0 rows and is bound by the round-trip time of its gathers).  Same accumulators,
// same order per row as group_score: same bits.
template <class P, int R, int U4 = 4>
__device__ __forceinline__ void group_score_multi(const ScanArgs &a, const unsigned char *qp, const uint32_t (&ids)[R], int t, float (&out)[R]) {
    constexpr int NRA = P::NRAUX > 0 ? P::NRAUX : 1;
    // steps in flight per row: R = 4 rows x U4 steps (16 pieces per lane at U4 = 4: the SQ walk then holds 146 registers = 3 waves per SIMD; 12 at U4 = 3: 116 =
    // 4 waves, and 6-step rows - d = 768 codes - load no clamped duplicate; measured in profiles/r5_sq_walk_*), R = 2 rows x 6 steps
    constexpr int U = R >= 4 ? U4 : 6;
    const int piece = lane_piece(t);
    const int piece_off = piece * 16;
    const bool piece_in_rem = piece < (int)a.rem_pieces;
    const unsigned char *rp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rp[r] = reinterpret_cast<const unsigned char *>(a.rows) + (uint64_t)ids[r] * a.row_stride + piece_off;
    typename P::acc_t acc[1][R][P::NACC];
    typename P::acc_t raux[R][NRA];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int k = 0; k < P::NACC; ++k) acc[0][r][k] = 0;
#pragma unroll
        for (int k = 0; k < NRA; ++k) raux[r][k] = 0;
    }
    const uint32_t nseg = a.nseg;
    for (uint32_t s0 = 0; s0 < nseg; s0 += U) {
        uint4 v[U][R];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t s = s0 + (uint32_t)u < nseg ? s0 + (uint32_t)u : nseg - 1;
#pragma unroll
            for (int r = 0; r < R; ++r) v[u][r] = *reinterpret_cast<const uint4 *>(rp[r] + (uint64_t)s * 128);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (s0 + (uint32_t)u < nseg) scan_step<P, 1, R>(acc, raux, v[u], qp, (s0 + (uint32_t)u) * 128 + piece_off, 0, true);
    }
    if (a.rem_pieces) {
        uint4 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            v[r] = make_uint4(0, 0, 0, 0);
            if (piece_in_rem) v[r] = *reinterpret_cast<const uint4 *>(rp[r] + (uint64_t)a.nseg * 128);
        }
        scan_step<P, 1, R>(acc, raux, v, qp, a.nseg * 128 + piece_off, 0, piece_in_rem);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        out[r] = P::finish(acc[0][r], raux[r], qp, reinterpret_cast<const unsigned char *>(a.rows) + (uint64_t)ids[r] * a.row_stride, ids[r], a);
}

template <class P>
__global__ __launch_bounds__(PAIR_BLOCK) void pair_kernel(const ScanArgs a, const PairSel sel, uint64_t n_items) {
    constexpr int NW = PAIR_BLOCK / WAVE;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = lane & 7, g = lane >> 3;
    const unsigned char *queries = reinterpret_cast<const unsigned char *>(a.queries);
    // Two shapes of the same loop.  gridDim.y == 1: the items as one sequence, strided over the grid.  gridDim.y > 1 (per-query slots with counts -
    // the verification lists of the prefilters, sized for the worst case): blockIdx.y owns a run of slots, its blocks walk the LIVE head of each
    // (count entries), so a slot of 16 384 entries that holds 100 costs what 100 entries cost.
    const bool lists = gridDim.y > 1;
    const uint64_t slots = lists ? n_items / sel.per_query : 1;
    for (uint64_t slot = lists ? blockIdx.y : 0; slot < slots; slot += gridDim.y) {
        const uint64_t first = lists ? slot * sel.per_query : 0;
        const uint64_t n_live = sel.limit ? (n_items < (uint64_t)*sel.limit ? n_items : (uint64_t)*sel.limit) : n_items;      // (a pool filled on the device: its count)
        const uint64_t end = lists ? first + (sel.counts[slot] < sel.per_query ? sel.counts[slot] : sel.per_query) : n_live;
        for (uint64_t base = first + ((uint64_t)blockIdx.x * NW + wave) * 8; base < end; base += (uint64_t)gridDim.x * NW * 8) {
            const uint64_t item = base + g;
            bool valid = item < end;
            uint32_t qi = valid ? sel.query_of(item) : 0;
            valid = valid && sel.live(item, qi);
            uint32_t id = valid ? a.ids[item] : 0;
            if (valid && (id >= a.n_rows || qi >= a.nq)) {
                *a.err_flag = 1;
                valid = false;
                id = 0;
                qi = 0;
            }
            // (a slot's dead tail - per-query lists are sized for the worst case, counts[] says how much is live - costs nothing: a wave whose eight items are all dead moves on)
            if (!__ballot(valid)) continue;
            // (a pair is one gathered row: the kernel is a chain of round trips, so twelve row pieces per lane are requested at once)
            const float score = group_score<P, 12>(a, queries + (uint64_t)qi * a.q_stride, id, t);
            if (valid && t == 0) a.scores[item] = score;
        }
    }
}

template <class S>
__global__ __launch_bounds__(PAIR_BLOCK) void pair_small_kernel(const ScanArgs a, const PairSel sel, uint64_t n_items) {
    const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
    const unsigned char *queries = reinterpret_cast<const unsigned char *>(a.queries);
    for (uint64_t item = (uint64_t)blockIdx.x * PAIR_BLOCK + threadIdx.x; item < n_items; item += (uint64_t)gridDim.x * PAIR_BLOCK) {
        const uint32_t qi = sel.query_of(item);
        if (!sel.live(item, qi)) continue;
        const uint32_t id = a.ids[item];
        if (id >= a.n_rows || qi >= a.nq) {
            *a.err_flag = 1;
            continue;
        }
        a.scores[item] = S::score(queries + (uint64_t)qi * a.q_stride, rows + (uint64_t)id * a.row_stride, id, a);
    }
}

// Launch functors handed to the per-dtype dispatchers: one switch over (dtype, distance) serves the
// tiled scan and the pair kernel.
struct ScanLauncher {
    hipStream_t st;
    int qt;
    ScanMode mode;
    int num_cus;
    uint32_t *grid;
    template <class P> int32_t row(const ScanArgs &a) const { return launch_policy<P>(st, qt, mode, a, num_cus, grid); }
    template <class S> int32_t small(const ScanArgs &a) const { return launch_small<S>(st, mode, a, num_cus, grid); }
};
struct PairLauncher {
    hipStream_t st;
    PairSel sel;
    uint64_t n_items;
    int num_cus;
    template <class P> int32_t row(const ScanArgs &a) const {
        if (n_items == 0) return QMX_OK;
        uint64_t want = (n_items + 31) / 32;
        uint32_t grid = (uint32_t)(want < (uint64_t)num_cus * 8 ? want : (uint64_t)num_cus * 8);
        dim3 g3(grid);
        if (!sel.qsel && sel.per_query >= 1024 && sel.counts && n_items % sel.per_query == 0 && n_items / sel.per_query >= 2) {
            // long slots with counts: blocks per slot (32 items per block trip), rows of the grid over the slots
            const uint64_t slots = n_items / sel.per_query;
            const uint32_t gy = (uint32_t)(slots < 1024 ? slots : 1024);
            const uint32_t per_slot = (uint32_t)std::min<uint64_t>((sel.per_query + 31) / 32, std::max<uint64_t>(1, (uint64_t)num_cus * 16 / gy));
            g3 = dim3(per_slot, gy);
        }
        ::qmx::clear_stale_error();
        hipLaunchKernelGGL((pair_kernel<P>), g3, dim3(PAIR_BLOCK), 0, st, a, sel, n_items);
        QMX_HIP(hipGetLastError());
        return QMX_OK;
    }
    template <class S> int32_t small(const ScanArgs &a) const {
        if (n_items == 0) return QMX_OK;
        uint64_t want = (n_items + PAIR_BLOCK - 1) / PAIR_BLOCK;
        uint32_t grid = (uint32_t)(want < (uint64_t)num_cus * 8 ? want : (uint64_t)num_cus * 8);
        ::qmx::clear_stale_error();
        hipLaunchKernelGGL((pair_small_kernel<S>), dim3(grid), dim3(PAIR_BLOCK), 0, st, a, sel, n_items);
        QMX_HIP(hipGetLastError());
        return QMX_OK;
    }
};

}  // namespace qmx

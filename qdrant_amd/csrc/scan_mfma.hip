// scan_mfma.hip — f32 dot / cosine brute-force scan for LARGE query tiles (8..32 queries per pass of the
// stored block) on the f32 matrix cores, with the bits of the x86 AVX2+FMA reference.
//
// Same reference loops as scan_common.hpp (BatchFilteredSearcher::peek_top_iter,
// lib/segment/src/index/hnsw_index/point_scorer.rs:423-472 over dot_similarity_avx,
// lib/segment/src/spaces/simple_avx.rs:167-213).  With >= 8 queries per pass the scan is a dense
// contraction [rows x dim] . [dim x Q]; the VALU kernel is LDS/VALU-bound there (Q = 16: 12.5 ms per
// 30.72 GB scan) while the HBM floor is ~4.9 ms.
//
// Why this is still bit-exact.  dot_similarity_avx keeps 32 independent fma chains per (row, query):
// chain c = 8 r + j (AVX register r, SIMD lane j) accumulates element 32 i + c of every 32-float step i in
// order, then folds the chains with four_way_hsum + hsum256_ps_avx (simple_avx.rs:10-28).
// v_mfma_f32_4x4x1_16b_f32 runs 16 independent 4x4 outer-product blocks with K = 1, i.e. ONE fused
// multiply-add per output element per instruction: D[blk][m][n] = fma(A[blk][m], B[blk][n], C[blk][m][n]).
// Mapping block -> chain, m -> stored row, n -> query, and issuing the instruction once per 32-float step
// makes every accumulator element exactly one AVX chain.  The fold is done with the reference's tree.
//
// Lane roles.  lane = x + 4 u0 + 8 u2 + 16 u1 + 32 rh  (MFMA block = lane >> 2):
//   u  = u0 + 2 u1 + 4 u2 = 0..7  which 16-byte piece of the 128-byte step the lane loads (columns 4u .. 4u+3)
//   rh = 0..1  which half of the 8-row tile (rows 4 rh .. 4 rh + 3)
//   x  = 0..3  A operand: stored row 4 rh + x;   B operand: query 4 g + x  (g = query group)
// One global_load_dwordx4 per lane per step feeds 4 MFMAs (t = 0..3: column 4u + t) per query group;
// the 8 lanes of a row read one full 128-byte line, every fetched byte is used once, as in the VALU scan.
// The query piece comes from the LDS tile with one conflict-free ds_read_b128 per query group per step.
// acc[g][t] (float4 = rows 4rh .. 4rh+3) of lane (u, rh, x) is chain c = 4u + t of query 4g + x, AVX register
// r = c >> 3 = u1 + 2 u2, SIMD lane j = c & 7 = 4 u0 + t.  The reference's tree, with every exchange step also
// halving the set of (row, query) results a lane is responsible for:
//   a + b, c + d        partner u1 ^ 1 = lane ^ 16   v_permlane16_swap   keeps rows {0,1} / {2,3}
//   (a+b) + (c+d)       partner u2 ^ 1 = lane ^ 8    DPP row_ror:8       keeps row 2 u1 + u2
//   hi128 + lo128       partner u0 ^ 1 = lane ^ 4    ds_swizzle          keeps query groups [0,NG/2) / [NG/2,NG)
//   (lr0+lr1)+(lr2+lr3) over t, in-lane
// so each lane ends with NG/2 finished scores (row 4rh + 2u1 + u2, queries 4 (gp + u0 NG/2) + x).
#include "scan_common.hpp"

namespace qmx {

typedef float f32x4 __attribute__((ext_vector_type(4)));


template <int XORMASK>
__device__ __forceinline__ float swz_xor(float v) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (XORMASK << 10) | 0x1F));
}
__device__ __forceinline__ float ror8(float v) {   // lane l <-> l ^ 8 inside each row of 16 lanes
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /* row_ror:8 */, 0xF, 0xF, true));
}
// lanes of even 16-lane rows get a_mine + a_partner, lanes of odd rows b_mine + b_partner (partner = lane ^ 16)
__device__ __forceinline__ float swap16_add(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// QW queries per wave, QSPLIT waves share one row stream (each scores its own QW queries of the
// QW * QSPLIT-query tile; the second read of a row line hits L1 / L2), D row loads in flight per lane.
template <int QW, int QSPLIT, int D, bool NT, int NWAVES, bool FAST, bool QH, bool HAS_IDS, int MODE>
__device__ __forceinline__ void scan_f32_mfma_body(const ScanArgs &a, unsigned char *smem) {
    constexpr int MF_BLOCK = NWAVES * 64, MF_NW = NWAVES;
    // QH = false: lane bit 5 (`rh`) picks the row half of an 8-row tile, a query group is 4 queries;
    // QH = true : lane bit 5 picks the QUERY half of a group of 8, the tile has 4 rows (each row piece is loaded by two
    //             lanes of the same instruction: one fetch): twice the queries per accumulator register
    constexpr int GQ = QH ? 8 : 4;              // queries per group
    constexpr int TR = QH ? 4 : 8;              // rows per tile
    constexpr int NG = QW / GQ;
    constexpr int QT = QW * QSPLIT;
    constexpr int NSTREAM = MF_NW / QSPLIT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qs = wave % QSPLIT;          // which QW-query slice of the tile this wave scores
    const int stream = wave / QSPLIT;      // which row stream of the block
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.queries);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const uint32_t n16 = (uint32_t)QT * a.q_stride / 16;
        for (uint32_t i = tid; i < n16; i += MF_BLOCK) dst[i] = src[i];
    }
    __syncthreads();

    const int x = lane & 3;
    const int u0 = (lane >> 2) & 1, u2 = (lane >> 3) & 1, u1 = (lane >> 4) & 1;
    const int u = u0 + 2 * u1 + 4 * u2;
    const int rh = lane >> 5;
    const int rofs = QH ? 0 : 4 * rh;           // first row (inside the tile) of the lane's 4 rows
    const int qofs = QH ? 4 * rh + x : x;       // the lane's query inside a group
    constexpr int NGH = NG / 2;                 // finished (row, query) results per lane and tile
    const int my_m = 2 * u1 + u2;               // ... for row rofs + my_m
    const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
    const uint32_t q0 = (uint32_t)qs * QW;                       // first query of this wave
    const unsigned char *qbase = smem + (q0 + (uint32_t)qofs) * a.q_stride + (uint32_t)u * 16;   // + g * GQ * q_stride + s * 128
    const uint32_t gstride = (uint32_t)GQ * a.q_stride;
    const int top = (int)a.top;

    uint64_t list[QW];
    uint64_t thr[NGH];    // k-th best key of the lane's own queries
    float thr_f[NGH];     // ... its score (-inf while the list is not full)
    int my_q[NGH];        // ... which are (wave-local index) 4 (gp + u0 NGH) + x
#pragma unroll
    for (int q = 0; q < QW; ++q) list[q] = 0;
    uint64_t gk[NGH];     // score part of the pre-scan's bound of the lane's queries (api_search.hip search_enqueue; 0 = none): equal scores pass
#pragma unroll
    for (int gp = 0; gp < NGH; ++gp) {
        my_q[gp] = GQ * (gp + u0 * NGH) + qofs;
        const uint32_t gq = (uint32_t)qs * QW + (uint32_t)my_q[gp];
        gk[gp] = (MODE == SCAN_TOPK && a.gthr && gq < a.nq) ? (a.gthr[gq] & 0xFFFFFFFF00000000ull) : 0ull;
        thr[gp] = gk[gp];
        thr_f[gp] = gk[gp] ? key_score(gk[gp]) : -__builtin_inff();
    }

    const uint32_t gw = blockIdx.x * NSTREAM + stream;
    const uint32_t tw = gridDim.x * NSTREAM;
    const uint64_t n_tiles = (a.n_cand + TR - 1) / TR;
    const uint32_t nseg = a.nseg;

    // row pointer of this lane for a tile (the lane streams row 4rh + x of it); out-of-range tiles alias tile 0
    auto lane_row_ptr = [&](uint64_t tile) -> const unsigned char * {
        uint64_t c = tile * TR + (uint32_t)(rofs + x);
        if (c >= a.n_cand) c = 0;
        uint32_t id = HAS_IDS ? a.ids[c] : (uint32_t)c;
        if (HAS_IDS && id >= a.n_rows) id = 0;
        return rows + (uint64_t)id * a.row_stride + (uint32_t)u * 16;
    };
    // D row pieces in flight per lane at all times: while chunk k is consumed, chunk k+1 (or the first chunk of
    // the NEXT tile) is already on its way.
    auto load_piece = [](const unsigned char *p) -> uint4 {
        if (NT) {   // streamed once: keep the lines out of the way of the query tile / partial lists in L2
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
            return make_uint4(v[0], v[1], v[2], v[3]);
        }
        return *reinterpret_cast<const uint4 *>(p);
    };
    uint4 cur[D], nxt[D];
    const unsigned char *rp = lane_row_ptr(gw < n_tiles ? gw : 0);
#pragma unroll
    for (int d = 0; d < D; ++d) cur[d] = load_piece(rp + (uint64_t)((uint32_t)d < nseg ? d : 0) * 128);

    // every wave of the block runs the same number of iterations (QSPLIT > 1 synchronises per tile so that the
    // waves sharing a row stream ask for the same lines at the same time: the second request hits L1 / L2)
    const uint64_t n_iters = (n_tiles + tw - 1) / tw;
    for (uint64_t it = 0; it < n_iters; ++it) {
        const uint64_t tile = gw + it * tw;
        if (QSPLIT > 1) __syncthreads();
        if (tile >= n_tiles) continue;
        // the 4 rows of this lane's half tile (rows 4rh + m)
        uint32_t rid[4];
        bool valid[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const uint64_t c = tile * TR + (uint32_t)(rofs + m);
            valid[m] = c < a.n_cand;
            const uint64_t cc = valid[m] ? c : 0;
            uint32_t id = HAS_IDS ? a.ids[cc] : (uint32_t)cc;
            if (HAS_IDS && id >= a.n_rows) {
                if (valid[m]) *a.err_flag = 1;
                id = 0;
                valid[m] = false;
            }
            rid[m] = id;
        }
        const unsigned char *rp_next = lane_row_ptr(tile + tw < n_tiles ? tile + tw : 0);

        f32x4 acc[NG][4];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[g][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

        uint4 qq[2][NG];     // query pieces of step s (qq[s & 1]) and s + 1, read one step ahead
#pragma unroll
        for (int g = 0; g < NG; ++g) qq[0][g] = *reinterpret_cast<const uint4 *>(qbase + (uint32_t)g * gstride);

        // one 32-float step: 4 MFMAs per query group on the row piece `vv` and the query pieces `qv`
        auto step_mfma = [&](const uint4 &vv, const uint4 (&qv)[NG]) {
            const float v0 = __uint_as_float(vv.x), v1 = __uint_as_float(vv.y), v2 = __uint_as_float(vv.z), v3 = __uint_as_float(vv.w);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                // _mm256_fmadd_ps(v1, v2, sum) of chain 4u + t for rows 4rh..4rh+3 x queries q0+4g..q0+4g+3
                acc[g][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(v0, __uint_as_float(qv[g].x), acc[g][0], 0, 0, 0);
                acc[g][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(v1, __uint_as_float(qv[g].y), acc[g][1], 0, 0, 0);
                acc[g][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(v2, __uint_as_float(qv[g].z), acc[g][2], 0, 0, 0);
                acc[g][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(v3, __uint_as_float(qv[g].w), acc[g][3], 0, 0, 0);
            }
        };
        if constexpr (FAST) {
            // nseg is a multiple of 2 D: chunks alternate between the two register buffers (no copies, no guards, constant
            // offsets inside a chunk); `cur` holds chunk c, `nxt` chunk c + 1, the refill of `cur` is chunk c + 2 of this
            // tile or chunk 0 of the next one.  The query piece of step s + 1 is read while step s multiplies; the read
            // past the last step lands in the slack behind the tile entry and is never used.
            const uint32_t nchunk = nseg / D;
            for (uint32_t c = 0; c < nchunk; c += 2) {
                const unsigned char *pb = rp + (uint64_t)(c + 1) * D * 128;
#pragma unroll
                for (int d = 0; d < D; ++d) nxt[d] = load_piece(pb + d * 128);
                const unsigned char *qc = qbase + (c * D + 1) * 128;
#pragma unroll
                for (int d = 0; d < D; ++d) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) qq[(d + 1) & 1][g] = *reinterpret_cast<const uint4 *>(qc + (uint32_t)g * gstride + d * 128);
                    step_mfma(cur[d], qq[d & 1]);
                }
                const bool more = c + 2 < nchunk;
                const unsigned char *pa = more ? rp + (uint64_t)(c + 2) * D * 128 : rp_next;
#pragma unroll
                for (int d = 0; d < D; ++d) cur[d] = load_piece(pa + d * 128);
                const unsigned char *qd = qbase + ((c + 1) * D + 1) * 128;
#pragma unroll
                for (int d = 0; d < D; ++d) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) qq[(d + 1) & 1][g] = *reinterpret_cast<const uint4 *>(qd + (uint32_t)g * gstride + d * 128);
                    step_mfma(nxt[d], qq[d & 1]);
                }
            }
        } else {
            for (uint32_t s0 = 0; s0 < nseg; s0 += D) {
                const bool last_chunk = s0 + D >= nseg;
                const unsigned char *np = last_chunk ? rp_next : rp;
                const uint32_t ns0 = last_chunk ? 0 : s0 + D;
    #pragma unroll
                for (int d = 0; d < D; ++d) {
                    const uint32_t sn = ns0 + d < nseg ? ns0 + d : nseg - 1;
                    nxt[d] = load_piece(np + (uint64_t)sn * 128);
                }
    #pragma unroll
                for (int d = 0; d < D; ++d) {
                    const uint32_t sq = s0 + d + 1 < nseg ? s0 + d + 1 : 0;     // query pieces of the next step
    #pragma unroll
                    for (int g = 0; g < NG; ++g)
                        qq[(d + 1) & 1][g] = *reinterpret_cast<const uint4 *>(qbase + (uint32_t)g * gstride + sq * 128);
                    if (s0 + d < nseg) step_mfma(cur[d], qq[d & 1]);
                }
    #pragma unroll
                for (int d = 0; d < D; ++d) cur[d] = nxt[d];
            }
        }
        rp = rp_next;

        // ---- fold the 32 chains in the reference's order (four_way_hsum, hsum256_ps_avx) ----
        float s2[NG][4];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float r01 = swap16_add(acc[g][t][0], acc[g][t][2]);      // u1 = 0: row 0, u1 = 1: row 2   (a + b | c + d)
                const float r23 = swap16_add(acc[g][t][1], acc[g][t][3]);      // u1 = 0: row 1, u1 = 1: row 3
                const float keep = u2 ? r23 : r01, send = u2 ? r01 : r23;
                s2[g][t] = keep + ror8(send);                                  // (a + b) + (c + d), row 2 u1 + u2
            }
        const uint32_t res_row = my_m == 0 ? rid[0] : my_m == 1 ? rid[1] : my_m == 2 ? rid[2] : rid[3];
        const bool res_valid = my_m == 0 ? valid[0] : my_m == 1 ? valid[1] : my_m == 2 ? valid[2] : valid[3];
#pragma unroll
        for (int gp = 0; gp < NGH; ++gp) {
            float lr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float keep = u0 ? s2[gp + NGH][t] : s2[gp][t], send = u0 ? s2[gp][t] : s2[gp + NGH][t];
                lr[t] = keep + swz_xor<4>(send);                               // hi128 + lo128
            }
            float score = (lr[0] + lr[1]) + (lr[2] + lr[3]);
            const uint32_t q = q0 + (uint32_t)my_q[gp];
            if (a.tail_start < a.dim) {                                        // scalar tail: mul then add (simple_avx.rs:208-211)
                const float *qf = reinterpret_cast<const float *>(smem + q * a.q_stride);
                const float *vf = reinterpret_cast<const float *>(rows + (uint64_t)res_row * a.row_stride);
                for (uint32_t i = a.tail_start; i < a.dim; ++i) score += qf[i] * vf[i];
            }
            const bool mine = res_valid && q < a.nq;
            if (MODE == SCAN_SCORES) {
                if (mine) a.scores[(uint64_t)q * a.scores_stride + (tile * TR + (uint32_t)(rofs + my_m))] = score;
            } else {
                // cheap reject on the score alone; ties with the k-th score and NaN (greatest in OrderedFloat) fall through
                // to the exact key compare
                bool c = mine && !(score < thr_f[gp]);
                if (__ballot(c)) {
                    const uint64_t key = make_key(score, res_row);
                    c = c && key > thr[gp] && a.del.live(res_row) && (!a.key_bound || key < a.key_bound[q]);
                    uint64_t mask = __ballot(c);
                    while (mask) {
                        const int src = __builtin_ctzll(mask);
                        mask &= mask - 1;
                        const uint64_t nk = readlane_u64(key, src);
                        const int ql = __builtin_amdgcn_readlane(my_q[gp], src);
#pragma unroll
                        for (int qq = 0; qq < QW; ++qq) {
                            if (ql == qq) {
                                if (nk > readlane_u64(list[qq], top - 1)) {
                                    wave_list_insert(list[qq], nk, lane);
                                    const uint64_t nt = readlane_u64(list[qq], top - 1);
#pragma unroll
                                    for (int gg = 0; gg < NGH; ++gg)
                                        if (my_q[gg] == qq && nt > gk[gg]) {
                                            thr[gg] = nt;
                                            thr_f[gg] = key_score(nt);
                                        }
                                }
                            }
                        }
                    }
                }
            }
        }
    }

    if (MODE == SCAN_SCORES) return;

    // ---- block merge: the NSTREAM wave lists of each query -> 1 list, one global write per block ----
    __syncthreads();
    uint64_t *lds_keys = reinterpret_cast<uint64_t *>(smem);
    const uint32_t utop = a.top;
#pragma unroll
    for (int q = 0; q < QW; ++q)
        if (lane < top) lds_keys[((uint32_t)stream * QT + q0 + q) * utop + lane] = list[q];
    __syncthreads();
    for (uint32_t q = wave; q < a.nq; q += MF_NW) {
        uint64_t merged = 0;
        for (int sw = 0; sw < NSTREAM; ++sw) {
            const uint64_t key = lane < top ? lds_keys[((uint32_t)sw * QT + q) * utop + lane] : 0;
            uint64_t mk = __ballot(key > readlane_u64(merged, top - 1));
            while (mk) {
                const int src = __builtin_ctzll(mk);
                mk &= mk - 1;
                const uint64_t nk = readlane_u64(key, src);
                if (nk > readlane_u64(merged, top - 1)) wave_list_insert(merged, nk, lane);
            }
        }
        if (lane < top) a.partial[((uint64_t)blockIdx.x * QT + q) * utop + lane] = merged;
    }
}

template <int QW, int QSPLIT, int D, bool NT, int NWAVES, bool FAST, bool QH, bool HAS_IDS, int MODE>
__global__ __launch_bounds__(NWAVES * 64) void scan_f32_mfma_kernel(const ScanArgs a0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (MODE == SCAN_SCORES) {
        // score mode takes any number of queries: the block scores its rows against one QW * QSPLIT-query tile after the other (one launch
        // instead of one per tile: the sample pre-scans of a 128-query batch are four latency-bound passes over ~10 k rows, which
        // stay in L2 between the passes)
        constexpr uint32_t QT = QW * QSPLIT;
        // (gridDim.y > 1: the launcher found room for every query tile's blocks at once - a short candidate list against many queries - and each
        // block scores ONE tile: the passes run side by side instead of one after the other)
        const uint32_t g_first = gridDim.y > 1 ? blockIdx.y * QT : 0u, g_end = gridDim.y > 1 ? g_first + 1u : a0.nq;
        for (uint32_t g0 = g_first; g0 < g_end && g0 < a0.nq; g0 += QT) {
            ScanArgs a = a0;
            a.queries = reinterpret_cast<const unsigned char *>(a0.queries) + (size_t)g0 * a0.q_stride;
            a.scores = a0.scores + (uint64_t)g0 * a0.scores_stride;
            a.nq = a0.nq - g0 < QT ? a0.nq - g0 : QT;
            if (g0 != g_first) __syncthreads();        // every wave is done with the previous tile's entries
            scan_f32_mfma_body<QW, QSPLIT, D, NT, NWAVES, FAST, QH, HAS_IDS, MODE>(a, smem);
        }
    } else {
        scan_f32_mfma_body<QW, QSPLIT, D, NT, NWAVES, FAST, QH, HAS_IDS, MODE>(a0, smem);
    }
}

template <int QW, int QSPLIT, int D, bool NT, int NWAVES, bool FAST, bool QH, bool HAS_IDS, int MODE>
static int32_t launch_mfma_inst(hipStream_t st, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    constexpr int QT = QW * QSPLIT;
    constexpr int MF_BLOCK = NWAVES * 64, MF_NW = NWAVES;
    constexpr int NSTREAM = MF_NW / QSPLIT;
    size_t lds = (size_t)QT * a.q_stride + 256;   // slack: the FAST path reads one step past the last entry
    if (MODE == SCAN_TOPK) {
        const size_t lk = (size_t)NSTREAM * QT * a.top * sizeof(uint64_t);
        if (lk > lds) lds = lk;
    }
    lds = (lds + 15) & ~(size_t)15;
    QMX_REQUIRE(lds <= 160 * 1024, QMX_ERR_NOT_SUPPORTED, "query tile needs %zu B of LDS (> 160 KiB)", lds);
    auto kfn = scan_f32_mfma_kernel<QW, QSPLIT, D, NT, NWAVES, FAST, QH, HAS_IDS, MODE>;
    static thread_local DeviceOnce attr_once;
    if (attr_once.need()) {
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.mark();
    }
    int per_cu = 0;
    QMX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, MF_BLOCK, lds));
    if (per_cu < 1) per_cu = 1;
    const uint64_t n_tiles = (a.n_cand + (QH ? 3 : 7)) / (QH ? 4 : 8);
    const uint64_t want = (n_tiles + NSTREAM - 1) / NSTREAM;
    const uint64_t cap = (uint64_t)num_cus * per_cu;
    uint32_t grid = (uint32_t)(want < cap ? want : cap);
    if (grid < 1) grid = 1;
    if (grid_out) {
        if (*grid_out && MODE == SCAN_TOPK && grid > *grid_out) grid = *grid_out;
        *grid_out = grid;
    }
    uint32_t grid_y = 1;
    if (MODE == SCAN_SCORES && a.nq > (uint32_t)QT) {
        const uint32_t groups = (a.nq + QT - 1) / QT;
        if ((uint64_t)grid * groups <= cap) grid_y = groups;
    }
    ::qmx::clear_stale_error();
    QMX_NOTE_KERNEL(kfn);
    hipLaunchKernelGGL(kfn, dim3(grid, grid_y), dim3(MF_BLOCK), lds, st, a);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

template <int QW, int QSPLIT, int D, bool NT, int NWAVES = 8, bool FAST = false, bool QH = false>
static int32_t launch_mfma_qt(hipStream_t st, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid) {
    const bool ids = a.ids != nullptr;
    if (mode == SCAN_TOPK)
        return ids ? launch_mfma_inst<QW, QSPLIT, D, NT, NWAVES, FAST, QH, true, SCAN_TOPK>(st, a, num_cus, grid)
                   : launch_mfma_inst<QW, QSPLIT, D, NT, NWAVES, FAST, QH, false, SCAN_TOPK>(st, a, num_cus, grid);
    return ids ? launch_mfma_inst<QW, QSPLIT, D, NT, NWAVES, FAST, QH, true, SCAN_SCORES>(st, a, num_cus, grid)
               : launch_mfma_inst<QW, QSPLIT, D, NT, NWAVES, FAST, QH, false, SCAN_SCORES>(st, a, num_cus, grid);
}

// qt in {8, 16, 32}; f32 rows, dot (or cosine on normalised rows), dim >= 32
int32_t launch_scan_f32_mfma(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    switch (qt) {
        // measured on MI355X, C2 (10 M x 768): nontemporal row loads and 12 steps in flight per lane are worth ~3 %
        case 8: return launch_mfma_qt<8, 1, 4, true>(st, mode, a, num_cus, grid_out);
        case 16:
            if (mfma16_scan_ok(16, mode, a)) return launch_scan_f32_mfma16(st, 16, a, num_cus, grid_out);
            // rows of a multiple of 384 floats (768, 1536, ...): guard-free ping-pong main loop
            if (a.nseg % 12 == 0) return launch_mfma_qt<16, 1, 6, true, 8, true>(st, mode, a, num_cus, grid_out);
            return launch_mfma_qt<16, 1, 12, true>(st, mode, a, num_cus, grid_out);
        case 32: {
            if (mfma16_scan_ok(32, mode, a)) return launch_scan_f32_mfma16(st, 32, a, num_cus, grid_out);
#ifdef QMX_TUNING
            static const int variant = getenv("QMX_MFMA_VARIANT") ? atoi(getenv("QMX_MFMA_VARIANT")) : 0;   // tuning experiments
            if (variant == 1) return launch_mfma_qt<32, 1, 6, true, 8, false, true>(st, mode, a, num_cus, grid_out);   // query-half layout, generic loop
            if (variant == 2 && a.nseg % 12 == 0) return launch_mfma_qt<32, 1, 6, true, 8, true, true>(st, mode, a, num_cus, grid_out);   // + ping-pong
            if (variant == 3 && a.nseg % 24 == 0) return launch_mfma_qt<32, 1, 12, true, 8, true, true>(st, mode, a, num_cus, grid_out);
#endif
            if (a.nseg % 12 == 0) return launch_mfma_qt<16, 2, 6, true, 8, true>(st, mode, a, num_cus, grid_out);
            return launch_mfma_qt<16, 2, 12, true>(st, mode, a, num_cus, grid_out);
        }
    }
    if (qt == 64 && mfma16_scan_ok(64, mode, a)) return launch_scan_f32_mfma16(st, 64, a, num_cus, grid_out);   // api_*.hip only asks when it applies
    set_error("unsupported MFMA query tile %d", qt);
    return QMX_ERR_BAD_ARG;
}

}  // namespace qmx

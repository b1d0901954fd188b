// scan_sqw.hip - EncodedVectorsU8 (scalar int8), brute-force top-k for LARGE query batches: 128 queries per pass of the code block - the structure of
// scan_tq4w.hip without a decode (an SQ row's code bytes ARE the int8 operands).
//
// Same reference loops as scan_sq_mfma.hip SqOps: BatchFilteredSearcher::peek_top_iter (lib/segment/src/index/hnsw_index/point_scorer.rs:423-472) over
// EncodedVectorsU8::score_point_avx (lib/quantization/src/encoded_vectors_u8.rs:471-490 -> cpp/avx2.c:25-63 impl_score_dot_avx) and postprocess_score
// (:100-103): the exact integer dot of the codes, then multiplier * dot + query_offset + vector_offset, left to right, not fused.  Dot / cosine / euclid
// with a positive multiplier (the usual sign: alpha^2, or 2 alpha^2 for the inverted L2), rows of a multiple of 64 codes below 1041 (sq_mfma_ok: the AVX2
// leaf's f32 lane sums stay exact).
//
// Why.  The 32-query kernel (scan_sq_mfma.hip) streams the 7.7 GB of a 10 M x 768 block once per 32 queries at 0.72 - 0.74 of HBM: 128 queries cost four
// passes, 5.3 ms.  Here a wave owns 32 rows of a 256-row tile, its lanes fetch exactly the 16-byte operand pieces v_mfma_i32_16x16x64_i8 wants from them
// (plain 16-byte loads into the operand registers themselves, four 64-code stages ahead), and multiplies them with all 128 queries (their codes as B-operand images in LDS, loaded from a
// 96 KiB image in L2): one pass of the block per 128 queries, bound by the HBM stream of the codes.
//
// Scores are exact, so the pass needs no band (scan_tq4w.hip: the same tail): a pair is a candidate when its score is not below the k-th best score of a
// strided sample of the block; the fast reject runs on integers: multiplier * dot + query_offset + vector_offset >= T  <=>  dot + vector_offset / multiplier
// >= (T - query_offset) / multiplier, so with B[row] = ceil(vector_offset / multiplier) + 1 (an int32 column made once per segment) and A[query] = the
// right side rounded down less a margin that covers the three f32 roundings of the expression, a pair can only reach the threshold when dot + B[row] >=
// A[query].  (A bound on the segment's LARGEST vector_offset alone admits everything on rows whose offsets spread wider than their scores: iid unit rows.)
#include <type_traits>

#include "scan_common.hpp"

namespace qmx {

typedef int i32x4q __attribute__((ext_vector_type(4)));

constexpr int SW_THREADS = 512;
constexpr int SW_BM = 256;                                   // rows per tile
constexpr int SW_QT = 128;                                   // queries per pass
constexpr int SW_B_UNITS = SW_QT * 4;                        // 16-byte units of the queries' stage: 128 queries x 64 codes = 8 KiB
constexpr int SW_LDS = 2 * SW_B_UNITS * 16 + SW_QT * 4;      // the queries' two buffers + the 128 integer bounds (the codes go to registers)
constexpr uint32_t SW_WCAP = 8192;                           // candidates one wave may list per pass

// unit index of (16-query tile t, piece kq, query-in-tile m) inside a stage of the queries' image: scan_split.hip sp_unit's swizzle, conflict-free for the operand reads
__device__ __forceinline__ uint32_t sw_unit(uint32_t t, uint32_t kq, uint32_t m) { return (t * 4 + kq) * 16 + (m ^ (2 * kq)); }

__device__ __forceinline__ void sw_stage_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct SqWideArgs {
    const uint4 *bq;          // [nch][SW_B_UNITS] the queries' operand images (sqw_pack_kernel)
    uint32_t nch;             // stages per tile: codes of a row / 64
    uint32_t nq;              // live queries (<= 128)
    const int32_t *thr_i;     // [128] A[query]: a pair with dot + B[row] below this cannot reach the query's threshold
    const int32_t *bi;        // [n] B[row] (sqw_stats_kernel)
    uint4 *wlist;             // [waves][wcap] (dot, row, query, 0); sqw_finish_kernel turns them into (key lo, key hi, query, 0)
    uint32_t *wcnt;           // [waves] entries each wave wanted to append (may run past wcap: overflow)
    uint32_t wcap;
};

// ---- once per segment: bi[row] = ceil(vector_offset / multiplier) + 1 (clamped to +-2^30), stats[0] = max |vector_offset| (uint bits of a non-negative
// float), stats[1] != 0: an offset that is not finite ----
__global__ __launch_bounds__(256) void sqw_stats_kernel(const float *off, uint64_t n, float multiplier, int32_t *bi, uint32_t *stats) {
    float hi = 0.0f;
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const float v = off[i];
        bad = bad || !(__builtin_fabsf(v) < __builtin_inff());
        hi = __builtin_fmaxf(hi, __builtin_fabsf(v));
        const double b = __builtin_ceil((double)v / (double)multiplier) + 1.0;
        bi[i] = b >= 1073741824.0 ? 1073741824 : b <= -1073741824.0 ? -1073741824 : (b == b ? (int32_t)b : 1073741824);
    }
    for (int o = 32; o >= 1; o >>= 1) hi = __builtin_fmaxf(hi, __shfl_xor(hi, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&stats[0], __float_as_uint(hi));
    if (bad) atomicOr(&stats[1], 1u);
}

// ---- once per 128-query tile: one block per query slot.  The query's codes in the B-operand images, its integer reject bound, what the finish needs ----
__global__ __launch_bounds__(256) void sqw_pack_kernel(const unsigned char *queries, uint32_t q_stride, uint32_t aux_off, uint32_t nq, uint32_t nch,
                                                       const uint64_t *gthr, float multiplier, float off_absmax, uint32_t dim, uint4 *bq, int32_t *thr_i, float *qinfo,
                                                       float *band, uint32_t *cand_cnt, uint32_t n_cnt) {
    const uint32_t qi = blockIdx.x;
    const bool live = qi < nq;
    if (qi == 0)
        for (uint32_t i = threadIdx.x; i < n_cnt; i += 256) cand_cnt[i] = 0;
    const unsigned char *entry = queries + (uint64_t)qi * q_stride;
    for (uint32_t u = threadIdx.x; u < nch * 4; u += 256) {
        const uint32_t kc = u >> 2, p = u & 3u;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (live) v = *reinterpret_cast<const uint4 *>(entry + (uint64_t)kc * 64 + p * 16);
        bq[(uint64_t)kc * SW_B_UNITS + sw_unit(qi >> 4, p, qi & 15u)] = v;
    }
    if (threadIdx.x != 0) return;
    float q_off = 0.0f, tf = __builtin_inff(), bd = 0.0f;
    int32_t ti = 0x7FFFFFFF;                                         // a dead slot passes nothing
    if (live) {
        q_off = reinterpret_cast<const QueryAux *>(entry + aux_off)->f0;
        const uint64_t k = gthr[qi];
        // no bound (the sample holds fewer than k live rows) or a query offset that is not finite: no candidates, and the infinite band sends the query -
        // alone - to the 32-query scan (sp_select_kernel)
        bd = __builtin_inff();
        const float t = k ? key_score(k) : 0.0f;
        if (k != 0 && t == t && __builtin_fabsf(q_off) < __builtin_inff()) {
            tf = t;
            bd = 0.0f;
            // (m * dot + q_off) + v_off >= T  <=>  dot + v_off / m >= (T - q_off) / m.  Three f32 roundings on the left, each within 2^-24 of its
            // operands' size (|m * dot| <= m 127^2 dim, |q_off|, |v_off| <= off_absmax); the row's column entry rounds v_off / m up and adds one
            const double eps = 3.8146972656e-6, m = (double)multiplier;
            const double sizes = m * 16129.0 * (double)dim + __builtin_fabs((double)q_off) + (double)off_absmax + __builtin_fabs((double)t);
            const double a_thr = ((double)t - (double)q_off) / m - (sizes * eps / m + 3.0);
            ti = a_thr <= -2147483647.0 ? (int32_t)0x80000000 : a_thr >= 2147483520.0 ? 0x7FFFFFFF : (int32_t)__builtin_floor(a_thr);
            if (!(a_thr == a_thr)) { ti = 0x7FFFFFFF; tf = __builtin_inff(); bd = __builtin_inff(); }
        }
    }
    thr_i[qi] = ti;
    band[qi] = bd;
    qinfo[qi] = q_off;
    qinfo[SW_QT + qi] = tf;
}

// The scan.  Block = 8 waves, one block per CU, persistent over 256-row tiles; a tile = nch stages of 64 codes.  A wave OWNS 32 rows of the tile: lane
// (m, kg) loads, per stage, the 16 code bytes [16 kg, +16) of rows m and 16 + m - exactly its operand registers of the stage's two row tiles - and
// multiplies with all 128 queries: 16 matrix instructions and 8 operand reads per wave and stage.
// The loads are `global_load_dwordx4` issued by inline asm INTO the operand registers, four stages ahead, and waited for by the kernel's own `s_waitcnt
// vmcnt` (an asm statement that names the registers it releases as in-out operands, so no use or copy of them can be scheduled in front of it):
//   * hipcc's own waits would be conservative across the loop's back edge (scan_tq4w.hip met `vmcnt(0)` at every stage);
//   * LDS-DMA (`global_load_lds_dwordx4`), which rounds 3 - 5 of this library use for such streams, costs the issuing wave 150 - 260 cycles per 1 KiB
//     request here - three requests per 16 matrix instructions: measured, the loop without its code requests ran 0.99 ms per 10 M x 768 pass, with them
//     1.64 - 1.78 whatever their depth (three 128-code stages one or two ahead, six 64-code stages five ahead).  A plain load is one issue slot.
// vmcnt is IN-ORDER: the wait of stage g lets the last 11 loads stay in flight - the codes of stages g + 1 .. g + 4 and the queries of g + 2 .. g + 4 -, which
// lands the codes of stage g and the queries of stage g + 1; the latter go to LDS by one ds_write_b128 per lane (the queries' buffers, two of them, are all
// the waves share: one barrier per stage).  64 KiB of codes per CU on their way.
__device__ __forceinline__ void sw_gload16(i32x4q &dst, const unsigned char *sbase, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
constexpr int SW_AHEAD_STAGES = 4;                           // stages the loads run ahead of the matrix work
constexpr int SW_SLOTS = SW_AHEAD_STAGES + 1;                // register sets of a stage's loads (the running stage's + those in flight)
__global__ __launch_bounds__(SW_THREADS, 1) void scan_sqw_kernel(const ScanArgs a, const SqWideArgs s) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4 *lds = reinterpret_cast<uint4 *>(smem_raw);
    int32_t *thr_lds = reinterpret_cast<int32_t *>(smem_raw + (size_t)2 * SW_B_UNITS * 16);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t n_tiles = (a.n_cand + SW_BM - 1) / SW_BM;
    const uint32_t nch = s.nch;
    const uint64_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    if (my_tiles == 0) {
        if (lane == 0) s.wcnt[blockIdx.x * (SW_THREADS / 64) + (uint32_t)w] = 0;
        return;
    }
    if (tid < SW_QT) thr_lds[tid] = s.thr_i[tid];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // from here on the kernel counts its vector-memory traffic itself
    const uint32_t kq_r = (uint32_t)lane >> 4, m_r = (uint32_t)lane & 15u;
    const uint32_t b_rd = sw_unit(0, kq_r, m_r);
    const uint32_t lane_off = (uint32_t)lane * 16u;
    uint4 *const b_lds = lds;
    const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
    const uint64_t last_row = a.n_cand - 1;
    const uint32_t row_stride32 = (uint32_t)a.row_stride;
    const uint32_t rl0 = (uint32_t)w * 32u + m_r, rl1 = rl0 + 16u;
    const uint32_t coff0 = rl0 * row_stride32 + kq_r * 16u, coff1 = rl1 * row_stride32 + kq_r * 16u;

    auto uniform_ptr = [&](uint64_t v) {
        return reinterpret_cast<const unsigned char *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
                                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
    };
    // One stage's three loads, in the order the waits rely on: this wave's 1 KiB of the queries' image of stage kc, then the lane's code pieces of row tiles
    // 0 and 1 of stage kc of the block's it-th tile (rows past the block: the last row's bytes, their scores are dropped)
    auto load_stage = [&](uint64_t it, uint32_t kc, i32x4q &rq, i32x4q &rc0, i32x4q &rc1) __attribute__((always_inline)) {
        const unsigned char *qsrc = uniform_ptr((uint64_t)(uintptr_t)(s.bq + (uint64_t)kc * SW_B_UNITS) + (uint32_t)w * 1024u);
        const uint64_t row0 = (blockIdx.x + it * gridDim.x) * SW_BM;
        const unsigned char *csrc = uniform_ptr((uint64_t)(uintptr_t)(rows + row0 * a.row_stride + kc * 64u));
        uint32_t o0 = coff0, o1 = coff1;
        const uint64_t room = last_row - row0;                      // (row0 <= last_row: the tile exists)
        if (room < SW_BM - 1) {                                     // the block's last, partial tile (wave-uniform)
            const uint32_t r0 = (uint64_t)rl0 < room ? rl0 : (uint32_t)room, r1 = (uint64_t)rl1 < room ? rl1 : (uint32_t)room;
            o0 = r0 * row_stride32 + kq_r * 16u;
            o1 = r1 * row_stride32 + kq_r * 16u;
        }
        sw_gload16(rq, qsrc, lane_off);
        sw_gload16(rc0, csrc, o0);
        sw_gload16(rc1, csrc, o1);
    };

    i32x4q acc[2][8];
    uint4 *const wl = s.wlist + (uint64_t)(blockIdx.x * (SW_THREADS / 64) + (uint32_t)w) * s.wcap;
    uint32_t wcount = 0;
    const uint32_t n_rows32 = (uint32_t)a.n_cand;
    // B of the lane's eight rows of the running tile (rows 4 kq_r .. + 3 of both 16-row tiles of the wave), loaded on the tile's first stage for its epilogue
    i32x4q bi8[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    auto load_bi = [&](uint64_t it) __attribute__((always_inline)) {
        const uint64_t row0 = (blockIdx.x + it * gridDim.x) * SW_BM;
        const unsigned char *src = uniform_ptr((uint64_t)(uintptr_t)(s.bi + row0));
        const uint64_t room = last_row - row0;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            uint32_t r = (uint32_t)w * 32u + (uint32_t)mt * 16u + 4 * kq_r;
            if ((uint64_t)r + 3 > room) r = room >= 3 ? (uint32_t)room - 3u : 0u;      // (entries of rows past the block are never used; keep the 16 bytes inside the column)
            sw_gload16(bi8[mt], src, r * 4u);
        }
    };

    // The epilogue of a tile: dot + B[row] >= A[query], narrowing by wave-uniform steps (query tile, then the lane's rows)
    auto epilogue = [&](uint64_t it) __attribute__((always_inline)) {
        const uint64_t tile = blockIdx.x + it * gridDim.x;
        const uint32_t row0 = (uint32_t)(tile * SW_BM) + (uint32_t)w * 32u + 4 * kq_r;
        uint32_t hits8 = 0;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            int mx = (int)0x80000000;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int v = acc[mt][nt][j] + bi8[mt][j];
                    mx = v > mx ? v : mx;
                }
            if (mx >= thr_lds[nt * 16 + (int)m_r]) hits8 |= 1u << nt;
        }
        if (!__ballot(hits8 != 0)) return;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            if (!__ballot((hits8 >> nt) & 1u)) continue;
            const uint32_t q = (uint32_t)nt * 16 + m_r;
            const int ti = thr_lds[q];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int v = acc[mt][nt][j];
                    const uint32_t row = row0 + (uint32_t)mt * 16 + (uint32_t)j;
                    const bool c = v + bi8[mt][j] >= ti && row < n_rows32 && q < s.nq;
                    const uint64_t hits = __ballot(c);
                    if (hits) {
                        const uint32_t at = wcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(hits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hits, 0u));
                        if (c && at < s.wcap) wl[at] = make_uint4((uint32_t)v, row, q, 0u);
                        wcount += (uint32_t)__builtin_popcountll(hits);
                    }
                }
            }
        }
    };

    const uint64_t n_stages = my_tiles * nch;
    // the stage the loads are at, as (tile, kc), by running counters; past the block's last stage they repeat it (the waits count loads, not bytes)
    uint64_t itr = 0;
    uint32_t kcr = 0;
    auto advance_r = [&]() {
        if (kcr + 1 < nch) ++kcr;
        else if (itr + 1 < my_tiles) { kcr = 0; ++itr; }
    };
    i32x4q rq[SW_SLOTS], rc[SW_SLOTS][2];
    // ---- prologue: the loads of stages 0 .. 3; the queries of stage 0 into their buffer ----
#pragma unroll
    for (int d = 0; d < SW_AHEAD_STAGES; ++d) {
        load_stage(itr, kcr, rq[d], rc[d][0], rc[d][1]);
        advance_r();
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rq[0]), "+v"(rc[0][0]), "+v"(rc[0][1]) : : "memory");
    *reinterpret_cast<i32x4q *>(b_lds + (uint32_t)w * 64u + (uint32_t)lane) = rq[0];
    sw_stage_barrier();
    uint64_t it = 0;
    uint32_t kc = 0, bslot = 0;
    // one stage; I = g mod SW_SLOTS selects the register sets at compile time
    auto one_stage = [&](auto IC) __attribute__((always_inline)) {
        constexpr int I = decltype(IC)::value, IN = (I + 1) % SW_SLOTS, IL = (I + SW_AHEAD_STAGES) % SW_SLOTS;
        const bool first = kc == 0;
        if (first) {
            if (it) {
                // (the column entries were loaded a tile ago: 3 (nch - 1) loads were issued behind them before the last wait, which let 11 stay in flight)
                if (3 * (nch - 1) < 11) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bi8[0]), "+v"(bi8[1]) : : "memory");
                else asm volatile("" : "+v"(bi8[0]), "+v"(bi8[1]) : : "memory");
                epilogue(it - 1);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) acc[mt][nt] = (i32x4q){0, 0, 0, 0};
        }
        // stage g + 4 -> the register set stage g - 1 ran on
        load_stage(itr, kcr, rq[IL], rc[IL][0], rc[IL][1]);
        advance_r();
        // all but the last 11 loads have landed: this stage's codes and the next stage's queries among them (a tile's two column loads inside that window
        // only make the wait cover more)
        asm volatile("s_waitcnt vmcnt(11)" : "+v"(rq[IN]), "+v"(rc[I][0]), "+v"(rc[I][1]) : : "memory");
        static_assert(3 * SW_AHEAD_STAGES - 1 == 11, "the wait above");
        *reinterpret_cast<i32x4q *>(b_lds + (bslot ^ 1u) * SW_B_UNITS + (uint32_t)w * 64u + (uint32_t)lane) = rq[IN];      // the next stage's queries: this wave's 1 KiB
        const uint4 *bb = b_lds + bslot * SW_B_UNITS + b_rd;
        constexpr int SW_AHEAD = 3;
        i32x4q bv[SW_AHEAD + 1];
#pragma unroll
        for (int k = 0; k < SW_AHEAD; ++k) bv[k] = *reinterpret_cast<const i32x4q *>(bb + k * 64);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            if (nt + SW_AHEAD < 8) bv[(nt + SW_AHEAD) % (SW_AHEAD + 1)] = *reinterpret_cast<const i32x4q *>(bb + (nt + SW_AHEAD) * 64);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(rc[I][mt], bv[nt % (SW_AHEAD + 1)], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (first) load_bi(it);      // (after the epilogue that read the previous tile's)
        sw_stage_barrier();
        if (++kc == nch) { kc = 0; ++it; }
        bslot ^= 1u;
    };
    for (uint64_t g = 0; g < n_stages; g += SW_SLOTS) {
        one_stage(std::integral_constant<int, 0>{});
        if (g + 1 < n_stages) one_stage(std::integral_constant<int, 1>{});
        if (g + 2 < n_stages) one_stage(std::integral_constant<int, 2>{});
        if (g + 3 < n_stages) one_stage(std::integral_constant<int, 3>{});
        if (g + 4 < n_stages) one_stage(std::integral_constant<int, 4>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bi8[0]), "+v"(bi8[1]) : : "memory");
    epilogue(my_tiles - 1);
    if (lane == 0) s.wcnt[blockIdx.x * (SW_THREADS / 64) + (uint32_t)w] = wcount;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- after the scan: an entry (dot, row, query) becomes (key lo, key hi, query) when its score - SqOps::finish (scan_sq_mfma.hip), operation for operation -
// is not below the query's threshold score (ties pass), else an entry the regroup skips (query 0xFFFFFFFF).  One wave per list. ----
__global__ __launch_bounds__(256) void sqw_finish_kernel(uint4 *wlist, const uint32_t *wcnt, uint32_t wcap, uint32_t n_lists, float multiplier, const float *row_offsets,
                                                         const float *qinfo) {
    const uint32_t l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= n_lists) return;
    uint32_t cnt = wcnt[l];
    cnt = cnt < wcap ? cnt : wcap;
    uint4 *list = wlist + (uint64_t)l * wcap;
    for (uint32_t i = threadIdx.x & 63u; i < cnt; i += 64) {
        const uint4 e = list[i];
        const uint32_t row = e.y, q = e.z;
        const float m1 = multiplier * (float)(int32_t)e.x;
        const float mq = m1 + qinfo[q];
        const float score = mq + row_offsets[row];
        if (!(score < qinfo[SW_QT + q])) {
            const uint64_t key = make_key(score, row);
            list[i] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), q, 0u);
        } else {
            list[i] = make_uint4(0u, 0u, 0xFFFFFFFFu, 0u);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
bool sqw_shape_ok(const ScanArgs &a) {
    return a.dim >= 64 && a.dim % 64 == 0 && a.row_stride % 16 == 0 && a.row_stride * (SW_BM - 1) + a.dim < (1ull << 31) && a.ids == nullptr &&
           a.top <= MAX_TOP_FAST && a.row_offsets != nullptr && a.sq_multiplier > 0.0f && a.sq_multiplier < __builtin_inff() && a.n_cand >= 1 &&
           a.n_cand < 0xFFFFFFFFull && (uint64_t)127 * 127 * a.dim < (1ull << 24);
}
size_t sqw_query_bytes(uint32_t dim) { return (size_t)(dim / 64) * SW_B_UNITS * 16; }
size_t sqw_wlists_counts_bytes(int num_cus) { return ((size_t)num_cus * (SW_THREADS / 64) * 4 + 255) / 256 * 256; }
size_t sqw_wlists_bytes(int num_cus) { return sqw_wlists_counts_bytes(num_cus) + (size_t)num_cus * (SW_THREADS / 64) * SW_WCAP * 16; }
uint32_t sqw_wcap() { return SW_WCAP; }

int32_t launch_sqw_stats(hipStream_t st, const float *d_off, uint64_t n, float multiplier, int32_t *d_bi, uint32_t *d_stats) {
    ::qmx::clear_stale_error();
    const uint32_t grid = (uint32_t)std::min<uint64_t>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(sqw_stats_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, d_off, n, multiplier, d_bi, d_stats);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

int32_t launch_sqw_pack(hipStream_t st, const ScanArgs &a, const uint64_t *d_gthr, float off_absmax, void *d_bq, int32_t *d_thr_i, float *d_qinfo, float *d_band,
                        uint32_t *d_cand_cnt, uint32_t n_cnt) {
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sqw_pack_kernel, dim3(SW_QT), dim3(256), 0, st, reinterpret_cast<const unsigned char *>(a.queries), a.q_stride, a.aux_off, a.nq, a.dim / 64,
                       d_gthr, a.sq_multiplier, off_absmax, a.dim, (uint4 *)d_bq, d_thr_i, d_qinfo, d_band, d_cand_cnt, n_cnt);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// d_wlists: [counts: sqw_wlists_counts_bytes][lists]; *grid_out = blocks launched (8 lists each)
int32_t launch_scan_sqw(hipStream_t st, const ScanArgs &a, const void *d_bq, const int32_t *d_thr_i, const int32_t *d_bi, const float *d_qinfo, int num_cus,
                        void *d_wlists, uint32_t *grid_out) {
    QMX_REQUIRE(sqw_shape_ok(a), QMX_ERR_NOT_SUPPORTED, "SQ wide scan: shape not supported");
    SqWideArgs s;
    s.bq = (const uint4 *)d_bq;
    s.nch = a.dim / 64;
    s.nq = a.nq;
    s.thr_i = d_thr_i;
    s.bi = d_bi;
    s.wcnt = (uint32_t *)d_wlists;
    s.wlist = (uint4 *)((unsigned char *)d_wlists + sqw_wlists_counts_bytes(num_cus));
    s.wcap = SW_WCAP;
    const uint64_t n_tiles = (a.n_cand + SW_BM - 1) / SW_BM;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)num_cus, n_tiles);
    static thread_local DeviceOnce once;
    ::qmx::clear_stale_error();
    auto kfn = scan_sqw_kernel;
    if (once.need()) {
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS));
        once.mark();
    }
    QMX_NOTE_KERNEL(kfn);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(SW_THREADS), SW_LDS, st, a, s);
    QMX_HIP(hipGetLastError());
    const uint32_t n_lists = grid * (SW_THREADS / 64);
    hipLaunchKernelGGL(sqw_finish_kernel, dim3((n_lists + 3) / 4), dim3(256), 0, st, s.wlist, s.wcnt, s.wcap, n_lists, a.sq_multiplier, a.row_offsets, d_qinfo);
    QMX_HIP(hipGetLastError());
    if (grid_out) *grid_out = grid;
    return QMX_OK;
}

}  // namespace qmx

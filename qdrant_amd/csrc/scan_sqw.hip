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
// passes, 5.0 ms.  Here a wave owns 32 rows of a 256-row tile, its lanes fetch exactly the 16-byte operand pieces v_mfma_i32_16x16x64_i8 wants from them
// (plain 16-byte loads into the operand registers themselves, two 128-code groups ahead), and multiplies them with all 128 queries (their codes as
// B-operand images RESIDENT in LDS: 96 KiB at 768 codes, copied once per block - no barrier in the loop): one pass of the block per 128 queries, 1.33 ms =
// 0.72 of HBM on the codes.
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
constexpr int SW_MAX_CODES = 1024;                           // sq_mfma_ok: 127^2 x codes < 2^24
constexpr int SW_LDS_MAX = SW_MAX_CODES / 64 * SW_B_UNITS * 16 + SW_QT * 4;      // the queries' whole image (8 KiB per 64 codes) + the 128 integer bounds: 128.5 KiB
constexpr uint32_t SW_WCAP = 8192;                           // candidates one wave may list per pass

// unit index of (16-query tile t, piece kq, query-in-tile m) inside a stage of the queries' image: scan_split.hip sp_unit's swizzle, conflict-free for the operand reads
__device__ __forceinline__ uint32_t sw_unit(uint32_t t, uint32_t kq, uint32_t m) { return (t * 4 + kq) * 16 + (m ^ (2 * kq)); }

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

// (default cache policy: a load instruction touches sixteen 64-byte HALF-lines whose other halves the group's next load is meant to hit - with `nt` the pass takes 1.49 ms
// instead of 1.37, profiles/r6_nt_policy_experiment.txt; the int8-copy scans, whose loads consume whole lines, gain from it: scan_split.hip)
__device__ __forceinline__ void sw_gload16(i32x4q &dst, const unsigned char *sbase, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

// ---- The scan.  Block = 8 waves, one block per CU, persistent over 256-row tiles; a tile = nch stages of 64 codes.  The queries' WHOLE operand image is
// RESIDENT in LDS (nch x 8 KiB: 96 KiB at 768 codes, 128 KiB at the 1 024 sq_mfma_ok admits), copied once per block.  After that copy the waves share
// nothing, so there is no stage barrier and no lockstep: a wave OWNS 32 rows of the tile, lane (m, kg) loads, per stage, the 16 code bytes [16 kg, +16) of
// rows m and 16 + m - exactly its operand registers of the stage's two row tiles - and multiplies them with all 128 queries: 16 matrix instructions and 8
// operand reads per wave and stage; while one wave of a SIMD waits for its codes the other multiplies.
// The loads are `global_load_dwordx4` issued by inline asm INTO the operand registers and waited for by the kernel's own `s_waitcnt vmcnt` (asm statements
// that name the registers they release as in-out operands, so no use or copy of them can be scheduled in front of the wait; hipcc's own waits would be
// conservative across the loop's back edge).  They are issued by GROUPS of G stages, AHEAD groups in front of the matrix work: the two loads of a row's
// 128-byte line leave the wave back to back instead of a stage apart.  vmcnt is IN-ORDER: the wait of group g lets the last 2 G AHEAD loads stay in flight.
// Measured at 10 M x 768, 128 queries (profiles/r6_sqw_resident.md): staged queries + a barrier per stage 1.70 - 1.77 ms; resident, G = 1: 1.45 whatever
// the depth (AHEAD 4 .. 10); G = 2 / 4 / 6: 1.32 - 1.36 whatever the depth (AHEAD 1 .. 4); twelve waves per CU (168 registers: spills) 1.47. ----
template <int I, int N, class F>
__device__ __forceinline__ void sw_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sw_static_for<I + 1, N>(f);
    }
}
// vmcnt(n) for a wave-uniform even n <= 14 (the immediate has to be one)
__device__ __forceinline__ void sw_wait_vm_dyn(uint32_t n) {
    switch (n >> 1) {
#define SW_W(k) case k: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * k) : "memory"); break;
        SW_W(1) SW_W(2) SW_W(3) SW_W(4) SW_W(5) SW_W(6) SW_W(7)
#undef SW_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
template <int G, int AHEAD>
__global__ __launch_bounds__(SW_THREADS, 1) void scan_sqw_kernel(const ScanArgs a, const SqWideArgs s) {
    constexpr int WAVES = SW_THREADS / 64, T = SW_THREADS, BM = SW_BM;
    static_assert(BM == WAVES * 32, "a wave owns 32 rows of a tile");
    constexpr int SLOTS = AHEAD + 1;
    static_assert(2 * G * AHEAD <= 62, "vmcnt is six bits");
    static_assert(2 * G * (AHEAD - 1) <= 14, "sw_wait_vm_dyn covers the short-row case up to 14");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4 *lds = reinterpret_cast<uint4 *>(smem_raw);
    const uint32_t nch = s.nch;
    const uint32_t ngr = (nch + G - 1) / G;                          // groups per tile
    int32_t *thr_lds = reinterpret_cast<int32_t *>(smem_raw + (size_t)nch * SW_B_UNITS * 16);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t n_tiles = (a.n_cand + BM - 1) / BM;
    const uint64_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    if (my_tiles == 0) {
        if (lane == 0) s.wcnt[blockIdx.x * (WAVES) + (uint32_t)w] = 0;
        return;
    }
#pragma unroll 4
    for (uint32_t u = (uint32_t)tid; u < nch * SW_B_UNITS; u += T) lds[u] = s.bq[u];
    if (tid < SW_QT) thr_lds[tid] = s.thr_i[tid];
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // from here on the kernel counts its vector-memory traffic itself
    const uint32_t kq_r = (uint32_t)lane >> 4, m_r = (uint32_t)lane & 15u;
    const uint4 *const b_rd = lds + sw_unit(0, kq_r, m_r);
    const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
    const uint64_t last_row = a.n_cand - 1;
    const uint32_t row_stride32 = (uint32_t)a.row_stride;
    const uint32_t rl0 = (uint32_t)w * 32u + m_r, rl1 = rl0 + 16u;

    auto uniform_ptr = [&](uint64_t v) {
        return reinterpret_cast<const unsigned char *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
                                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
    };
    // One group's 2 G loads: the lane's code pieces of row tiles 0 and 1 of group b of the block's it-th tile (rows past the block: the last row's bytes,
    // their scores are dropped; stages past the row: the group's last piece again - the waits count loads)
    auto load_group = [&](uint64_t it, uint32_t b, i32x4q (&rc)[G][2]) __attribute__((always_inline)) {
        const uint64_t row0 = (blockIdx.x + it * gridDim.x) * BM;
        const unsigned char *csrc = uniform_ptr((uint64_t)(uintptr_t)(rows + row0 * a.row_stride + b * (G * 64u)));
        const uint32_t gsz = nch - b * G < (uint32_t)G ? nch - b * G : (uint32_t)G;
        uint32_t r0 = rl0, r1 = rl1;
        const uint64_t room = last_row - row0;                      // (row0 <= last_row: the tile exists)
        if (room < BM - 1) {                                     // the block's last, partial tile (wave-uniform)
            r0 = (uint64_t)rl0 < room ? rl0 : (uint32_t)room;
            r1 = (uint64_t)rl1 < room ? rl1 : (uint32_t)room;
        }
        const uint32_t o0 = r0 * row_stride32 + kq_r * 16u, o1 = r1 * row_stride32 + kq_r * 16u;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const uint32_t jj = (uint32_t)j < gsz ? (uint32_t)j : gsz - 1u;
            sw_gload16(rc[j][0], csrc, o0 + jj * 64u);
            sw_gload16(rc[j][1], csrc, o1 + jj * 64u);
        }
    };

    i32x4q acc[2][8];
    uint4 *const wl = s.wlist + (uint64_t)(blockIdx.x * (WAVES) + (uint32_t)w) * s.wcap;
    uint32_t wcount = 0;
    const uint32_t n_rows32 = (uint32_t)a.n_cand;
    // B of the lane's eight rows of the running tile (rows 4 kq_r .. + 3 of both 16-row tiles of the wave)
    i32x4q bi8[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    auto load_bi = [&](uint64_t it) __attribute__((always_inline)) {
        const uint64_t row0 = (blockIdx.x + it * gridDim.x) * BM;
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
        const uint32_t row0 = (uint32_t)(tile * BM) + (uint32_t)w * 32u + 4 * kq_r;
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

    const uint64_t n_groups = my_tiles * ngr;
    // the group the loads are at, as (tile, b), by running counters; past the block's last group they repeat it (the waits count loads, not bytes)
    uint64_t itr = 0;
    uint32_t br = 0;
    auto advance_r = [&]() {
        if (br + 1 < ngr) ++br;
        else if (itr + 1 < my_tiles) { br = 0; ++itr; }
    };
    i32x4q rc[SLOTS][G][2];
    // ---- prologue: the loads of groups 0 .. AHEAD - 1 ----
    sw_static_for<0, AHEAD>([&](auto DC) __attribute__((always_inline)) {
        load_group(itr, br, rc[decltype(DC)::value]);
        advance_r();
    });
    uint64_t it = 0;
    uint32_t b = 0;
    // one group; I = g mod SLOTS selects the register sets at compile time
    auto one_group = [&](auto IC) __attribute__((always_inline)) {
        constexpr int I = decltype(IC)::value, IL = (I + AHEAD) % SLOTS;
        if (b == 0) {
            if (it) {
                // the finished tile's column entries were loaded on ITS first group, in front of that group's code loads: 2 G ngr loads have been issued
                // behind them since, of which the last wait let 2 G AHEAD stay in flight
                if (ngr < (uint32_t)AHEAD) sw_wait_vm_dyn(2u * G * ngr);
                asm volatile("" : "+v"(bi8[0]), "+v"(bi8[1]) : : "memory");
                epilogue(it - 1);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) acc[mt][nt] = (i32x4q){0, 0, 0, 0};
            load_bi(it);
        }
        // group g + AHEAD -> the register set group g - 1 ran on
        load_group(itr, br, rc[IL]);
        advance_r();
        const uint32_t gsz = nch - b * G < (uint32_t)G ? nch - b * G : (uint32_t)G;
        // all but the last 2 G AHEAD loads have landed: this group's codes among them
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * G * AHEAD) : "memory");
#pragma unroll
        for (int J = 0; J < G; ++J) asm volatile("" : "+v"(rc[I][J][0]), "+v"(rc[I][J][1]) : : "memory");
#pragma unroll
        for (int J = 0; J < G; ++J) {
            if ((uint32_t)J < gsz) {
                const uint4 *bb = b_rd + (b * G + J) * SW_B_UNITS;
                constexpr int SW_AHEAD = 3;
                i32x4q bv[SW_AHEAD + 1];
#pragma unroll
                for (int k = 0; k < SW_AHEAD; ++k) bv[k] = *reinterpret_cast<const i32x4q *>(bb + k * 64);
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) {
                    if (nt + SW_AHEAD < 8) bv[(nt + SW_AHEAD) % (SW_AHEAD + 1)] = *reinterpret_cast<const i32x4q *>(bb + (nt + SW_AHEAD) * 64);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(rc[I][J][mt], bv[nt % (SW_AHEAD + 1)], acc[mt][nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (++b == ngr) { b = 0; ++it; }
    };
    for (uint64_t g = 0; g < n_groups; g += SLOTS) {
        sw_static_for<0, SLOTS>([&](auto IC) __attribute__((always_inline)) {
            if (g + decltype(IC)::value < n_groups) one_group(IC);
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bi8[0]), "+v"(bi8[1]) : : "memory");
    epilogue(my_tiles - 1);
    if (lane == 0) s.wcnt[blockIdx.x * (WAVES) + (uint32_t)w] = wcount;
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
           a.n_cand < 0xFFFFFFFFull && (uint64_t)127 * 127 * a.dim < (1ull << 24) && a.dim <= (uint32_t)SW_MAX_CODES;
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
    const size_t lds_bytes = (size_t)s.nch * SW_B_UNITS * 16 + SW_QT * 4;      // the queries' image + the 128 integer bounds
    static thread_local DeviceOnce once;
    ::qmx::clear_stale_error();
    auto kfn = scan_sqw_kernel<2, 2>;
    if (once.need()) {
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS_MAX));
        once.mark();
    }
    QMX_NOTE_KERNEL(kfn);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(SW_THREADS), lds_bytes, st, a, s);
    QMX_HIP(hipGetLastError());
    const uint32_t n_lists = grid * (SW_THREADS / 64);
    hipLaunchKernelGGL(sqw_finish_kernel, dim3((n_lists + 3) / 4), dim3(256), 0, st, s.wlist, s.wcnt, s.wcap, n_lists, a.sq_multiplier, a.row_offsets, d_qinfo);
    QMX_HIP(hipGetLastError());
    if (grid_out) *grid_out = grid;
    return QMX_OK;
}

}  // namespace qmx

// scan_tq4w.hip - EncodedVectorsTQ, 4 bits per value, brute-force top-k for LARGE query batches: 128 queries per pass of the code block, every row's
// 4-bit codes decoded ONCE per pass into the int8 operand registers of the matrix cores.
//
// Same reference loops as scan_tq.hip / scan_sq_mfma.hip TqOps<4>: BatchFilteredSearcher::peek_top_iter
// (lib/segment/src/index/hnsw_index/point_scorer.rs:423-472) over Query4bitSimd::dotprod (lib/quantization/src/turboquant/simd/query4bit/mod.rs:
// dot_raw = sum q_signed * c_u, q_signed = 128 high + low) and score_precomputed (turboquant/quantization.rs:569-620).  Every score is the exact integer
// dot of the row's codebook bytes with the two signed digits of the query, finished by TqOps<4>::finish's f32 expression: the same bits.
//
// Why.  The 32-query kernel (scan_sq_mfma.hip) decodes a row's nibbles once per 32 queries: 336 of its 660 vector instructions per 16-row tile are the
// decode (10 M x 768: 1.21 - 1.36 ms per 32 queries = 0.36 - 0.40 of HBM, decode-bound).  Here a wave owns 32 rows of a 256-row tile, decodes their codes
// straight into the operand registers its own matrix instructions read, and multiplies them with 128 queries x 2 digits (the queries' digits arrive as
// B-operand images by LDS-DMA from a 2 x 96 KiB image in L2): 64 matrix instructions, 32 operand reads and ~105 decode instructions per wave and stage.
// What bounds it is no longer HBM (3.84 GB per 128 queries) but the matrix cores and the issue slots beside them: 2 digits x 2 x 128 x 768 operations per
// row = 0.78 ms per 10 M rows at the int8 peak of 2.4 GHz, 0.97 ms at the 1.93 GHz the chip holds under this load; a 16-cycle matrix instruction leaves
// four issue slots per SIMD and the decode, the operand reads and the copy requests need ~3.5 of them (profiles/r6_tqw_*: 2.0 ms per 128 queries against
// 4.86 ms through the 32-query kernel).
//
// Scores are exact, so the pass needs no band: a pair is a CANDIDATE when its score is not below the k-th best score of a strided sample of the block
// (api_search.hip: the pre-scan every wide path starts with; ties pass), candidates go to per-wave lists in global memory (scan_i8copy_kernel's), are
// regrouped per query, the k best keys per query are selected, re-scored by the pair kernel and sorted - the tail of the int8 prefilter, with a band of
// zero.  The fast reject runs on integers: with sf in [sf_min, sf_max] over the segment (and l2 >= l2_min) a pair can only reach the threshold when
// low + 128 high >= thr_i[q], a bound tq4w_pack_kernel derives per query with the rounding of finish() on its side; and since |low| <= 64 C1 (C1 = the
// largest sum of |codebook bytes| of a row of the block), only when high >= (thr_i[q] - 64 C1) / 128 - the first test, on half the accumulators.
#include <type_traits>

#include "scan_common.hpp"

namespace qmx {

typedef int i32x4w __attribute__((ext_vector_type(4)));

constexpr int TW_THREADS = 512;
constexpr int TW_BM = 256;                                   // rows per tile
constexpr int TW_QT = 128;                                   // queries per pass
constexpr int TW_B_DIGIT_UNITS = TW_QT * 2 * 4;              // 16-byte units of one digit of the queries' stage (128 queries x {even, odd dims} x 4 pieces): 16 KiB
constexpr int TW_B_UNITS = 2 * TW_B_DIGIT_UNITS;             // low digits, high digits: 32 KiB
constexpr int TW_RP_UNITS = TW_BM * 8;                       // ... of a stage pair's codes: 256 rows x 128 bytes = 32 KiB
constexpr int TW_LDS = (2 * TW_B_UNITS + 2 * TW_RP_UNITS) * 16 + 2 * TW_QT * 4;      // 64 + 64 + 1 = 129 KiB
constexpr uint32_t TW_WCAP = 8192;                           // candidates one wave may list per pass

// unit index of (16-row or 16-query tile t, half hl, k-group kq, row-in-tile m): scan_split.hip sp_unit, the layout both operand reads are conflict-free in
__device__ __forceinline__ uint32_t tw_unit(uint32_t t, uint32_t hl, uint32_t kq, uint32_t m) { return ((t * 2 + hl) * 4 + kq) * 16 + (m ^ (2 * kq)); }

typedef __attribute__((address_space(3))) unsigned char tw_lds_byte;
// 1 KiB of global memory (wave-uniform base, lane i fetches bytes 16 i ..) straight into LDS at the wave-uniform byte address lds_dst (scan_split.hip sp_glds16)
__device__ __forceinline__ void tw_glds16(const unsigned char *src, uint32_t lane_off, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(src), "s"(lds_dst) : "memory");
}
// LDS traffic is all that crosses a stage barrier (scan_split.hip sp_stage_barrier: __syncthreads() would drain the row stream)
__device__ __forceinline__ void tw_stage_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct TqWideArgs {
    const uint4 *bq;          // [nch][2 digits][TW_B_DIGIT_UNITS] the queries' operand images (tq4w_pack_kernel)
    uint32_t nch;             // stages per tile: code bytes of a row / 64 (even)
    uint32_t nq;              // live queries (<= 128)
    const int32_t *thr_i;     // [128] a pair whose low + 128 high is below this cannot reach the query's threshold; [128 .. 256): the same bound on the high sum
                              // alone (|low sum| <= 64 C1, C1 = the block's largest sum of |codebook bytes| of a row)
    const float *qinfo;       // [4][128] f0, ec, qlsq, thr_f (the threshold score: ties pass)
    uint4 *wlist;             // [waves][wcap] (key lo, key hi, query, 0)
    uint32_t *wcnt;           // [waves] entries each wave wanted to append (may run past wcap: overflow)
    uint32_t wcap;
};

// ---- once per segment: stats[0] = min sf, [1] = max sf, [2] = min l2, [5] = max l2 (uint bits of positive floats order like the floats), [3] != 0: a value that is
// not a positive finite number (no wide pass for this block) ----
__global__ __launch_bounds__(256) void tq4w_stats_kernel(const float *sf, const float *l2, uint64_t n, uint32_t *stats) {
    float lo = __builtin_inff(), hi = 0.0f, l2lo = __builtin_inff(), l2hi = 0.0f;
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const float v = sf[i];
        bad = bad || !(v > 0.0f && v < __builtin_inff());
        lo = __builtin_fminf(lo, v);
        hi = __builtin_fmaxf(hi, v);
        if (l2) {
            const float w = l2[i];
            bad = bad || !(w >= 0.0f && w < __builtin_inff());
            l2lo = __builtin_fminf(l2lo, w);
            l2hi = __builtin_fmaxf(l2hi, w);
        }
    }
    for (int o = 32; o >= 1; o >>= 1) {
        lo = __builtin_fminf(lo, __shfl_xor(lo, o, 64));
        hi = __builtin_fmaxf(hi, __shfl_xor(hi, o, 64));
        l2lo = __builtin_fminf(l2lo, __shfl_xor(l2lo, o, 64));
        l2hi = __builtin_fmaxf(l2hi, __shfl_xor(l2hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0 && !bad) {
        atomicMin(&stats[0], __float_as_uint(lo));
        atomicMax(&stats[1], __float_as_uint(hi));
        if (l2) {
            atomicMin(&stats[2], __float_as_uint(l2lo));
            atomicMax(&stats[5], __float_as_uint(l2hi));
        }
    }
    if (bad) atomicOr(&stats[3], 1u);
}

// ---- once per segment: stats[4] = C1 = max over the rows of sum_i |codebook byte of code i| (all code bytes of the device row, the zero padding included).
// One wave per row at a time; |c| through the same byte permutes as the decode, summed by v_sad_u8 ----
__global__ __launch_bounds__(256) void tq4w_c1_kernel(const unsigned char *rows, uint64_t row_stride, uint64_t n, uint32_t code_bytes, uint32_t *stats) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4;
    uint32_t best = 0;
    for (uint64_t r = wave; r < n; r += nwaves) {
        const unsigned char *row = rows + r * row_stride;
        uint32_t sum = 0;
        for (uint32_t b = (uint32_t)lane * 16u; b < code_bytes; b += 1024u) {
            const uint4 v = *reinterpret_cast<const uint4 *>(row + b);
            const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int sh = 0; sh < 8; sh += 4) {
                    const uint32_t sel = (x[k] >> sh) & 0x07070707u;
                    const uint32_t lo = __builtin_amdgcn_perm(0x06121F2Cu, 0x3B4C6180u, sel), hi = __builtin_amdgcn_perm(0x7F614C3Bu, 0x2C1F1206u, sel);   // |c| of codes 0..7, 8..15
                    const uint32_t ab = __builtin_amdgcn_perm(hi, lo, ((x[k] >> (sh + 1)) & 0x04040404u) | 0x03020100u);
                    sum = __builtin_amdgcn_sad_u8(ab, 0u, sum);
                }
        }
        for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
        best = sum > best ? sum : best;
    }
    if (lane == 0) atomicMax(&stats[4], best);
}

// ---- once per 128-query tile: one block per query slot.  The query's digits in the B-operand images, its integer reject bound, what finish() needs ----
// A query entry (scan_tq.hip tq_query_encode_kernel) holds, per 16-byte row piece P, 64 bytes: [low digits of the even dims][low, odd][high, even][high, odd];
// stage kc of the scan covers row pieces 4 kc .. 4 kc + 3, one MFMA the even (or the odd) dims of the four: unit (query tile, eo, p, query) of digit D.
__global__ __launch_bounds__(256) void tq4w_pack_kernel(const unsigned char *queries, uint32_t q_stride, uint32_t aux_off, uint32_t nq, uint32_t nch,
                                                        const uint64_t *gthr, float sf_min, float sf_max, float l2_min, float l2_max, uint32_t c1, int is_l2, int high_only, uint4 *bq, int32_t *thr_i,
                                                        float *qinfo, float *band, uint32_t *cand_cnt, uint32_t n_cnt) {
    const uint32_t qi = blockIdx.x;
    const bool live = qi < nq;
    if (qi == 0)
        for (uint32_t i = threadIdx.x; i < n_cnt; i += 256) cand_cnt[i] = 0;
    const unsigned char *entry = queries + (uint64_t)qi * q_stride;
    for (uint32_t u = threadIdx.x; u < nch * 16; u += 256) {
        const uint32_t kc = u >> 4, dg = (u >> 3) & 1u, eo = (u >> 2) & 1u, p = u & 3u;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (live) v = *reinterpret_cast<const uint4 *>(entry + (uint64_t)(4 * kc + p) * 64 + dg * 32 + eo * 16);
        bq[(uint64_t)kc * TW_B_UNITS + dg * TW_B_DIGIT_UNITS + tw_unit(qi >> 4, eo, p, qi & 15u)] = v;
    }
    if (threadIdx.x != 0) return;
    float f0 = 0.0f, ec = 0.0f, qlsq = 0.0f, tf = __builtin_inff(), bd = 0.0f;
    int32_t ti = 0x7FFFFFFF;                                         // a dead slot passes nothing
    if (live) {
        const QueryAux *aux = reinterpret_cast<const QueryAux *>(entry + aux_off);
        f0 = aux->f0;
        ec = __uint_as_float(aux->pad[3]);
        const float ql = __uint_as_float(aux->pad[0]);
        qlsq = ql * ql;
        const uint64_t k = gthr[qi];
        // no bound (the sample holds fewer than k live rows) or a query the integer bound cannot be derived for (a zero query: f0 = 0): no candidates, and
        // the infinite band sends the query - alone - to the 32-query scan (sp_select_kernel)
        bd = __builtin_inff();
        const float t = k ? key_score(k) : 0.0f;
        if (k != 0 && t == t && f0 > 0.0f && f0 < __builtin_inff() && ec == ec) {
            tf = t;
            bd = 0.0f;
            ti = (int32_t)0x80000000;
            // dot / cosine: score = dot * sf >= T.   L2 (inverted): -((qlsq + l2^2) - (2 dot) sf) >= T  =>  dot * sf >= (qlsq + l2_min^2 + T) / 2 = W
            // (finish() rounds a handful of times, every operation monotone: 2^-18 of slack on each quantity covers them all)
            const double eps = 3.8146972656e-6;
            double w;
            if (is_l2) {
                const double base = ((double)qlsq + (double)l2_min * (double)l2_min) * (1.0 - eps);
                w = 0.5 * (base + (double)t - __builtin_fabs((double)t) * eps);
            } else {
                w = (double)t;
            }
            w -= __builtin_fabs(w) * eps;
            const double dot_thr = w > 0.0 ? w / (double)sf_max : w / (double)sf_min;          // the smallest dot that can still reach W with a row's sf
            const double s_thr = (dot_thr - (double)ec) / (double)f0 - ((__builtin_fabs(dot_thr) + 2.0 * __builtin_fabs((double)ec)) * eps / (double)f0 + 4.0);
            if (s_thr == s_thr) ti = s_thr <= -2147483647.0 ? (int32_t)0x80000000 : s_thr >= 2147483520.0 ? 0x7FFFFFFF : (int32_t)__builtin_floor(s_thr);
            if (high_only) {
                // the pass multiplies the HIGH digits only: its score differs from the exact one by f0 * low * sf (twice that under L2), |low| <= 64 C1 -
                // the band of the selection - plus the roundings of both f32 expressions (sizes: the threshold, the band, ec * sf, |q|^2 + |v|^2 under L2)
                const double lowmax = 64.0 * (double)c1 * (double)f0 * (double)sf_max * (is_l2 ? 2.0 : 1.0);
                const double sizes = __builtin_fabs((double)t) + lowmax + 2.0 * __builtin_fabs((double)ec) * (double)sf_max +
                                     (is_l2 ? (double)qlsq + (double)l2_max * (double)l2_max : 0.0);
                const double b = lowmax * 1.0001 + sizes * 2.0e-5;
                bd = b < 3.0e38 ? (float)b * 1.000001f : __builtin_inff();
                if (!(bd < 3.0e38f)) { ti = 0x7FFFFFFF; tf = __builtin_inff(); }
            }
        }
    }
    thr_i[qi] = ti;
    {   // 128 high + low >= ti with |low| <= 64 c1  =>  high >= (ti - 64 c1) / 128
        int32_t th = ti;
        if (ti != 0x7FFFFFFF && ti != (int32_t)0x80000000) {
            const int64_t num = (int64_t)ti - 64 * (int64_t)c1;
            const int64_t fl = num >= 0 ? num / 128 : -((-num + 127) / 128);
            th = fl < -2147483647ll ? (int32_t)0x80000000 : (int32_t)fl;
        }
        thr_i[TW_QT + qi] = th;
    }
    band[qi] = bd;
    qinfo[qi] = f0;
    qinfo[TW_QT + qi] = ec;
    qinfo[2 * TW_QT + qi] = qlsq;
    qinfo[3 * TW_QT + qi] = tf;
}

// The codebook bytes of the LOW (SH = 0) or HIGH (SH = 4) nibbles of x's four bytes (tq_policies.hpp tq4_lookup with the nibble masks folded in: selector
// byte = nibble & 7 into each half of the table, then byte i of the low half or - where bit 3 of the nibble is set - of the high half): 6 / 7 instructions
template <int SH>
__device__ __forceinline__ uint32_t tw_lut4(uint32_t x) {
    const uint32_t s = (SH ? x >> SH : x) & 0x07070707u;
    const uint32_t lo = __builtin_amdgcn_perm(0xFAEEE1D4u, 0xC5B49F80u, s), hi = __builtin_amdgcn_perm(0x7F614C3Bu, 0x2C1F1206u, s);
    return __builtin_amdgcn_perm(hi, lo, ((x >> (SH + 1)) & 0x04040404u) | 0x03020100u);
}

// The scan.  Block = 8 waves, one block per CU, persistent over 256-row tiles; a tile = nch stages of 128 coordinates (64 code bytes per row).
// A wave OWNS 32 rows of the tile and multiplies them with all 128 queries: lane (m, kg) of the wave fetches the 16 code bytes [16 kg, 16 kg + 16) of
// rows m and 16 + m of the stage and decodes them into exactly the operand registers v_mfma_i32_16x16x64_i8 wants from it - the even dims of that piece
// for one instruction, the odd dims for the next (the k order inside an instruction is free as long as both operands agree: tq4w_pack_kernel lays the
// queries' digits out the same way).  So the decoded rows never travel through LDS: no stores, no operand reads for the A side, and the decode is paid
// once per 128 queries.  (The first version kept a decoded 256-row stage in LDS, shared by 4 x 2 waves: its 32 KiB of ds_write_b128 per stage went through
// the 79 B/clk store path and cost as much as half a stage's matrix work - profiles/r6_tqw_*.)
// EVERYTHING the loop fetches arrives by LDS-DMA and is counted by the kernel itself (one plain vector load inside the loop and the compiler's own
// conservative `s_waitcnt vmcnt(0)` drains the streams at every stage).  Per stage g a wave
//   1. reads its own 2 x 16 code bytes of stage g + 1 from its lane-private staging area,
//   2. runs 16 groups of (two operand reads of the queries' image, four matrix instructions, one table lookup of the next stage's decode), with the
//      stage's six copy requests behind the first groups: four 1 KiB pieces of the queries' images of stage g + 1 (from the L2-resident image) and two
//      1 KiB pieces of its rows' codes three to four stages ahead (the two 64-byte halves of a row's 128-byte line are asked for back to back),
//   3. waits for the queries of stage g + 1 and meets the others at the stage barrier (the queries' buffers are all the waves share).
// LDS: queries 2 x 32 KiB, code staging 2 stage pairs x 32 KiB, the 128 queries' two integer bounds 1 KiB.
// The epilogue of a tile lists (low + 128 high, row, query) of every pair that meets the query's integer bound; tq4w_finish_kernel turns the entries into
// keys (TqOps<4>::finish, the exact compare with the threshold score) before the regroup.
template <bool HO /* the high digits only: half the matrix work, scores within the band tq4w_pack_kernel states; the survivors are re-scored exactly */>
__global__ __launch_bounds__(TW_THREADS, 1) void scan_tq4w_kernel(const ScanArgs a, const TqWideArgs s) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4 *lds = reinterpret_cast<uint4 *>(smem_raw);
    int32_t *thr_lds = reinterpret_cast<int32_t *>(smem_raw + (size_t)(2 * TW_B_UNITS + 2 * TW_RP_UNITS) * 16);      // [256]: thr_i, thr_h
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t n_tiles = (a.n_cand + TW_BM - 1) / TW_BM;
    const uint32_t nch = s.nch, npair = nch / 2;
    const uint64_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    if (my_tiles == 0) {
        if (lane == 0) s.wcnt[blockIdx.x * (TW_THREADS / 64) + (uint32_t)w] = 0;
        return;
    }
    if (tid < 2 * TW_QT) thr_lds[tid] = s.thr_i[tid];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // from here on the kernel counts its vector-memory traffic itself
    const uint32_t kq_r = (uint32_t)lane >> 4, m_r = (uint32_t)lane & 15u;
    const uint32_t b_rd = tw_unit(0, 0, kq_r, m_r);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(tw_lds_byte *)smem_raw;
    const uint32_t lane_off = (uint32_t)lane * 16u;
    uint4 *const b_lds = lds, *const r_lds = lds + 2 * TW_B_UNITS;
    const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
    const uint64_t last_row = a.n_cand - 1;
    const uint32_t row_stride32 = (uint32_t)a.row_stride;
    // the lane's rows inside the tile (one per 16-row tile mt of the wave), its piece of a stage
    const uint32_t rl0 = (uint32_t)w * 32u + m_r, rl1 = rl0 + 16u;
    const uint32_t coff0 = rl0 * row_stride32 + kq_r * 16u, coff1 = rl1 * row_stride32 + kq_r * 16u;

    auto uniform_ptr = [&](uint64_t v) {
        return reinterpret_cast<const unsigned char *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
                                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
    };
    // the queries' images of stage kc -> B buffer `slot`: this wave's 4 KiB of the 32, four 1 KiB pieces
    const unsigned char *rq_src = nullptr;
    uint32_t rq_dst = 0;
    auto queries_begin = [&](uint32_t kc, uint32_t slot) {
        // (HO: the high digits' half of the image alone, 2 KiB per wave)
        const uint32_t off = HO ? TW_B_DIGIT_UNITS * 16u + (uint32_t)w * 2048u : (uint32_t)w * 4096u;
        rq_src = uniform_ptr((uint64_t)(uintptr_t)(s.bq + (uint64_t)kc * TW_B_UNITS) + off);
        rq_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + (slot * TW_B_UNITS) * 16u + off));
    };
    auto queries_piece = [&](int i) { tw_glds16(rq_src + i * 1024, lane_off, (uint32_t)__builtin_amdgcn_readfirstlane((int)(rq_dst + i * 1024))); };
    // this wave's codes of stage pair `kp` of the block's it-th tile -> staging slot `slot`: [stage of the pair][mt][wave][lane], a lane's own 16 bytes;
    // pieces 0, 1 = the first stage of the pair (mt 0, 1), pieces 2, 3 = the second (rows past the block: the last row's bytes, their scores are dropped)
    const unsigned char *rc_src = nullptr;
    uint32_t rc_dst = 0, rc_o0 = coff0, rc_o1 = coff1;
    auto codes_begin = [&](uint64_t it, uint32_t kp, uint32_t slot) {
        const uint64_t row0 = (blockIdx.x + it * gridDim.x) * TW_BM;
        rc_src = uniform_ptr((uint64_t)(uintptr_t)(rows + row0 * a.row_stride + kp * 128u));
        rc_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + (2 * TW_B_UNITS + slot * TW_RP_UNITS) * 16u + (uint32_t)w * 1024u));
        rc_o0 = coff0;
        rc_o1 = coff1;
        const uint64_t room = last_row - row0;                      // (row0 <= last_row: the tile exists)
        if (room < TW_BM - 1) {                                     // the block's last, partial tile (wave-uniform)
            const uint32_t r0 = (uint64_t)rl0 < room ? rl0 : (uint32_t)room, r1 = (uint64_t)rl1 < room ? rl1 : (uint32_t)room;
            rc_o0 = r0 * row_stride32 + kq_r * 16u;
            rc_o1 = r1 * row_stride32 + kq_r * 16u;
        }
    };
    auto codes_piece = [&](int i) { tw_glds16(rc_src + (i >> 1) * 64, (i & 1) ? rc_o1 : rc_o0, (uint32_t)__builtin_amdgcn_readfirstlane((int)(rc_dst + (uint32_t)i * 8192u))); };
    // the lane's two code units of stage `sp` of the pair in staging slot `slot`
    auto read_codes = [&](uint32_t slot, uint32_t sp, uint4 &c0, uint4 &c1) {
        const uint4 *src = r_lds + slot * TW_RP_UNITS + sp * 1024u + (uint32_t)tid;
        c0 = src[0];
        c1 = src[512];
    };

    i32x4w accl[2][8], acch[2][8];
    uint4 *const wl = s.wlist + (uint64_t)(blockIdx.x * (TW_THREADS / 64) + (uint32_t)w) * s.wcap;
    uint32_t wcount = 0;
    const uint32_t n_rows32 = (uint32_t)a.n_cand;

    // The epilogue of a tile.  Level 1 on the high sums alone (64 of the lane's 128 accumulators): |low sum| <= 64 C1 whatever the row; then the whole
    // sum, narrowing by wave-uniform steps (query tile, 16-row tile, the four rows of a lane) as scan_i8copy_kernel does.
    auto epilogue = [&](uint64_t it) __attribute__((always_inline)) {
        const uint64_t tile = blockIdx.x + it * gridDim.x;
        const uint32_t row0 = (uint32_t)(tile * TW_BM) + (uint32_t)w * 32u + 4 * kq_r;
        uint32_t hits8 = 0;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            int mx = (int)0x80000000;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) mx = acch[mt][nt][j] > mx ? acch[mt][nt][j] : mx;
            if (mx >= thr_lds[TW_QT + nt * 16 + (int)m_r]) hits8 |= 1u << nt;
        }
        if (!__ballot(hits8 != 0)) return;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            if (!__ballot((hits8 >> nt) & 1u)) continue;
            const uint32_t q = (uint32_t)nt * 16 + m_r;
            const int ti = thr_lds[HO ? TW_QT + q : q];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                int m4 = (int)0x80000000;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int v = HO ? acch[mt][nt][j] : (acch[mt][nt][j] << 7) + accl[mt][nt][j];
                    m4 = v > m4 ? v : m4;
                }
                if (!__ballot(m4 >= ti)) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int v = HO ? acch[mt][nt][j] : (acch[mt][nt][j] << 7) + accl[mt][nt][j];
                    const uint32_t row = row0 + (uint32_t)mt * 16 + (uint32_t)j;
                    const bool c = v >= ti && row < n_rows32 && q < s.nq;
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

    // operand registers of the current stage (even / odd dims of the lane's piece, rows m and 16 + m) and - being decoded - of the next
    uint4 ae[2], ao[2], ne[2], no[2];
    uint32_t slot_piece0 = 0;      // the first of the two code pieces the running stage asks for
    auto decode_all = [&](const uint4 &c0, const uint4 &c1) {
        const uint32_t x0[4] = {c0.x, c0.y, c0.z, c0.w}, x1[4] = {c1.x, c1.y, c1.z, c1.w};
        uint32_t e0[4], o0[4], e1[4], o1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            e0[k] = tw_lut4<0>(x0[k]);
            o0[k] = tw_lut4<4>(x0[k]);
            e1[k] = tw_lut4<0>(x1[k]);
            o1[k] = tw_lut4<4>(x1[k]);
        }
        ae[0] = make_uint4(e0[0], e0[1], e0[2], e0[3]);
        ao[0] = make_uint4(o0[0], o0[1], o0[2], o0[3]);
        ae[1] = make_uint4(e1[0], e1[1], e1[2], e1[3]);
        ao[1] = make_uint4(o1[0], o1[1], o1[2], o1[3]);
    };
    auto as_i32x4 = [](const uint4 &v) { return (i32x4w){(int)v.x, (int)v.y, (int)v.z, (int)v.w}; };

    // One stage on the queries' buffer `slot`: 16 groups (query tile nt, even / odd dims) of four matrix instructions; behind group k one table lookup
    // of the next stage's decode (word k of the lane's 16: [mt][even / odd][word]) and, behind the first six, the stage's copy requests.
    auto stage = [&](uint32_t slot, const uint4 &c0, const uint4 &c1, const uint4 (&ce)[2], const uint4 (&co)[2], uint4 (&de)[2], uint4 (&dd)[2]) __attribute__((always_inline)) {
        const uint4 *bb = b_lds + slot * TW_B_UNITS + b_rd;
        const uint32_t xs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        uint32_t dw[16];
        // the queries' operands run TW_AHEAD groups ahead of the matrix instructions that use them (a group lasts 64 - 128 cycles, an LDS read under load longer)
        constexpr int TW_AHEAD = 3;
        i32x4w bl[TW_AHEAD + 1], bh[TW_AHEAD + 1];
        auto b_read = [&](int k) {
            const int n2 = k >> 1, e2 = k & 1;
            if (!HO) bl[k % (TW_AHEAD + 1)] = *reinterpret_cast<const i32x4w *>(bb + n2 * 128 + e2 * 64);
            bh[k % (TW_AHEAD + 1)] = *reinterpret_cast<const i32x4w *>(bb + TW_B_DIGIT_UNITS + n2 * 128 + e2 * 64);
        };
#pragma unroll
        for (int k = 0; k < TW_AHEAD; ++k) b_read(k);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int nt = k >> 1, eo = k & 1;
            if (k + TW_AHEAD < 16) b_read(k + TW_AHEAD);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const i32x4w av = as_i32x4(eo ? co[mt] : ce[mt]);
                if (!HO) accl[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bl[k % (TW_AHEAD + 1)], accl[mt][nt], 0, 0, 0);
                acch[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bh[k % (TW_AHEAD + 1)], acch[mt][nt], 0, 0, 0);
            }
            {   // word k of the next stage's operands: mt = k >> 3, even / odd = (k >> 2) & 1, word = k & 3
                const uint32_t x = xs[(k >> 3) * 4 + (k & 3)];
                dw[k] = ((k >> 2) & 1) ? tw_lut4<4>(x) : tw_lut4<0>(x);
            }
            // the stage's copy requests, queries first (the end-of-stage wait relies on the order)
            constexpr int NQP = HO ? 2 : 4;
            if (k < NQP) queries_piece(k);
            else if (k < NQP + 2) codes_piece((int)(slot_piece0 + (uint32_t)(k - NQP)));
            __builtin_amdgcn_sched_barrier(0);
        }
        de[0] = make_uint4(dw[0], dw[1], dw[2], dw[3]);
        dd[0] = make_uint4(dw[4], dw[5], dw[6], dw[7]);
        de[1] = make_uint4(dw[8], dw[9], dw[10], dw[11]);
        dd[1] = make_uint4(dw[12], dw[13], dw[14], dw[15]);
    };

    const uint64_t n_stages = my_tiles * nch, n_pairs = my_tiles * npair;
    // (pair P as (tile, pair of the tile); past the block's last pair the requests repeat it: the waits count requests, not bytes)
    auto pair_at = [&](uint64_t P, uint64_t &it_out, uint32_t &kp_out) {
        const uint64_t Pc = P < n_pairs ? P : n_pairs - 1;
        it_out = Pc / npair;
        kp_out = (uint32_t)(Pc % npair);
    };
    // ---- prologue: the codes of pairs 0 and 1, the queries of stage 0; stage 0 decoded ----
    {
        uint64_t itp;
        uint32_t kpp;
        pair_at(0, itp, kpp);
        codes_begin(itp, kpp, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) codes_piece(i);
        pair_at(1, itp, kpp);
        codes_begin(itp, kpp, 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) codes_piece(i);
        queries_begin(0, 0);
#pragma unroll
        for (int i = 0; i < (HO ? 2 : 4); ++i) queries_piece(i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint4 c0, c1;
        read_codes(0, 0, c0, c1);
        decode_all(c0, c1);
    }
    tw_stage_barrier();
    uint64_t it = 0, P = 0, it2 = 0;
    uint32_t kc = 0, kp2 = 0;
    pair_at(2, it2, kp2);
    // one stage of the loop; `sp` = its place inside its pair = the parity of g (nch is even) = the queries' buffer it reads; the operand registers
    // ping-pong between two sets (cur -> the matrix instructions, nxt <- the decode), so the loop body is a pair of stages
    auto one_stage = [&](const uint32_t sp, const uint4 (&ce)[2], const uint4 (&co)[2], uint4 (&de)[2], uint4 (&dd)[2]) __attribute__((always_inline)) {
        if (kc == 0) {
            if (it) epilogue(it - 1);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) {
                    if (!HO) accl[mt][nt] = (i32x4w){0, 0, 0, 0};
                    acch[mt][nt] = (i32x4w){0, 0, 0, 0};
                }
        }
        // the lane's codes of the next stage: the second stage of this pair, or the first of the next (landed: every request but the last two was waited
        // for at the end of the previous stage, and these are at least three stages old)
        uint4 c0, c1;
        if (sp == 0) read_codes((uint32_t)P & 1u, 1, c0, c1);
        else read_codes(((uint32_t)P + 1u) & 1u, 0, c0, c1);
        const uint32_t kc1 = kc + 1 == nch ? 0 : kc + 1;
        queries_begin(kc1, sp ^ 1u);                        // the next stage's -> the buffer the previous stage was read from (everybody is past that barrier)
        {   // pair P + 2 -> the staging slot of pair P: its first stage's units were read a stage pair ago, its second stage's just now (sp = 0) - the
            // requests of this stage overwrite the first stage's half (sp = 0: pieces 0, 1) or the second's (sp = 1: pieces 2, 3)
            codes_begin(it2, kp2, (uint32_t)P & 1u);      // (pair P + 2 as (tile, pair of the tile): running counters, advanced when P is)
            slot_piece0 = sp * 2u;
        }
        stage(sp, c0, c1, ce, co, de, dd);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");    // the queries of the next stage (the two code requests behind them may be on their way)
        tw_stage_barrier();
        if (++kc == nch) { kc = 0; ++it; }
        if (sp == 1) {
            ++P;
            if (kp2 + 1 < npair) ++kp2;
            else if (it2 + 1 < my_tiles) { kp2 = 0; ++it2; }
        }
    };
    for (uint64_t g = 0; g < n_stages; g += 2) {
        one_stage(0, ae, ao, ne, no);
        one_stage(1, ne, no, ae, ao);
    }
    epilogue(my_tiles - 1);
    if (lane == 0) s.wcnt[blockIdx.x * (TW_THREADS / 64) + (uint32_t)w] = wcount;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // nothing may land in LDS after the block is gone
}

// ---- after the scan: an entry (low + 128 high, row, query) becomes (key lo, key hi, query) when its score - TqOps<4>::finish (scan_sq_mfma.hip), operation
// for operation - is not below the query's threshold score (ties pass), else an entry the regroup skips (query 0xFFFFFFFF).  One wave per list. ----
__global__ __launch_bounds__(256) void tq4w_finish_kernel(uint4 *wlist, const uint32_t *wcnt, uint32_t wcap, uint32_t n_lists, const float *sf, const float *l2,
                                                          uint32_t invert, const float *qinfo, const float *band /* or nullptr: the entries are whole sums */) {
    const uint32_t l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= n_lists) return;
    uint32_t cnt = wcnt[l];
    cnt = cnt < wcap ? cnt : wcap;
    uint4 *list = wlist + (uint64_t)l * wcap;
    for (uint32_t i = threadIdx.x & 63u; i < cnt; i += 64) {
        const uint4 e = list[i];
        const uint32_t row = e.y, q = e.z;
        const float f0 = qinfo[q], ec = qinfo[TW_QT + q], qlsq = qinfo[2 * TW_QT + q], tf = qinfo[3 * TW_QT + q];
        const float sumf = band ? (float)(128 * (int32_t)e.x) : (float)(int32_t)e.x;      // (tq_i32: |low + 128 high| < 2^31; with a band: the high sum alone)
        const float dot = f0 * sumf + ec;
        const float sfr = sf[row];
        float score;
        if (l2) {
            const float len = l2[row];
            const float y = len * len, z = (2.0f * dot) * sfr;
            score = (qlsq + y) - z;
        } else {
            score = dot * sfr;
        }
        score = invert ? -score : score;
        if (!(score < (band ? tf - band[q] : tf))) {
            const uint64_t key = make_key(score, row);
            list[i] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), q, 0u);
        } else {
            list[i] = make_uint4(0u, 0u, 0xFFFFFFFFu, 0u);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
bool tq4w_shape_ok(const ScanArgs &a) {
    return a.tq_bits == 4 && a.tq_i32 && a.dim >= 128 && a.dim % 128 == 0 && a.row_stride % 16 == 0 && a.ids == nullptr && a.top <= MAX_TOP_FAST &&
           (a.tq_l2 != nullptr) == (a.tq_invert != 0) && a.n_cand >= 1 && a.n_cand < 0xFFFFFFFFull;
}
size_t tq4w_query_bytes(uint32_t code_bytes) { return (size_t)(code_bytes / 64) * TW_B_UNITS * 16; }
size_t tq4w_wlists_counts_bytes(int num_cus) { return ((size_t)num_cus * (TW_THREADS / 64) * 4 + 255) / 256 * 256; }
size_t tq4w_wlists_bytes(int num_cus) { return tq4w_wlists_counts_bytes(num_cus) + (size_t)num_cus * (TW_THREADS / 64) * TW_WCAP * 16; }
uint32_t tq4w_wcap() { return TW_WCAP; }

int32_t launch_tq4w_stats(hipStream_t st, const float *d_sf, const float *d_l2, const void *d_rows, uint64_t row_stride, uint32_t code_bytes, uint64_t n, uint32_t *d_stats /* [8] */) {
    ::qmx::clear_stale_error();
    const uint32_t grid = (uint32_t)std::min<uint64_t>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(tq4w_stats_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, d_sf, d_l2, n, d_stats);
    QMX_HIP(hipGetLastError());
    hipLaunchKernelGGL(tq4w_c1_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, reinterpret_cast<const unsigned char *>(d_rows), row_stride, n, code_bytes, d_stats);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

int32_t launch_tq4w_pack(hipStream_t st, const ScanArgs &a, const uint64_t *d_gthr, float sf_min, float sf_max, float l2_min, float l2_max, uint32_t c1, int high_only,
                         void *d_bq, int32_t *d_thr_i,
                         float *d_qinfo, float *d_band, uint32_t *d_cand_cnt, uint32_t n_cnt) {
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(tq4w_pack_kernel, dim3(TW_QT), dim3(256), 0, st, reinterpret_cast<const unsigned char *>(a.queries), a.q_stride, a.aux_off, a.nq, a.dim / 64,
                       d_gthr, sf_min, sf_max, l2_min, l2_max, c1, a.tq_l2 ? 1 : 0, high_only, (uint4 *)d_bq, d_thr_i, d_qinfo, d_band, d_cand_cnt, n_cnt);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// d_wlists: [counts: tq4w_wlists_counts_bytes][lists]; *grid_out = blocks launched (8 lists each)
int32_t launch_scan_tq4w(hipStream_t st, const ScanArgs &a, const void *d_bq, const int32_t *d_thr_i, const float *d_qinfo, const float *d_band_high_only, int num_cus,
                         void *d_wlists, uint32_t *grid_out) {
    QMX_REQUIRE(tq4w_shape_ok(a), QMX_ERR_NOT_SUPPORTED, "TurboQuant wide scan: shape not supported");
    TqWideArgs s;
    s.bq = (const uint4 *)d_bq;
    s.nch = a.dim / 64;
    s.nq = a.nq;
    s.thr_i = d_thr_i;
    s.qinfo = d_qinfo;
    s.wcnt = (uint32_t *)d_wlists;
    s.wlist = (uint4 *)((unsigned char *)d_wlists + tq4w_wlists_counts_bytes(num_cus));
    s.wcap = TW_WCAP;
    const uint64_t n_tiles = (a.n_cand + TW_BM - 1) / TW_BM;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)num_cus, n_tiles);
    static thread_local DeviceOnce once_both, once_high;
    ::qmx::clear_stale_error();
    if (d_band_high_only) {
        auto kfn = scan_tq4w_kernel<true>;
        if (once_high.need()) {
            QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS));
            once_high.mark();
        }
        QMX_NOTE_KERNEL(kfn);
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(TW_THREADS), TW_LDS, st, a, s);
    } else {
        auto kfn = scan_tq4w_kernel<false>;
        if (once_both.need()) {
            QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS));
            once_both.mark();
        }
        QMX_NOTE_KERNEL(kfn);
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(TW_THREADS), TW_LDS, st, a, s);
    }
    QMX_HIP(hipGetLastError());
    const uint32_t n_lists = grid * (TW_THREADS / 64);
    hipLaunchKernelGGL(tq4w_finish_kernel, dim3((n_lists + 3) / 4), dim3(256), 0, st, s.wlist, s.wcnt, s.wcap, n_lists, a.tq_sf, a.tq_l2, a.tq_invert, d_qinfo, d_band_high_only);
    QMX_HIP(hipGetLastError());
    if (grid_out) *grid_out = grid;
    return QMX_OK;
}

}  // namespace qmx

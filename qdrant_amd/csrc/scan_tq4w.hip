// scan_tq4w.hip - EncodedVectorsTQ, 4 bits per value, brute-force top-k for LARGE query batches: 128 queries per pass of the code block, the 4-bit
// codes decoded ONCE per 256-row tile into the int8 operand images of the matrix cores.
//
// Same reference loops as scan_tq.hip / scan_sq_mfma.hip TqOps<4>: BatchFilteredSearcher::peek_top_iter
// (lib/segment/src/index/hnsw_index/point_scorer.rs:423-472) over Query4bitSimd::dotprod (lib/quantization/src/turboquant/simd/query4bit/mod.rs:
// dot_raw = sum q_signed * c_u, q_signed = 128 high + low) and score_precomputed (turboquant/quantization.rs:569-620).  Every score is the exact integer
// dot of the row's codebook bytes with the two signed digits of the query, finished by TqOps<4>::finish's f32 expression: the same bits.
//
// Why.  The 32-query kernel (scan_sq_mfma.hip) decodes a row's nibbles in the registers of the wave that multiplies them: 336 of its 660 vector
// instructions per 16-row tile are the decode, paid once per 32 queries (10 M x 768: 1.36 ms per 32 queries = 0.36 of HBM, decode-bound).  Here a block
// decodes a stage - 256 rows x 128 coordinates = 16 KiB of codes - once into LDS as the A-operand image scan_i8copy_kernel streams from its copy
// (scan_split.hip sp_unit), the queries' digits arrive as B-operand images (LDS-DMA from a 2 x 96 KiB image in L2), and the eight waves multiply 64 rows x
// 64 queries x 2 digits each: 64 matrix instructions per wave and stage.  What bounds it is no longer HBM (3.84 GB per 128 queries) but the matrix cores
// and the LDS: 2 digits x 2 x 128 x 768 operations per row = 0.78 ms per 10 M rows at the int8 peak, 256 KiB of LDS traffic per stage.
//
// Scores are exact, so the pass needs no band: a pair is a CANDIDATE when its score is not below the k-th best score of a strided sample of the block
// (api_search.hip: the pre-scan every wide path starts with; ties pass), candidates go to per-wave lists in global memory (scan_i8copy_kernel's), are
// regrouped per query, the k best keys per query are selected, re-scored by the pair kernel and sorted - the tail of the int8 prefilter, with a band of
// zero.  The fast reject runs on integers: with sf in [sf_min, sf_max] over the segment (and l2 >= l2_min) a pair can only reach the threshold when
// low + 128 high >= thr_i[q], a bound tq4w_pack_kernel derives per query with the rounding of finish() on its side.
#include "scan_common.hpp"

namespace qmx {

typedef int i32x4w __attribute__((ext_vector_type(4)));

constexpr int TW_THREADS = 512;
constexpr int TW_BM = 256;                                   // rows per tile
constexpr int TW_QT = 128;                                   // queries per pass
constexpr int TW_A_UNITS = TW_BM * 2 * 4;                    // 16-byte units of a decoded stage: 256 rows x {even, odd dims} x 4 pieces = 32 KiB
constexpr int TW_B_DIGIT_UNITS = TW_QT * 2 * 4;              // ... of one digit of the queries' stage: 16 KiB
constexpr int TW_B_UNITS = 2 * TW_B_DIGIT_UNITS;             // low digits, high digits: 32 KiB
constexpr int TW_R_UNITS = TW_BM * 4;                        // ... of a stage's codes: 256 rows x 4 pieces = 16 KiB
constexpr int TW_LDS = (2 * TW_A_UNITS + 2 * TW_B_UNITS + 2 * TW_R_UNITS) * 16;      // 64 + 64 + 32 = 160 KiB: the whole LDS of the CU
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
    const int32_t *thr_i;     // [128] a pair whose low + 128 high is below this cannot reach the query's threshold
    const float *qinfo;       // [4][128] f0, ec, qlsq, thr_f (the threshold score: ties pass)
    uint4 *wlist;             // [waves][wcap] (key lo, key hi, query, 0)
    uint32_t *wcnt;           // [waves] entries each wave wanted to append (may run past wcap: overflow)
    uint32_t wcap;
};

// ---- once per segment: stats[0] = min sf, [1] = max sf, [2] = min l2 (uint bits of positive floats order like the floats), [3] != 0: a value that is
// not a positive finite number (no wide pass for this block) ----
__global__ __launch_bounds__(256) void tq4w_stats_kernel(const float *sf, const float *l2, uint64_t n, uint32_t *stats) {
    float lo = __builtin_inff(), hi = 0.0f, l2lo = __builtin_inff();
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
        }
    }
    for (int o = 32; o >= 1; o >>= 1) {
        lo = __builtin_fminf(lo, __shfl_xor(lo, o, 64));
        hi = __builtin_fmaxf(hi, __shfl_xor(hi, o, 64));
        l2lo = __builtin_fminf(l2lo, __shfl_xor(l2lo, o, 64));
    }
    if ((threadIdx.x & 63) == 0 && !bad) {
        atomicMin(&stats[0], __float_as_uint(lo));
        atomicMax(&stats[1], __float_as_uint(hi));
        if (l2) atomicMin(&stats[2], __float_as_uint(l2lo));
    }
    if (bad) atomicOr(&stats[3], 1u);
}

// ---- once per 128-query tile: one block per query slot.  The query's digits in the B-operand images, its integer reject bound, what finish() needs ----
// A query entry (scan_tq.hip tq_query_encode_kernel) holds, per 16-byte row piece P, 64 bytes: [low digits of the even dims][low, odd][high, even][high, odd];
// stage kc of the scan covers row pieces 4 kc .. 4 kc + 3, one MFMA the even (or the odd) dims of the four: unit (query tile, eo, p, query) of digit D.
__global__ __launch_bounds__(256) void tq4w_pack_kernel(const unsigned char *queries, uint32_t q_stride, uint32_t aux_off, uint32_t nq, uint32_t nch,
                                                        const uint64_t *gthr, float sf_min, float sf_max, float l2_min, int is_l2, uint4 *bq, int32_t *thr_i,
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
        }
    }
    thr_i[qi] = ti;
    band[qi] = bd;
    qinfo[qi] = f0;
    qinfo[TW_QT + qi] = ec;
    qinfo[2 * TW_QT + qi] = qlsq;
    qinfo[3 * TW_QT + qi] = tf;
}

// 4-bit selectors, one per byte -> the codebook bytes (tq_policies.hpp tq4_lookup)
__device__ __forceinline__ uint32_t tw_lut4(uint32_t sel) {
    const uint32_t s = sel & 0x07070707u;
    const uint32_t lo = __builtin_amdgcn_perm(0xFAEEE1D4u, 0xC5B49F80u, s), hi = __builtin_amdgcn_perm(0x7F614C3Bu, 0x2C1F1206u, s);
    return __builtin_amdgcn_perm(hi, lo, ((sel >> 1) & 0x04040404u) | 0x03020100u);
}

// The scan.  Block = 8 waves = 4 row quarters (wm) x 2 query halves (wn), one block per CU, persistent over 256-row tiles; a tile = nch stages of 128
// coordinates (64 code bytes per row).  EVERYTHING the loop fetches arrives by LDS-DMA and is counted by the kernel itself (one plain vector load inside the
// loop and the compiler's own conservative `s_waitcnt vmcnt(0)` drains the streams at every stage - measured: 3.2 us per stage, an HBM round trip).
// Per stage g a wave
//   1. asks for the queries' images of stage g + 1 (four 1 KiB copies from the L2-resident image) and for ITS 2 KiB of the codes of stage g + 2 (two copies
//      of 16 rows x 64 bytes: the 128 units its own threads decode, so no barrier stands between a copy's arrival and its decode),
//   2. multiplies stage g: 24 ds_read_b128, 64 v_mfma_i32_16x16x64_i8 - (even dims, odd dims) x (low digit, high digit) - into 2 x 16 accumulator tiles,
//   3. decodes the codes of stage g + 1 into the other A buffer (two ds_read_b128, ~110 vector instructions, four ds_write_b128 per thread),
//   4. waits for the queries of stage g + 1 and meets the others at the stage barrier.
// LDS: decoded rows 2 x 32 KiB, queries 2 x 32 KiB, codes 2 x 16 KiB = 160 KiB.
// The epilogue of a tile lists (low + 128 high, row, query) of every pair that meets the query's integer bound; tq4w_finish_kernel turns the entries into
// keys (TqOps<4>::finish, the exact compare with the threshold score) before the regroup.
template <bool L2_UNUSED>
__global__ __launch_bounds__(TW_THREADS, 1) void scan_tq4w_kernel(const ScanArgs a, const TqWideArgs s) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4 *lds = reinterpret_cast<uint4 *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t n_tiles = (a.n_cand + TW_BM - 1) / TW_BM;
    const uint32_t nch = s.nch;
    const uint64_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    if (my_tiles == 0) {
        if (lane == 0) s.wcnt[blockIdx.x * (TW_THREADS / 64) + (uint32_t)w] = 0;
        return;
    }
    const uint32_t wm = (uint32_t)w & 3u, wn = (uint32_t)w >> 2;
    const uint32_t kq_r = (uint32_t)lane >> 4, m_r = (uint32_t)lane & 15u;
    const uint32_t a_rd = tw_unit(wm * 4, 0, kq_r, m_r), b_rd = tw_unit(wn * 4, 0, kq_r, m_r);
    int thr_i[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) thr_i[nt] = s.thr_i[wn * 64 + nt * 16 + m_r];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // from here on the kernel counts its vector-memory traffic itself
    const uint32_t lds0 = (uint32_t)(uintptr_t)(tw_lds_byte *)smem_raw;
    const uint32_t lane_off = (uint32_t)lane * 16u;
    uint4 *const a_lds = lds, *const b_lds = lds + 2 * TW_A_UNITS, *const r_lds = lds + 2 * TW_A_UNITS + 2 * TW_B_UNITS;
    const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
    const uint64_t last_row = a.n_cand - 1;
    // the codes a thread decodes: rows (tid >> 2) and 128 + (tid >> 2) of the tile, piece tid & 3 of the stage = units tid and 512 + tid of the stage's
    // codes in LDS ([row][piece]); a wave's copies fetch exactly its threads' units: rows 16 w .. + 15 and 128 + 16 w .. + 15
    const uint32_t my_r = (uint32_t)tid >> 2, my_p = (uint32_t)tid & 3u;
    const uint32_t wr0 = tw_unit(my_r >> 4, 0, my_p, my_r & 15u), wr1 = tw_unit(8 + (my_r >> 4), 0, my_p, my_r & 15u);     // (the odd dims: + 64 units)
    const uint32_t row_stride32 = (uint32_t)a.row_stride;

    auto uniform_ptr = [&](uint64_t v) {
        return reinterpret_cast<const unsigned char *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
                                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
    };
    // the queries' images of stage kc -> B buffer `slot`: this wave's 4 KiB of the 32
    auto request_queries = [&](uint32_t kc, uint32_t slot) {
        const unsigned char *src = uniform_ptr((uint64_t)(uintptr_t)(s.bq + (uint64_t)kc * TW_B_UNITS) + (uint32_t)w * 4096u);
        const uint32_t dst = lds0 + (2 * TW_A_UNITS + slot * TW_B_UNITS) * 16u + (uint32_t)w * 4096u;
#pragma unroll
        for (int i = 0; i < 4; ++i) tw_glds16(src + i * 1024, lane_off, dst + i * 1024);
    };
    // this wave's share of the codes of stage kc of the block's it-th tile -> code buffer `slot` (rows past the block: the last row's bytes, scores dropped)
    auto request_codes = [&](uint64_t it, uint32_t kc, uint32_t slot) {
        const uint64_t row0 = (blockIdx.x + it * gridDim.x) * TW_BM;
        const unsigned char *src = uniform_ptr((uint64_t)(uintptr_t)(rows + row0 * a.row_stride + kc * 64u));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t rl = (uint32_t)h * 128u + my_r;
            const uint64_t room = last_row - row0;                  // (row0 <= last_row: the tile exists)
            const uint32_t rc = (uint64_t)rl < room ? rl : (uint32_t)room;
            const uint32_t dst = lds0 + (2 * TW_A_UNITS + 2 * TW_B_UNITS) * 16u + slot * 16384u + (uint32_t)h * 8192u + (uint32_t)w * 1024u;
            tw_glds16(src, rc * row_stride32 + my_p * 16u, dst);
        }
    };
    auto decode_stage = [&](uint32_t rslot, uint32_t aslot) {
        const uint4 *src = r_lds + rslot * 1024 + (uint32_t)tid;
        const uint4 v0 = src[0], v1 = src[512];
        uint4 *dst = a_lds + aslot * TW_A_UNITS;
        const uint32_t x0[4] = {v0.x, v0.y, v0.z, v0.w}, x1[4] = {v1.x, v1.y, v1.z, v1.w};
        uint32_t e0[4], o0[4], e1[4], o1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            e0[k] = tw_lut4(x0[k] & 0x0F0F0F0Fu);
            o0[k] = tw_lut4((x0[k] >> 4) & 0x0F0F0F0Fu);
            e1[k] = tw_lut4(x1[k] & 0x0F0F0F0Fu);
            o1[k] = tw_lut4((x1[k] >> 4) & 0x0F0F0F0Fu);
        }
        dst[wr0] = make_uint4(e0[0], e0[1], e0[2], e0[3]);
        dst[wr0 + 64] = make_uint4(o0[0], o0[1], o0[2], o0[3]);
        dst[wr1] = make_uint4(e1[0], e1[1], e1[2], e1[3]);
        dst[wr1 + 64] = make_uint4(o1[0], o1[1], o1[2], o1[3]);
    };

    i32x4w accl[4][4], acch[4][4];
    uint4 *const wl = s.wlist + (uint64_t)(blockIdx.x * (TW_THREADS / 64) + (uint32_t)w) * s.wcap;
    uint32_t wcount = 0;
    const uint32_t n_rows32 = (uint32_t)a.n_cand;

    // The epilogue of a tile: the integer bound, narrowing by wave-uniform steps (query tile, 16-row tile, the four rows of a lane) as scan_i8copy_kernel does
    auto epilogue = [&](uint64_t it) {
        const uint64_t tile = blockIdx.x + it * gridDim.x;
        const uint32_t row0 = (uint32_t)(tile * TW_BM) + wm * 64 + 4 * kq_r;
        bool hit[4];
        bool maybe = false;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            int mx = (int)0x80000000;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int v = (acch[mt][nt][j] << 7) + accl[mt][nt][j];
                    mx = v > mx ? v : mx;
                }
            hit[nt] = mx >= thr_i[nt];
            maybe = maybe || hit[nt];
        }
        if (!__ballot(maybe)) return;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (!__ballot(hit[nt])) continue;
            const uint32_t q = wn * 64 + (uint32_t)nt * 16 + m_r;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                int m4 = (int)0x80000000;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int v = (acch[mt][nt][j] << 7) + accl[mt][nt][j];
                    m4 = v > m4 ? v : m4;
                }
                if (!__ballot(m4 >= thr_i[nt])) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int v = (acch[mt][nt][j] << 7) + accl[mt][nt][j];
                    const uint32_t row = row0 + (uint32_t)mt * 16 + (uint32_t)j;
                    const bool c = v >= thr_i[nt] && row < n_rows32 && q < s.nq;
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

    // one stage: the matrix work on buffers `slot`.  The operand reads run one query tile ahead of the matrix instructions and no further (the scheduler,
    // left alone, hoists all 24 reads of the stage in front of them: 96 registers the 128 accumulators leave no room for)
    auto multiply = [&](uint32_t slot) {
        const uint4 *ab = a_lds + slot * TW_A_UNITS + a_rd;
        const uint4 *bb = b_lds + slot * TW_B_UNITS + b_rd;
#pragma unroll
        for (int eo = 0; eo < 2; ++eo) {
            i32x4w av[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) av[mt] = *reinterpret_cast<const i32x4w *>(ab + mt * 128 + eo * 64);
            i32x4w bl = *reinterpret_cast<const i32x4w *>(bb + eo * 64);
            i32x4w bh = *reinterpret_cast<const i32x4w *>(bb + TW_B_DIGIT_UNITS + eo * 64);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                i32x4w nl = bl, nh = bh;
                if (nt < 3) {
                    nl = *reinterpret_cast<const i32x4w *>(bb + (nt + 1) * 128 + eo * 64);
                    nh = *reinterpret_cast<const i32x4w *>(bb + TW_B_DIGIT_UNITS + (nt + 1) * 128 + eo * 64);
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    accl[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[mt], bl, accl[mt][nt], 0, 0, 0);
                    acch[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[mt], bh, acch[mt][nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                bl = nl;
                bh = nh;
            }
        }
    };

    const uint64_t n_stages = my_tiles * nch;
    // ---- prologue: the queries of stage 0, the codes of stages 0 and 1; stage 0 decoded ----
    request_queries(0, 0);
    request_codes(0, 0, 0);
    {
        const bool two = n_stages > 1;          // (a block with a single stage asks for it twice: the waits below count requests, not bytes)
        request_codes(two && nch == 1 ? 1 : 0, two && nch > 1 ? 1 : 0, 1);
    }
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // the codes of stage 0 (this wave's units) have landed
    decode_stage(0, 0);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    tw_stage_barrier();
    // (the stage two ahead, as (tile, kc), and the one one ahead)
    uint64_t it2 = 0;
    uint32_t kc2 = 2;
    while (kc2 >= nch && it2 + 1 < my_tiles) { kc2 -= nch; ++it2; }
    if (kc2 >= nch) kc2 = nch - 1;                          // (no such stage: its request repeats the last one)
    uint64_t it = 0;
    uint32_t kc = 0;
    for (uint64_t g = 0; g < n_stages; ++g) {
        const uint32_t slot = (uint32_t)g & 1u;
        if (kc == 0) {
            if (it) epilogue(it - 1);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    accl[mt][nt] = (i32x4w){0, 0, 0, 0};
                    acch[mt][nt] = (i32x4w){0, 0, 0, 0};
                }
        }
        const uint32_t kc1 = kc + 1 == nch ? 0 : kc + 1;
        request_queries(kc1, slot ^ 1u);                    // stage g + 1 -> the buffer stage g - 1 was read from (everybody is past that barrier)
        request_codes(it2, kc2, slot);                      // stage g + 2 -> the buffer this wave decoded its units of stage g from, one stage ago
        multiply(slot);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // all but this stage's six requests have landed: this wave's codes of stage g + 1 among them
        decode_stage(slot ^ 1u, slot ^ 1u);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");    // the queries of stage g + 1
        tw_stage_barrier();
        if (++kc == nch) { kc = 0; ++it; }
        if (kc2 + 1 < nch) ++kc2;
        else if (it2 + 1 < my_tiles) { kc2 = 0; ++it2; }
    }
    epilogue(my_tiles - 1);
    if (lane == 0) s.wcnt[blockIdx.x * (TW_THREADS / 64) + (uint32_t)w] = wcount;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // nothing may land in LDS after the block is gone
}

// ---- after the scan: an entry (low + 128 high, row, query) becomes (key lo, key hi, query) when its score - TqOps<4>::finish (scan_sq_mfma.hip), operation
// for operation - is not below the query's threshold score (ties pass), else an entry the regroup skips (query 0xFFFFFFFF).  One wave per list. ----
__global__ __launch_bounds__(256) void tq4w_finish_kernel(uint4 *wlist, const uint32_t *wcnt, uint32_t wcap, uint32_t n_lists, const float *sf, const float *l2,
                                                          uint32_t invert, const float *qinfo) {
    const uint32_t l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= n_lists) return;
    uint32_t cnt = wcnt[l];
    cnt = cnt < wcap ? cnt : wcap;
    uint4 *list = wlist + (uint64_t)l * wcap;
    for (uint32_t i = threadIdx.x & 63u; i < cnt; i += 64) {
        const uint4 e = list[i];
        const uint32_t row = e.y, q = e.z;
        const float f0 = qinfo[q], ec = qinfo[TW_QT + q], qlsq = qinfo[2 * TW_QT + q], tf = qinfo[3 * TW_QT + q];
        const float sumf = (float)(int32_t)e.x;               // (tq_i32: |low + 128 high| < 2^31)
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
        if (!(score < tf)) {
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

int32_t launch_tq4w_stats(hipStream_t st, const float *d_sf, const float *d_l2, uint64_t n, uint32_t *d_stats) {
    ::qmx::clear_stale_error();
    const uint32_t grid = (uint32_t)std::min<uint64_t>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(tq4w_stats_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, d_sf, d_l2, n, d_stats);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

int32_t launch_tq4w_pack(hipStream_t st, const ScanArgs &a, const uint64_t *d_gthr, float sf_min, float sf_max, float l2_min, void *d_bq, int32_t *d_thr_i,
                         float *d_qinfo, float *d_band, uint32_t *d_cand_cnt, uint32_t n_cnt) {
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(tq4w_pack_kernel, dim3(TW_QT), dim3(256), 0, st, reinterpret_cast<const unsigned char *>(a.queries), a.q_stride, a.aux_off, a.nq, a.dim / 64,
                       d_gthr, sf_min, sf_max, l2_min, a.tq_l2 ? 1 : 0, (uint4 *)d_bq, d_thr_i, d_qinfo, d_band, d_cand_cnt, n_cnt);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// d_wlists: [counts: tq4w_wlists_counts_bytes][lists]; *grid_out = blocks launched (8 lists each)
int32_t launch_scan_tq4w(hipStream_t st, const ScanArgs &a, const void *d_bq, const int32_t *d_thr_i, const float *d_qinfo, int num_cus, void *d_wlists,
                         uint32_t *grid_out) {
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
    static thread_local DeviceOnce once;
    ::qmx::clear_stale_error();
    auto kfn = scan_tq4w_kernel<false>;
    if (once.need()) {
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS));
        once.mark();
    }
    QMX_NOTE_KERNEL(kfn);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(TW_THREADS), TW_LDS, st, a, s);
    QMX_HIP(hipGetLastError());
    const uint32_t n_lists = grid * (TW_THREADS / 64);
    hipLaunchKernelGGL(tq4w_finish_kernel, dim3((n_lists + 3) / 4), dim3(256), 0, st, s.wlist, s.wcnt, s.wcap, n_lists, a.tq_sf, a.tq_l2, a.tq_invert, d_qinfo);
    QMX_HIP(hipGetLastError());
    if (grid_out) *grid_out = grid;
    return QMX_OK;
}

}  // namespace qmx

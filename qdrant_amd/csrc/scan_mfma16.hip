// scan_mfma16.hip — f32 dot / cosine brute-force top-k for a 32-query tile on v_mfma_f32_16x16x4_f32, chain-major,
// with the bits of the x86 AVX2+FMA reference (dot_similarity_avx, lib/segment/src/spaces/simple_avx.rs:167-213 under
// BatchFilteredSearcher::peek_top_iter, lib/segment/src/index/hnsw_index/point_scorer.rs:423-472).
//
// Why another kernel.  scan_mfma.hip maps one AVX chain to one 4x4x1 MFMA block: every instruction needs a fresh A and a
// fresh B register for 512 flops and a cross-lane fold per tile; at 32 queries the CU runs out of issue slots (9.5 ms per
// 30.72 GB scan, the HBM floor is ~5 ms).  v_mfma_f32_16x16x4_f32 does 2048 flops per A / B register pair and is still an
// exact f32 fma chain: D[m][n] = fma(A[m][3], B[3][n], fma(A[m][2], B[2][n], fma(A[m][1], B[1][n], fma(A[m][0], B[0][n], C[m][n]))))
// (the parity tests pin that order).  dot_similarity_avx keeps 32 independent chains per (row, query): chain c = 8 r + j (AVX
// register r, SIMD lane j) takes element 32 i + c of every 32-float step i in order.  Chains are independent, so they can be
// run ONE AFTER THE OTHER: the 16x16 accumulator tile of chain c holds (16 stored rows) x (16 queries), the K stream of
// the instruction is elements c, c + 32, c + 64, ... of the rows and of the queries.  Folding the 32 chains is then
// ELEMENTWISE on accumulator tiles in the reference's tree (four_way_hsum, hsum256_ps_avx, simple_avx.rs:10-28) - no
// cross-lane traffic at all.
//
// Work split of a 256-thread block (4 waves, one per SIMD; one block per CU, persistent over 16-row tiles):
//   wave w owns SIMD lanes j = w and w + 4 of all four AVX registers = chains {w, w+4} + 8 r: 8 chains x 2 query tiles
//   of 16 = 16 accumulator tiles (64 VGPRs).  Its share of the QUERIES (B operands: 8 chains x dim / 128 K-steps x 2
//   tiles) lives in registers for the whole kernel (96 VGPRs at dim 768): the queries are never re-read.
//   It folds T_w = (s1 + s2) + (s3 + s4), T_{w+4} likewise, lr_w = T_{w+4} + T_w (extractf128(x, 1) + cast(x)), and the
//   four waves exchange lr_0..lr_3 through 8 KiB of LDS: score = (lr_0 + lr_1) + (lr_2 + lr_3)  (_mm_hadd_ps, p1 + p2).
// Rows stream HBM -> LDS with global_load_lds_dwordx4 (1 KiB of ONE row per wave instruction: fully coalesced, no
// staging VGPRs) into a ring of 8 stages of 16 rows x 256 floats; 6-7 stages (~100 KiB per CU) are always in flight, counted
// with s_waitcnt vmcnt(24) - never 0 inside the loop.  The A operand of (chain c, K-step) is one ds_read_b32: lane
// (m = lane & 15, k = lane >> 4) reads element c + 32 (4 step + k) of row m.  Every byte of the block is read from HBM
// once and from LDS once.
//
// Restrictions (anything else takes the scan_mfma.hip path): top-k mode over the whole block (no id list), dim 256, 512 or
// 768 (the query registers of longer rows do not fit next to a second block on the CU), 16-byte aligned rows.
#include "scan_common.hpp"

namespace qmx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;

constexpr int M16_NBUF = 4;                 // ring stages per block; two blocks share a CU
constexpr int M16_ROWP = 1024 + 16;         // LDS pitch of a 1 KiB row chunk
constexpr int M16_STAGE = 16 * M16_ROWP;
constexpr int M16_XCH = 4 * 2 * 4 * 64 * 4; // lr exchange: [wave][query tile][reg][lane] f32
constexpr int M16_PF_SINK = 256;             // where the L2 prefetch loads land (never read)
constexpr int M16_PF_AHEAD = 3;              // stages the L2 prefetch runs ahead of the ring refill
constexpr int M16_LDS = M16_NBUF * M16_STAGE + M16_XCH + M16_PF_SINK;
constexpr int M16_QT = 32;

// 1 KiB of one row (wave-uniform base `row`, lane i fetches bytes 16 i .. 16 i + 15) -> LDS at the wave-uniform byte address
// lds_dst (lane i lands at lds_dst + 16 i).  Invisible to the compiler's s_waitcnt bookkeeping on purpose: the loop below counts
// these loads itself.
__device__ __forceinline__ void glds16(const unsigned char *row, uint32_t lane_off, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(row), "s"(lds_dst) : "memory");
}
// one dword of 64 lines -> a 256-byte LDS sink: pulls the lines into L2 ahead of the ring refill (no register destination: nothing
// for the compiler to track, nothing that can land late in a live register)
__device__ __forceinline__ void gpf4(const unsigned char *base, uint32_t lane_off, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(base), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// bit `id` of a device bitmap through the scalar cache (id is wave-uniform): no vector-memory counter involved
__device__ __forceinline__ bool bit_get_uniform(const uint64_t *words, uint32_t id) {
    const uint64_t *p = words + (id >> 6);
    uint64_t wv;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wv) : "s"(p) : "memory");
    return (wv >> (id & 63)) & 1;
}
__device__ __forceinline__ bool live_uniform(const DeletedView &d, uint32_t id) {   // DeletedView::live for a uniform id
    const bool vdel = (d.vec_deleted && id < d.n_vec_bits) ? bit_get_uniform(d.vec_deleted, id) : false;
    const bool pdel = d.point_deleted ? (id < d.n_point_bits ? bit_get_uniform(d.point_deleted, id) : true) : !(id < d.n_rows);
    const bool ok = d.allowed ? (id < d.n_allowed_bits && bit_get_uniform(d.allowed, id)) : true;
    return !vdel && !pdel && ok;
}

constexpr bool getenv_pin = false;
template <int KS /* dim / 256 */, int DBG = 0 /* tuning experiments: 1 no MFMA, 2 no row loads, 3 no epilogue */, bool PF = false /* L2 prefetch stream: measured slower */>
__global__ __launch_bounds__(256, 2) void scan_f32_mfma16_kernel(const ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kk = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte *)smem;
    const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
    const int top = (int)a.top;

    // ---- this wave's slice of the 32 queries, in B-operand layout: bq[ci][s][t] = query 16 t + n, element chain(ci) + 32 (4 s + kk)
    // ci = 2 r + jj  ->  chain 8 r + w + 4 jj
    float bq[8][KS * 2][2];
    {
        const float *qf = reinterpret_cast<const float *>(a.queries);
        const uint32_t qs = a.q_stride / 4;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
#pragma unroll
            for (int s = 0; s < KS * 2; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    bq[ci][s][t] = qf[(uint32_t)(16 * t + n) * qs + (uint32_t)(8 * (ci >> 1) + 4 * (ci & 1) + w) + 32u * (uint32_t)(4 * s + kk)];
    }
    // epilogue role of the wave: query tile tq = w & 1 (queries 16 tq + n), accumulator registers j0, j0 + 1 (rows 4 kk + j0 + p)
    const int tq = w & 1, j0 = 2 * (w >> 1);
    const int my_q = 16 * tq + n;
    const bool has_kb = a.key_bound != nullptr;       // bound of a later pass of a top > 64 search (keys must stay below it)
    uint64_t kb = has_kb ? a.key_bound[my_q < (int)a.nq ? my_q : 0] : 0;
    // make every query register "used" here: the compiler then waits for its own loads before the loop instead of at their first
    // use inside it (a vmcnt(N) there would drain the row stream, which it does not know about)
#pragma unroll
    for (int ci = 0; ci < 8; ++ci)
#pragma unroll
        for (int s = 0; s < KS * 2; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) asm volatile("" : "+v"(bq[ci][s][t]));
    asm volatile("" : "+v"(kb));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // from here on the only vector loads in flight are the row stream

    uint64_t list[16];        // this wave's top list of its 16 queries (lane i = i-th best key)
    float thr_f = -__builtin_inff();
#pragma unroll
    for (int q = 0; q < 16; ++q) list[q] = 0;

    const uint64_t n_tiles = (a.n_cand + 15) / 16;
    const uint64_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;

    // ---- the row stream.  Stages are issued strictly in order: stage = (tile iteration, 1 KiB chunk kc of its rows); this wave
    // fetches rows 4 w .. 4 w + 3 of the tile.  Stages past the end re-read row 0 (never consumed): the in-flight count stays
    // constant and the loop needs no tail.  All of it is scalar work: row bases are wave-uniform.
    const uint32_t lane_off = (uint32_t)lane * 16u;
    uint64_t ld_it = 0;            // tile iteration of the next stage to issue
    uint32_t ld_kc = 0;            // ... and its chunk
    uint32_t ld_slot = 0;
    const unsigned char *ld_row[4];
    auto ld_new_tile = [&]() {
        const uint64_t tile = blockIdx.x + ld_it * gridDim.x;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint64_t c = tile * 16 + (uint32_t)(4 * w + i);
            if (c >= a.n_cand) c = a.n_cand - 1;           // rows past the end of a partial last tile: any valid row, results dropped
            if (ld_it >= my_tiles) c = 0;
            ld_row[i] = reinterpret_cast<const unsigned char *>(uniform_u64((uint64_t)(rows + c * a.row_stride)));
        }
    };
    auto issue_row = [&](int i) {                          // row 4 w + i of the stage being issued
        if (DBG != 2) glds16(ld_row[i] + ld_kc * 1024u, lane_off, lds0 + ld_slot * M16_STAGE + (uint32_t)(4 * w + i) * M16_ROWP);
    };
    auto ld_advance = [&]() {                              // after the 4 rows of a stage
        ld_slot = (ld_slot + 1) & (M16_NBUF - 1);
        if (++ld_kc == KS) {
            ld_kc = 0;
            ++ld_it;
            ld_new_tile();
        }
    };
    // L2 prefetch, PF_AHEAD stages ahead of the refill: lane l of wave w touches 128-byte line l & 7 of the 1 KiB chunk of row
    // 4 w + (l & 31) / 8 (lanes 32..63 repeat 0..31: same lines, one request).  The HBM latency is then paid here, off the consumer's path: the
    // refill itself hits L2, so a late consumer (fold / selection / barrier skew) catches up at compute speed.
    const bool pf_on = PF;
    uint64_t pf_it = 0;
    uint32_t pf_kc = 0;
    const unsigned char *pf_base = rows;      // wave-uniform: row 4 w of the tile being prefetched
    // (a block of fewer than 4 rows only ever touches row 0)
    const uint32_t pf_lane_off = (a.n_cand < 4 ? 0u : (((uint32_t)lane & 31u) >> 3) * (uint32_t)a.row_stride) + ((uint32_t)lane & 7u) * 128u;
    auto pf_new_tile = [&]() {
        const uint64_t tile = blockIdx.x + pf_it * gridDim.x;
        uint64_t c = tile * 16 + (uint32_t)(4 * w);
        if (c + 4 > a.n_cand || pf_it >= my_tiles) c = 0;        // partial last tile / past the end: touch rows 0..3 instead
        pf_base = reinterpret_cast<const unsigned char *>(uniform_u64((uint64_t)(rows + c * a.row_stride)));
    };
    auto pf_issue = [&]() {
        if (!pf_on) return;
        if (DBG != 2) gpf4(pf_base + pf_kc * 1024u, pf_lane_off, lds0 + M16_NBUF * M16_STAGE + M16_XCH);
        if (++pf_kc == KS) {
            pf_kc = 0;
            ++pf_it;
            pf_new_tile();
        }
    };
    ld_new_tile();
    if (pf_on) {
        pf_new_tile();
        for (int p = 0; p < M16_PF_AHEAD; ++p) pf_issue();     // stages 0 .. PF_AHEAD - 1 are fetched by the refill itself; harmless
    }
    for (int p = 0; p < M16_NBUF; ++p) {
        pf_issue();
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_row(i);
        ld_advance();
    }

    const uint32_t a_off = (uint32_t)n * M16_ROWP + (uint32_t)kk * 128u + (uint32_t)w * 4u;   // lane part of the A-operand address
    float *xch = reinterpret_cast<float *>(smem + M16_NBUF * M16_STAGE);
    // A operand of (chain index ci, K-step ks) of a stage: element chain(ci) + 32 (4 ks + kk) of the 256-float chunk of row n
    auto lds_a = [&](const unsigned char *stage_base, int ks, int ci) -> float {
        return *reinterpret_cast<const float *>(stage_base + (8 * (ci >> 1) + 4 * (ci & 1)) * 4 + ks * 512);
    };
    float ax0[8], ax1[8];     // K-steps 0 and 1 of the current stage
    constexpr int PER_STAGE = PF ? 5 : 4;      // vector-memory instructions per stage and wave: [prefetch,] 4 rows
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER_STAGE * (M16_NBUF - 1)) : "memory");   // stage 0 of this wave has landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) ax0[ci] = lds_a(smem + a_off, 0, ci);

    // Second half of a tile's epilogue, run one stage barrier after the tile's lr values were written to xch: the wave finishes
    // accumulator register j0 + p of query tile tq: row 4 (lane >> 4) + j0 + p of the tile, query 16 tq + (lane & 15).
    auto finalize = [&](uint64_t tile, int p) {
        const int j = j0 + p;
        const float lr0 = xch[((0 * 2 + tq) * 4 + j) * 64 + lane], lr1 = xch[((1 * 2 + tq) * 4 + j) * 64 + lane];
        const float lr2 = xch[((2 * 2 + tq) * 4 + j) * 64 + lane], lr3 = xch[((3 * 2 + tq) * 4 + j) * 64 + lane];
        const float score = (lr0 + lr1) + (lr2 + lr3);                              // hadd: p1 = lr0 + lr1, p2 = lr2 + lr3; p1 + p2
        const uint64_t c = tile * 16 + (uint32_t)(4 * kk + j);
        const uint32_t res_row = (uint32_t)c;
        const bool mine = c < a.n_cand && my_q < (int)a.nq;
        // cheap reject on the score alone; ties with the k-th score and NaN (greatest in OrderedFloat) fall through to the key compare
        bool cnd = mine && !(score < thr_f);
        if (__ballot(cnd)) {
            const uint64_t key = make_key(score, res_row);
            cnd = cnd && (!has_kb || key < kb);      // an equal score still loses or wins on the id inside the list compare
            uint64_t mask = __ballot(cnd);
            while (mask) {
                const int src = __builtin_ctzll(mask);
                mask &= mask - 1;
                const uint64_t nk = readlane_u64(key, src);
                const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)res_row, src);
                if (!live_uniform(a.del, id)) continue;
                const int ql = src & 15;                                            // the source lane's query inside the tile
#pragma unroll
                for (int qq = 0; qq < 16; ++qq) {
                    if (ql == qq) {
                        if (nk > readlane_u64(list[qq], top - 1)) {
                            wave_list_insert(list[qq], nk, lane);
                            const uint64_t nt = readlane_u64(list[qq], top - 1);
                            if (n == qq) thr_f = nt ? key_score(nt) : -__builtin_inff();
                        }
                    }
                }
            }
        }
    };

    // The non-matrix work of a stage (LDS reads of the next operands, the refill of the ring, the scalar address arithmetic) is
    // spread BETWEEN the 32 MFMAs in program order - a wave issues in order, so whatever sits behind the last MFMA of a run waits
    // for the whole run - and the other block on the CU (2 blocks x 4 waves = 2 waves per SIMD, free-running) fills what is left.
    uint32_t slot = 0;        // ring slot of the stage being multiplied
    for (uint64_t it = 0; it < my_tiles; ++it) {
        const uint64_t tile = blockIdx.x + it * gridDim.x;
        f32x4 acc[8][2];
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[ci][t] = (f32x4){0.f, 0.f, 0.f, 0.f};   // _mm256_setzero_ps

#pragma unroll
        for (int kc = 0; kc < KS; ++kc) {
            const unsigned char *cur = smem + slot * M16_STAGE + a_off;
            slot = (slot + 1) & (M16_NBUF - 1);
            const unsigned char *nxt = smem + slot * M16_STAGE + a_off;
            // K-step 0 (operands read during the previous stage); the K-step 1 operands of this stage arrive underneath
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                if (DBG == 1) acc[ci][0][0] += ax0[ci];
                else {
                    acc[ci][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax0[ci], bq[ci][kc * 2][0], acc[ci][0], 0, 0, 0);
                    acc[ci][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax0[ci], bq[ci][kc * 2][1], acc[ci][1], 0, 0, 0);
                }
                if (ci < 4) {
                    ax1[2 * ci] = lds_a(cur, 1, 2 * ci);
                    ax1[2 * ci + 1] = lds_a(cur, 1, 2 * ci + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // this wave's 4 loads of the next stage have landed when at most the 4 (NBUF - 2) issued after them are outstanding; the
            // barrier makes that true for every wave's rows and says every wave holds all of the current stage in registers
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(PER_STAGE * (M16_NBUF - 2)) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // K-step 1; the refill of the slot just freed and the K-step 0 operands of the next stage go in between
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                if (DBG == 1) acc[ci][1][0] += ax1[ci];
                else {
                    acc[ci][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax1[ci], bq[ci][kc * 2 + 1][0], acc[ci][0], 0, 0, 0);
                    acc[ci][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax1[ci], bq[ci][kc * 2 + 1][1], acc[ci][1], 0, 0, 0);
                }
                if (ci < 4) {
                    if (ci == 0) pf_issue();
                    issue_row(ci);
                    if (ci == 3) ld_advance();
                } else {
                    ax0[2 * (ci - 4)] = lds_a(nxt, 0, 2 * (ci - 4));
                    ax0[2 * (ci - 4) + 1] = lds_a(nxt, 0, 2 * (ci - 4) + 1);
                }
                // (the tail of the tile's last stage is left to the scheduler: it pulls the fold's adds up between the MFMAs)
                if (!(kc == KS - 1 && ci >= 4) || getenv_pin) __builtin_amdgcn_sched_barrier(0);
                // the lr values of the PREVIOUS tile sit in xch since before this stage's barrier: finish that tile here, under
                // the matrix work of this one
                if (kc == 0 && (ci == 4 || ci == 5) && it > 0 && DBG != 3) {
                    finalize(tile - gridDim.x, ci - 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        if (DBG == 3) {
            if (it + 1 < my_tiles) continue;       // keep the accumulators alive: only the last tile folds
        }
        // ---- fold: four_way_hsum (a + b) + (c + d) per SIMD lane, then hi128 + lo128; register r = ci >> 1, jj = ci & 1.  The
        // exchange is read behind the NEXT stage barrier (finalize above / below): no barrier of its own.  With KS >= 2 the next
        // write of xch is behind a second stage barrier, which no wave passes before it has read xch; KS == 1 needs its own.
        if (KS == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 t_lo = (acc[0][t] + acc[2][t]) + (acc[4][t] + acc[6][t]);     // SIMD lane w
            const f32x4 t_hi = (acc[1][t] + acc[3][t]) + (acc[5][t] + acc[7][t]);     // SIMD lane w + 4
            const f32x4 lr = t_hi + t_lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) xch[((w * 2 + t) * 4 + j) * 64 + lane] = lr[j];
        }
    }
    if (my_tiles) {               // the last tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const uint64_t last = blockIdx.x + (my_tiles - 1) * gridDim.x;
        finalize(last, 0);
        finalize(last, 1);
    }

    // ---- block merge: the 2 wave lists of each query (waves tq, tq + 2) -> 1 list, one global write per block ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the phantom stages must have landed before the ring is reused
    __syncthreads();
    uint64_t *lds_keys = reinterpret_cast<uint64_t *>(smem);
    const uint32_t utop = a.top;
#pragma unroll
    for (int q = 0; q < 16; ++q)
        if (lane < top) lds_keys[((uint32_t)w * 16 + q) * utop + lane] = list[q];
    __syncthreads();
    for (uint32_t q = w; q < a.nq; q += 4) {
        uint64_t merged = 0;
        for (int half = 0; half < 2; ++half) {
            const uint32_t sw = (q >> 4) + 2 * half;
            const uint64_t key = lane < top ? lds_keys[(sw * 16 + (q & 15)) * utop + lane] : 0;
            uint64_t mk = __ballot(key > readlane_u64(merged, top - 1));
            while (mk) {
                const int src = __builtin_ctzll(mk);
                mk &= mk - 1;
                const uint64_t nk = readlane_u64(key, src);
                if (nk > readlane_u64(merged, top - 1)) wave_list_insert(merged, nk, lane);
            }
        }
        if (lane < top) a.partial[((uint64_t)blockIdx.x * M16_QT + q) * utop + lane] = merged;
    }
}

template <int KS, int DBG = 0, bool PF = false>
static int32_t launch_m16(hipStream_t st, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    auto kfn = scan_f32_mfma16_kernel<KS, DBG, PF>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const uint64_t n_tiles = (a.n_cand + 15) / 16;
    int per_cu = 0;
    QMX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, (size_t)M16_LDS));
    if (per_cu < 1) per_cu = 1;
    const uint64_t cap = (uint64_t)num_cus * per_cu;                                         // persistent blocks: 2 per CU
    uint32_t grid = (uint32_t)(n_tiles < cap ? n_tiles : cap);
    if (grid < 1) grid = 1;
    if (grid_out) {
        if (*grid_out && grid > *grid_out) grid = *grid_out;
        *grid_out = grid;
    }
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), (size_t)M16_LDS, st, a);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

bool mfma16_scan_ok(ScanMode mode, const ScanArgs &a) {
    return getenv("QMX_NO_MFMA16") == nullptr && mode == SCAN_TOPK && a.ids == nullptr && a.rem_pieces == 0 && a.tail_start == a.dim && a.nseg % 8 == 0 && a.nseg / 8 >= 1 &&
           a.nseg / 8 <= 3 && a.row_stride % 16 == 0 && a.top <= 64;
}

// 32-query tile, top-k over the whole block; the caller checked mfma16_scan_ok
int32_t launch_scan_f32_mfma16(hipStream_t st, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    switch (a.nseg / 8) {
        case 1: return launch_m16<1>(st, a, num_cus, grid_out);
        case 2: return launch_m16<2>(st, a, num_cus, grid_out);
        case 3: {
            const char *dbg = getenv("QMX_M16_DBG");
            if (dbg && dbg[0] == '1') return launch_m16<3, 1>(st, a, num_cus, grid_out);
            if (dbg && dbg[0] == '2') return launch_m16<3, 2>(st, a, num_cus, grid_out);
            if (dbg && dbg[0] == '3') return launch_m16<3, 3>(st, a, num_cus, grid_out);
            if (dbg && dbg[0] == '4') return launch_m16<3, 0, true>(st, a, num_cus, grid_out);
            return launch_m16<3>(st, a, num_cus, grid_out);
        }
    }
    set_error("mfma16 scan: unsupported row length");
    return QMX_ERR_BAD_ARG;
}

}  // namespace qmx

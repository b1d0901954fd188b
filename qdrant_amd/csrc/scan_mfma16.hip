// scan_mfma16.hip — f32 dot / cosine brute-force top-k for 32- and 64-query tiles on v_mfma_f32_16x16x4_f32, chain-major,
// with the bits of the x86 AVX2+FMA reference (dot_similarity_avx, lib/segment/src/spaces/simple_avx.rs:167-213 under
// BatchFilteredSearcher::peek_top_iter, lib/segment/src/index/hnsw_index/point_scorer.rs:423-472).
//
// Why another kernel.  scan_mfma.hip maps one AVX chain to one 4x4x1 MFMA block: every instruction needs a fresh A and a
// fresh B register for 512 flops and a cross-lane fold per tile; at 32 queries the CU runs out of issue slots (9.9 ms per
// 30.72 GB scan, the HBM floor is ~5 ms).  v_mfma_f32_16x16x4_f32 does 2048 flops per A / B register pair and is still an
// exact f32 fma chain: D[m][n] = fma(A[m][3], B[3][n], fma(A[m][2], B[2][n], fma(A[m][1], B[1][n], fma(A[m][0], B[0][n], C[m][n]))))
// (the parity tests pin that order).  dot_similarity_avx keeps 32 independent chains per (row, query): chain c = 8 r + j (AVX
// register r, SIMD lane j) takes element 32 i + c of every 32-float step i in order.  Chains are independent, so they can be
// run ONE AFTER THE OTHER: the 16x16 accumulator tile of chain c holds (16 stored rows) x (16 queries), the K stream of
// the instruction is elements c, c + 32, c + 64, ... of the rows and of the queries.  Folding the 32 chains is then
// ELEMENTWISE on accumulator tiles in the reference's tree (four_way_hsum, hsum256_ps_avx, simple_avx.rs:10-28) - no
// cross-lane traffic at all.
//
// Four shapes of the same kernel (NW waves per block, NT = QT / 16 query tiles):
//   QT = 16, dim <= 1536: NW = 4, two blocks per CU, 8 chains x 1 query tile per wave (the pass is an HBM stream: 0.86 of peak).
//   QT = 32, dim <= 768:  NW = 4, two blocks per CU.  Wave w owns SIMD lanes j = w and w + 4 of the four AVX registers (8 chains).
//   QT = 64, dim <= 768:  NW = 8, one block per CU.   Wave w owns SIMD lane j = w (4 chains).
//   QT = 32, dim <= 1536: NW = 8, one block per CU, 4 chains x 2 query tiles per wave (the longer rows need the registers).
// Either way a wave holds its K-step x query-tile slices of the QUERIES in registers for the whole kernel (B operands,
// 96 VGPRs at dim 768 / 1536: the queries are never re-read) and up to 64 accumulator VGPRs.
// The fold: T_j = (s1 + s2) + (s3 + s4) per SIMD lane j (in-wave), lr_j = T_{j+4} + T_j (extractf128(x, 1) + cast(x);
// in-wave for NW = 4), then the waves exchange through LDS and score = (lr_0 + lr_1) + (lr_2 + lr_3) (_mm_hadd_ps, p1 + p2).
// Rows stream HBM -> LDS with global_load_lds_dwordx4 (1 KiB of ONE row per wave instruction: fully coalesced, no
// staging VGPRs) into a ring of stages of 16 rows x 256 floats, counted with s_waitcnt vmcnt(N) - never 0 inside the loop.
// The A operand of (chain c, K-step) is one ds_read_b32: lane (m = lane & 15, k = lane >> 4) reads element
// c + 32 (4 step + k) of row m.  Every byte of the block is read from HBM once and from LDS once.
//
// Restrictions (anything else takes the scan_mfma.hip path): top-k mode (whole block or a candidate id list), dim a multiple of
// 128 up to 2048 (64-query passes: up to 768), 16-byte aligned rows.  Rows of an even number of 512-byte K-steps stream in
// 1 KiB chunks (two K-steps per ring stage), the others (384, 640, ...) in 512-byte chunks, two rows per load.
#include "scan_common.hpp"

namespace qmx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;

constexpr int M16_ROWP = 1024 + 16;         // LDS pitch of a 1 KiB row chunk
constexpr int M16_STAGE = 16 * M16_ROWP;

// 1 KiB of one row (wave-uniform base `row`, lane i fetches bytes 16 i .. 16 i + 15) -> LDS at the wave-uniform byte address
// lds_dst (lane i lands at lds_dst + 16 i).  Invisible to the compiler's s_waitcnt bookkeeping on purpose: the loop below counts
// these loads itself.
__device__ __forceinline__ void glds16(const unsigned char *row, uint32_t lane_off, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(row), "s"(lds_dst) : "memory");
}
// the same with a per-lane source address
__device__ __forceinline__ void glds16v(const unsigned char *src, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// one dword through the scalar cache (wave-uniform address): candidate ids, without touching the vector-memory counter
__device__ __forceinline__ uint32_t sload_u32(const uint32_t *p) {
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
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

template <int NW, int NT, int KPS /* 512-byte K-steps of a row per ring stage: 2, or 1 when the row has an odd number of them */>
struct M16Shape {
    static constexpr int QT = 16 * NT;
    static constexpr int JW = 8 / NW;                  // SIMD lanes j per wave (2: j = w, w + 4;  1: j = w)
    static constexpr int CW = 4 * JW;                  // chains per wave: ci = r * JW + jj  ->  chain 8 r + w + 4 jj
    static constexpr int RPW = 16 / NW;                // rows of a stage fetched per wave
    static constexpr int STAGE = KPS == 2 ? M16_STAGE : M16_STAGE / 2;   // 16 rows x 1 KiB (pitch 1040) or 8 row pairs x (2 x 512 B) (pitch 1040)
    static constexpr int NBUF = (NW == 4 ? 4 : 7) * (KPS == 2 ? 1 : 2);   // ring stages (4 waves: two blocks share the CU's 160 KiB)
    static constexpr int XCH = NW * NT * 4 * 64 * 4;   // fold exchange: [wave][query tile][reg][lane] f32
    static constexpr int IDR = 16 * 16 * 4;            // candidate ids of the last 16 tiles [tile iteration & 15][row] (id-list scans)
    static constexpr int LDS = NBUF * STAGE + XCH + IDR;
    static constexpr int VPS = KPS == 2 ? RPW : RPW / 2;   // vector-memory instructions per stage and wave (a 1 KiB load holds one row chunk or two half chunks)
    static constexpr int PW = NT * 4 / NW;             // accumulator planes (query tile, register) a wave finishes per tile
    static_assert(PW == 1 || PW == 2, "the NT * 4 planes of a tile are split evenly over the waves");
};

template <int KSTEPS /* dim / 128: 512-byte K-steps per row */, int NW, int NT, int DBG = 0 /* tuning experiments (QMX_M16_DBG): 2 no row loads, 3 no fold / selection */,
          bool LAG = (NW == 8 && NT == 4 && KSTEPS == 6) /* half of the waves run one stage behind the others, see `lag` */,
          bool IDS = false /* candidates = a.ids[0 .. n_cand) instead of rows 0 .. n_cand) (payload-filtered scans, peek_top_iter) */>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void scan_f32_mfma16_kernel(const ScanArgs a) {
    if (a.run_if && *a.run_if == 0) return;           // the exact pass behind the split prefilter was not needed (scan_split.hip)
    constexpr int KPS = KSTEPS % 2 == 0 ? 2 : 1;      // K-steps per ring stage
    constexpr int KS = KSTEPS / KPS;                  // stages per 16-row tile
    typedef M16Shape<NW, NT, KPS> S;
    constexpr int QT = S::QT, JW = S::JW, CW = S::CW, RPW = S::RPW, NBUF = S::NBUF, PW = S::PW;
    constexpr int M16_STAGE_B = S::STAGE;
    static_assert(!LAG || KPS == 2, "the lagging-wave schedule is written for two-step stages");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kk = lane >> 4;
    // LAG: the two waves of a SIMD would otherwise fold / select at the same time (they run between the same barriers) and leave the
    // matrix pipe idle meanwhile.  Waves NW / 2 .. NW - 1 (the second wave of every SIMD) therefore consume the ring ONE STAGE
    // BEHIND the others: their fold falls under the other wave's MFMAs and vice versa.  The barrier sequence B_0, B_1, ... is shared:
    // a leading wave multiplies stage k around B_k, a lagging wave stage k - 1.  What changes: the slot refilled after B_k is that of
    // stage k - 1 (everyone is done with it), so one stage less is in flight; a lagging wave passes B_0 before its loop and a leading
    // wave one barrier after its loop; the fold values of a tile are complete one barrier later, so leading waves finish the
    // previous tile during chunk 1 of the next one (lagging waves during chunk 0, as without LAG).  xch is single-buffered: safe for
    // KS = 3 (reads of tile T happen in real stage 3 T + 4, the next writes in 3 T + 5 and 3 T + 6).
    const bool lag = LAG && (((a.flags & 0x100u) ? (w & 1) : (w >= NW / 2)) != 0);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte *)smem;
    const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
    const int top = (int)a.top;
    auto chain_rel = [](int ci) { return 8 * (ci / JW) + 4 * (ci % JW); };     // chain(ci) - w

    // ---- this wave's slice of the queries, in B-operand layout: bq[ci][s][t] = query 16 t + n, element chain(ci) + 32 (4 s + kk)
    float bq[CW][KSTEPS][NT];
    {
        const float *qf = reinterpret_cast<const float *>(a.queries);
        const uint32_t qs = a.q_stride / 4;
#pragma unroll
        for (int ci = 0; ci < CW; ++ci)
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    bq[ci][s][t] = qf[(uint32_t)(16 * t + n) * qs + (uint32_t)(chain_rel(ci) + w) + 32u * (uint32_t)(4 * s + kk)];
    }
    // epilogue role of the wave: query tile tq (queries 16 tq + n), accumulator registers j0 .. j0 + PW - 1 (rows 4 kk + j0 + p)
    const int tq = w % NT, j0 = PW * (w / NT);
    const int my_q = 16 * tq + n;
    const bool has_kb = a.key_bound != nullptr;       // bound of a later pass of a top > 64 search (keys must stay below it)
    uint64_t kb = has_kb ? a.key_bound[my_q < (int)a.nq ? my_q : 0] : 0;
    // lower bound of the query's final k-th best score from the host's pre-scan of a prefix of the block (0 = none): rows scoring
    // below it are never candidates, so the wave lists start selective instead of each re-learning the threshold from its own rows
    uint32_t g_ord = a.gthr ? (uint32_t)(a.gthr[my_q < (int)a.nq ? my_q : 0] >> 32) : 0;
    // make every query register "used" here: the compiler then waits for its own loads before the loop instead of at their first
    // use inside it (a vmcnt(N) there would drain the row stream, which it does not know about)
#pragma unroll
    for (int ci = 0; ci < CW; ++ci)
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(bq[ci][s][t]));
    asm volatile("" : "+v"(kb));
    asm volatile("" : "+v"(g_ord));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // from here on the only vector loads in flight are the row stream

    uint64_t list[16];        // this wave's top list of its 16 queries (lane i = i-th best key)
    // reject threshold of the lane's own query: the score of the k-th best key of the wave's list once it is full, the pre-scan's bound
    // before that (-inf without one)
    float thr_f = g_ord ? ord_to_score(g_ord) : -__builtin_inff();
#pragma unroll
    for (int q = 0; q < 16; ++q) list[q] = 0;

    const uint64_t n_tiles = (a.n_cand + 15) / 16;
    const uint64_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;

    // ---- the row stream.  Stages are issued strictly in order: stage = (tile iteration, 1 KiB chunk kc of its rows); this wave
    // fetches rows RPW w .. RPW w + RPW - 1 of the tile.  Stages past the end re-read row 0 (never consumed): the in-flight count
    // stays constant and the loop needs no tail.  All of it is scalar work: row bases are wave-uniform.
    const uint32_t lane_off = (uint32_t)lane * 16u;
    uint32_t *idring = reinterpret_cast<uint32_t *>(smem + NBUF * M16_STAGE_B + S::XCH);
    uint64_t ld_it = 0;            // tile iteration of the next stage to issue
    uint32_t ld_kc = 0;            // ... and its chunk
    uint32_t ld_slot = 0;
    const unsigned char *ld_row[RPW];
    auto ld_new_tile = [&]() {
        const uint64_t tile = blockIdx.x + ld_it * gridDim.x;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            uint64_t c = tile * 16 + (uint32_t)(RPW * w + i);
            if (c >= a.n_cand) c = a.n_cand - 1;           // rows past the end of a partial last tile: any valid row, results dropped
            if (ld_it >= my_tiles) c = 0;
            if (IDS) {                                     // the candidate's row: its id travels to the tile's finalize through LDS
                uint32_t id = sload_u32(reinterpret_cast<const uint32_t *>(uniform_u64((uint64_t)(a.ids + c))));
                if (lane == 0) idring[((uint32_t)ld_it & 15u) * 16u + (uint32_t)(RPW * w + i)] = id;
                c = id < a.n_rows ? id : 0;                // an id past the storage is reported by finalize; never read
            }
            ld_row[i] = reinterpret_cast<const unsigned char *>(uniform_u64((uint64_t)(rows + c * a.row_stride)));
        }
    };
    // one 1 KiB load of the stage being issued: KPS == 2: the chunk of row RPW w + i; KPS == 1: the 512-byte chunks of the row pair
    // (RPW w + 2 i, + 1), lanes 0..31 / 32..63, landing side by side (pitch 1040 per pair keeps the 4-way bank pattern of the b32 reads)
    auto issue_row = [&](int i) {
        if (DBG == 2) return;
        if (KPS == 2) {
            glds16(ld_row[i] + ld_kc * 1024u, lane_off, lds0 + ld_slot * M16_STAGE_B + (uint32_t)(RPW * w + i) * M16_ROWP);
        } else {
            const unsigned char *src = (lane < 32 ? ld_row[(2 * i) % RPW] : ld_row[(2 * i + 1) % RPW]) + ld_kc * 512u + ((uint32_t)lane & 31u) * 16u;
            glds16v(src, lds0 + ld_slot * M16_STAGE_B + (uint32_t)(RPW / 2 * w + i) * M16_ROWP);
        }
    };
    auto ld_advance = [&]() {                              // after the rows of a stage
        ld_slot = ld_slot + 1 == NBUF ? 0 : ld_slot + 1;
        if (++ld_kc == KS) {
            ld_kc = 0;
            ++ld_it;
            ld_new_tile();
        }
    };
    ld_new_tile();
    const int n_prologue = LAG ? NBUF - 1 + (lag ? 1 : 0) : NBUF;     // LAG: every wave has issued stage k + NBUF - 1 once it is past B_k
    for (int p = 0; p < n_prologue; ++p) {
#pragma unroll
        for (int i = 0; i < S::VPS; ++i) issue_row(i);
        ld_advance();
    }

    // lane part of the A-operand address: row n of the stage, 128-byte segment kk of the K-step, column w
    const uint32_t a_off = (KPS == 2 ? (uint32_t)n * M16_ROWP : (uint32_t)(n >> 1) * M16_ROWP + (uint32_t)(n & 1) * 512u) + (uint32_t)kk * 128u + (uint32_t)w * 4u;
    float *xch = reinterpret_cast<float *>(smem + NBUF * M16_STAGE_B);
    // A operand of (chain index ci, K-step ks) of a stage: element chain(ci) + 32 (4 ks + kk) of the 256-float chunk of row n
    auto lds_a = [&](const unsigned char *stage_base, int ks, int ci) -> float {
        return *reinterpret_cast<const float *>(stage_base + chain_rel(ci) * 4 + ks * 512);
    };
    float ax0[CW], ax1[CW];   // K-steps 0 and 1 of the current stage
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(S::VPS * (NBUF - 1 - (LAG ? 1 : 0))) : "memory");   // stage 0 of this wave has landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int ci = 0; ci < CW; ++ci) ax0[ci] = lds_a(smem + a_off, 0, ci);

    // Second half of a tile's epilogue, run one stage barrier after the tile's fold values were written to xch: the wave finishes
    // accumulator register j0 + p of query tile tq: row 4 (lane >> 4) + j0 + p of the tile, query 16 tq + (lane & 15).
    auto finalize = [&](uint64_t it_of_tile, int p) {
        const uint64_t tile = blockIdx.x + it_of_tile * gridDim.x;
        const int j = j0 + p;
        auto x = [&](int src_wave) { return xch[((src_wave * NT + tq) * 4 + j) * 64 + lane]; };
        float score;
        if (JW == 2) {
            score = (x(0) + x(1)) + (x(2) + x(3));                                  // hadd: p1 = lr0 + lr1, p2 = lr2 + lr3; p1 + p2
        } else {
            const float lr0 = x(4 % NW) + x(0), lr1 = x(5 % NW) + x(1), lr2 = x(6 % NW) + x(2), lr3 = x(7 % NW) + x(3);   // extractf128(x, 1) + cast(x)
            score = (lr0 + lr1) + (lr2 + lr3);
        }
        const uint64_t c = tile * 16 + (uint32_t)(4 * kk + j);
        uint32_t res_row = (uint32_t)c;
        bool mine = c < a.n_cand && my_q < (int)a.nq;
        if (IDS) {
            res_row = idring[((uint32_t)it_of_tile & 15u) * 16u + (uint32_t)(4 * kk + j)];
            if (mine && res_row >= a.n_rows) {             // raw_scorer.rs would panic on such an id: report it
                *a.err_flag = 1;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (keeps the store out of the load counting; error path only)
                mine = false;
            }
        }
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
                            if (n == qq && nt && !(key_score(nt) < thr_f)) thr_f = key_score(nt);   // never below the pre-scan's bound
                        }
                    }
                }
            }
        }
    };

    // The non-matrix work of a stage (LDS reads of the next operands, the refill of the ring, the scalar address arithmetic) is
    // spread BETWEEN the 32 MFMAs in program order - a wave issues in order, so whatever sits behind the last MFMA of a run waits
    // for the whole run - and the other wave on the SIMD fills what is left.
    if (lag) {                // B_0: this wave's rows of stage 1 have landed (NBUF stages issued, NBUF - 2 of them after stage 1)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(S::VPS * (NBUF - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    uint32_t slot = 0;        // ring slot of the stage being multiplied
    for (uint64_t it = 0; it < my_tiles; ++it) {
        f32x4 acc[CW][NT];
#pragma unroll
        for (int ci = 0; ci < CW; ++ci)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[ci][t] = (f32x4){0.f, 0.f, 0.f, 0.f};   // _mm256_setzero_ps

        if constexpr (KPS == 1) {
            // One K-step per stage (rows of an odd number of 512-byte K-steps): barrier first - it says the next stage has landed and
            // that every wave holds the current one in registers - then the refill of the current stage's slot and the LDS reads of the
            // next stage's operands go between the MFMAs of the current one.
#pragma unroll
            for (int kc = 0; kc < KS; ++kc) {
                slot = slot + 1 == NBUF ? 0 : slot + 1;
                const unsigned char *nxt = smem + slot * M16_STAGE_B + a_off;
                if (DBG != 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(S::VPS * (NBUF - 2)) : "memory");
                if (DBG != 4 && DBG != 6) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#pragma unroll
                for (int ci = 0; ci < CW; ++ci) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[ci][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax0[ci], bq[ci][kc][t], acc[ci][t], 0, 0, 0);
                    if (ci < S::VPS) issue_row(ci);
                    if (ci == S::VPS - 1) ld_advance();
                    if (ci >= CW / 2) {
                        ax1[2 * (ci - CW / 2)] = lds_a(nxt, 0, 2 * (ci - CW / 2));
                        ax1[2 * (ci - CW / 2) + 1] = lds_a(nxt, 0, 2 * (ci - CW / 2) + 1);
                    }
                    if (!(kc == KS - 1 && ci >= CW / 2)) __builtin_amdgcn_sched_barrier(0);
                    if (DBG != 3 && ci >= CW / 2 && ci < CW / 2 + PW && it > 0 && kc == 0) {
                        finalize(it - 1, ci - CW / 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int ci = 0; ci < CW; ++ci) ax0[ci] = ax1[ci];
            }
        } else {
#pragma unroll
            for (int kc = 0; kc < KS; ++kc) {
                const unsigned char *cur = smem + slot * M16_STAGE_B + a_off;
                slot = slot + 1 == NBUF ? 0 : slot + 1;
                const unsigned char *nxt = smem + slot * M16_STAGE_B + a_off;
                // K-step 0 (operands read during the previous stage); the K-step 1 operands of this stage arrive underneath
#pragma unroll
                for (int ci = 0; ci < CW; ++ci) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[ci][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax0[ci], bq[ci][kc * 2][t], acc[ci][t], 0, 0, 0);
                    if (ci < CW / 2) {
                        ax1[2 * ci] = lds_a(cur, 1, 2 * ci);
                        ax1[2 * ci + 1] = lds_a(cur, 1, 2 * ci + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // this wave's loads of the next stage have landed when at most the RPW (NBUF - 2) issued after them are outstanding; the
                // barrier makes that true for every wave's rows and says every wave holds all of the current stage in registers
                if (DBG != 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(S::VPS * (NBUF - 2 - (LAG ? 1 : 0))) : "memory");
                if (DBG != 4 && DBG != 6) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                // K-step 1; the refill of the slot just freed and the K-step 0 operands of the next stage go in between
#pragma unroll
                for (int ci = 0; ci < CW; ++ci) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[ci][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax1[ci], bq[ci][kc * 2 + 1][t], acc[ci][t], 0, 0, 0);
                    if (ci < CW / 2) {
                        // the refill: RPW rows over the first CW / 2 slots (RPW == CW / 2 in both shapes)
                        if (ci < RPW) issue_row(ci);
                        if (ci == CW / 2 - 1) ld_advance();
                    } else {
                        ax0[2 * (ci - CW / 2)] = lds_a(nxt, 0, 2 * (ci - CW / 2));
                        ax0[2 * (ci - CW / 2) + 1] = lds_a(nxt, 0, 2 * (ci - CW / 2) + 1);
                    }
                    // (the tail of the tile's last stage is left to the scheduler: it pulls the fold's adds up between the MFMAs)
                    if (!(kc == KS - 1 && ci >= CW / 2)) __builtin_amdgcn_sched_barrier(0);
                    // the fold values of the PREVIOUS tile sit in xch since before this stage's barrier: finish that tile here, under
                    // the matrix work of this one
                    if (DBG != 3 && ci >= CW / 2 && ci < CW / 2 + PW && it > 0 && ((kc == 0 && (!LAG || lag)) || (LAG && kc == 1 && !lag))) {
                        finalize(it - 1, ci - CW / 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }

        }

        // ---- fold: four_way_hsum (a + b) + (c + d) per SIMD lane [then hi128 + lo128]; AVX register r = ci / JW, jj = ci % JW.
        // The exchange is read behind the NEXT stage barrier (finalize above / below): no barrier of its own.  With KS >= 2 the
        // next write of xch is behind a second stage barrier, which no wave passes before it has read xch; KS == 1 needs its own.
        if (DBG == 3 && it + 1 < my_tiles) continue;      // (keeps the accumulators alive: only the last tile folds)
        if (KS == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 v;
            if (JW == 2) {
                const f32x4 t_lo = (acc[0][t] + acc[2][t]) + (acc[4 % CW][t] + acc[6 % CW][t]);   // SIMD lane w
                const f32x4 t_hi = (acc[1][t] + acc[3][t]) + (acc[5 % CW][t] + acc[7 % CW][t]);   // SIMD lane w + 4
                v = t_hi + t_lo;                                                                    // lr_w
            } else {
                v = (acc[0][t] + acc[1][t]) + (acc[2][t] + acc[3][t]);                              // T_w
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) xch[((w * NT + t) * 4 + j) * 64 + lane] = v[j];
        }
    }
    if (LAG && !lag) {            // the lagging waves' last stage
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (my_tiles) {               // the last tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        finalize(my_tiles - 1, 0);
        if (PW == 2) finalize(my_tiles - 1, 1);
    }

    // ---- block merge: the NW / NT wave lists of each query (waves tq, tq + NT, ...) -> 1 list, one global write per block ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the phantom stages must have landed before the ring is reused
    __syncthreads();
    uint64_t *lds_keys = reinterpret_cast<uint64_t *>(smem);
    const uint32_t utop = a.top;
#pragma unroll
    for (int q = 0; q < 16; ++q)
        if (lane < top) lds_keys[((uint32_t)w * 16 + q) * utop + lane] = list[q];
    __syncthreads();
    for (uint32_t q = w; q < a.nq; q += NW) {
        uint64_t merged = 0;
        for (int h = 0; h < NW / NT; ++h) {
            const uint32_t sw = (q >> 4) + NT * h;
            const uint64_t key = lane < top ? lds_keys[(sw * 16 + (q & 15)) * utop + lane] : 0;
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

template <int KSTEPS, int NW, int NT, int DBG = 0, bool LAG = (NW == 8 && NT == 4 && KSTEPS == 6), bool IDS = false>
static int32_t launch_m16(hipStream_t st, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    if constexpr (!IDS && DBG == 0 && LAG == (NW == 8 && NT == 4 && KSTEPS == 6)) {
        if (a.ids) {
            // (4 waves x 2 query tiles at 5 or 6 K-steps have no register left for the id plumbing: the 8-wave 32-query shape takes it)
            if constexpr (KSTEPS >= 5 && NW == 4 && NT == 2) return launch_m16<KSTEPS, 8, 2, 0, false, true>(st, a, num_cus, grid_out);
            else return launch_m16<KSTEPS, NW, NT, 0, LAG, true>(st, a, num_cus, grid_out);
        }
    }
    typedef M16Shape<NW, NT, (KSTEPS % 2 == 0 ? 2 : 1)> S;
    auto kfn = scan_f32_mfma16_kernel<KSTEPS, NW, NT, DBG, LAG, IDS>;
    static thread_local DeviceOnce attr_once;
    if (attr_once.need()) {
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.mark();
    }
    const uint64_t n_tiles = (a.n_cand + 15) / 16;
    int per_cu = 0;
    QMX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, NW * 64, (size_t)S::LDS));
    if (per_cu < 1) per_cu = 1;
    const uint64_t cap = (uint64_t)num_cus * per_cu;                                         // persistent blocks
    uint32_t grid = (uint32_t)(n_tiles < cap ? n_tiles : cap);
    if (grid < 1) grid = 1;
    if (grid_out) {
        if (*grid_out && grid > *grid_out) grid = *grid_out;
        *grid_out = grid;
    }
    ScanArgs b = a;
#ifdef QMX_TUNING
    if (getenv("QMX_M16_LAG_ODD")) b.flags |= 0x100u;       // tuning experiment: odd waves lag instead of the upper half
#endif
    ::qmx::clear_stale_error();
    QMX_NOTE_KERNEL(kfn);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(NW * 64), (size_t)S::LDS, st, b);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// qt = 16, 32 or 64; rows of dim = 128 k floats: k <= 12, 14 or 16 (k <= 6 for 64 queries: the query registers)
// row lengths the kernel is built for: dim = 128 k floats, k <= 12, 14 or 16 (k <= 6 for 64 queries: the query registers)
bool mfma16_dim_ok(int qt, uint32_t dim) {
    if (option(OPT_NO_MFMA16) || dim % 128 != 0) return false;
    const uint32_t k = dim / 128;
    return k >= 1 && k <= (qt == 64 ? 6u : 16u) && !(k > 12 && k % 2 == 1);
}

bool mfma16_scan_ok(int qt, ScanMode mode, const ScanArgs &a) {
    if (option(OPT_NO_MFMA16)) return false;
    return (qt == 16 || qt == 32 || qt == 64) && mode == SCAN_TOPK && a.rem_pieces == 0 && a.tail_start == a.dim && mfma16_dim_ok(qt, a.dim) &&
           a.row_stride % 16 == 0 && a.top <= 64;
}

template <int NW, int NT, int K0, int K1>
static int32_t launch_m16_steps(hipStream_t st, int ksteps, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    if constexpr (K0 <= K1) {
        if constexpr (!(K0 > 12 && K0 % 2 == 1)) {      // (13 / 15 one-step stages: the stage loop no longer unrolls)
            if (ksteps == K0) return launch_m16<K0, NW, NT>(st, a, num_cus, grid_out);
        }
        return launch_m16_steps<NW, NT, K0 + 1, K1>(st, ksteps, a, num_cus, grid_out);
    } else {
        set_error("mfma16 scan: unsupported row length");
        return QMX_ERR_BAD_ARG;
    }
}

// top-k over the whole block or a candidate id list; the caller checked mfma16_scan_ok
int32_t launch_scan_f32_mfma16(hipStream_t st, int qt, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    const int ksteps = (int)(a.nseg / 4);
    if (qt == 16) return launch_m16_steps<4, 1, 1, 16>(st, ksteps, a, num_cus, grid_out);
    if (qt == 32) {
        if (ksteps <= 6) return launch_m16_steps<4, 2, 1, 6>(st, ksteps, a, num_cus, grid_out);
        return launch_m16_steps<8, 2, 7, 16>(st, ksteps, a, num_cus, grid_out);
    }
    if (qt == 64) {
#ifdef QMX_TUNING   // instrumented instantiations (some skip waits or barriers and return wrong results): never in the shipped library
        if (ksteps == 6) {
            const char *dbg = getenv("QMX_M16_DBG");
            if (dbg && dbg[0] == '2') return launch_m16<6, 8, 4, 2>(st, a, num_cus, grid_out);
            if (dbg && dbg[0] == '3') return launch_m16<6, 8, 4, 3>(st, a, num_cus, grid_out);
            if (dbg && dbg[0] == '4') return launch_m16<6, 8, 4, 4, false>(st, a, num_cus, grid_out);   // (wrong results) no stage wait, no stage barrier
            if (dbg && dbg[0] == '6') return launch_m16<6, 8, 4, 6, false>(st, a, num_cus, grid_out);   // (wrong results) no stage barrier
            if (dbg && dbg[0] == '7') return launch_m16<6, 8, 4, 3, false>(st, a, num_cus, grid_out);   // no fold / selection, lock-step
            if (dbg && dbg[0] == '5') return launch_m16<6, 8, 4, 0, false>(st, a, num_cus, grid_out);   // lock-step waves
        }
#endif
        return launch_m16_steps<8, 4, 1, 6>(st, ksteps, a, num_cus, grid_out);
    }
    set_error("mfma16 scan: unsupported shape");
    return QMX_ERR_BAD_ARG;
}

}  // namespace qmx
